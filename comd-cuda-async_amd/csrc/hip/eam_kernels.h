// eam_kernels.h -- EAM force kernels (three passes) for gfx950.
//
// Physics of the reference's EAM_Force_thread_atom<step> / EAM_Force_cta_cell<step>
// (gpu_eam_thread_atom.h:32-140, gpu_eam_cta_cell.h:34-278) and the pass-2 kernels
// (gpu_eam_thread_atom.h:269-287, gpu_eam_cta_cell.h:280-295):
//   pass 1: e_i = 1/2 sum phi(r), rhobar_i = sum rho(r), f_i -= phi'(r) d/r          (0 < r^2 <= rc^2)
//   pass 2: dfEmbed_i = F'(rhobar_i), e_i += F(rhobar_i)
//   pass 3: f_i -= (F'_i + F'_j) rho'(r) d/r
// Tables are the reference's default quadratic interpolation (gpu_common.h:48-86); the -P spline mode is out of scope.
//
//  thread_atom : one thread per cell slot (cell*cap + i), tables read through L1/L2.
//  cta_cell    : 4 waves per workgroup, each wave owns one cell at a time.  phi/rho tables live in LDS for the whole
//                (persistent) workgroup; the wave stages the positions (and F' in pass 3) of its 27 stencil cells,
//                compacted, in LDS, spreads the candidates over its lanes, compacts accepted pairs through
//                ballot/mbcnt into an LDS queue and evaluates them at full lane occupancy; per-atom sums leave
//                through a ds_bpermute butterfly.
#pragma once
#include "device_common.h"

struct EamArgs {
   const double* __restrict__ rx; const double* __restrict__ ry; const double* __restrict__ rz;
   double* __restrict__ fx; double* __restrict__ fy; double* __restrict__ fz; double* __restrict__ e;
   double* __restrict__ rhobar; double* __restrict__ dfEmbed;
   const int* __restrict__ nAtoms;
   const int* __restrict__ nbr;
   const int* __restrict__ cells;
   int nCells, cap;
   double rc2;
   InterpolationObjectGpu phi, rho, f;
};

// ---------------------------------------------------------------------------------------------------
// thread per slot; grid ceil(nCells*cap/256) x 256
template <int STEP>
__global__ __launch_bounds__(256)
void EAM_Force_thread_atom(EamArgs a)
{
   const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const int ci = (int)(tid / a.cap);
   const int ia = (int)(tid - (long)ci * a.cap);
   if (ci >= a.nCells) return;
   const int iBox = a.cells ? a.cells[ci] : ci;
   if (ia >= a.nAtoms[iBox]) return;
   const size_t iOff = (size_t)iBox * a.cap + ia;

   const TableView phiT = makeTable(a.phi, a.phi.values), rhoT = makeTable(a.rho, a.rho.values);
   const double xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
   double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0, rb = 0.0, dfi = 0.0;
   if (STEP == 3) { fx = a.fx[iOff]; fy = a.fy[iOff]; fz = a.fz[iOff]; dfi = a.dfEmbed[iOff]; }

   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
   for (int k = 0; k < 27; ++k) {
      const int jBox = nb[k];
      const int nj = a.nAtoms[jBox];
      const size_t base = (size_t)jBox * a.cap;
      for (int j = 0; j < nj; ++j) {
         double dx = xi - a.rx[base + j], dy = yi - a.ry[base + j], dz = zi - a.rz[base + j];
         double r2 = dx*dx + dy*dy + dz*dz;
         if (r2 <= a.rc2 && r2 > 0.0) {
            double ir = rsqrt64(r2), r = r2 * ir;
            double rho, drho, dphi;
            interpolate(rhoT, r, rho, drho);
            if (STEP == 1) { double phi; interpolate(phiT, r, phi, dphi); e += phi; rb += rho; }
            else           { dphi = (dfi + a.dfEmbed[base + j]) * drho; }
            dphi *= ir;
            fx -= dphi * dx; fy -= dphi * dy; fz -= dphi * dz;
         }
      }
   }
   a.fx[iOff] = fx; a.fy[iOff] = fy; a.fz[iOff] = fz;
   if (STEP == 1) { a.e[iOff] = 0.5 * e; a.rhobar[iOff] = rb; }
}

// pass 2, shared by both methods: embedding energy and its derivative
__global__ __launch_bounds__(256)
void EAM_Force_embed(EamArgs a)
{
   const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const int ci = (int)(tid / a.cap);
   const int ia = (int)(tid - (long)ci * a.cap);
   if (ci >= a.nCells) return;
   const int iBox = a.cells ? a.cells[ci] : ci;
   if (ia >= a.nAtoms[iBox]) return;
   const size_t iOff = (size_t)iBox * a.cap + ia;
   const TableView fT = makeTable(a.f, a.f.values);
   double F, dF;
   interpolate(fT, a.rhobar[iOff], F, dF);
   a.dfEmbed[iOff] = dF;
   a.e[iOff] += F;
}

// ---------------------------------------------------------------------------------------------------
// wave per cell, tables in LDS.
#define EAM_CTA_THREADS 256
#define EAM_CTA_WAVES   4
#define EAM_CTA_MAXCAND 512          // candidates (atoms in the 27 stencil cells) one wave can stage
#define EAM_CTA_QUEUE   128

template <int STEP>
__global__ __launch_bounds__(EAM_CTA_THREADS)
void EAM_Force_cta_cell(EamArgs a, int* __restrict__ status)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   // layout: rho table | phi table (pass 1) | per wave: x,y,z[,df] candidates, queue, offsets
   const int nRhoPad = a.rho.n + 3, nPhiPad = (STEP == 1) ? a.phi.n + 3 : 0;
   double* sRho = (double*)ldsRaw;
   double* sPhi = sRho + nRhoPad;
   double* waveBase = sPhi + nPhiPad;
   const int perWaveDoubles = (STEP == 1 ? 3 : 4) * EAM_CTA_MAXCAND;
   const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
   double* sx = waveBase + (size_t)wave * (perWaveDoubles + (EAM_CTA_QUEUE * 2 + 32 * 4) / 8);
   double* sy = sx + EAM_CTA_MAXCAND;
   double* sz = sy + EAM_CTA_MAXCAND;
   double* sdf = sz + EAM_CTA_MAXCAND;                       // pass 3 only
   unsigned short* q = (unsigned short*)(sx + perWaveDoubles);
   int* sOff = (int*)(q + EAM_CTA_QUEUE);                    // [28]... stored as 32 ints

   for (int t = threadIdx.x; t < nRhoPad; t += EAM_CTA_THREADS) sRho[t] = a.rho.values[t];
   if (STEP == 1) for (int t = threadIdx.x; t < nPhiPad; t += EAM_CTA_THREADS) sPhi[t] = a.phi.values[t];
   __syncthreads();
   const TableView rhoT = makeTable(a.rho, sRho), phiT = makeTable(a.phi, sPhi);

   for (int ci = blockIdx.x * EAM_CTA_WAVES + wave; ci < a.nCells; ci += gridDim.x * EAM_CTA_WAVES) {
      const int iBox = uniform(a.cells ? a.cells[ci] : ci);
      const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
      // candidate offsets of the 27 stencil cells (own cell first)
      int myBox = lane < 27 ? nb[lane] : 0;
      int cnt = lane < 27 ? a.nAtoms[myBox] : 0;
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
      if (lane < 27) sOff[lane] = incl - cnt;
      if (lane == 26) sOff[27] = incl;
      __builtin_amdgcn_wave_barrier();
      const int nCand = uniform(sOff[27]);
      const int ni = uniform(sOff[1]);                       // own cell's count
      if (nCand > EAM_CTA_MAXCAND) { if (lane == 0) atomicOr(&status[0], 2); continue; }
      for (int k = 0; k < 27; ++k) {
         const int jBox = uniform(nb[k]);
         const int off = uniform(sOff[k]), nj = uniform(sOff[k + 1]) - off;
         for (int j = lane; j < nj; j += 64) {
            size_t o = (size_t)jBox * a.cap + j;
            sx[off + j] = a.rx[o]; sy[off + j] = a.ry[o]; sz[off + j] = a.rz[o];
            if (STEP == 3) sdf[off + j] = a.dfEmbed[o];
         }
      }
      __builtin_amdgcn_wave_barrier();

      for (int i = 0; i < ni; ++i) {
         const double xi = sx[i], yi = sy[i], zi = sz[i];
         const double dfi = (STEP == 3) ? sdf[i] : 0.0;
         double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0, rb = 0.0;
         int qn = 0;
         auto evalPair = [&](int jj) {
            double dx = xi - sx[jj], dy = yi - sy[jj], dz = zi - sz[jj];
            double r2 = dx*dx + dy*dy + dz*dz;
            double ir = rsqrt64(r2), r = r2 * ir;
            double rho, drho, dphi;
            interpolate(rhoT, r, rho, drho);
            if (STEP == 1) { double phi; interpolate(phiT, r, phi, dphi); e += phi; rb += rho; }
            else           { dphi = (dfi + sdf[jj]) * drho; }
            dphi *= ir;
            fx -= dphi * dx; fy -= dphi * dy; fz -= dphi * dz;
         };
         for (int j0 = 0; j0 < nCand; j0 += 64) {
            const int j = j0 + lane;
            bool hit = false;
            if (j < nCand) {
               double dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
               double r2 = dx*dx + dy*dy + dz*dz;
               hit = (r2 <= a.rc2) && (r2 > 0.0);
            }
            const unsigned long long m = __ballot(hit);
            if (hit) {
               int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
               q[pos] = (unsigned short)j;
            }
            qn += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (qn >= 64) {
               evalPair(q[lane]);
               qn -= 64;
               const unsigned short carry = q[64 + lane];
               __builtin_amdgcn_wave_barrier();
               if (lane < qn) q[lane] = carry;
               __builtin_amdgcn_wave_barrier();
            }
         }
         if (lane < qn) evalPair(q[lane]);
         __builtin_amdgcn_wave_barrier();
         fx = waveSum(fx); fy = waveSum(fy); fz = waveSum(fz);
         if (STEP == 1) { e = waveSum(e); rb = waveSum(rb); }
         if (lane == 0) {
            const size_t io = (size_t)iBox * a.cap + i;
            if (STEP == 1) { a.fx[io] = fx; a.fy[io] = fy; a.fz[io] = fz; a.e[io] = 0.5 * e; a.rhobar[io] = rb; }
            else           { a.fx[io] += fx; a.fy[io] += fy; a.fz[io] += fz; }
         }
      }
      __builtin_amdgcn_wave_barrier();
   }
}

static inline size_t eamCtaLdsBytes(int step, int nRho, int nPhi)
{
   size_t tables = (size_t)(nRho + 3 + (step == 1 ? nPhi + 3 : 0)) * 8;
   size_t perWave = (size_t)(step == 1 ? 3 : 4) * EAM_CTA_MAXCAND * 8 + EAM_CTA_QUEUE * 2 + 32 * 4;
   return tables + EAM_CTA_WAVES * perWave;
}
