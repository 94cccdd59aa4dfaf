// lj_kernels.h -- Lennard-Jones force kernels for gfx950.
//
// Same physics as the reference's LJ_Force_thread_atom (gpu_lj_thread_atom.h:29-143) and
// LJ_Force_cta_cell (gpu_lj_cta_cell.h:29-122): full 27-cell stencil, 0 < r^2 <= rc^2,
// f_i = eps * sum r6*r2inv*(48 r6 - 24) * d,  e_i = 4 eps * sum 1/2 (r6 (r6 - 1) - eShift).
// Different machine mapping (see DESIGN.md "Kernels"):
//
//  thread_atom : one wave = 64 consecutive slots of ONE link cell, so the neighbour atom j is the same for
//                all lanes -> j positions come through the scalar unit (s_load into SGPRs), the vector unit
//                only does the distance test and the pair evaluation.  No LDS, no atom lists.
//  cta_cell    : one workgroup per link cell; the positions of all 27 stencil cells are staged, compacted,
//                in LDS once; each wave owns one i atom at a time and spreads the candidates over its 64
//                lanes; accepted pairs are compacted (ballot + mbcnt) into a per-wave LDS queue and evaluated
//                64 at a time at full lane occupancy; per-atom force/energy leave the wave through a
//                ds_bpermute butterfly.
#pragma once
#include "device_common.h"

struct LjArgs {
   const double* __restrict__ rx; const double* __restrict__ ry; const double* __restrict__ rz;
   double* __restrict__ fx; double* __restrict__ fy; double* __restrict__ fz; double* __restrict__ e;
   const int* __restrict__ nAtoms;
   const int* __restrict__ nbr;        // [nLocal*27], self first
   const int* __restrict__ cells;      // optional cell list
   int nCells, cap;
   double rc2, s6, eShift, eps;
};

// One accepted pair.  With u = s6 / r^6:  e_pair = u (u - 1) - eShift,  f_pair = 24 u (2u - 1) / r^2 * d.
// The constant factors (24 eps on the force, 4 eps * 1/2 on the energy) are applied once per atom by the caller.
// ENERGY = false drops the energy ops: e[] is only consumed by computeEnergy, i.e. by the last step of a timestep() call.
template <bool ENERGY>
__device__ __forceinline__ void ljPair(double dx, double dy, double dz, double r2, const LjArgs& a,
                                       double& fx, double& fy, double& fz, double& e)
{
   const double ir2 = rcp64(r2);
   const double u = a.s6 * ir2 * ir2 * ir2;
   if (ENERGY) e += __builtin_fma(u, u - 1.0, -a.eShift);
   const double fr = u * ir2 * __builtin_fma(u, 2.0, -1.0);
   fx = __builtin_fma(fr, dx, fx); fy = __builtin_fma(fr, dy, fy); fz = __builtin_fma(fr, dz, fz);
}

// one neighbour cell against the wave's 64 i atoms; SELF adds the r2 > 0 guard of the own cell
template <bool SELF, bool ENERGY>
__device__ __forceinline__ void ljCellLoop(const LjArgs& a, int jBox, double xi, double yi, double zi,
                                           double& fx, double& fy, double& fz, double& e)
{
   const int nj = uniform(a.nAtoms[jBox]);
   const double* __restrict__ px = a.rx + (size_t)jBox * a.cap;
   const double* __restrict__ py = a.ry + (size_t)jBox * a.cap;
   const double* __restrict__ pz = a.rz + (size_t)jBox * a.cap;
   // j is wave-uniform: fetch 8 neighbours per scalar-load batch (3 x s_load_dwordx16), then test them
   int j = 0;
   for (; j + 8 <= nj; j += 8) {
      double xs[8], ys[8], zs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { xs[u] = px[j + u]; ys[u] = py[j + u]; zs[u] = pz[j + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
         double dx = xi - xs[u], dy = yi - ys[u], dz = zi - zs[u];
         double r2 = dx*dx + dy*dy + dz*dz;
         bool hit = SELF ? (r2 <= a.rc2 && r2 > 0.0) : (r2 <= a.rc2);
         if (hit) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
      }
   }
   for (; j < nj; ++j) {
      double dx = xi - px[j], dy = yi - py[j], dz = zi - pz[j];
      double r2 = dx*dx + dy*dy + dz*dz;
      bool hit = SELF ? (r2 <= a.rc2 && r2 > 0.0) : (r2 <= a.rc2);
      if (hit) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
   }
}

// ---------------------------------------------------------------------------------------------------
// Generic chunk: the wave owns m <= 64 atoms (slots chunk*64 .. chunk*64+m-1 of iBox).  With m <= 32 the atoms are replicated
// G = 64/m (<= 4) times across the lanes and replica g takes the stencil cells k = g, g+G, g+2G, ..., so the wave finishes in
// ~1/G of the neighbour iterations; replicas are summed with ds_bpermute.  The neighbour is not wave-uniform here, so
// positions come through per-lane loads.  Used for the under-filled tail wave of a cell (20 of 64 lanes for 148 atoms)
// and for chunks beyond the host's occupancy estimate.
template <bool ENERGY>
__device__ __forceinline__ void ljChunkGeneric(const LjArgs& a, int iBox, int ni, int chunk, int lane)
{
   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
   const int m = ni - chunk * 64 < 64 ? ni - chunk * 64 : 64;
   const int G = 64 / m < 4 ? 64 / m : 4;
   const int g = lane / m, ai = lane - g * m;
   const bool valid = g < G;
   const size_t iOff = (size_t)iBox * a.cap + chunk * 64 + (valid ? ai : 0);
   const double xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
   double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
   for (int t = 0; t * G < 27; ++t) {
      const int k = t * G + g;
      const bool okk = valid && k < 27;
      const int jBox = okk ? nb[k] : iBox;
      const int nj = okk ? a.nAtoms[jBox] : 0;
      const size_t base = (size_t)jBox * a.cap;
      for (int j = 0; __any(j < nj); ++j) {
         if (j < nj) {
            const double dx = xi - a.rx[base + j], dy = yi - a.ry[base + j], dz = zi - a.rz[base + j];
            const double r2 = dx*dx + dy*dy + dz*dz;
            if (r2 <= a.rc2 && r2 > 0.0) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
         }
      }
   }
   double tx = fx, ty = fy, tz = fz, te = e;
   for (int r = 1; r < G; ++r) {                       // all lanes take part; only lanes < m keep the result
      const int src = (ai + r * m) & 63;
      tx += bpermute64(fx, src); ty += bpermute64(fy, src); tz += bpermute64(fz, src);
      if (ENERGY) te += bpermute64(e, src);
   }
   if (lane < m) {
      const double fs = 24.0 * a.eps;
      a.fx[iOff] = tx * fs; a.fy[iOff] = ty * fs; a.fz[iOff] = tz * fs;
      if (ENERGY) a.e[iOff] = te * 2.0 * a.eps;
   }
}

// ---------------------------------------------------------------------------------------------------
// thread per atom, wave per 64-slot chunk of a cell.  Requires cap % 64 == 0.
// grid: one workgroup of `wavesPerCell` waves per cell, where wavesPerCell = ceil((largest occupancy + slack) / 64) as
// last seen by the host (SimGpu.max_atoms_cell) -- 3 waves for 5-sigma LJ Cu instead of cap/64 = 4, so no wave is born dead.
// A cell that outgrew that estimate is still complete: its waves take the extra chunks through the generic path.
template <bool ENERGY>
__global__ __launch_bounds__(256)
void LJ_Force_thread_atom(LjArgs a, int wavesPerCell)
{
   const int lane = threadIdx.x & 63;
   // one workgroup per cell when wavesPerCell <= 4 (blockDim = 64 * wavesPerCell); otherwise 4-wave workgroups laid flat over (cell, chunk)
   const int gw = uniform(xcdRemap(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
   const int ci = gw / wavesPerCell;
   const int chunk = gw - ci * wavesPerCell;
   if (ci >= a.nCells) return;
   const int iBox = uniform(a.cells ? a.cells[ci] : ci);
   const int ni = uniform(a.nAtoms[iBox]);
   if (chunk * 64 >= ni) return;
   const int m = uniform(ni - chunk * 64 < 64 ? ni - chunk * 64 : 64);     // atoms this wave owns

   if (m <= 32) {
      ljChunkGeneric<ENERGY>(a, iBox, ni, chunk, lane);
   } else {
      // full wave: neighbour j is wave-uniform -> positions arrive through the scalar unit
      const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
      const int iSlot = chunk * 64 + lane;
      const bool active = iSlot < ni;
      const size_t iOff = (size_t)iBox * a.cap + (active ? iSlot : ni - 1);   // idle lanes shadow the last atom
      const double xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
      double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
      // (a software-pipelined variant -- scalar loads of batch b+1 issued before batch b is evaluated, 4 neighbours per batch to fit
      // two batches in SGPRs -- measured 9 % slower: 4.30 vs 3.95 ms; the 8-wide batches below rely on the other waves for latency cover)
      ljCellLoop<true, ENERGY>(a, iBox, xi, yi, zi, fx, fy, fz, e);
      for (int k = 1; k < 27; ++k) ljCellLoop<false, ENERGY>(a, uniform(nb[k]), xi, yi, zi, fx, fy, fz, e);
      if (active) {
         const double fs = 24.0 * a.eps;
         a.fx[iOff] = fx * fs; a.fy[iOff] = fy * fs; a.fz[iOff] = fz * fs;
         if (ENERGY) a.e[iOff] = e * 2.0 * a.eps;          // 4 eps * 1/2 per pair
      }
   }
   // NOTE: no store may precede the scalar-path loads above on any path through this kernel, or the compiler gives up proving the
   // position arrays unclobbered and silently replaces the s_load_dwordx16 stream by per-lane global_load (measured: 4.7 -> 6.2 ms).
   // Everything below runs after the wave's first chunk is stored.  Check `grep -c s_load_dwordx16` in the ISA after edits here.
   for (int c = chunk + wavesPerCell; c * 64 < ni; c += wavesPerCell) ljChunkGeneric<ENERGY>(a, iBox, ni, c, lane);
}

// ---------------------------------------------------------------------------------------------------
// workgroup per cell, 512 threads = 8 waves.  Dynamic LDS: 3 * LJ_CTA_MAXCAND doubles of candidate
// positions + one 128-entry pair queue per wave.
#define LJ_CTA_THREADS 512
#define LJ_CTA_WAVES   (LJ_CTA_THREADS / 64)
#define LJ_CTA_MAXCAND 6400
#define LJ_CTA_QUEUE   128
#define LJ_CTA_LDS_BYTES (3 * LJ_CTA_MAXCAND * 8 + LJ_CTA_WAVES * LJ_CTA_QUEUE * 2 + 32 * 4)

__global__ __launch_bounds__(LJ_CTA_THREADS)
void LJ_Force_cta_cell(LjArgs a, int* __restrict__ status)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   double* sx = (double*)ldsRaw;
   double* sy = sx + LJ_CTA_MAXCAND;
   double* sz = sy + LJ_CTA_MAXCAND;
   unsigned short* qAll = (unsigned short*)(sz + LJ_CTA_MAXCAND);
   int* sOff = (int*)(qAll + LJ_CTA_WAVES * LJ_CTA_QUEUE);          // [28] candidate offsets per stencil cell

   const int lane = threadIdx.x & 63;
   const int wave = uniform(threadIdx.x >> 6);
   unsigned short* q = qAll + wave * LJ_CTA_QUEUE;

   const int ci = xcdRemap(blockIdx.x, gridDim.x);
   const int iBox = a.cells ? a.cells[ci] : ci;
   const int ni = a.nAtoms[iBox];
   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;

   // stage: offsets (one wave), then positions of every stencil cell, compacted, own cell first
   if (threadIdx.x < 64) {
      int cnt = lane < 27 ? a.nAtoms[nb[lane]] : 0;
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
      if (lane < 27) sOff[lane] = incl - cnt;
      if (lane == 26) sOff[27] = incl;
   }
   __syncthreads();
   const int nCand = sOff[27];
   if (nCand > LJ_CTA_MAXCAND) {            // cannot happen for cap*27 <= MAXCAND; flag instead of corrupting LDS
      if (threadIdx.x == 0) atomicOr(&status[0], 2);
      return;
   }
   for (int k = 0; k < 27; ++k) {
      const int jBox = nb[k];
      const int off = sOff[k], nj = sOff[k + 1] - off;
      for (int j = threadIdx.x; j < nj; j += LJ_CTA_THREADS) {
         size_t o = (size_t)jBox * a.cap + j;
         sx[off + j] = a.rx[o]; sy[off + j] = a.ry[o]; sz[off + j] = a.rz[o];
      }
   }
   __syncthreads();

   for (int i = wave; i < ni; i += LJ_CTA_WAVES) {          // wave-uniform i; own cell occupies candidates [0, ni)
      const double xi = sx[i], yi = sy[i], zi = sz[i];
      double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
      int qn = 0;                                           // wave-uniform queue fill
      for (int j0 = 0; j0 < nCand; j0 += 64) {
         const int j = j0 + lane;
         bool hit = false;
         if (j < nCand) {
            double dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
            double r2 = dx*dx + dy*dy + dz*dz;
            hit = (r2 <= a.rc2) && (r2 > 0.0);              // same guard as the reference (no self pair, no divide by zero)
         }
         const unsigned long long m = __ballot(hit);
         if (hit) {
            int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            q[pos] = (unsigned short)j;
         }
         qn += __popcll(m);
         __builtin_amdgcn_wave_barrier();
         if (qn >= 64) {                                    // evaluate one full batch at 64/64 lanes
            const int jj = q[lane];
            double dx = xi - sx[jj], dy = yi - sy[jj], dz = zi - sz[jj];
            ljPair<true>(dx, dy, dz, dx*dx + dy*dy + dz*dz, a, fx, fy, fz, e);
            qn -= 64;
            const unsigned short carry = q[64 + lane];      // move the overflow to the front
            __builtin_amdgcn_wave_barrier();
            if (lane < qn) q[lane] = carry;
            __builtin_amdgcn_wave_barrier();
         }
      }
      if (lane < qn) {                                      // tail batch
         const int jj = q[lane];
         double dx = xi - sx[jj], dy = yi - sy[jj], dz = zi - sz[jj];
         ljPair<true>(dx, dy, dz, dx*dx + dy*dy + dz*dz, a, fx, fy, fz, e);
      }
      __builtin_amdgcn_wave_barrier();
      fx = waveSum(fx); fy = waveSum(fy); fz = waveSum(fz); e = waveSum(e);
      if (lane == 0) {
         const size_t io = (size_t)iBox * a.cap + i;
         const double fs = 24.0 * a.eps;
         a.fx[io] = fx * fs; a.fy[io] = fy * fs; a.fz[io] = fz * fs;
         a.e[io] = e * 2.0 * a.eps;
      }
   }
}
