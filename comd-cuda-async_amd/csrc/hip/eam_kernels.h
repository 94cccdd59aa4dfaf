// eam_kernels.h -- EAM force kernels (three passes) for gfx950.
//
// Physics of the reference's EAM_Force_thread_atom<step> / EAM_Force_cta_cell<step>
// (gpu_eam_thread_atom.h:32-140, gpu_eam_cta_cell.h:34-278) and the pass-2 kernels
// (gpu_eam_thread_atom.h:269-287, gpu_eam_cta_cell.h:280-295):
//   pass 1: e_i = 1/2 sum phi(r), rhobar_i = sum rho(r), f_i -= phi'(r) d/r          (0 < r^2 <= rc^2)
//   pass 2: dfEmbed_i = F'(rhobar_i), e_i += F(rhobar_i)
//   pass 3: f_i -= (F'_i + F'_j) rho'(r) d/r
// Tables: the reference's default quadratic interpolation in r (gpu_common.h:48-86), or with SPLINE its cubic splines in r^2
// (-P, gpu_common.h:95-129: value and (1/r) d/dr straight from r^2, read through L2); F(rhobar) is quadratic in both modes.
//
//  thread_atom : one thread per cell slot (cell*cap + i), tables read through L1/L2.
//  cta_cell    : persistent workgroups of 4 waves, each wave owns one cell at a time.  phi/rho tables live in LDS for the
//                workgroup's lifetime; the wave stages the positions of its 27 stencil cells, compacted, in LDS, spreads
//                the candidates over its lanes, compacts accepted pairs through v_cmp/mbcnt into an LDS queue and evaluates
//                them at full lane occupancy; the sums of two atoms at a time leave through v_permlane32_swap + DPP.
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
   InterpolationSplineObjectGpu phiS, rhoS;
};

// ---------------------------------------------------------------------------------------------------
// thread per slot; grid ceil(nCells*cap/256) x 256
template <int STEP, bool SPLINE>
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
            double rho, drho, dphi;
            if (SPLINE) {                                  // drho, dphi are (1/r) d/dr already
               interpolateSpline(a.rhoS, r2, rho, drho);
               if (STEP == 1) { double phi; interpolateSpline(a.phiS, r2, phi, dphi); e += phi; rb += rho; }
               else           { dphi = (dfi + a.dfEmbed[base + j]) * drho; }
            } else {
               double ir = rsqrt64(r2), r = r2 * ir;
               interpolate(rhoT, r, rho, drho);
               if (STEP == 1) { double phi; interpolate(phiT, r, phi, dphi); e += phi; rb += rho; }
               else           { dphi = (dfi + a.dfEmbed[base + j]) * drho; }
               dphi *= ir;
            }
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
//
// Per cell the wave (1) prefix-sums the occupancies of the 27 stencil cells, (2) stages their positions (and F' in
// pass 3), compacted, into its LDS slice -- all global loads of a group of rounds are issued before the first is consumed,
// so the HBM/L2 latency is paid a few times per cell, not 27 times -- (4) for each i atom: 3 LDS reads + 7 VALU ops per
// candidate tile of 64, v_cmp mask -> mbcnt
// compaction into a 128-entry LDS queue, evaluation of the accepted pairs at full lane occupancy, (5) reduces the per-lane
// partial sums of TWO atoms at a time: v_permlane32_swap puts one atom in each half of the wave, DPP row ops finish (no LDS).
#define EAM_CTA_THREADS 256
#define EAM_CTA_WAVES   4
#define EAM_CTA_MAXCAND 384                        // stencil atoms a wave can stage (FCC Cu, Cu_u6 cutoff, 80^3: 256..365, mean 283); 3 workgroups per CU
#define EAM_CTA_MAXCAND_WIDE 640                   // longer cutoffs (Mishin Cu01: 5.51 A, mean 393 per stencil); tables then stay in L2
#define EAM_CTA_QUEUE   128

// A cell whose 27-cell stencil holds more than EAM_CTA_MAXCAND atoms (small boxes have larger cells) is handled by the
// same wave in the thread-per-atom form: lane = i atom, neighbours streamed from global memory, same tables.
template <int STEP, bool SPLINE>
__device__ __forceinline__ void eamCellDirect(const EamArgs& a, int iBox, int lane, const TableView& rhoT, const TableView& phiT, bool sameGrid)
{
   const int ni = a.nAtoms[iBox];
   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
   for (int i = lane; i < ni; i += 64) {
      const size_t iOff = (size_t)iBox * a.cap + i;
      const double xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
      double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0, rb = 0.0, dfi = 0.0;
      if (STEP == 3) { fx = a.fx[iOff]; fy = a.fy[iOff]; fz = a.fz[iOff]; dfi = a.dfEmbed[iOff]; }
      for (int k = 0; k < 27; ++k) {
         const int jBox = nb[k];
         const int nj = a.nAtoms[jBox];
         const size_t base = (size_t)jBox * a.cap;
         for (int j = 0; j < nj; ++j) {
            const double dx = xi - a.rx[base + j], dy = yi - a.ry[base + j], dz = zi - a.rz[base + j];
            const double r2 = dx*dx + dy*dy + dz*dz;
            if (r2 <= a.rc2 && r2 > 0.0) {
               double rho, drho, dphi;
               if (SPLINE) {
                  interpolateSpline(a.rhoS, r2, rho, drho);
                  if (STEP == 1) { double phi; interpolateSpline(a.phiS, r2, phi, dphi); e += phi; rb += rho; }
                  else           { dphi = (dfi + a.dfEmbed[base + j]) * drho; }
               } else {
                  const double ir = rsqrt64(r2), r = r2 * ir;
                  if (STEP == 1) {
                     double phi;
                     if (sameGrid) interpolatePair(rhoT.v, rhoT, r, phi, dphi, rho, drho);
                     else { interpolate(rhoT, r, rho, drho); interpolate(phiT, r, phi, dphi); }
                     e += phi; rb += rho;
                  } else {
                     interpolate(rhoT, r, rho, drho);
                     dphi = (dfi + a.dfEmbed[base + j]) * drho;
                  }
                  dphi *= ir;
               }
               fx -= dphi * dx; fy -= dphi * dy; fz -= dphi * dz;
            }
         }
      }
      a.fx[iOff] = fx; a.fy[iOff] = fy; a.fz[iOff] = fz;
      if (STEP == 1) { a.e[iOff] = 0.5 * e; a.rhobar[iOff] = rb; }
   }
}

template <int STEP, int MAXCAND, bool LDS_TABLES, bool SPLINE>
__global__ __launch_bounds__(EAM_CTA_THREADS)
void EAM_Force_cta_cell_pairs(EamArgs a, int* __restrict__ status)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   constexpr int NV = (STEP == 1) ? 5 : 3;                   // values reduced per atom: f (3) [+ e, rhobar]
   constexpr int NC = 3;                                     // doubles staged per candidate: r; pass 3 also keeps its global slot (int)
   // tables too large for the LDS (setfl files: 10000 samples each) are read through L2 instead
   const int nRhoPad = a.rho.n + 3, nPhiPad = (STEP == 1) ? a.phi.n + 3 : 0;
   double* sRho = (double*)ldsRaw;
   double* sPhi = sRho + (LDS_TABLES ? nRhoPad : 0);
   double* waveBase = sPhi + (LDS_TABLES ? nPhiPad : 0);
   constexpr int perWaveDoubles = NC * MAXCAND + (STEP == 3 ? MAXCAND / 2 : 0) + (2 * EAM_CTA_QUEUE * 2 + 64 * 4) / 8;
   const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
   double* sx = waveBase + (size_t)wave * perWaveDoubles;
   double* sy = sx + MAXCAND;
   double* sz = sy + MAXCAND;
   int* sSlot = (int*)(sz + MAXCAND);               // pass 3 only: global slot of each candidate, to fetch F'_j for accepted pairs
   unsigned short* qBase = (unsigned short*)(sx + NC * MAXCAND + (STEP == 3 ? MAXCAND / 2 : 0));   // one pair queue per atom of a group
   int* sOff = (int*)(qBase + 2 * EAM_CTA_QUEUE);           // [32] exclusive candidate offsets of the stencil cells
   int* sBox = sOff + 32;                                    // [32] their cell ids

   // pass 1 on a shared r grid (funcfl): one interleaved {phi, rho} table; otherwise two separate tables
   const bool sameGrid = (STEP == 1) && LDS_TABLES && a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   if (!LDS_TABLES) {
   } else if (sameGrid) {
      for (int t = threadIdx.x; t < nRhoPad; t += EAM_CTA_THREADS) { sRho[2 * t] = a.phi.values[t]; sRho[2 * t + 1] = a.rho.values[t]; }
   } else {
      for (int t = threadIdx.x; t < nRhoPad; t += EAM_CTA_THREADS) sRho[t] = a.rho.values[t];
      if (STEP == 1) for (int t = threadIdx.x; t < nPhiPad; t += EAM_CTA_THREADS) sPhi[t] = a.phi.values[t];
   }
   __syncthreads();
   const TableView rhoT = makeTable(a.rho, LDS_TABLES ? sRho : a.rho.values), phiT = makeTable(a.phi, LDS_TABLES ? sPhi : a.phi.values);

   // cap is a power of two <= 64 for EAM (chooseMaxAtoms): `cellsPerRound` stencil cells are staged per round of 64 lanes
   const int capShift = 31 - __builtin_clz(a.cap);
   const int cellsPerRound = 64 >> capShift;
   const int myK = lane >> capShift, myJ = lane & (a.cap - 1);

   // XCD-aware persistent walk.  Workgroup b runs on XCD b % 8 (round-robin dispatch); give XCD x the contiguous cell range
   // [x*N/8, (x+1)*N/8) and let its workgroups sweep that range side by side, 4 consecutive cells (one per wave) each,
   // so the stencil planes they share stay in that XCD's 4 MiB L2 (before: 14-18x re-fetch of the positions, rocprof FETCH_SIZE).
   const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = (gridDim.x + 7 - xcd) >> 3;
   const int cellLo = (int)((long)a.nCells * xcd / 8), cellHi = (int)((long)a.nCells * (xcd + 1) / 8);
   int overrun = 0;
   for (int ci = cellLo + lb * EAM_CTA_WAVES + wave; ci < cellHi; ci += nlb * EAM_CTA_WAVES) {
      const int iBox = uniform(a.cells ? a.cells[ci] : ci);
      const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
      // (1) candidate offsets, own cell first
      {
         const int box = lane < 27 ? nb[lane] : 0;
         const int cnt = lane < 27 ? a.nAtoms[box] : 0;
         int incl = cnt;
#pragma unroll
         for (int d = 1; d < 32; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
         if (lane < 28) { sOff[lane] = incl - cnt; sBox[lane] = box; }     // lane 27: cnt = 0 -> sOff[27] = total
      }
      __builtin_amdgcn_wave_barrier();
      const int nCand = uniform(sOff[27]);
      const int ni = uniform(sOff[1]);
      if (nCand > MAXCAND) { eamCellDirect<STEP, SPLINE>(a, iBox, lane, rhoT, phiT, sameGrid); continue; }

      // (2) stage positions [and F'] of the stencil cells, in groups of GROUP rounds with all loads in flight together
      constexpr int GROUP = 4;                      // 4 rounds x (3-4 loads) in flight; 7 rounds cost 60 more VGPRs and a wave per SIMD
      for (int k0 = 0; k0 < 27; k0 += GROUP * cellsPerRound) {
         double vx[GROUP], vy[GROUP], vz[GROUP];
         int dst[GROUP], src[GROUP];
#pragma unroll
         for (int g = 0; g < GROUP; ++g) {
            const int k = k0 + g * cellsPerRound + myK;
            dst[g] = -1;
            if (k < 27) {
               const int off = sOff[k], n = sOff[k + 1] - off;
               if (myJ < n) {
                  const size_t o = (size_t)sBox[k] * a.cap + myJ;
                  vx[g] = a.rx[o]; vy[g] = a.ry[o]; vz[g] = a.rz[o];
                  src[g] = (int)o;
                  dst[g] = off + myJ;
               }
            }
         }
#pragma unroll
         for (int g = 0; g < GROUP; ++g)
            if (dst[g] >= 0) {
               sx[dst[g]] = vx[g]; sy[dst[g]] = vy[g]; sz[dst[g]] = vz[g];
               if (STEP == 3) sSlot[dst[g]] = src[g];
            }
      }
      __builtin_amdgcn_wave_barrier();

      const int nTiles = (nCand + 63) >> 6;

      // (4)+(5) two i atoms per round, side by side: one read of each candidate serves both distance tests, and the two accepted-pair
      // evaluations are one branch-free block (a lane without a pair evaluates r = cutoff, weighted 0), i.e. two independent chains of
      // LDS table reads and fp64 arithmetic per lane
      for (int i0 = 0; i0 < ni; i0 += 2) {
         double part[2][NV];
#pragma unroll
         for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v) part[u][v] = 0.0;
         const bool twoAtoms = i0 + 1 < ni;                          // wave-uniform
         const int iA = i0, iB = twoAtoms ? i0 + 1 : i0;
         const double xA = sx[iA], yA = sy[iA], zA = sz[iA];
         const double xB = sx[iB], yB = sy[iB], zB = sz[iB];
         const double dfA = (STEP == 3) ? a.dfEmbed[(size_t)iBox * a.cap + iA] : 0.0;
         const double dfB = (STEP == 3) ? a.dfEmbed[(size_t)iBox * a.cap + iB] : 0.0;
         unsigned short* qA = qBase;
         unsigned short* qB = qBase + EAM_CTA_QUEUE;
         int qnA = 0, qnB = 0;
         for (int t = 0; t < nTiles; ++t) {
            const int c = t * 64 + lane;
            const int cc = c < nCand ? c : 0;                        // lanes past the list re-read candidate 0 and are masked
            const double px = sx[cc], py = sy[cc], pz = sz[cc];
            const double ax = xA - px, ay = yA - py, az = zA - pz;
            const double bx = xB - px, by = yB - py, bz = zB - pz;
            const double r2A = ax*ax + ay*ay + az*az, r2B = bx*bx + by*by + bz*bz;
            const bool hitA = (r2A <= a.rc2) && (r2A > 0.0) && (c < nCand);
            const bool hitB = twoAtoms && (r2B <= a.rc2) && (r2B > 0.0) && (c < nCand);
            const unsigned long long mA = __ballot(hitA), mB = __ballot(hitB);
            // a store only while the index is inside the queue: an atom with more in-cutoff neighbours than the queue holds loses the excess
            // (flagged below) instead of overwriting the other queue and the wave's stencil table
            const int kA = qnA + __builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u));
            const int kB = qnB + __builtin_amdgcn_mbcnt_hi((unsigned)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB, 0u));
            if (hitA && kA < EAM_CTA_QUEUE) qA[kA] = (unsigned short)c;
            if (hitB && kB < EAM_CTA_QUEUE) qB[kB] = (unsigned short)c;
            qnA += __popcll(mA); qnB += __popcll(mB);
         }
         // More neighbours inside the cutoff than a queue holds (3x FCC Cu): the excess was dropped above;
         // the results are void, the flag makes comdCheckStatus stop the run.
         // (reported once, after the cell walk: a store in here costs 4 % -- the loads behind it lose their freedom to move)
         if (qnA > EAM_CTA_QUEUE) { overrun = 1; qnA = EAM_CTA_QUEUE; }
         if (qnB > EAM_CTA_QUEUE) { overrun = 1; qnB = EAM_CTA_QUEUE; }
         // accepted pairs, 64 per batch and atom (FCC Cu: 42 neighbours inside the Cu_u6 cutoff, 54 inside Mishin's -> one batch)
         const int qnMax = qnA > qnB ? qnA : qnB;
         for (int b = 0; b < qnMax; b += 64) {
            const bool hA = b + lane < qnA, hB = b + lane < qnB;
            const int jA = hA ? qA[b + lane] : iA, jB = hB ? qB[b + lane] : iB;      // no pair: the atom itself (r2 = 0), replaced by the cutoff below
            const double ax = xA - sx[jA], ay = yA - sy[jA], az = zA - sz[jA];
            const double bx = xB - sx[jB], by = yB - sy[jB], bz = zB - sz[jB];
            const double sA = hA ? ax*ax + ay*ay + az*az : a.rc2, sB = hB ? bx*bx + by*by + bz*bz : a.rc2;
            double rhoA, drhoA, dphiA, rhoB, drhoB, dphiB;
            if (SPLINE) {                                   // cubic splines in r^2: (1/r) d/dr directly, no square root
               interpolateSpline(a.rhoS, sA, rhoA, drhoA); interpolateSpline(a.rhoS, sB, rhoB, drhoB);
               if (STEP == 1) {
                  double phiA, phiB;
                  interpolateSpline(a.phiS, sA, phiA, dphiA); interpolateSpline(a.phiS, sB, phiB, dphiB);
                  part[0][3] += hA ? phiA : 0.0; part[0][4] += hA ? rhoA : 0.0;
                  part[1][3] += hB ? phiB : 0.0; part[1][4] += hB ? rhoB : 0.0;
               } else {
                  dphiA = (dfA + a.dfEmbed[sSlot[jA]]) * drhoA; dphiB = (dfB + a.dfEmbed[sSlot[jB]]) * drhoB;
               }
               dphiA = hA ? dphiA : 0.0; dphiB = hB ? dphiB : 0.0;
            } else {
               const double irA = rsqrt64(sA), irB = rsqrt64(sB);
               const double rA = sA * irA, rB = sB * irB;
               if (STEP == 1) {
                  double phiA, phiB;
                  if (sameGrid) { interpolatePair(sRho, rhoT, rA, phiA, dphiA, rhoA, drhoA); interpolatePair(sRho, rhoT, rB, phiB, dphiB, rhoB, drhoB); }
                  else { interpolate(rhoT, rA, rhoA, drhoA); interpolate(phiT, rA, phiA, dphiA); interpolate(rhoT, rB, rhoB, drhoB); interpolate(phiT, rB, phiB, dphiB); }
                  part[0][3] += hA ? phiA : 0.0; part[0][4] += hA ? rhoA : 0.0;
                  part[1][3] += hB ? phiB : 0.0; part[1][4] += hB ? rhoB : 0.0;
               } else {
                  interpolate(rhoT, rA, rhoA, drhoA); interpolate(rhoT, rB, rhoB, drhoB);
                  dphiA = (dfA + a.dfEmbed[sSlot[jA]]) * drhoA; dphiB = (dfB + a.dfEmbed[sSlot[jB]]) * drhoB;
               }
               dphiA = hA ? dphiA * irA : 0.0; dphiB = hB ? dphiB * irB : 0.0;
            }
            part[0][0] -= dphiA * ax; part[0][1] -= dphiA * ay; part[0][2] -= dphiA * az;
            part[1][0] -= dphiB * bx; part[1][1] -= dphiB * by; part[1][2] -= dphiB * bz;
         }
         // two-atom reduction (permlane32 swap + DPP): lane 31 holds the totals of atom i0, lane 63 those of atom i0 + 1
         double tot[NV];
#pragma unroll
         for (int v = 0; v < NV; ++v) tot[v] = pairSum(part[0][v], part[1][v]);
         if ((lane & 31) == 31) {
            const int i = i0 + (lane >> 5);
            if (i < ni) {
               const size_t io = (size_t)iBox * a.cap + i;
               if (STEP == 1) { a.fx[io] = tot[0]; a.fy[io] = tot[1]; a.fz[io] = tot[2]; a.e[io] = 0.5 * tot[3]; a.rhobar[io] = tot[4]; }
               else           { a.fx[io] += tot[0]; a.fy[io] += tot[1]; a.fz[io] += tot[2]; }
            }
         }
      }
      __builtin_amdgcn_wave_barrier();
   }
   if (overrun && lane == 0) atomicOr(&status[3], 1);
}

static inline size_t eamCtaTableBytes(int step, int nRho, int nPhi) { return (size_t)(nRho + 3 + (step == 1 ? nPhi + 3 : 0)) * 8; }

static inline size_t eamCtaLdsBytes(int step, int maxCand, size_t tableBytes)
{
   size_t perWave = (size_t)3 * maxCand * 8 + (step == 3 ? maxCand * 4 : 0) + 2 * EAM_CTA_QUEUE * 2 + 64 * 4;
   return tableBytes + EAM_CTA_WAVES * perWave;
}
