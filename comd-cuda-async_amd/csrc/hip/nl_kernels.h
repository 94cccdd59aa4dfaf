// nl_kernels.h -- Verlet neighbour lists on the device (methods thread_atom_nl / warp_atom_nl).
//
// What the reference does with gpu_neighborList.c + gpu_kernels.cu:1087-1110 (updateNeighborListRequiredKernel),
// :1486-2029 (buildNeighborListGpu and the *_nl force kernels) and its gid hash table (hashTable.c): lists hold every
// atom within cutoff + skin, link cells are sized cutoff + skin, and the lists stay valid until some atom has moved
// more than skin/2 since the build.  Between builds no atom changes its slot (no re-binning, CoMD.c:257-268 /
// timestep.c:278-352), so here a list entry names a slot, never a gid, and the halo copies are refreshed in place by a
// positional exchange (the slot-ordered path of the dF/drho exchange) -- no hash table.
//
// Three list formats (NeighborListGpu.slabFormat), all row-major over the atoms of a cell so that a wave reads row k of its
// atoms as one line:
//   1  LJ:  16-bit offsets into the LDS staging of one group of 9 stencil cells   (LJ_Force_nl_slabs, second half of this file)
//   2  EAM: 16-bit record numbers in a wave's LDS staging of the whole 27-cell stencil (EAM_Force_nl_lds, end of this file)
//   0  plain: list[(cell * maxNbr + k) * cap + i] = global slot of the k-th neighbour, gathered from global memory -- the fallback
//      for cells of more than 512 slots and for EAM tables that do not fit the LDS (setfl); kernels directly below.
#pragma once
#include "device_common.h"
#include "lj_kernels.h"
#include "eam_kernels.h"

struct NlView {
   int* __restrict__ list;
   int* __restrict__ count;            // [nLocalBoxes * cap]
   int  maxNbr;
};

// thread per slot of the listed cells
__device__ __forceinline__ bool nlSlot(const int* __restrict__ cells, const int* __restrict__ nAtoms, int nCells, int cap, int& iBox, int& i)
{
   const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const int ci = (int)(tid / cap);
   if (ci >= nCells) return false;
   i = (int)(tid - (long)ci * cap);
   iBox = cells ? cells[ci] : ci;
   return i < nAtoms[iBox];
}

// ---- build: every atom within rBuild of atom i (27-cell stencil, self excluded), positions snapshotted into lastR ----------
__global__ __launch_bounds__(256)
void BuildNeighborList(const real_t* __restrict__ rx, const real_t* __restrict__ ry, const real_t* __restrict__ rz,
                       const int* __restrict__ nAtoms, const int* __restrict__ nbr, const int* __restrict__ cells, int nCells, int cap,
                       NlView nl, real_t rBuild2, real_t* __restrict__ lastX, real_t* __restrict__ lastY, real_t* __restrict__ lastZ,
                       int* __restrict__ status)
{
   int iBox, i;
   if (!nlSlot(cells, nAtoms, nCells, cap, iBox, i)) return;
   const size_t iSlot = (size_t)iBox * cap + i;
   const real_t xi = rx[iSlot], yi = ry[iSlot], zi = rz[iSlot];
   lastX[iSlot] = xi; lastY[iSlot] = yi; lastZ[iSlot] = zi;
   int* __restrict__ row = nl.list + (size_t)iBox * nl.maxNbr * cap + i;
   int n = 0;
   for (int k = 0; k < 27; ++k) {
      const int jBox = nbr[(size_t)iBox * 27 + k];
      const int nj = nAtoms[jBox];
      const size_t base = (size_t)jBox * cap;
      for (int j = 0; j < nj; ++j) {
         const real_t dx = xi - rx[base + j], dy = yi - ry[base + j], dz = zi - rz[base + j];
         const real_t r2 = dx*dx + dy*dy + dz*dz;
         if (r2 <= rBuild2 && base + j != iSlot) {
            if (n < nl.maxNbr) row[(size_t)n * cap] = (int)(base + j);
            ++n;
         }
      }
   }
   if (n > nl.maxNbr) { atomicOr(&status[3], 2); n = nl.maxNbr; }
   nl.count[iSlot] = n;
}

// gpu_kernels.cu:1087-1110: has any local atom moved more than skin/2 since the build?
__global__ __launch_bounds__(256)
void NeighborListUpdateRequired(const real_t* __restrict__ rx, const real_t* __restrict__ ry, const real_t* __restrict__ rz,
                                const real_t* __restrict__ lastX, const real_t* __restrict__ lastY, const real_t* __restrict__ lastZ,
                                const int* __restrict__ nAtoms, int nLocalBoxes, int cap, real_t skinHalf2, int* __restrict__ flag)
{
   int iBox, i;
   if (!nlSlot(nullptr, nAtoms, nLocalBoxes, cap, iBox, i)) return;
   const size_t s = (size_t)iBox * cap + i;
   const real_t dx = rx[s] - lastX[s], dy = ry[s] - lastY[s], dz = rz[s] - lastZ[s];
   if (dx*dx + dy*dy + dz*dz > skinHalf2) *flag = 1;
}

// ---- LJ over the list ----------------------------------------------------------------------------------------------------
template <bool ENERGY>
__global__ __launch_bounds__(256)
void LJ_Force_thread_atom_nl(LjArgs a, NlView nl)
{
   int iBox, i;
   if (!nlSlot(a.cells, a.nAtoms, a.nCells, a.cap, iBox, i)) return;
   const size_t iSlot = (size_t)iBox * a.cap + i;
   const real_t xi = a.rx[iSlot], yi = a.ry[iSlot], zi = a.rz[iSlot];
   const int n = nl.count[iSlot];
   const int* __restrict__ row = nl.list + (size_t)iBox * nl.maxNbr * a.cap + i;
   real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0);
   int k = 0;
   for (; k + 4 <= n; k += 4) {                 // four gathers in flight
      int j[4]; real_t dx[4], dy[4], dz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) j[u] = row[(size_t)(k + u) * a.cap];
#pragma unroll
      for (int u = 0; u < 4; ++u) { dx[u] = xi - a.rx[j[u]]; dy[u] = yi - a.ry[j[u]]; dz[u] = zi - a.rz[j[u]]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const real_t r2 = dx[u]*dx[u] + dy[u]*dy[u] + dz[u]*dz[u];
         if (r2 <= a.rc2) ljPair<ENERGY>(dx[u], dy[u], dz[u], r2, a, fx, fy, fz, e);
      }
   }
   for (; k < n; ++k) {
      const int j = row[(size_t)k * a.cap];
      const real_t dx = xi - a.rx[j], dy = yi - a.ry[j], dz = zi - a.rz[j];
      const real_t r2 = dx*dx + dy*dy + dz*dz;
      if (r2 <= a.rc2) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
   }
   const real_t fs = LJ_FORCE_SCALE(a);
   a.fx[iSlot] = fx * fs; a.fy[iSlot] = fy * fs; a.fz[iSlot] = fz * fs;
   if (ENERGY) a.e[iSlot] = e * R(2.0) * a.eps;
}

// ---- EAM passes 1 and 3 over the list; tables in LDS when they fit (funcfl), else through L2 (setfl) ----------------------------
template <int STEP, bool LDS_TABLES, bool SPLINE>
__global__ __launch_bounds__(256)
void EAM_Force_thread_atom_nl(EamArgs a, NlView nl)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   real_t* sRho = (real_t*)ldsRaw;
   const int nRhoPad = a.rho.n + 3, nPhiPad = (STEP == 1) ? a.phi.n + 3 : 0;
   real_t* sPhi = sRho + nRhoPad;
   const bool sameGrid = (STEP == 1) && LDS_TABLES && a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   if (LDS_TABLES) {
      if (sameGrid) {
         for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) { sRho[2 * t] = a.phi.values[t]; sRho[2 * t + 1] = a.rho.values[t]; }
      } else {
         for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) sRho[t] = a.rho.values[t];
         if (STEP == 1) for (int t = threadIdx.x; t < nPhiPad; t += blockDim.x) sPhi[t] = a.phi.values[t];
      }
      __syncthreads();
   }
   const TableView rhoT = makeTable(a.rho, LDS_TABLES ? sRho : a.rho.values), phiT = makeTable(a.phi, LDS_TABLES ? sPhi : a.phi.values);

   int iBox, i;
   if (!nlSlot(a.cells, a.nAtoms, a.nCells, a.cap, iBox, i)) return;
   const size_t iSlot = (size_t)iBox * a.cap + i;
   const real_t xi = a.rx[iSlot], yi = a.ry[iSlot], zi = a.rz[iSlot];
   const int n = nl.count[iSlot];
   const int* __restrict__ row = nl.list + (size_t)iBox * nl.maxNbr * a.cap + i;
   real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0), rb = R(0.0), dfi = R(0.0);
   if (STEP == 3) { fx = a.fx[iSlot]; fy = a.fy[iSlot]; fz = a.fz[iSlot]; dfi = a.dfEmbed[iSlot]; }
   for (int k0 = 0; k0 < n; k0 += 4) {
      int j[4]; real_t dx[4], dy[4], dz[4], dfj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) j[u] = (k0 + u < n) ? row[(size_t)(k0 + u) * a.cap] : (int)iSlot;     // padding pairs have r2 = 0 and are rejected
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         dx[u] = xi - a.rx[j[u]]; dy[u] = yi - a.ry[j[u]]; dz[u] = zi - a.rz[j[u]];
         if (STEP == 3) dfj[u] = a.dfEmbed[j[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const real_t r2 = dx[u]*dx[u] + dy[u]*dy[u] + dz[u]*dz[u];
         if (r2 <= a.rc2 && r2 > R(0.0)) {
            real_t rho, drho, dphi;
            if (SPLINE) {
               interpolateSpline(a.rhoS, r2, rho, drho);
               if (STEP == 1) { real_t phi; interpolateSpline(a.phiS, r2, phi, dphi); e += phi; rb += rho; }
               else           { dphi = (dfi + dfj[u]) * drho; }
            } else {
               const real_t ir = rsqrtR(r2), r = r2 * ir;
               if (STEP == 1) {
                  real_t phi;
                  if (sameGrid) interpolatePair(sRho, rhoT, r, phi, dphi, rho, drho);
                  else { interpolate(rhoT, r, rho, drho); interpolate(phiT, r, phi, dphi); }
                  e += phi; rb += rho;
               } else {
                  interpolate(rhoT, r, rho, drho);
                  dphi = (dfi + dfj[u]) * drho;
               }
               dphi *= ir;
            }
            fx -= dphi * dx[u]; fy -= dphi * dy[u]; fz -= dphi * dz[u];
         }
      }
   }
   a.fx[iSlot] = fx; a.fy[iSlot] = fy; a.fz[iSlot] = fz;
   if (STEP == 1) { a.e[iSlot] = R(0.5) * e; a.rhobar[iSlot] = rb; }
}

// ====================================================================================================================
// LJ at 5 sigma: ~730 listed neighbours per atom.  Gathering them from global memory is address-rate bound (one lane address per
// clock per CU: 5.8 ms per step at 80^3, slower than the cell kernel), so the list is split into NL_GROUPS groups of 9 stencil
// cells (groupCell below), the workgroup of a cell stages one group's positions in the LDS at a time and the entries are
// 16-bit indices (3 * position: the x of an {x,y,z} record) into that staging.
// list16[((cell*NL_GROUPS + g) * rows + k) * cap + i], count[(cell*NL_GROUPS + g) * cap + i].
#ifndef NL_GROUPS
#define NL_GROUPS 3
#endif
#ifndef NL_BATCH
#define NL_BATCH 8
#endif

#define NL_GROUP_CELLS (27 / NL_GROUPS)

struct NlSlabView {
   unsigned short* __restrict__ list;
   int* __restrict__ count;
   int  rows;
};

// Cell kk of group g, as an index into the self-first neighbour table.  With NL_DIAGONAL the three groups are the classes of
// (dx + dy + dz) mod 3 of the stencil offsets: each holds near and far cells alike (group 0: the cell itself, six edge and two
// corner neighbours; groups 1 and 2: three face, three edge and three corner neighbours each), so every atom finds a similar share of
// its list in every group wherever it sits in its cell, and neither the lanes of a wave nor the waves of a workgroup wait for each
// other at the per-group barriers (x-plane groups: an atom near a face has most of its list on that side; 2.41 ms per force call at
// 80^3 against 2.06 ms with these).
#ifndef NL_DIAGONAL
#define NL_DIAGONAL (NL_GROUPS == 3)
#endif
__device__ __forceinline__ int groupCell(int g, int kk)
{
#if NL_DIAGONAL
   // positions p = 9a + 3b + c with (a + b + c) % 3 == g, mapped through p -> (p < 13 ? p + 1 : p == 13 ? 0 : p)
   constexpr unsigned char tab[3][9] = { { 1, 6, 8, 12, 0, 15, 19, 21, 26 }, { 2, 4, 9, 10, 14, 16, 20, 22, 24 }, { 3, 5, 7, 11, 13, 17, 18, 23, 25 } };
   return tab[g][kk];
#else
   const int p = g * NL_GROUP_CELLS + kk;              // x-major stencil order (self at position 13): x-planes (3 groups) or z-columns (9)
   return p < 13 ? p + 1 : p == 13 ? 0 : p;
#endif
}

// Build: workgroup per cell, thread per atom, the same staging as the force kernel; every thread walks the staged records of the
// group in order (a wave-uniform LDS address: the read is a broadcast) and appends its hits.  blockDim.x >= every cell's occupancy
// (launched with the cell capacity), LDS = 3 * groupCapacity doubles.  Fetching the candidates with per-lane global loads instead
// costs 22 ms per build at 80^3 (address rate of uniform loads); this form 2-3 ms.
__global__ __launch_bounds__(512)
void BuildNeighborListSlabs(const real_t* __restrict__ rx, const real_t* __restrict__ ry, const real_t* __restrict__ rz,
                            const int* __restrict__ nAtoms, const int* __restrict__ nbr, int nCells, int cap,
                            NlSlabView nl, real_t rBuild2, real_t* __restrict__ lastX, real_t* __restrict__ lastY, real_t* __restrict__ lastZ,
                            int* __restrict__ stats, int* __restrict__ status, int bankOrder, int groupCap)
{
   extern __shared__ __attribute__((aligned(16))) real_t ldsPos[];
   real_t* __restrict__ sp = ldsPos;
   const int iBox = blockIdx.x;
   const int i = threadIdx.x;
   const int ni = nAtoms[iBox];
   const bool active = i < ni;
   const size_t iSlot = (size_t)iBox * cap + (active ? i : 0);
   const real_t xi = rx[iSlot], yi = ry[iSlot], zi = rz[iSlot];
   if (active) { lastX[iSlot] = xi; lastY[iSlot] = yi; lastZ[iSlot] = zi; }
   if (i == 0) atomicMax(&stats[1], ni);
   bool over = false;
   for (int g = 0; g < NL_GROUPS; ++g) {
      if (g) __syncthreads();
      int total = 0, selfAt = -1;                      // records of the group; where this cell's own atoms start in it (-1: not in this group)
      {
         real_t vx[NL_GROUP_CELLS], vy[NL_GROUP_CELLS], vz[NL_GROUP_CELLS];
         int dst[NL_GROUP_CELLS];
#pragma unroll
         for (int kk = 0; kk < NL_GROUP_CELLS; ++kk) {
            const int k = groupCell(g, kk);
            const int jBox = nbr[(size_t)iBox * 27 + k];
            const int nj = nAtoms[jBox];
            if (k == 0) selfAt = total;
            const size_t js = (size_t)jBox * cap + (i < nj ? i : 0);
            vx[kk] = rx[js]; vy[kk] = ry[js]; vz[kk] = rz[js];
            dst[kk] = i < nj ? total + i : -1;
            total += nj;
         }
#pragma unroll
         for (int kk = 0; kk < NL_GROUP_CELLS; ++kk)
            if (dst[kk] >= 0) { sp[3 * dst[kk]] = vx[kk]; sp[3 * dst[kk] + 1] = vy[kk]; sp[3 * dst[kk] + 2] = vz[kk]; }
      }
      if (i == 0) atomicMax(&stats[0], total);
      __syncthreads();
      if (active && !bankOrder) {
         unsigned short* __restrict__ row = nl.list + ((size_t)(iBox * NL_GROUPS + g) * nl.rows) * cap + i;
         const int me = selfAt >= 0 ? selfAt + i : -1;
         int n = 0;
         for (int t = 0; t < total; ++t) {
            const real_t dx = xi - sp[3 * t], dy = yi - sp[3 * t + 1], dz = zi - sp[3 * t + 2];
            if (dx*dx + dy*dy + dz*dz <= rBuild2 && t != me) {
               if (n < nl.rows) row[(size_t)n * cap] = (unsigned short)(3 * t);      // index of the x of its {x,y,z} record
               ++n;
            }
         }
         if (n > nl.rows) { over = true; n = nl.rows; }
         nl.count[(size_t)(iBox * NL_GROUPS + g) * cap + i] = n;
      }
      if (bankOrder) {
         // [round 4] The force kernel gathers entry k of 64 different rows with one ds_read_b64 per coordinate, and a row in record order sends neighbouring
         // lanes to neighbouring records: SQ_LDS_BANK_CONFLICT was 53 % of the LDS cycles.  An entry e (= 3 t) names the 8-byte word e of the staging, so its
         // bank pair is e mod 16: order every row so that entry k of lane i has class (i + k) mod 16 -- the 16 lanes of a quarter wave then read 16 different
         // bank pairs at every step -- by dealing the row's entries round-robin over the classes, starting at class i mod 16 and skipping a class that has run
         // out (the tail of a row loses the alignment; the classes of a row of 240 hold 15 +- 4).  Two sweeps over the records: count per class, then place.
         unsigned short* sCnt = (unsigned short*)(sp + 3 * (size_t)groupCap);      // [16][blockDim]: hits of this lane per class
         unsigned short* sRun = sCnt + 16 * blockDim.x;                            // [16][blockDim]: ... placed so far
         for (int c = 0; c < 16; ++c) { sCnt[c * blockDim.x + i] = 0; sRun[c * blockDim.x + i] = 0; }
         const int me = selfAt >= 0 ? selfAt + i : -1;
         int n = 0;
         if (active) {
            for (int t = 0; t < total; ++t) {
               const real_t dx = xi - sp[3 * t], dy = yi - sp[3 * t + 1], dz = zi - sp[3 * t + 2];
               if (dx*dx + dy*dy + dz*dz <= rBuild2 && t != me) { ++sCnt[((3 * t) & 15) * blockDim.x + i]; ++n; }
            }
            if (n > nl.rows) over = true;
            unsigned short* __restrict__ row = nl.list + ((size_t)(iBox * NL_GROUPS + g) * nl.rows) * cap + i;
            int cnt[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) cnt[c] = sCnt[c * blockDim.x + i];
            const int s0 = i & 15;
            for (int t = 0; t < total; ++t) {
               const real_t dx = xi - sp[3 * t], dy = yi - sp[3 * t + 1], dz = zi - sp[3 * t + 2];
               if (dx*dx + dy*dy + dz*dz <= rBuild2 && t != me) {
                  const int c = (3 * t) & 15;
                  const int m = sRun[c * blockDim.x + i]++;
                  // entries dealt before the m-th of class c: every class c' contributes min(cnt[c'], m + 1) when it comes before c in the cycle that starts
                  // at s0, min(cnt[c'], m) when it comes after
                  const int rc_ = (c - s0) & 15;
                  int pos = m;
#pragma unroll
                  for (int cc = 0; cc < 16; ++cc) {
                     const int before = (((cc - s0) & 15) < rc_) ? 1 : 0;
                     const int lim = m + before;
                     pos += cc == c ? 0 : (cnt[cc] < lim ? cnt[cc] : lim);
                  }
                  if (pos < nl.rows) row[(size_t)pos * cap] = (unsigned short)(3 * t);
               }
            }
            nl.count[(size_t)(iBox * NL_GROUPS + g) * cap + i] = n < nl.rows ? n : nl.rows;
         }
      }
   }
   if (over) atomicOr(&status[3], 2);
}

// workgroup per cell, thread per atom; blockDim.x = fullest cell rounded up to whole waves (<= 512); LDS = 3 * groupAtoms doubles
template <bool ENERGY>
__global__ __launch_bounds__(512)
void LJ_Force_nl_slabs(LjArgs a, NlSlabView nl, int groupAtoms)
{
   extern __shared__ __attribute__((aligned(16))) real_t ldsPos[];      // {x, y, z} records: one address per neighbour, ds_read offsets 0/8/16
   real_t* __restrict__ sp = ldsPos;
   (void)groupAtoms;
   const int iBox = a.cells ? a.cells[blockIdx.x] : blockIdx.x;
   const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
   const int ni = uniform(a.nAtoms[iBox]);
   // A wave owns the atoms 64 wave .. 64 wave + m - 1 of the cell.  An under-filled wave (the tail of a cell: 20 of 64 lanes at 148 atoms, and its
   // instruction stream is as long as a full wave's) replicates its atoms G times across the lanes: replica g walks the rows g, g + G, ... of each
   // list, the partial sums meet through ds_bpermute at the end.  A full wave is the case G = 1.
   const int m = ni - 64 * wave < 64 ? (ni - 64 * wave > 0 ? ni - 64 * wave : 0) : 64;
   const int G = (m > 0 && m <= 32) ? (64 / m < 4 ? 64 / m : 4) : 1;
   const int g = G > 1 ? lane / m : 0, ai = G > 1 ? lane - g * m : lane;
   const int i = 64 * wave + ai;                              // the atom of this lane
   const bool active = ai < m && g < G;
   const size_t iSlot = (size_t)iBox * a.cap + (active ? i : 0);
   const real_t xi = a.rx[iSlot], yi = a.ry[iSlot], zi = a.rz[iSlot];
   real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0);
   for (int grp = 0; grp < NL_GROUPS; ++grp) {
      if (grp) __syncthreads();                       // everyone is done reading the previous group
      {  // all cells' loads in flight together, then the LDS stores (one global round trip per group); thread t stages slot t of every cell
         real_t vx[NL_GROUP_CELLS], vy[NL_GROUP_CELLS], vz[NL_GROUP_CELLS];
         int dst[NL_GROUP_CELLS];
         int off = 0;
#pragma unroll
         for (int kk = 0; kk < NL_GROUP_CELLS; ++kk) {
            const int jBox = a.nbr[(size_t)iBox * 27 + groupCell(grp, kk)];
            const int nj = a.nAtoms[jBox];
            const size_t js = (size_t)jBox * a.cap + (tid < nj ? tid : 0);      // unconditional loads (slot 0 always exists) so they all issue at once
            vx[kk] = a.rx[js]; vy[kk] = a.ry[js]; vz[kk] = a.rz[js];
            dst[kk] = tid < nj ? off + tid : -1;
            off += nj;
         }
#pragma unroll
         for (int kk = 0; kk < NL_GROUP_CELLS; ++kk)
            if (dst[kk] >= 0) { sp[3 * dst[kk]] = vx[kk]; sp[3 * dst[kk] + 1] = vy[kk]; sp[3 * dst[kk] + 2] = vz[kk]; }
      }
      __syncthreads();
      if (active) {
         const int nAll = nl.count[(size_t)(iBox * NL_GROUPS + grp) * a.cap + i];
         const int n = nAll > g ? (nAll - g + G - 1) / G : 0;               // rows of this replica: g, g + G, ...
         const unsigned short* __restrict__ row = nl.list + ((size_t)(iBox * NL_GROUPS + grp) * nl.rows + g) * a.cap + i;
         const size_t step = (size_t)G * a.cap;
         // NL_BATCH rows per trip, the next trip's rows already in flight (the list streams from HBM), all LDS gathers of a trip
         // issued before the arithmetic: one LDS round trip per batch, not per pair
         int k = 0;
         int jn[NL_BATCH];
         if (n >= NL_BATCH) {
#pragma unroll
            for (int u = 0; u < NL_BATCH; ++u) jn[u] = row[(size_t)u * step];
         }
         for (; k + NL_BATCH <= n; k += NL_BATCH) {
            int j[NL_BATCH];
#pragma unroll
            for (int u = 0; u < NL_BATCH; ++u) j[u] = jn[u];
            if (k + 2 * NL_BATCH <= n) {
#pragma unroll
               for (int u = 0; u < NL_BATCH; ++u) jn[u] = row[(size_t)(k + NL_BATCH + u) * step];
            }
            real_t dx[NL_BATCH], dy[NL_BATCH], dz[NL_BATCH];
#pragma unroll
            for (int u = 0; u < NL_BATCH; ++u) { dx[u] = xi - sp[j[u]]; dy[u] = yi - sp[j[u] + 1]; dz[u] = zi - sp[j[u] + 2]; }
#pragma unroll
            for (int u = 0; u < NL_BATCH; ++u) {
               const real_t r2 = dx[u]*dx[u] + dy[u]*dy[u] + dz[u]*dz[u];
               if (r2 <= a.rc2) ljPair<ENERGY>(dx[u], dy[u], dz[u], r2, a, fx, fy, fz, e);      // (evaluating all 8 branch-free, misses weighted 0: 2.09 vs 2.07 ms)
            }
         }
         for (; k < n; ++k) {
            const int j = row[(size_t)k * step];
            const real_t dx = xi - sp[j], dy = yi - sp[j + 1], dz = zi - sp[j + 2];
            const real_t r2 = dx*dx + dy*dy + dz*dz;
            if (r2 <= a.rc2) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
         }
      }
   }
   if (G > 1) {                                               // (wave-uniform) all lanes take part; lanes < m keep the sum of the replicas
      if (!active) { fx = fy = fz = e = R(0.0); }
      real_t tx = fx, ty = fy, tz = fz, te = e;
      for (int r = 1; r < G; ++r) {
         const int src = (ai + r * m) & 63;
         tx += bpermuteR(fx, src); ty += bpermuteR(fy, src); tz += bpermuteR(fz, src);
         if (ENERGY) te += bpermuteR(e, src);
      }
      fx = tx; fy = ty; fz = tz; e = te;
   }
   if (active && g == 0) {
      const real_t fs = LJ_FORCE_SCALE(a);
      a.fx[iSlot] = fx * fs; a.fy[iSlot] = fy * fs; a.fz[iSlot] = fz * fs;
      if (ENERGY) a.e[iSlot] = e * R(2.0) * a.eps;
   }
}

// ====================================================================================================================
// EAM: ~57 listed neighbours per atom (Cu_u6, skin 10 %), ~14 atoms per cell: the whole 27-cell stencil is 373 atoms, 9-12 KB as
// {x,y,z[,F']} records, so ONE WAVE stages it in its own LDS slice and walks the lists of its cell with no barrier at all.
// Lanes: 4 per atom (lane = 4*i + q takes rows k = q, q+4, ... of atom i; a quad-permute DPP sum joins them), 16 atoms per round.
// Entries are 16-bit record numbers in staging order (the self-first neighbour table, each cell's atoms in slot order).
// list16[(cell * rows + k) * cap + i], count[cell * cap + i].  cap <= 64.
#define EAM_NL_WAVES 4

// Build, same shape as the force kernel below: one wave per cell stages the stencil in its LDS slice; the four lanes of atom i test
// four staged records per trip and a ballot packs their hits into consecutive rows (record order), so the rows q, q+4, ... each
// lane reads back in the force kernel are an even quarter of the list.
__host__ __device__ static inline size_t eamBuildWaveBytes(int stencilCapacity, int rows)
{
   return ((size_t)3 * stencilCapacity * sizeof(real_t) + 256 + (size_t)16 * rows * 2 + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(64 * EAM_NL_WAVES)
void BuildNeighborListCell16(const real_t* __restrict__ rx, const real_t* __restrict__ ry, const real_t* __restrict__ rz,
                             const int* __restrict__ nAtoms, const int* __restrict__ nbr, int nCells, int cap,
                             NlSlabView nl, real_t rBuild2, real_t* __restrict__ lastX, real_t* __restrict__ lastY, real_t* __restrict__ lastZ,
                             int* __restrict__ stats, int* __restrict__ status, int stencilCapacity)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
   // per wave: [3 * stencilCapacity] records, [32] offsets, [32] cells, [16 atoms][rows] 16-bit entries
   real_t* __restrict__ sp = (real_t*)(ldsRaw + (size_t)wave * eamBuildWaveBytes(stencilCapacity, nl.rows));
   int* sOff = (int*)(sp + 3 * stencilCapacity);
   int* sBox = sOff + 32;
   unsigned short* sHit = (unsigned short*)(sBox + 32);
   const int q = lane & 3, ia = lane >> 2;
   const int nWaves = gridDim.x * EAM_NL_WAVES;
   const int gw = blockIdx.x * EAM_NL_WAVES + wave;
   const int per = (nCells + nWaves - 1) / nWaves;
   bool over = false, tooSmall = false;
   int maxTotal = 0, maxNi = 0;
   for (int iBox = gw * per; iBox < (gw + 1) * per && iBox < nCells; ++iBox) {
      {
         const int box = lane < 27 ? nbr[(size_t)iBox * 27 + lane] : 0;
         const int cnt = lane < 27 ? nAtoms[box] : 0;
         int incl = cnt;
#pragma unroll
         for (int d = 1; d < 32; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
         if (lane < 28) { sOff[lane] = incl - cnt; sBox[lane] = box; }
      }
      __builtin_amdgcn_wave_barrier();
      const int total = uniform(sOff[27]), ni = uniform(sOff[1]);
      maxTotal = total > maxTotal ? total : maxTotal; maxNi = ni > maxNi ? ni : maxNi;      // one atomic per wave at the end, not per cell
      if (total > stencilCapacity) { tooSmall = true; continue; }      // the host sees stats[0] > capacity and repeats the build (worst-case capacity: flagged)
      for (int t0 = 0; t0 < total; t0 += 256) {
         real_t vx[4], vy[4], vz[4];
#pragma unroll
         for (int g = 0; g < 4; ++g) {
            const int t = t0 + g * 64 + lane;
            const int tt = t < total ? t : 0;
            int lo = 0;
#pragma unroll
            for (int step = 16; step >= 1; step >>= 1) { const int m = lo + step; if (m <= 26 && sOff[m] <= tt) lo = m; }
            const size_t o = (size_t)sBox[lo] * cap + (tt - sOff[lo]);
            vx[g] = rx[o]; vy[g] = ry[o]; vz[g] = rz[o];
         }
#pragma unroll
         for (int g = 0; g < 4; ++g) {
            const int t = t0 + g * 64 + lane;
            if (t < total) { sp[3 * t] = vx[g]; sp[3 * t + 1] = vy[g]; sp[3 * t + 2] = vz[g]; }
         }
      }
      __builtin_amdgcn_wave_barrier();
      for (int i0 = 0; i0 < ni; i0 += 16) {
         const int i = i0 + ia;
         const bool have = i < ni;
         const int ii = have ? i : 0;
         const size_t iSlot = (size_t)iBox * cap + ii;
         const real_t xi = sp[3 * ii], yi = sp[3 * ii + 1], zi = sp[3 * ii + 2];
         if (have && q == 0) { lastX[iSlot] = xi; lastY[iSlot] = yi; lastZ[iSlot] = zi; }
         unsigned short* __restrict__ row = nl.list + ((size_t)iBox * nl.rows) * cap + ii;
         int n = 0;                                          // hits of atom i so far (the same in its four lanes)
         for (int t0 = 0; t0 < total; t0 += 16) {            // 16 records per trip, four per lane of the quad; rows keep record order
            bool hit[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                     // the four distance tests are independent: 12 LDS reads in flight
               const int t = t0 + 4 * u + q;
               const int tt = t < total ? t : 0;
               const real_t dx = xi - sp[3 * tt], dy = yi - sp[3 * tt + 1], dz = zi - sp[3 * tt + 2];
               hit[u] = have && t < total && dx*dx + dy*dy + dz*dz <= rBuild2 && t != i;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
               const unsigned nib = (unsigned)(__ballot(hit[u]) >> (lane & ~3)) & 0xFu;
               const int k = n + __popc(nib & ((1u << q) - 1u));
               if (hit[u] && k < nl.rows) sHit[ia * nl.rows + k] = (unsigned short)(t0 + 4 * u + q);      // collected in the LDS first ...
               n += __popc(nib);
            }
         }
         if (n > nl.rows) { over = true; n = nl.rows; }
         if (have && q == 0) nl.count[iSlot] = n;
         // ... and written out four rows per instruction, 16 atoms x 2 bytes contiguous in each (a store per hit from the loop
         // above touches 64 different lines per instruction and made the build store-address bound: 4.1 -> 2 ms)
         __builtin_amdgcn_wave_barrier();
         int nMax = have ? n : 0;
#pragma unroll
         for (int m = 32; m >= 4; m >>= 1) { const int o = __shfl_xor(nMax, m); nMax = o > nMax ? o : nMax; }
         for (int k0 = 0; k0 < nMax; k0 += 4) {
            const int k = k0 + q;
            if (have && k < n) row[(size_t)k * cap] = sHit[ia * nl.rows + k];
         }
         __builtin_amdgcn_wave_barrier();
      }
      __builtin_amdgcn_wave_barrier();
   }
   if (lane == 0) { atomicMax(&stats[0], maxTotal); atomicMax(&stats[1], maxNi); }
   if (over || (tooSmall && stencilCapacity >= 1536)) atomicOr(&status[3], 2);
}

// quad sum: every lane of an aligned group of 4 ends with the group's total
__device__ __forceinline__ real_t quadSum(real_t v)
{
   v += dppMoveR<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
   v += dppMoveR<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
   return v;
}

// LDS layouts in bytes (records in the build's precision; offsets, cell ids and 16-bit rows in their own units), every part a multiple of 16
__host__ __device__ static inline size_t eamTableBytesAligned(size_t tableWords) { return (tableWords * sizeof(real_t) + 15) & ~(size_t)15; }
__host__ __device__ static inline size_t eamNlWaveBytes(int rec, int stencilAtoms)
{
   return ((size_t)rec * stencilAtoms * sizeof(real_t) + 256 + 16 * 64 * 2 + 15) & ~(size_t)15;      // records, [32] offsets + [32] cells, [16][64] row stash
}

template <int STEP, bool SPLINE>
__global__ __launch_bounds__(64 * EAM_NL_WAVES)
void EAM_Force_nl_lds(EamArgs a, NlSlabView nl, int stencilAtoms)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   constexpr int REC = (STEP == 3) ? 4 : 3;                  // doubles per staged atom: x, y, z [, F']
   const int nRhoPad = a.rho.n + 3;
   const bool sameGrid = !SPLINE && (STEP == 1) && a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   real_t* sRho = (real_t*)ldsRaw;                           // pass 1 on one r grid: interleaved {phi, rho}; else rho then phi
   real_t* sPhi = sRho + nRhoPad;
   const int tableDoubles = SPLINE ? 0 : (STEP == 1) ? 2 * nRhoPad + (sameGrid ? 0 : (a.phi.n + 3 - nRhoPad)) : nRhoPad;      // -P: spline coefficients stay in L2
   if (SPLINE) {
   } else if (sameGrid) {
      for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) { sRho[2 * t] = a.phi.values[t]; sRho[2 * t + 1] = a.rho.values[t]; }
   } else {
      for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) sRho[t] = a.rho.values[t];
      if (STEP == 1) for (int t = threadIdx.x; t < a.phi.n + 3; t += blockDim.x) sPhi[t] = a.phi.values[t];
   }
   __syncthreads();
   const TableView rhoT = makeTable(a.rho, sRho), phiT = makeTable(a.phi, sPhi);

   const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
   real_t* __restrict__ sp = (real_t*)(ldsRaw + eamTableBytesAligned(tableDoubles) + (size_t)wave * eamNlWaveBytes(REC, stencilAtoms));
   int* sOff = (int*)(sp + REC * stencilAtoms);              // [32]: exclusive record offsets of the 27 cells, [27] = total
   int* sBox = sOff + 32;                                    // [32]
   unsigned short* sEnt = (unsigned short*)(sBox + 32);      // [16][64]: the rows a lane fetched for the current round
   const int q = lane & 3, ia = lane >> 2;                   // list part, atom of the round

   // cells are dealt to waves in contiguous runs so that neighbouring cells (shared stencil lines) meet in one L2
   const int nWaves = gridDim.x * EAM_NL_WAVES;
   const int gw = xcdRemap(blockIdx.x, gridDim.x) * EAM_NL_WAVES + wave;
   const int per = (a.nCells + nWaves - 1) / nWaves;
   const int ciEnd = (gw + 1) * per < a.nCells ? (gw + 1) * per : a.nCells;
   constexpr int CH = 16;                                    // list rows a lane keeps in flight (rows q, q+4, ..., q+60 of its atom)
   constexpr int SR = 8;                                     // staged records a lane keeps in flight (512 per wave; larger stencils: a second, blocking trip)

   // Software pipeline over the wave's cells.  While cell c is evaluated out of the LDS, the loads of cell c+1 (stencil records,
   // list rows) are in flight into registers, and the stencil description of cell c+2 (two dependent reads: neighbour table,
   // occupancies) is on its way: a cell costs its arithmetic, not three global round trips.
   auto cellOf = [&](int ci) { return a.cells ? a.cells[ci] : ci; };
   int boxB = 0, cntB = 0;                                   // description of the cell after the one being loaded
   int boxA = 0, cntA = 0;                                   // description of the cell being loaded
   const int ciBegin = gw * per;
   if (ciBegin < ciEnd)     { const int c = cellOf(ciBegin);     boxA = lane < 27 ? a.nbr[(size_t)c * 27 + lane] : 0; cntA = lane < 27 ? a.nAtoms[boxA] : 0; }
   if (ciBegin + 1 < ciEnd) { const int c = cellOf(ciBegin + 1); boxB = lane < 27 ? a.nbr[(size_t)c * 27 + lane] : 0; cntB = lane < 27 ? a.nAtoms[boxB] : 0; }

   real_t vx[SR], vy[SR], vz[SR], vd[SR];
   int ent[CH];
   int nMine = 0, totalL = 0, niL = 0, iBoxL = 0;            // of the cell whose loads are in flight

   // scan the description in (boxA, cntA) into sOff/sBox, then issue that cell's record and row loads
   auto issueLoads = [&]() {
      {
         int incl = cntA;
#pragma unroll
         for (int d = 1; d < 32; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
         if (lane < 28) { sOff[lane] = incl - cntA; sBox[lane] = boxA; }      // lane 27: cnt = 0 -> sOff[27] = total
      }
      __builtin_amdgcn_wave_barrier();
      totalL = uniform(sOff[27]); niL = uniform(sOff[1]); iBoxL = uniform(boxA);
      {
         const int ii = ia < niL ? ia : 0;
         nMine = ia < niL ? nl.count[(size_t)iBoxL * a.cap + ii] : 0;
         const unsigned short* __restrict__ row = nl.list + ((size_t)iBoxL * nl.rows) * a.cap + ii;
#pragma unroll
         for (int u = 0; u < CH; ++u) { const int k = q + 4 * u; ent[u] = k < nl.rows ? row[(size_t)k * a.cap] : 0; }      // in bounds whatever n is
      }
#pragma unroll
      for (int g = 0; g < SR; ++g) {
         const int t = g * 64 + lane;
         const int tt = t < totalL ? t : 0;
         int lo = 0;                                                    // largest k in [0, 26] with sOff[k] <= tt
#pragma unroll
         for (int step = 16; step >= 1; step >>= 1) { const int m = lo + step; if (m <= 26 && sOff[m] <= tt) lo = m; }
         const size_t o = (size_t)sBox[lo] * a.cap + (tt - sOff[lo]);
         vx[g] = a.rx[o]; vy[g] = a.ry[o]; vz[g] = a.rz[o];
         if (STEP == 3) vd[g] = a.dfEmbed[o];
      }
   };
   if (ciBegin < ciEnd) issueLoads();

   for (int ci = ciBegin; ci < ciEnd; ++ci) {
      // (a) the loads of cell ci have been issued: land them in the LDS
      const int total = totalL, ni = niL, iBox = iBoxL;
      const int n0 = nMine;
#pragma unroll
      for (int g = 0; g < SR; ++g) {
         const int t = g * 64 + lane;
         if (t < total) {
            real_t* r = sp + REC * t;
            r[0] = vx[g]; r[1] = vy[g]; r[2] = vz[g];
            if (STEP == 3) r[3] = vd[g];
         }
      }
      for (int t = SR * 64 + lane; t < total; t += 64) {     // stencils beyond 512 atoms: blocking trip (sOff/sBox still describe cell ci)
         int lo = 0;
#pragma unroll
         for (int step = 16; step >= 1; step >>= 1) { const int m = lo + step; if (m <= 26 && sOff[m] <= t) lo = m; }
         const size_t o = (size_t)sBox[lo] * a.cap + (t - sOff[lo]);
         real_t* r = sp + REC * t;
         r[0] = a.rx[o]; r[1] = a.ry[o]; r[2] = a.rz[o];
         if (STEP == 3) r[3] = a.dfEmbed[o];
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) sEnt[u * 64 + lane] = (unsigned short)ent[u];
      __builtin_amdgcn_wave_barrier();

      // (b) start cell ci+1 (its description arrived during cell ci-1) and ask for the description of ci+2
      if (ci + 1 < ciEnd) {
         boxA = boxB; cntA = cntB;
         if (ci + 2 < ciEnd) { const int c = cellOf(ci + 2); boxB = lane < 27 ? a.nbr[(size_t)c * 27 + lane] : 0; cntB = lane < 27 ? a.nAtoms[boxB] : 0; }
         issueLoads();
      }

      // (c) evaluate cell ci from the LDS
      auto fetchRows = [&](int i0, int& nOut) {              // later rounds of a cell with more than 16 atoms: blocking
         const int i = i0 + ia;
         const int ii = i < ni ? i : 0;
         nOut = i < ni ? nl.count[(size_t)iBox * a.cap + ii] : 0;
         const unsigned short* __restrict__ row = nl.list + ((size_t)iBox * nl.rows) * a.cap + ii;
         for (int u = 0; u < CH; ++u) { const int k = q + 4 * u; sEnt[u * 64 + lane] = k < nl.rows ? row[(size_t)k * a.cap] : (unsigned short)0; }
      };
      for (int i0 = 0; i0 < ni; i0 += 16) {
         int n = n0;
         if (i0) fetchRows(i0, n);
         const int i = i0 + ia;
         const bool have = i < ni;
         const int ii = have ? i : 0;
         const real_t xi = sp[REC * ii], yi = sp[REC * ii + 1], zi = sp[REC * ii + 2];      // own cell is staged first: record i
         const real_t dfi = (STEP == 3) ? sp[REC * ii + 3] : R(0.0);
         real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0), rb = R(0.0);
         auto pairEval = [&](int j) {
            const real_t* r0 = sp + REC * j;
            const real_t dx = xi - r0[0], dy = yi - r0[1], dz = zi - r0[2];
            const real_t r2 = dx*dx + dy*dy + dz*dz;
            if (r2 <= a.rc2 && r2 > R(0.0)) {
               real_t rho, drho, dphi;
               if (SPLINE) {                                 // (1/r) d/dr straight from r^2
                  interpolateSpline(a.rhoS, r2, rho, drho);
                  if (STEP == 1) { real_t phi; interpolateSpline(a.phiS, r2, phi, dphi); e += phi; rb += rho; }
                  else           { dphi = (dfi + r0[3 % REC]) * drho; }
               } else {
                  const real_t ir = rsqrtR(r2), r = r2 * ir;
                  if (STEP == 1) {
                     real_t phi;
                     if (sameGrid) interpolatePair(sRho, rhoT, r, phi, dphi, rho, drho);
                     else { interpolate(rhoT, r, rho, drho); interpolate(phiT, r, phi, dphi); }
                     e += phi; rb += rho;
                  } else {
                     interpolate(rhoT, r, rho, drho);
                     dphi = (dfi + r0[3 % REC]) * drho;
                  }
                  dphi *= ir;
               }
               fx -= dphi * dx; fy -= dphi * dy; fz -= dphi * dz;
            }
         };
         // two pairs per trip, branch-free (a miss is evaluated at r = cutoff and weighted 0): two independent chains of LDS table
         // reads and fp64 arithmetic in flight per lane -- with 2 waves per SIMD the chains hide each other's latency
         auto pairEval2 = [&](int j0, int j1, bool two) {
            const real_t* r0 = sp + REC * j0; const real_t* r1 = sp + REC * j1;
            const real_t dx0 = xi - r0[0], dy0 = yi - r0[1], dz0 = zi - r0[2];
            const real_t dx1 = xi - r1[0], dy1 = yi - r1[1], dz1 = zi - r1[2];
            const real_t q0 = dx0*dx0 + dy0*dy0 + dz0*dz0, q1 = dx1*dx1 + dy1*dy1 + dz1*dz1;
            const bool h0 = q0 <= a.rc2 && q0 > R(0.0), h1 = two && q1 <= a.rc2 && q1 > R(0.0);
            const real_t s0 = h0 ? q0 : a.rc2, s1 = h1 ? q1 : a.rc2;
            real_t rho0, drho0, dphi0, rho1, drho1, dphi1;
            if (SPLINE) {
               interpolateSpline(a.rhoS, s0, rho0, drho0); interpolateSpline(a.rhoS, s1, rho1, drho1);
               if (STEP == 1) {
                  real_t phi0, phi1;
                  interpolateSpline(a.phiS, s0, phi0, dphi0); interpolateSpline(a.phiS, s1, phi1, dphi1);
                  e += (h0 ? phi0 : R(0.0)) + (h1 ? phi1 : R(0.0));
                  rb += (h0 ? rho0 : R(0.0)) + (h1 ? rho1 : R(0.0));
               } else {
                  dphi0 = (dfi + r0[3 % REC]) * drho0; dphi1 = (dfi + r1[3 % REC]) * drho1;
               }
               dphi0 = h0 ? dphi0 : R(0.0); dphi1 = h1 ? dphi1 : R(0.0);
            } else {
               const real_t ir0 = rsqrtR(s0), ir1 = rsqrtR(s1);
               const real_t d0 = s0 * ir0, d1 = s1 * ir1;
               if (STEP == 1) {
                  real_t phi0, phi1;
                  if (sameGrid) { interpolatePair(sRho, rhoT, d0, phi0, dphi0, rho0, drho0); interpolatePair(sRho, rhoT, d1, phi1, dphi1, rho1, drho1); }
                  else { interpolate(rhoT, d0, rho0, drho0); interpolate(phiT, d0, phi0, dphi0); interpolate(rhoT, d1, rho1, drho1); interpolate(phiT, d1, phi1, dphi1); }
                  e += (h0 ? phi0 : R(0.0)) + (h1 ? phi1 : R(0.0));
                  rb += (h0 ? rho0 : R(0.0)) + (h1 ? rho1 : R(0.0));
               } else {
                  interpolate(rhoT, d0, rho0, drho0); interpolate(rhoT, d1, rho1, drho1);
                  dphi0 = (dfi + r0[3 % REC]) * drho0; dphi1 = (dfi + r1[3 % REC]) * drho1;
               }
               dphi0 = h0 ? dphi0 * ir0 : R(0.0); dphi1 = h1 ? dphi1 * ir1 : R(0.0);
            }
            fx -= dphi0 * dx0; fy -= dphi0 * dy0; fz -= dphi0 * dz0;
            fx -= dphi1 * dx1; fy -= dphi1 * dy1; fz -= dphi1 * dz1;
         };
         const int mine = n > q ? (n - q + 3) >> 2 : 0;      // this lane's rows: q, q+4, ... < n
         const int inRegs = mine < CH ? mine : CH;
         for (int u = 0; u < inRegs; u += 2) pairEval2(sEnt[u * 64 + lane], u + 1 < inRegs ? sEnt[(u + 1) * 64 + lane] : ii, u + 1 < inRegs);
         if (mine > CH) {                                    // longer lists than the registers hold (not with the default sizing)
            const unsigned short* __restrict__ row = nl.list + ((size_t)iBox * nl.rows) * a.cap + ii;
            for (int u = CH; u < mine; ++u) pairEval(row[(size_t)(q + 4 * u) * a.cap]);
         }
         fx = quadSum(fx); fy = quadSum(fy); fz = quadSum(fz);
         if (STEP == 1) { e = quadSum(e); rb = quadSum(rb); }
         if (have && q == 0) {
            const size_t io = (size_t)iBox * a.cap + i;
            if (STEP == 1) { a.fx[io] = fx; a.fy[io] = fy; a.fz[io] = fz; a.e[io] = R(0.5) * e; a.rhobar[io] = rb; }
            else           { a.fx[io] += fx; a.fy[io] += fy; a.fz[io] += fz; }
         }
      }
      __builtin_amdgcn_wave_barrier();
   }
}

static inline size_t eamNlLdsBytes(int step, int nRho, int nPhi, bool sameGrid, int stencilAtoms, bool spline)
{
   const int rec = step == 3 ? 4 : 3;
   const size_t tableWords = spline ? 0 : step == 1 ? (size_t)2 * (nRho + 3) + (sameGrid ? 0 : (nPhi + 3 - (nRho + 3))) : (size_t)(nRho + 3);
   return eamTableBytesAligned(tableWords) + (size_t)EAM_NL_WAVES * eamNlWaveBytes(rec, stencilAtoms);
}

// ====================================================================================================================
// EAM_Force_cta_cell (method cta_cell, no Verlet lists): the list kernel above with the list built on the fly.
//
// Round 1's cta_cell kernel tested one atom
// pair against 64 candidates per instruction and compacted the hits with ballot + mbcnt into an LDS queue: per pair a chain of
// LDS read -> arithmetic -> ballot -> LDS write -> LDS gather -> table gather -> cross-lane reduction, ~4000 cycles long, with three
// waves per SIMD to hide it -- VALU 44 % busy, 47 % of the wave time parked (profiles/r01_summary.md).  Here a wave owns a cell the
// way EAM_Force_nl_lds does: the 27-cell stencil is staged in the wave's LDS slice, FOUR LANES serve one atom, 16 atoms per round.
//   build : two atoms at a time, all 64 lanes test 64 staged records per trip (one read of a record serves both atoms); ballot + mbcnt
//           append the hits in record order to the atom's row of a [16][rows] LDS table -- the atom's in-cutoff neighbours, exactly;
//   force : lane q evaluates rows q, q+4, ... two per trip, branch-free; quad-permute DPP adds the four partials;
//   pipe  : while cell c is built and evaluated out of the LDS, the records of c+1 are in flight into registers and the
//           description of c+2 (two dependent reads) is on its way.
// No cross-lane traffic beyond the quad, no queue shared by 64 lanes, and per lane long runs of independent work.
#define EAM_ROW_WORDS 128                  // hand-over pass 1 -> pass 3: per atom [16 lanes][8 trips] words of two 16-bit record numbers
__host__ __device__ static inline size_t eamCtaWaveBytes(int rec, int stencilAtoms, int rows)
{
   // records (+ F' in pass 3), [32] offsets + [32] cells; pass 1 (rows > 0) adds [16 atoms][rows] 16-bit record numbers and [16] row lengths
   return ((size_t)rec * stencilAtoms * sizeof(real_t) + 256 + (rows > 0 ? (size_t)16 * rows * 2 + 64 : 0) + 15) & ~(size_t)15;
}

template <int STEP, bool LDS_TABLES, bool SPLINE>
__global__ __launch_bounds__(512)
void EAM_Force_cta_cell(EamArgs a, int stencilAtoms, int rows, unsigned* __restrict__ rowsG, unsigned short* __restrict__ rowCountG,
                        int fuseEmbed, int* __restrict__ status)
{
   // atoms per round: a lane's share of a row is at most 8 trips x 2 = 16 entries, so an atom needs ceil(rows / 16) lanes
   // (rows <= 64, funcfl Cu: 4 lanes, 16 atoms per round; Mishin's longer cutoff, rows 88: 6 lanes, 10 atoms per round)
   const int roundAtoms = 64 / ((rows + 15) / 16) < 16 ? 64 / ((rows + 15) / 16) : 16;
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   constexpr int REC = 3;                                    // doubles per staged atom: x, y, z (24-byte stride: conflict-free wave-wide reads);
   constexpr int RECD = (STEP == 3) ? 1 : 0;                 // pass 3 keeps F' of the staged atoms in an array of its own
   const int nRhoPad = a.rho.n + 3;
   const int wavesPerBlock = blockDim.x >> 6;
   const bool sameGrid = (STEP == 1) && LDS_TABLES && a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   real_t* sRho = (real_t*)ldsRaw;
   real_t* sPhi = sRho + nRhoPad;
   int tableDoubles = 0;
   if (LDS_TABLES) {
      tableDoubles = (STEP == 1) ? 2 * nRhoPad + (sameGrid ? 0 : (a.phi.n + 3 - nRhoPad)) : nRhoPad;
      if (sameGrid) {
         for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) { sRho[2 * t] = a.phi.values[t]; sRho[2 * t + 1] = a.rho.values[t]; }
      } else {
         for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) sRho[t] = a.rho.values[t];
         if (STEP == 1) for (int t = threadIdx.x; t < a.phi.n + 3; t += blockDim.x) sPhi[t] = a.phi.values[t];
      }
      __syncthreads();
   }
   const TableView rhoT = makeTable(a.rho, LDS_TABLES ? sRho : a.rho.values), phiT = makeTable(a.phi, LDS_TABLES ? sPhi : a.phi.values);

   const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
   real_t* __restrict__ sp = (real_t*)(ldsRaw + eamTableBytesAligned(tableDoubles) + (size_t)wave * eamCtaWaveBytes(REC + RECD, stencilAtoms, STEP == 1 ? rows : 0));
   real_t* __restrict__ sd = sp + REC * stencilAtoms;        // [stencilAtoms] F' (pass 3)
   int* sOff = (int*)(sp + (REC + RECD) * stencilAtoms);     // [32]: exclusive record offsets of the 27 cells, [27] = total
   int* sBox = sOff + 32;                                    // [32]
   unsigned short* sHit = (unsigned short*)(sBox + 32);      // pass 1 only: [16][rows] rows under construction
   int* sCnt = (int*)(sHit + 16 * rows);                     // pass 1 only: [16] neighbours of the round's atoms

   // cells are dealt to waves in contiguous runs so that neighbouring cells (shared stencil lines) meet in one L2
   const int nWaves = gridDim.x * wavesPerBlock;
   const int gw = xcdRemap(blockIdx.x, gridDim.x) * wavesPerBlock + wave;
   const int per = (a.nCells + nWaves - 1) / nWaves;
   const int ciBegin = gw * per;
   const int ciEnd = (gw + 1) * per < a.nCells ? (gw + 1) * per : a.nCells;
   constexpr int SR = 6;                                     // staged records a lane keeps in flight (384 per wave; larger stencils: a second, blocking trip)

   auto cellOf = [&](int ci) { return a.cells ? a.cells[ci] : ci; };
   int boxB = 0, cntB = 0, boxA = 0, cntA = 0;               // descriptions of the cell after next / of the cell being loaded
   if (ciBegin < ciEnd)     { const int c = cellOf(ciBegin);     boxA = lane < 27 ? a.nbr[(size_t)c * 27 + lane] : 0; cntA = lane < 27 ? a.nAtoms[boxA] : 0; }
   if (ciBegin + 1 < ciEnd) { const int c = cellOf(ciBegin + 1); boxB = lane < 27 ? a.nbr[(size_t)c * 27 + lane] : 0; cntB = lane < 27 ? a.nAtoms[boxB] : 0; }

   real_t vx[SR], vy[SR], vz[SR], vd[SR];
   // pass 3: the lane's own share of its atom's row, as pass 1 left it -- eight words, one per trip of the evaluation loop, each the two
   // record numbers of that trip -- and the row length; fetched a cell ahead (two 16-byte loads per lane) and kept in registers: no row
   // table in the LDS, which is what lets a third workgroup share the CU in pass 3
   uint4 rowLo = make_uint4(0u, 0u, 0u, 0u), rowHi = rowLo;
   int rowCnt = 0;
   int totalL = 0, niL = 0, iBoxL = 0;
   auto fetchRows = [&](int iBox, int ni, int i0) {
      const int nRound = ni - i0 < roundAtoms ? ni - i0 : roundAtoms;
      int L = 64 / (nRound > 0 ? nRound : 1); if (L > 16) L = 16;
      const int ia = (lane * ((65536 + L - 1) / L)) >> 16;
      const int q = lane - ia * L;
      rowLo = make_uint4(0u, 0u, 0u, 0u); rowHi = rowLo; rowCnt = 0;
      if (ia < nRound) {
         const size_t slot = (size_t)iBox * a.cap + i0 + ia;
         const uint4* __restrict__ w = reinterpret_cast<const uint4*>(rowsG + slot * EAM_ROW_WORDS + q * 8);
         rowLo = w[0]; rowHi = w[1];
         rowCnt = rowCountG[slot];
      }
   };
   auto issueLoads = [&]() {
      {
         int incl = cntA;
#pragma unroll
         for (int d = 1; d < 32; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
         if (lane < 28) { sOff[lane] = incl - cntA; sBox[lane] = boxA; }      // lane 27: cnt = 0 -> sOff[27] = total
      }
      __builtin_amdgcn_wave_barrier();
      totalL = uniform(sOff[27]); niL = uniform(sOff[1]); iBoxL = uniform(boxA);
#pragma unroll
      for (int g = 0; g < SR; ++g) {
         const int t = g * 64 + lane;
         const int tt = t < totalL ? t : 0;
         int lo = 0;                                                    // largest k in [0, 26] with sOff[k] <= tt
#pragma unroll
         for (int step = 16; step >= 1; step >>= 1) { const int m = lo + step; if (m <= 26 && sOff[m] <= tt) lo = m; }
         const size_t o = (size_t)sBox[lo] * a.cap + (tt - sOff[lo]);
         vx[g] = a.rx[o]; vy[g] = a.ry[o]; vz[g] = a.rz[o];
         if (STEP == 3) vd[g] = a.dfEmbed[o];
      }
      if (STEP == 3) fetchRows(iBoxL, niL, 0);
   };
   if (ciBegin < ciEnd) issueLoads();

   bool over = false;
   for (int ci = ciBegin; ci < ciEnd; ++ci) {
      // (a) the loads of cell ci have been issued: land them in the LDS
      const int total = totalL, ni = niL, iBox = iBoxL;
      const bool fits = total <= stencilAtoms;
      uint4 wLo = rowLo, wHi = rowHi;                        // pass 3: rows of the first 16 atoms (prefetched with the records)
      int nMine = rowCnt;
      if (fits) {
#pragma unroll
         for (int g = 0; g < SR; ++g) {
            const int t = g * 64 + lane;
            if (t < total) {
               real_t* r = sp + REC * t;
               r[0] = vx[g]; r[1] = vy[g]; r[2] = vz[g];
               if (STEP == 3) sd[t] = vd[g];
            }
         }
         for (int t = SR * 64 + lane; t < total; t += 64) {     // stencils beyond 384 atoms: blocking trip (sOff/sBox still describe cell ci)
            int lo = 0;
#pragma unroll
            for (int step = 16; step >= 1; step >>= 1) { const int m = lo + step; if (m <= 26 && sOff[m] <= t) lo = m; }
            const size_t o = (size_t)sBox[lo] * a.cap + (t - sOff[lo]);
            real_t* r = sp + REC * t;
            r[0] = a.rx[o]; r[1] = a.ry[o]; r[2] = a.rz[o];
            if (STEP == 3) sd[t] = a.dfEmbed[o];
         }
      }
      __builtin_amdgcn_wave_barrier();

      // pass 1, build of a round's rows: one atom at a time against ALL the wave's records -- lane = record; for round 0 the SR trips' records
      // still sit in the registers they arrived in (the build runs before the next cell's loads reuse them), the atom's position is a broadcast
      // read -- ballot + mbcnt append the hits, in record order, to the atom's row.  (Round 2 first re-read the records from the LDS for every
      // pair of atoms: 18 VALU + 1.5 LDS instructions per atom and trip, now 13 + 0.)  Later rounds (cells of more than 16 atoms) and
      // stencils beyond SR trips read the LDS.
      auto buildRound = [&](const int i0, const bool fromRegs) {
         const int nRound = ni - i0 < roundAtoms ? ni - i0 : roundAtoms;
         for (int pa = 0; pa < nRound; pa += 2) {             // two atoms per sweep: two independent ballot -> count -> write chains
            const int iA = i0 + pa, iB = pa + 1 < nRound ? iA + 1 : iA;
            const real_t xA = sp[REC * iA], yA = sp[REC * iA + 1], zA = sp[REC * iA + 2];
            const real_t xB = sp[REC * iB], yB = sp[REC * iB + 1], zB = sp[REC * iB + 2];
            unsigned short* __restrict__ rowA = sHit + pa * rows;
            unsigned short* __restrict__ rowB = rowA + rows;      // (with an odd atom count the last B repeats A and lands in an unused row)
            int nA = 0, nB = 0;
            int tDone = 0;
            auto sweep = [&](const int t, const real_t px, const real_t py, const real_t pz, const bool first) {
               const real_t ax = xA - px, ay = yA - py, az = zA - pz;
               const real_t bx = xB - px, by = yB - py, bz = zB - pz;
               const real_t r2A = ax * ax + ay * ay + az * az, r2B = bx * bx + by * by + bz * bz;
               // records 0 .. ni-1 are the cell itself (first trip): only they can be the atom (r2 = 0); lanes past the list hold record 0
               const bool liveT = t < total;
               const bool hitA = liveT && r2A <= a.rc2 && (!first || t != iA);
               const bool hitB = liveT && r2B <= a.rc2 && (!first || t != iB);
               const unsigned long long mA = __builtin_amdgcn_ballot_w64(hitA), mB = __builtin_amdgcn_ballot_w64(hitB);
               const int kA = nA + __builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u));
               const int kB = nB + __builtin_amdgcn_mbcnt_hi((unsigned)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB, 0u));
               if (hitA && kA < rows) rowA[kA] = (unsigned short)t;
               if (hitB && kB < rows) rowB[kB] = (unsigned short)t;
               nA += __popcll(mA); nB += __popcll(mB);
            };
            if (fromRegs) {
#pragma unroll
               for (int g = 0; g < SR; ++g)
                  if (g * 64 < total) sweep(g * 64 + lane, vx[g], vy[g], vz[g], g == 0);      // wave-uniform condition
               tDone = SR * 64;
            }
            for (int t0 = tDone; t0 < total; t0 += 64) {
               const int t = t0 + lane;
               const int tt = t < total ? t : 0;
               sweep(t, sp[REC * tt], sp[REC * tt + 1], sp[REC * tt + 2], true);
            }
            if (lane == 0) { sCnt[pa] = nA; sCnt[pa + 1] = nB; }
         }
      };
      if (STEP == 1 && fits) buildRound(0, true);

      // (b) start cell ci+1 (its description arrived during cell ci-1) and ask for the description of ci+2
      if (ci + 1 < ciEnd) {
         boxA = boxB; cntA = cntB;
         if (ci + 2 < ciEnd) { const int c = cellOf(ci + 2); boxB = lane < 27 ? a.nbr[(size_t)c * 27 + lane] : 0; cntB = lane < 27 ? a.nAtoms[boxB] : 0; }
         issueLoads();
      }
      if (!fits) {      // a stencil larger than the LDS slice (small boxes have larger cells): thread-per-atom form, same tables
         eamCellDirect<STEP, SPLINE>(a, iBox, lane, rhoT, phiT, sameGrid, fuseEmbed);
         __builtin_amdgcn_wave_barrier();
         continue;
      }

      // (c) up to 16 atoms per round: build their neighbour rows in the LDS, then evaluate them
      for (int i0 = 0; i0 < ni; i0 += roundAtoms) {
         // LANES PER ATOM: as many as the round's atoms leave room for (FCC Cu at 80^3: cells of 9 atoms -> 7 lanes each, 6 -> 10, 13 or 14 -> 4),
         // so a sparsely filled cell does not idle two lanes out of five; lane = L * atom + q, lane q takes rows q, q + L, ...
         const int nRound = ni - i0 < roundAtoms ? ni - i0 : roundAtoms;
         int L = 64 / nRound; if (L > 16) L = 16;
         const int ia = (lane * ((65536 + L - 1) / L)) >> 16;          // lane / L (exact for lane < 64)
         const int q = lane - ia * L;
         const int i = i0 + ia;
         const bool have = ia < nRound;
         const int ii = have ? i : 0;
         const real_t xi = sp[REC * ii], yi = sp[REC * ii + 1], zi = sp[REC * ii + 2];      // own cell is staged first: record i
         const real_t dfi = (STEP == 3) ? sd[ii] : R(0.0);
         // pass 3 adds to the forces of pass 1: ask for them now, a whole round of arithmetic before they are needed
         const size_t io = (size_t)iBox * a.cap + ii;
         real_t f0x = R(0.0), f0y = R(0.0), f0z = R(0.0);
         if (STEP == 3 && have && q == 0) { f0x = a.fx[io]; f0y = a.fy[io]; f0z = a.fz[io]; }
         unsigned short* __restrict__ myRow = sHit + (have ? ia : 0) * rows;
         if (STEP == 3) {
            // pass 3: the rows pass 1 left behind (same staging order: cells and occupancies do not change inside a force evaluation);
            // cells of more than 16 atoms fetch the later rounds' here (rowLo/rowHi hold the next cell's by now)
            if (i0 != 0) {
               const uint4 keepLo = rowLo, keepHi = rowHi; const int keepCnt = rowCnt;
               fetchRows(iBox, ni, i0);
               wLo = rowLo; wHi = rowHi; nMine = rowCnt;
               rowLo = keepLo; rowHi = keepHi; rowCnt = keepCnt;
            }
         } else {
            if (i0 != 0) buildRound(i0, false);             // (round 0 was built before the next cell's loads went out)
            __builtin_amdgcn_wave_barrier();
         }
         int n = have ? (STEP == 1 ? sCnt[ia] : nMine) : 0;   // in-cutoff neighbours of atom i (the same in its L lanes)
         if (n > rows) { over = true; n = rows; }
         __builtin_amdgcn_wave_barrier();
         if (STEP == 1 && lane < nRound) rowCountG[(size_t)iBox * a.cap + i0 + lane] = (unsigned short)(sCnt[lane] < rows ? sCnt[lane] : rows);
         unsigned* __restrict__ myWords = rowsG + ((size_t)iBox * a.cap + ii) * EAM_ROW_WORDS + q * 8;      // pass 1 writes, trip by trip
         const unsigned wReg[8] = { wLo.x, wLo.y, wLo.z, wLo.w, wHi.x, wHi.y, wHi.z, wHi.w };

         real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0), rb = R(0.0);
         // two pairs per trip, branch-free (a missing second pair is evaluated at r = cutoff and weighted 0)
         auto evalTrip = [&](const int j0, const int j1, const bool h1) {
            const real_t* r0 = sp + REC * j0; const real_t* r1 = sp + REC * j1;
            const real_t dx0 = xi - r0[0], dy0 = yi - r0[1], dz0 = zi - r0[2];
            const real_t dx1 = xi - r1[0], dy1 = yi - r1[1], dz1 = zi - r1[2];
            const real_t s0 = dx0*dx0 + dy0*dy0 + dz0*dz0;
            const real_t s1 = h1 ? dx1*dx1 + dy1*dy1 + dz1*dz1 : a.rc2;
            real_t rho0, drho0, dphi0, rho1, drho1, dphi1;
            if (SPLINE) {                                    // -P: cubic splines in r^2 give (1/r) d/dr directly, no square root
               interpolateSpline(a.rhoS, s0, rho0, drho0); interpolateSpline(a.rhoS, s1, rho1, drho1);
               if (STEP == 1) {
                  real_t phi0, phi1;
                  interpolateSpline(a.phiS, s0, phi0, dphi0); interpolateSpline(a.phiS, s1, phi1, dphi1);
                  e += phi0 + (h1 ? phi1 : R(0.0));
                  rb += rho0 + (h1 ? rho1 : R(0.0));
               } else {
                  dphi0 = (dfi + sd[STEP == 3 ? j0 : 0]) * drho0; dphi1 = (dfi + sd[STEP == 3 ? j1 : 0]) * drho1;
               }
               dphi1 = h1 ? dphi1 : R(0.0);
            } else {
               const real_t ir0 = rsqrtR(s0), ir1 = rsqrtR(s1);
               const real_t d0 = s0 * ir0, d1 = s1 * ir1;
               if (STEP == 1) {
                  real_t phi0, phi1;
                  if (sameGrid) { interpolatePair(sRho, rhoT, d0, phi0, dphi0, rho0, drho0); interpolatePair(sRho, rhoT, d1, phi1, dphi1, rho1, drho1); }
                  else { interpolate(rhoT, d0, rho0, drho0); interpolate(phiT, d0, phi0, dphi0); interpolate(rhoT, d1, rho1, drho1); interpolate(phiT, d1, phi1, dphi1); }
                  e += phi0 + (h1 ? phi1 : R(0.0));
                  rb += rho0 + (h1 ? rho1 : R(0.0));
               } else {
                  interpolate(rhoT, d0, rho0, drho0); interpolate(rhoT, d1, rho1, drho1);
                  dphi0 = (dfi + sd[STEP == 3 ? j0 : 0]) * drho0; dphi1 = (dfi + sd[STEP == 3 ? j1 : 0]) * drho1;
               }
               dphi0 = dphi0 * ir0; dphi1 = h1 ? dphi1 * ir1 : R(0.0);
            }
            fx -= dphi0 * dx0; fy -= dphi0 * dy0; fz -= dphi0 * dz0;
            fx -= dphi1 * dx1; fy -= dphi1 * dy1; fz -= dphi1 * dz1;
         };
         // this lane's rows: q, q + L, q + 2L, ... < n, two per trip (rows <= 64 and L >= 4: at most 8 trips)
         if (STEP == 1) {
            int trip = 0;
            for (int k0 = q; k0 < n; k0 += 2 * L, ++trip) {
               const bool h1 = k0 + L < n;
               const int j0 = myRow[k0], j1 = h1 ? myRow[k0 + L] : ii;
               myWords[trip] = (unsigned)j0 | ((unsigned)j1 << 16);         // pass 3 reads its rows back lane by lane, trip by trip
               evalTrip(j0, j1, h1);
            }
         } else {
#pragma unroll
            for (int trip = 0; trip < 8; ++trip) {           // (unrolled: the words sit in registers)
               const int k0 = q + 2 * L * trip;
               if (k0 < n) {
                  const bool h1 = k0 + L < n;
                  evalTrip((int)(wReg[trip] & 0xffffu), h1 ? (int)(wReg[trip] >> 16) : ii, h1);
               }
            }
         }
         // the L lanes of an atom are consecutive: a shift-down tree adds them into the first (quad-permute DPP when L is 4)
         if (L == 4) {
            fx = quadSum(fx); fy = quadSum(fy); fz = quadSum(fz);
            if (STEP == 1) { e = quadSum(e); rb = quadSum(rb); }
         } else {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
               const bool take = q + d < L;
               const real_t tx = __shfl_down(fx, d), ty = __shfl_down(fy, d), tz = __shfl_down(fz, d);
               fx += take ? tx : R(0.0); fy += take ? ty : R(0.0); fz += take ? tz : R(0.0);
               if (STEP == 1) { const real_t te = __shfl_down(e, d), tr = __shfl_down(rb, d); e += take ? te : R(0.0); rb += take ? tr : R(0.0); }
            }
         }
         if (have && q == 0) {
            if (STEP == 1) {
               real_t ei = R(0.5) * e;
               if (fuseEmbed) {                               // pass 2 for this atom (EAM_Force_embed): needs only its own rhobar
                  real_t F, dF;
                  interpolate(makeTable(a.f, a.f.values), rb, F, dF);
                  a.dfEmbed[io] = dF;
                  ei += F;
               }
               a.fx[io] = fx; a.fy[io] = fy; a.fz[io] = fz; a.e[io] = ei; a.rhobar[io] = rb;
            } else { a.fx[io] = f0x + fx; a.fy[io] = f0y + fy; a.fz[io] = f0z + fz; }
         }
         __builtin_amdgcn_wave_barrier();
      }
   }
   if (__builtin_amdgcn_ballot_w64(over) != 0ull && lane == 0) atomicOr(&status[3], 1);
}

static inline size_t eamCtaCellLdsBytes(int step, int nRho, int nPhi, bool ldsTables, bool sameGrid, int stencilAtoms, int rows, int wavesPerBlock)
{
   const int rec = step == 3 ? 4 : 3;
   size_t tableWords = 0;
   if (ldsTables) tableWords = step == 1 ? (size_t)2 * (nRho + 3) + (sameGrid ? 0 : (nPhi + 3 - (nRho + 3))) : (size_t)(nRho + 3);
   return eamTableBytesAligned(tableWords) + (size_t)wavesPerBlock * eamCtaWaveBytes(rec, stencilAtoms, step == 1 ? rows : 0);
}
