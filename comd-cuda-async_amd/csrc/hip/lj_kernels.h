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
//                The under-filled tail wave of a cell replicates its atoms across the lanes, one stencil subset per replica.
//  cta_cell    : one workgroup per link cell, one thread per atom (the reference's shape); the stencil cells are staged
//                in LDS three at a time and read back as wave-wide broadcasts, eight neighbours per trip.
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
// CTA per link cell (the reference's LJ_Force_cta_cell shape, gpu_lj_cta_cell.h:29-122): one workgroup per cell, one thread
// per atom, neighbour positions staged in LDS and read back as wave-wide broadcasts.  The 27 stencil cells are staged in nine
// slabs of three cells (<= 3 * cap atoms, 3 x 8 B each = 18 KB at cap 256), small enough that LDS never limits how many workgroups share
// a CU; the reference stages 128 atoms at a time behind two barriers per tile, here a slab costs two barriers per ~440 atoms.
// Inside a slab every lane walks the same j sequence, so each ds_read_b128 (two neighbours per read) is a broadcast.
#define LJ_CTA_CELLS     3                 // stencil cells staged per slab: 27 / 3 = 9 slabs, 13 KB of LDS -> the CU fills up with workgroups
static inline size_t ljCtaLdsBytes(int cap) { return (size_t)3 * (LJ_CTA_CELLS * cap + 8) * 8 + 16 * 4; }   // slab capacity = LJ_CTA_CELLS * cap atoms

// Pairlists (-L; the reference's LJ_Force_cta_cell_pairlist, gpu_lj_cta_cell.h:124-274): one bit per (wave, 8-neighbour trip) says
// whether ANY atom of the wave is within cutoff + skin of ANY of the trip's eight neighbours.  The bits are generated by the force
// call that follows a rebuild (PL = 1) and let every later call (PL = 2) skip the dead trips: no LDS reads, no distance tests.
// They stay valid as long as the Verlet lists would (no atom moved more than skin/2; slots frozen in between).
// words[((cell * wavesMax + wave) * LJ_CTA_SLABS + slab) * LJ_PL_WORDS + trip/32]
#define LJ_CTA_SLABS 9
#define LJ_PL_WORDS  8                     // 256 trips = 2048 staged atoms per slab at most (3 cells of <= 512)
struct LjPairlist { unsigned* __restrict__ words; int wavesMax; double plCut2; };

// all lanes read the same neighbour pair (LDS broadcast); NA = atoms per thread
template <int NA, int PL, bool SELF, bool ENERGY>
__device__ __forceinline__ void slabLoop(const double* sx, const double* sy, const double* sz, int nSlab, const LjArgs& a,
                                         const double (&xi)[2], const double (&yi)[2], const double (&zi)[2],
                                         double (&fx)[2], double (&fy)[2], double (&fz)[2], double (&e)[2],
                                         unsigned* __restrict__ plWords, double plCut2, const bool (&own)[2])
{
   // eight neighbours per trip: the twelve 16-byte LDS reads are issued together, then evaluated (the slab is padded to a multiple of 8)
   unsigned word = 0;
   for (int j = 0; j < nSlab; j += 8) {
      const int trip = j >> 3;
      if (PL == 2) {
         if ((trip & 31) == 0) word = (unsigned)__builtin_amdgcn_readfirstlane((int)plWords[trip >> 5]);
         if (!((word >> (trip & 31)) & 1u)) continue;
      }
      double2 X[4], Y[4], Z[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
         X[v] = *reinterpret_cast<const double2*>(sx + j + 2 * v);
         Y[v] = *reinterpret_cast<const double2*>(sy + j + 2 * v);
         Z[v] = *reinterpret_cast<const double2*>(sz + j + 2 * v);
      }
      bool near = false;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
#pragma unroll
         for (int u = 0; u < NA; ++u) {
            {
               const double dx = xi[u] - X[v].x, dy = yi[u] - Y[v].x, dz = zi[u] - Z[v].x;
               const double r2 = dx*dx + dy*dy + dz*dz;
               if (PL == 1) near = near || (own[u] && r2 <= plCut2);
               if (SELF ? (r2 <= a.rc2 && r2 > 0.0) : (r2 <= a.rc2)) ljPair<ENERGY>(dx, dy, dz, r2, a, fx[u], fy[u], fz[u], e[u]);
            }
            {
               const double dx = xi[u] - X[v].y, dy = yi[u] - Y[v].y, dz = zi[u] - Z[v].y;
               const double r2 = dx*dx + dy*dy + dz*dz;
               if (PL == 1) near = near || (own[u] && r2 <= plCut2);
               if (SELF ? (r2 <= a.rc2 && r2 > 0.0) : (r2 <= a.rc2)) ljPair<ENERGY>(dx, dy, dz, r2, a, fx[u], fy[u], fz[u], e[u]);
            }
         }
      }
      if (PL == 1) {
         if (__ballot(near) != 0ull) word |= 1u << (trip & 31);
         if ((trip & 31) == 31 || j + 8 >= nSlab) { if ((threadIdx.x & 63) == 0) plWords[trip >> 5] = word; word = 0; }
      }
   }
}

// ENERGY = false drops the energy arithmetic (comdSetEnergyNeeded); only the first slab holds the cell itself and needs the r2 > 0 guard
template <int PL, bool ENERGY>
__global__ __launch_bounds__(256)
void LJ_Force_cta_cell(LjArgs a, int* __restrict__ status, LjPairlist pl)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   const int slabCap = LJ_CTA_CELLS * a.cap + 8;            // cannot overflow: a cell never holds more than cap atoms
   double* sx = (double*)ldsRaw;
   double* sy = sx + slabCap;
   double* sz = sy + slabCap;
   int* sOff = (int*)(sz + slabCap);                        // offsets of the slab's cells

   const int ci = xcdRemap(blockIdx.x, gridDim.x);
   const int iBox = a.cells ? a.cells[ci] : ci;
   const int ni = a.nAtoms[iBox];
   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
   const int nThreads = blockDim.x;

   // each thread owns up to two atoms (cells of up to 2 * blockDim atoms): t and t + blockDim
   double xi[2], yi[2], zi[2], fx[2] = {0.0, 0.0}, fy[2] = {0.0, 0.0}, fz[2] = {0.0, 0.0}, e[2] = {0.0, 0.0};
   bool own[2];
#pragma unroll
   for (int u = 0; u < 2; ++u) {
      const int i = threadIdx.x + u * nThreads;
      own[u] = i < ni;
      const size_t io = (size_t)iBox * a.cap + (own[u] ? i : 0);
      xi[u] = a.rx[io]; yi[u] = a.ry[io]; zi[u] = a.rz[io];
   }
   const bool second = ni > nThreads;                        // workgroup-uniform

   for (int slab = 0; slab < 27 / LJ_CTA_CELLS; ++slab) {
      __syncthreads();                                       // previous slab fully consumed
      if (threadIdx.x < 64) {
         const int lane = threadIdx.x;
         const int cnt = lane < LJ_CTA_CELLS ? a.nAtoms[nb[slab * LJ_CTA_CELLS + lane]] : 0;
         int incl = cnt;
#pragma unroll
         for (int d = 1; d < 16; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
         if (lane <= LJ_CTA_CELLS) sOff[lane] = incl - cnt;    // lane LJ_CTA_CELLS: cnt = 0 -> total
      }
      __syncthreads();
      const int nSlab = sOff[LJ_CTA_CELLS];
      for (int q = 0; q < LJ_CTA_CELLS; ++q) {
         const int jBox = nb[slab * LJ_CTA_CELLS + q];
         const int off = sOff[q], nj = sOff[q + 1] - off;
         for (int j = threadIdx.x; j < nj; j += nThreads) {
            const size_t o = (size_t)jBox * a.cap + j;
            sx[off + j] = a.rx[o]; sy[off + j] = a.ry[o]; sz[off + j] = a.rz[o];
         }
      }
      if (threadIdx.x < 8 && nSlab + threadIdx.x < ((nSlab + 7) & ~7)) {          // pad to a multiple of 8 with far-away points
         sx[nSlab + threadIdx.x] = 1.0e30; sy[nSlab + threadIdx.x] = 1.0e30; sz[nSlab + threadIdx.x] = 1.0e30;
      }
      __syncthreads();

      unsigned* plWords = PL ? pl.words + (((size_t)iBox * pl.wavesMax + (threadIdx.x >> 6)) * LJ_CTA_SLABS + slab) * LJ_PL_WORDS : nullptr;
      if (slab == 0) {                                       // the self-first neighbour table puts the cell itself into slab 0
         if (!second) slabLoop<1, PL, true, ENERGY>(sx, sy, sz, nSlab, a, xi, yi, zi, fx, fy, fz, e, plWords, pl.plCut2, own);
         else         slabLoop<2, PL, true, ENERGY>(sx, sy, sz, nSlab, a, xi, yi, zi, fx, fy, fz, e, plWords, pl.plCut2, own);
      } else {
         if (!second) slabLoop<1, PL, false, ENERGY>(sx, sy, sz, nSlab, a, xi, yi, zi, fx, fy, fz, e, plWords, pl.plCut2, own);       // the common case: one atom per thread
         else         slabLoop<2, PL, false, ENERGY>(sx, sy, sz, nSlab, a, xi, yi, zi, fx, fy, fz, e, plWords, pl.plCut2, own);
      }
   }
   const double fs = 24.0 * a.eps;
#pragma unroll
   for (int u = 0; u < 2; ++u)
      if (own[u]) {
         const size_t io = (size_t)iBox * a.cap + threadIdx.x + u * nThreads;
         a.fx[io] = fx[u] * fs; a.fy[io] = fy[u] * fs; a.fz[io] = fz[u] * fs;
         if (ENERGY) a.e[io] = e[u] * 2.0 * a.eps;
      }
}

// ===================================================================================================================
// LJ_Force_cell_tiles: CTA per link cell, WAVE per atom pair, lanes = candidates.
//
// Why: with a thread per atom (kernels above) the neighbour j is tested by 64 atoms at once and the pair evaluation is issued whenever
// ANY of them is inside the cutoff -- for 5-sigma LJ that is every second candidate, with 14 % of the lanes doing useful work
// (profiles/r01_summary.md).  Here the roles are swapped.  The workgroup stages the stencil, one group of 9 cells at a time, in the
// LDS as {x,y,z} records in TILES of 32 (each stencil cell padded to whole tiles), with the bounding box of every tile.  A wave
// takes two atoms A, B of the cell at a time (lanes 0-31 work for A, 32-63 for B):
//   1. prune : lane = tile; distance from A / B to the tile's box -> 64-bit mask of tiles worth visiting (about half of them);
//   2. test  : per visited tile 32 candidates x 2 atoms, exact fp64 r^2; hits are compacted (ballot + mbcnt) into one 16-bit
//              LDS queue per atom -- the in-cutoff neighbours of the atom, nothing else;
//   3. force : the queues are evaluated 64 pairs per atom and trip at full lane occupancy (positions gathered from the LDS records);
//   4. sum   : v_permlane32_swap + DPP fold the two atoms' lane partials (pairSum), lane 31 / 63 add them to the cell's LDS accumulators.
// Per atom: ~34 tile visits x 32 tests instead of 4000 tests, 554 evaluations in 9 full trips instead of ~2000 mostly empty ones.
// Deterministic: tile order, lane order and the reduction tree are fixed, so runs stay bit-reproducible.
#define LJT_WAVES   8
#define LJT_THREADS (64 * LJT_WAVES)
#define LJT_QCAP    448                 // in-cutoff neighbours of ONE atom inside ONE group of 9 stencil cells (FCC Cu, 5 sigma: ~185 +- 60)
#define LJT_GROUP_CELLS 9

__device__ __forceinline__ int ljtGroupCell(int g, int kk)
{
   // the classes of (dx + dy + dz) mod 3 of the stencil offsets, as indices into the self-first neighbour table: every class mixes near
   // and far cells, so an atom finds a similar share of its neighbours in each (queue capacity, balance between the groups)
   constexpr unsigned char tab[3][9] = { { 1, 6, 8, 12, 0, 15, 19, 21, 26 }, { 2, 4, 9, 10, 14, 16, 20, 22, 24 }, { 3, 5, 7, 11, 13, 17, 18, 23, 25 } };
   return tab[g][kk];
}

struct LjTileGeom { int tilesPerCell; float rcPrune2; };      // tiles a stencil cell may occupy; pruning radius^2 with the fp32 slack

static inline size_t ljTileLdsBytes(int tilesPerCell)
{
   const size_t maxTiles = (size_t)LJT_GROUP_CELLS * tilesPerCell;
   return maxTiles * 32 * 3 * sizeof(double)                 // records
        + (size_t)tilesPerCell * 32 * 4 * sizeof(double)     // accumulators of the cell's own atoms
        + 6 * ((maxTiles + 63) & ~(size_t)63) * sizeof(float)   // tile boxes
        + (size_t)LJT_WAVES * 2 * LJT_QCAP * sizeof(unsigned short);
}

template <bool ENERGY>
__global__ __launch_bounds__(LJT_THREADS)
void LJ_Force_cell_tiles(LjArgs a, LjTileGeom geo, int* __restrict__ status)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   const int maxTiles = LJT_GROUP_CELLS * geo.tilesPerCell;
   const int tbStride = (maxTiles + 63) & ~63;
   double* __restrict__ recs = (double*)ldsRaw;                                   // [maxTiles * 32][3]
   double* __restrict__ acc = recs + (size_t)maxTiles * 32 * 3;                  // [tilesPerCell * 32][4]
   float* __restrict__ tb = (float*)(acc + (size_t)geo.tilesPerCell * 32 * 4);  // [6][tbStride]: min x,y,z, max x,y,z
   unsigned short* __restrict__ qAll = (unsigned short*)(tb + 6 * tbStride);

   const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
   const int wave = uniform(threadIdx.x >> 6);
   unsigned short* __restrict__ q = qAll + (size_t)wave * 2 * LJT_QCAP;         // [2][LJT_QCAP]

   const int ci = xcdRemap(blockIdx.x, gridDim.x);
   const int iBox = uniform(a.cells ? a.cells[ci] : ci);
   int ni = uniform(a.nAtoms[iBox]);
   if (ni == 0) return;                                                          // workgroup-uniform
   const int slotsPerCell = geo.tilesPerCell * 32;
   bool tooMany = ni > slotsPerCell, qOver = false;
   if (tooMany) ni = slotsPerCell;
   for (int t = threadIdx.x; t < slotsPerCell * 4; t += LJT_THREADS) acc[t] = 0.0;
   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;

   for (int g = 0; g < 3; ++g) {
      __syncthreads();                                       // the previous group's records are no longer read (and acc is zeroed)
      // ---- stage group g: every wave works out the tile layout for itself (lanes 0..8 = the 9 cells), tiles are dealt to the waves in pairs
      int cBox = 0, cN = 0, cT0 = 0;
      if (lane < LJT_GROUP_CELLS) { cBox = nb[ljtGroupCell(g, lane)]; cN = a.nAtoms[cBox]; }
      if (cN > slotsPerCell) { tooMany = true; cN = slotsPerCell; }
      const int cT = (cN + 31) >> 5;
      {
         int incl = cT;
#pragma unroll
         for (int d = 1; d < 16; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
         cT0 = incl - cT;
      }
      const int nTiles = uniform(__shfl(cT0 + cT, LJT_GROUP_CELLS - 1));
      for (int tp = wave; 2 * tp < nTiles; tp += LJT_WAVES) {
         const int tile = 2 * tp + half;
         int box = 0, n = 0, t0 = 0;
#pragma unroll
         for (int k = 0; k < LJT_GROUP_CELLS; ++k) {
            const int st = __builtin_amdgcn_readlane(cT0, k);
            if (tile >= st) { box = __builtin_amdgcn_readlane(cBox, k); n = __builtin_amdgcn_readlane(cN, k); t0 = st; }
         }
         const int slot = (tile - t0) * 32 + l31;
         const bool have = tile < nTiles && slot < n;
         double x = 1.0e30, y = 1.0e30, z = 1.0e30;           // padding: never inside a cutoff
         if (have) { const size_t o = (size_t)box * a.cap + slot; x = a.rx[o]; y = a.ry[o]; z = a.rz[o]; }
         if (tile < nTiles) { double* r = recs + (size_t)(tile * 32 + l31) * 3; r[0] = x; r[1] = y; r[2] = z; }
         // box of the tile's real atoms, in fp32 rounded outwards (half-wave butterflies)
         float lo[3] = { have ? (float)x : 3.0e38f, have ? (float)y : 3.0e38f, have ? (float)z : 3.0e38f };
         float hi[3] = { have ? (float)x : -3.0e38f, have ? (float)y : -3.0e38f, have ? (float)z : -3.0e38f };
#pragma unroll
         for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], m)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], m)); }
         if (l31 == 0 && tile < nTiles) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
               tb[c * tbStride + tile] = lo[c] - (fabsf(lo[c]) * 1.2e-7f + 1.0e-30f);
               tb[(3 + c) * tbStride + tile] = hi[c] + (fabsf(hi[c]) * 1.2e-7f + 1.0e-30f);
            }
         }
      }
      __syncthreads();

      // ---- the cell's atoms, two per wave pass
      for (int p = wave; 2 * p < ni; p += LJT_WAVES) {
         const int iA = 2 * p, iB = (2 * p + 1 < ni) ? 2 * p + 1 : iA;          // odd count: the last atom works alone, B shadows it
         const bool two = iB != iA;
         const size_t iOff = (size_t)iBox * a.cap + (half ? iB : iA);
         const double xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
         // 1. prune: lane = tile
         const float xAf = __shfl((float)xi, 0), yAf = __shfl((float)yi, 0), zAf = __shfl((float)zi, 0);
         const float xBf = __shfl((float)xi, 32), yBf = __shfl((float)yi, 32), zBf = __shfl((float)zi, 32);
         int nA = 0, nB = 0;
         for (int tb0 = 0; tb0 < nTiles; tb0 += 64) {
            const int tl = tb0 + lane;
            const int tt = tl < nTiles ? tl : 0;
            const float bx0 = tb[tt], by0 = tb[tbStride + tt], bz0 = tb[2 * tbStride + tt];
            const float bx1 = tb[3 * tbStride + tt], by1 = tb[4 * tbStride + tt], bz1 = tb[5 * tbStride + tt];
            const float ax = fmaxf(fmaxf(bx0 - xAf, xAf - bx1), 0.f), ay = fmaxf(fmaxf(by0 - yAf, yAf - by1), 0.f), az = fmaxf(fmaxf(bz0 - zAf, zAf - bz1), 0.f);
            const float bx = fmaxf(fmaxf(bx0 - xBf, xBf - bx1), 0.f), by = fmaxf(fmaxf(by0 - yBf, yBf - by1), 0.f), bz = fmaxf(fmaxf(bz0 - zBf, zBf - bz1), 0.f);
            const bool want = tl < nTiles && (ax * ax + ay * ay + az * az <= geo.rcPrune2 || bx * bx + by * by + bz * bz <= geo.rcPrune2);
            unsigned long long visit = __ballot(want);
            // 2. test the visited tiles: 32 candidates x {A, B}
            while (visit) {
               const int t = tb0 + (int)__builtin_ctzll(visit);
               visit &= visit - 1;
               const int rec = t * 32 + l31;
               const double* __restrict__ r = recs + (size_t)rec * 3;
               const double dx = xi - r[0], dy = yi - r[1], dz = zi - r[2];
               const double r2 = dx * dx + dy * dy + dz * dz;
               const bool hit = r2 <= a.rc2 && r2 > 0.0;
               const unsigned long long m = __ballot(hit);
               const unsigned mA = (unsigned)m, mB = (unsigned)(m >> 32);
               const int cA = __popc(mA), cB = __popc(mB);
               // queue slot: lanes 0-31 append to A's queue, lanes 32-63 to B's (mbcnt counts A's hits too for them)
               const int k = (half ? LJT_QCAP + nB - cA : nA) + __builtin_amdgcn_mbcnt_hi(mB, __builtin_amdgcn_mbcnt_lo(mA, 0u));
               const int kLimit = half ? 2 * LJT_QCAP : LJT_QCAP;
               if (hit && k < kLimit) q[k] = (unsigned short)rec;
               nA += cA; nB += cB;
            }
         }
         if (nA > LJT_QCAP) { qOver = true; nA = LJT_QCAP; }
         if (nB > LJT_QCAP) { qOver = true; nB = LJT_QCAP; }
         // 3. evaluate the queues densely: 64 pairs of A and 64 pairs of B per trip
         const double xA = __shfl(xi, 0), yA = __shfl(yi, 0), zA = __shfl(zi, 0);
         const double xB = __shfl(xi, 32), yB = __shfl(yi, 32), zB = __shfl(zi, 32);
         double fA[4] = { 0.0, 0.0, 0.0, 0.0 }, fB[4] = { 0.0, 0.0, 0.0, 0.0 };
         const int nMax = nA > nB ? nA : nB;
         for (int b = 0; b < nMax; b += 64) {
            const bool hA = b + lane < nA, hB = b + lane < nB;
            const int jA = hA ? q[b + lane] : 0, jB = hB ? q[LJT_QCAP + b + lane] : 0;
            const double* __restrict__ ra = recs + (size_t)jA * 3;
            const double* __restrict__ rb = recs + (size_t)jB * 3;
            const double ax = xA - ra[0], ay = yA - ra[1], az = zA - ra[2];
            const double bx = xB - rb[0], by = yB - rb[1], bz = zB - rb[2];
            const double sA = hA ? ax * ax + ay * ay + az * az : a.rc2, sB = hB ? bx * bx + by * by + bz * bz : a.rc2;
            const double iA2 = rcp64(sA), iB2 = rcp64(sB);
            const double uA = a.s6 * iA2 * iA2 * iA2, uB = a.s6 * iB2 * iB2 * iB2;
            if (ENERGY) { fA[3] += hA ? __builtin_fma(uA, uA - 1.0, -a.eShift) : 0.0; fB[3] += hB ? __builtin_fma(uB, uB - 1.0, -a.eShift) : 0.0; }
            const double wA = hA ? uA * iA2 * __builtin_fma(uA, 2.0, -1.0) : 0.0, wB = hB ? uB * iB2 * __builtin_fma(uB, 2.0, -1.0) : 0.0;
            fA[0] = __builtin_fma(wA, ax, fA[0]); fA[1] = __builtin_fma(wA, ay, fA[1]); fA[2] = __builtin_fma(wA, az, fA[2]);
            fB[0] = __builtin_fma(wB, bx, fB[0]); fB[1] = __builtin_fma(wB, by, fB[1]); fB[2] = __builtin_fma(wB, bz, fB[2]);
         }
         // 4. lane partials -> totals of A in lane 31, of B in lane 63 -> the cell's accumulators (this wave owns the pair in every group)
#pragma unroll
         for (int v = 0; v < (ENERGY ? 4 : 3); ++v) {
            const double tot = pairSum(fA[v], fB[v]);
            if (lane == 31) acc[iA * 4 + v] += tot;
            if (lane == 63 && two) acc[iB * 4 + v] += tot;
         }
      }
   }
   __syncthreads();
   const double fs = 24.0 * a.eps;
   for (int i = threadIdx.x; i < ni; i += LJT_THREADS) {
      const size_t io = (size_t)iBox * a.cap + i;
      a.fx[io] = acc[i * 4] * fs; a.fy[io] = acc[i * 4 + 1] * fs; a.fz[io] = acc[i * 4 + 2] * fs;
      if (ENERGY) a.e[io] = acc[i * 4 + 3] * 2.0 * a.eps;
   }
   if (tooMany && lane == 0) atomicOr(&status[0], 2);
   if (qOver && lane == 0) atomicOr(&status[3], 4);
}
