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
   const real_t* __restrict__ rx; const real_t* __restrict__ ry; const real_t* __restrict__ rz;
   real_t* __restrict__ fx; real_t* __restrict__ fy; real_t* __restrict__ fz; real_t* __restrict__ e;
   const int* __restrict__ nAtoms;
   const int* __restrict__ nbr;        // [nLocal*27], self first
   const int* __restrict__ cells;      // optional cell list
   int nCells, cap;
   real_t rc2, s6, eShift, eps;
};

// One accepted pair.  With u = s6 / r^6:  e_pair = u (u - 1) - eShift,  f_pair = 24 u (2u - 1) / r^2 * d.
// The constant factors (24 eps on the force, 4 eps * 1/2 on the energy) are applied once per atom by the caller.
// ENERGY = false drops the energy ops: e[] is only consumed by computeEnergy, i.e. by the last step of a timestep() call.
template <bool ENERGY>
__device__ __forceinline__ void ljPair(real_t dx, real_t dy, real_t dz, real_t r2, const LjArgs& a,
                                       real_t& fx, real_t& fy, real_t& fz, real_t& e)
{
   const real_t ir2 = rcpR(r2);
   const real_t u = a.s6 * ir2 * ir2 * ir2;
   if (ENERGY) e += fmaR(u, u - R(1.0), -a.eShift);
   const real_t fr = u * ir2 * fmaR(u, R(2.0), -R(1.0));
   fx = fmaR(fr, dx, fx); fy = fmaR(fr, dy, fy); fz = fmaR(fr, dz, fz);
}

// one neighbour cell against the wave's 64 i atoms; SELF adds the r2 > 0 guard of the own cell
template <bool SELF, bool ENERGY>
__device__ __forceinline__ void ljCellLoop(const LjArgs& a, int jBox, real_t xi, real_t yi, real_t zi,
                                           real_t& fx, real_t& fy, real_t& fz, real_t& e)
{
   const int nj = uniform(a.nAtoms[jBox]);
   const real_t* __restrict__ px = a.rx + (size_t)jBox * a.cap;
   const real_t* __restrict__ py = a.ry + (size_t)jBox * a.cap;
   const real_t* __restrict__ pz = a.rz + (size_t)jBox * a.cap;
   // j is wave-uniform: fetch 8 neighbours per scalar-load batch (3 x s_load_dwordx16), then test them
   int j = 0;
   for (; j + 8 <= nj; j += 8) {
      real_t xs[8], ys[8], zs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { xs[u] = px[j + u]; ys[u] = py[j + u]; zs[u] = pz[j + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
         real_t dx = xi - xs[u], dy = yi - ys[u], dz = zi - zs[u];
         real_t r2 = dx*dx + dy*dy + dz*dz;
         bool hit = SELF ? (r2 <= a.rc2 && r2 > R(0.0)) : (r2 <= a.rc2);
         if (hit) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
      }
   }
   for (; j < nj; ++j) {
      real_t dx = xi - px[j], dy = yi - py[j], dz = zi - pz[j];
      real_t r2 = dx*dx + dy*dy + dz*dz;
      bool hit = SELF ? (r2 <= a.rc2 && r2 > R(0.0)) : (r2 <= a.rc2);
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
   const real_t xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
   real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0);
   for (int t = 0; t * G < 27; ++t) {
      const int k = t * G + g;
      const bool okk = valid && k < 27;
      const int jBox = okk ? nb[k] : iBox;
      const int nj = okk ? a.nAtoms[jBox] : 0;
      const size_t base = (size_t)jBox * a.cap;
      for (int j = 0; __any(j < nj); ++j) {
         if (j < nj) {
            const real_t dx = xi - a.rx[base + j], dy = yi - a.ry[base + j], dz = zi - a.rz[base + j];
            const real_t r2 = dx*dx + dy*dy + dz*dz;
            if (r2 <= a.rc2 && r2 > R(0.0)) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
         }
      }
   }
   real_t tx = fx, ty = fy, tz = fz, te = e;
   for (int r = 1; r < G; ++r) {                       // all lanes take part; only lanes < m keep the result
      const int src = (ai + r * m) & 63;
      tx += bpermuteR(fx, src); ty += bpermuteR(fy, src); tz += bpermuteR(fz, src);
      if (ENERGY) te += bpermuteR(e, src);
   }
   if (lane < m) {
      const real_t fs = R(24.0) * a.eps;
      a.fx[iOff] = tx * fs; a.fy[iOff] = ty * fs; a.fz[iOff] = tz * fs;
      if (ENERGY) a.e[iOff] = te * R(2.0) * a.eps;
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
      const real_t xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
      real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0);
      // (a software-pipelined variant -- scalar loads of batch b+1 issued before batch b is evaluated, 4 neighbours per batch to fit
      // two batches in SGPRs -- measured 9 % slower: 4.30 vs 3.95 ms; the 8-wide batches below rely on the other waves for latency cover)
      ljCellLoop<true, ENERGY>(a, iBox, xi, yi, zi, fx, fy, fz, e);
      for (int k = 1; k < 27; ++k) ljCellLoop<false, ENERGY>(a, uniform(nb[k]), xi, yi, zi, fx, fy, fz, e);
      if (active) {
         const real_t fs = R(24.0) * a.eps;
         a.fx[iOff] = fx * fs; a.fy[iOff] = fy * fs; a.fz[iOff] = fz * fs;
         if (ENERGY) a.e[iOff] = e * R(2.0) * a.eps;          // 4 eps * 1/2 per pair
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
static inline size_t ljCtaLdsBytes(int cap) { return (size_t)3 * (LJ_CTA_CELLS * cap + 8) * sizeof(real_t) + 16 * 4; }   // slab capacity = LJ_CTA_CELLS * cap atoms

// Pairlists (-L; the reference's LJ_Force_cta_cell_pairlist, gpu_lj_cta_cell.h:124-274): one bit per (wave, 8-neighbour trip) says
// whether ANY atom of the wave is within cutoff + skin of ANY of the trip's eight neighbours.  The bits are generated by the force
// call that follows a rebuild (PL = 1) and let every later call (PL = 2) skip the dead trips: no LDS reads, no distance tests.
// They stay valid as long as the Verlet lists would (no atom moved more than skin/2; slots frozen in between).
// words[((cell * wavesMax + wave) * LJ_CTA_SLABS + slab) * LJ_PL_WORDS + trip/32]
#define LJ_CTA_SLABS 9
#define LJ_PL_WORDS  8                     // 256 trips = 2048 staged atoms per slab at most (3 cells of <= 512)
struct LjPairlist { unsigned* __restrict__ words; int wavesMax; real_t plCut2; };

// all lanes read the same neighbour pair (LDS broadcast); NA = atoms per thread
template <int NA, int PL, bool SELF, bool ENERGY>
__device__ __forceinline__ void slabLoop(const real_t* sx, const real_t* sy, const real_t* sz, int nSlab, const LjArgs& a,
                                         const real_t (&xi)[2], const real_t (&yi)[2], const real_t (&zi)[2],
                                         real_t (&fx)[2], real_t (&fy)[2], real_t (&fz)[2], real_t (&e)[2],
                                         unsigned* __restrict__ plWords, real_t plCut2, const bool (&own)[2])
{
   // eight neighbours per trip: the twelve 16-byte LDS reads are issued together, then evaluated (the slab is padded to a multiple of 8)
   unsigned word = 0;
   for (int j = 0; j < nSlab; j += 8) {
      const int trip = j >> 3;
      if (PL == 2) {
         if ((trip & 31) == 0) word = (unsigned)__builtin_amdgcn_readfirstlane((int)plWords[trip >> 5]);
         if (!((word >> (trip & 31)) & 1u)) continue;
      }
      real2 X[4], Y[4], Z[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
         X[v] = *reinterpret_cast<const real2*>(sx + j + 2 * v);
         Y[v] = *reinterpret_cast<const real2*>(sy + j + 2 * v);
         Z[v] = *reinterpret_cast<const real2*>(sz + j + 2 * v);
      }
      bool near = false;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
#pragma unroll
         for (int u = 0; u < NA; ++u) {
            {
               const real_t dx = xi[u] - X[v].x, dy = yi[u] - Y[v].x, dz = zi[u] - Z[v].x;
               const real_t r2 = dx*dx + dy*dy + dz*dz;
               if (PL == 1) near = near || (own[u] && r2 <= plCut2);
               if (SELF ? (r2 <= a.rc2 && r2 > R(0.0)) : (r2 <= a.rc2)) ljPair<ENERGY>(dx, dy, dz, r2, a, fx[u], fy[u], fz[u], e[u]);
            }
            {
               const real_t dx = xi[u] - X[v].y, dy = yi[u] - Y[v].y, dz = zi[u] - Z[v].y;
               const real_t r2 = dx*dx + dy*dy + dz*dz;
               if (PL == 1) near = near || (own[u] && r2 <= plCut2);
               if (SELF ? (r2 <= a.rc2 && r2 > R(0.0)) : (r2 <= a.rc2)) ljPair<ENERGY>(dx, dy, dz, r2, a, fx[u], fy[u], fz[u], e[u]);
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
   real_t* sx = (real_t*)ldsRaw;
   real_t* sy = sx + slabCap;
   real_t* sz = sy + slabCap;
   int* sOff = (int*)(sz + slabCap);                        // offsets of the slab's cells

   const int ci = xcdRemap(blockIdx.x, gridDim.x);
   const int iBox = a.cells ? a.cells[ci] : ci;
   const int ni = a.nAtoms[iBox];
   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
   const int nThreads = blockDim.x;

   // each thread owns up to two atoms (cells of up to 2 * blockDim atoms): t and t + blockDim
   real_t xi[2], yi[2], zi[2], fx[2] = {R(0.0), R(0.0)}, fy[2] = {R(0.0), R(0.0)}, fz[2] = {R(0.0), R(0.0)}, e[2] = {R(0.0), R(0.0)};
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
         sx[nSlab + threadIdx.x] = FAR_AWAY; sy[nSlab + threadIdx.x] = FAR_AWAY; sz[nSlab + threadIdx.x] = FAR_AWAY;
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
   const real_t fs = R(24.0) * a.eps;
#pragma unroll
   for (int u = 0; u < 2; ++u)
      if (own[u]) {
         const size_t io = (size_t)iBox * a.cap + threadIdx.x + u * nThreads;
         a.fx[io] = fx[u] * fs; a.fy[io] = fy[u] * fs; a.fz[io] = fz[u] * fs;
         if (ENERGY) a.e[io] = e[u] * R(2.0) * a.eps;
      }
}
