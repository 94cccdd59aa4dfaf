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
   real_t rc2, s6, s6x2, eShift, eps;
};

// One accepted pair.  With w = 1 / r^6:  e_pair = s6 w (s6 w - 1) - eShift,  f_pair = 24 s6 w (2 s6 w - 1) / r^2 * d.
// The constant factors (24 eps s6 on the force, 4 eps * 1/2 on the energy) are applied once per atom by the caller, which leaves
// q = 1/r^8 and g = 2 s6 w - 1 here: eleven VALU operations after v_rcp_f64.
// ENERGY = false drops the energy ops: e[] is only consumed by computeEnergy, i.e. by the last step of a timestep() call.
template <bool ENERGY>
__device__ __forceinline__ void ljPair(real_t dx, real_t dy, real_t dz, real_t r2, const LjArgs& a,
                                       real_t& fx, real_t& fy, real_t& fz, real_t& e)
{
   const real_t ir2 = rcpR(r2);
   const real_t t = ir2 * ir2;
   const real_t w = t * ir2;
   if (ENERGY) { const real_t u = a.s6 * w; e += fmaR(u, u - R(1.0), -a.eShift); }
   const real_t fr = (t * t) * fmaR(w, a.s6x2, -R(1.0));
   fx = fmaR(fr, dx, fx); fy = fmaR(fr, dy, fy); fz = fmaR(fr, dz, fz);
}
#define LJ_FORCE_SCALE(a) (R(24.0) * (a).eps * (a).s6)

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
// Wave candidate lists.  A full wave tests every atom of the 27-cell stencil against its 64 atoms (4000 candidates per atom for
// 5-sigma LJ Cu), but an atom further than the cutoff from the bounding box of those 64 atoms is a miss for every lane.
// LJ_WaveCandidates drops them before the force kernel runs: one wave per (cell, 64-slot chunk), 64 stencil atoms per trip (one
// per lane), point-to-box distance, ballot + mbcnt compaction, global slots appended in stencil order -- the cell's own atoms
// first, so the order of every lane's sum in a FULL wave is that of the plain stencil walk and its forces are bit-identical to it (a tail wave
// deals its list to replicas in parts other than the walk's: same pairs, partial sums in another order).
// The lists are rebuilt by every force call (no skin, nothing carried over): a pruning device, not a Verlet list.
// A list that would not fit its row is marked -1 and the force kernel walks the stencil for that wave.
// One candidate as the force kernel fetches it: the fourth field is rc^2, the same in every record.  Comparing r^2 against the record's copy
// costs nothing (a scalar operand either way) and keeps all 32 bytes live, so hipcc fetches a candidate with ONE s_load_dwordx8 instead of
// an x4 + x2 pair -- the scalar cache is the bottleneck of this path (three s_load_dwordx2 per candidate from the SoA arrays: 5.6 ms).
struct __attribute__((aligned(16))) LjPos4 { real_t x, y, z, rc2; };
struct LjWaveLists {
   unsigned* __restrict__ cand;        // [nLocalCells * wavesMax][candCap] byte offsets of the candidates in pos ((cell * capP + i) * sizeof(LjPos4))
   const LjPos4* __restrict__ pos;     // [nTotalCells * capP] packed records of slots 0 .. capP-1 of every cell (LJ_PackPositions)
   int2*     __restrict__ count;       // [nLocalCells * wavesMax] {candidates of the own cell, all candidates}; y < 0: no list
   int candCap, wavesMax;
   int capP;                           // slots per cell in pos (<= cap: the waves the lists are laid out for); a stencil with a fuller cell gets no lists
   const float4* __restrict__ posF;    // [nTotalCells * capP] the same records in single precision, relative to the corner of the local domain (the list build's copy)
   float rc2BoxF, growF;               // rc^2 and a factor on the half widths of the list build's fp32 boxes, each with a margin for the rounding of the box distance
};

// a list entry is the BYTE offset of the candidate in the packed position array (32 bits: base + zero-extended offset is the s_load
// soffset form, no address arithmetic per candidate)
__device__ __forceinline__ LjPos4 atByte(const LjPos4* __restrict__ base, unsigned byteOff) { return *(const LjPos4*)((const char*)base + byteOff); }

template <bool SELF, bool ENERGY>
__device__ __forceinline__ void ljListTest(const LjArgs& a, const LjPos4& q, real_t xi, real_t yi, real_t zi, real_t& fx, real_t& fy, real_t& fz, real_t& e)
{
   real_t dx = xi - q.x, dy = yi - q.y, dz = zi - q.z;
   real_t r2 = dx*dx + dy*dy + dz*dz;
   bool hit = SELF ? (r2 <= q.rc2 && r2 > R(0.0)) : (r2 <= q.rc2);        // (the record's own copy of rc^2: see LjPos4)
   if (hit) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
}

template <bool SELF, bool ENERGY>
__device__ __forceinline__ void ljListLoop(const LjArgs& a, const LjPos4* __restrict__ pos, const unsigned* __restrict__ L, int p, int pEnd,
                                           real_t xi, real_t yi, real_t zi, real_t& fx, real_t& fy, real_t& fz, real_t& e)
{
   // The list is wave-uniform: eight offsets per s_load_dwordx8, one s_load_dwordx8 per candidate record.  The offsets of batch b + 1 are
   // fetched together with the records of batch b, so a batch costs one exposed scalar-load latency, not two dependent ones.
   // (Records one half batch ahead as well -- two sets of four in flight -- needs 25 SGPR spill reloads per batch: not kept.)
   const int n8 = (pEnd - p) >> 3;
   if (n8 > 0) {
      const int last = p + 8 * (n8 - 1);                      // the prefetch past the last batch re-reads it (valid offsets, unused)
      unsigned id[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) id[u] = L[p + u];
      for (int b = 0; b < n8; ++b, p += 8) {
         LjPos4 q[8];
         unsigned idn[8];
         const int pn = p + 8 < last ? p + 8 : last;
#pragma unroll
         for (int u = 0; u < 8; ++u) q[u] = atByte(pos, id[u]);
#pragma unroll
         for (int u = 0; u < 8; ++u) idn[u] = L[pn + u];
#pragma unroll
         for (int u = 0; u < 8; ++u) ljListTest<SELF, ENERGY>(a, q[u], xi, yi, zi, fx, fy, fz, e);
#pragma unroll
         for (int u = 0; u < 8; ++u) id[u] = idn[u];
      }
   }
   for (; p < pEnd; ++p) ljListTest<SELF, ENERGY>(a, atByte(pos, L[p]), xi, yi, zi, fx, fy, fz, e);
}

// x, y, z of every occupied slot side by side, so that one scalar load fetches a candidate
__global__ __launch_bounds__(256)
void LJ_PackPositions(const real_t* __restrict__ rx, const real_t* __restrict__ ry, const real_t* __restrict__ rz, const int* __restrict__ nAtoms,
                      LjPos4* __restrict__ pos, float4* __restrict__ posF, int cap, int capP, int nCells, real_t rc2, real_t ox, real_t oy, real_t oz)
{
   const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const int c = (int)(t / capP), i = (int)(t - (long)c * capP);
   if (c >= nCells || i >= nAtoms[c]) return;
   const size_t o = (size_t)c * cap + i;
   LjPos4 v; v.x = rx[o]; v.y = ry[o]; v.z = rz[o]; v.rc2 = rc2;
   pos[(size_t)c * capP + i] = v;
   // the same position in single precision, relative to the corner of the local domain: what the LIST BUILD tests against its boxes (a pruning
   // test with a margin for exactly this rounding; the force kernel decides on the fp64 record)
   posF[(size_t)c * capP + i] = make_float4((float)(v.x - ox), (float)(v.y - oy), (float)(v.z - oz), 0.0f);
}

__device__ __forceinline__ real_t waveMinR(real_t v) { for (int d = 1; d < 64; d <<= 1) v = minR(v, __shfl_xor(v, d)); return v; }
__device__ __forceinline__ real_t waveMaxR(real_t v) { for (int d = 1; d < 64; d <<= 1) v = maxR(v, __shfl_xor(v, d)); return v; }
#ifndef COMD_SINGLE
__device__ __forceinline__ float waveMinR(float v) { for (int d = 1; d < 64; d <<= 1) v = __builtin_fminf(v, __shfl_xor(v, d)); return v; }
__device__ __forceinline__ float waveMaxR(float v) { for (int d = 1; d < 64; d <<= 1) v = __builtin_fmaxf(v, __shfl_xor(v, d)); return v; }
#endif

__device__ __forceinline__ double uniformR(double v)
{
   const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(v) & 0xffffffffll));
   const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(v) >> 32));
   return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float uniformR(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// One wave per cell builds the lists of all of the cell's waves in one sweep over the stencil (the sweep is L2 traffic: one per list made
// this kernel bandwidth-bound at 0.39 ms).  Boxes are held as centre + half width: the distance of a point to the box along an axis is
// max(0, |x - c| - h).  The whole test runs in SINGLE precision on the float4 records of LJ_PackPositions (one 16-byte load per stencil atom
// instead of three 8-byte ones, fp32 VALU instructions at twice the fp64 rate); w.rc2BoxF and w.growF carry the margin that covers it.
#define LJ_LIST_CHUNKS 4                      // lists built per sweep (cells of up to 256 atoms need one sweep)
#define LJ_LIST_TRIPS 3                       // 64-atom trips of a stencil cell held in registers
#define LJ_LIST_BATCH 3                       // stencil cells loaded together (27 = 9 batches)
__global__ __launch_bounds__(256)
void LJ_WaveCandidates(LjArgs a, LjWaveLists w, int wavesPerCell)
{
   const int lane = threadIdx.x & 63;
   const int ci = uniform(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
   if (ci >= a.nCells) return;
   const int iBox = uniform(a.cells ? a.cells[ci] : ci);
   const int ni = uniform(a.nAtoms[iBox]);
   int nChunks = (ni + 63) >> 6;                            // waves of the force kernel that look for a list: chunk < wavesPerCell, chunk < wavesMax
   if (nChunks > wavesPerCell) nChunks = wavesPerCell;
   if (nChunks > w.wavesMax) nChunks = w.wavesMax;
   if (nChunks <= 0) return;
   // the stencil's cell ids and occupancies, one per lane, read back with v_readlane (this kernel stores, so hipcc would not use scalar
   // loads for them, and two dependent vector loads per cell in front of the position loads are two exposed latencies per cell)
   const int myBox = lane < 27 ? a.nbr[(size_t)iBox * 27 + lane] : 0;
   const int myCount = lane < 27 ? a.nAtoms[myBox] : 0;
   if (__ballot(myCount > w.capP) != 0ull) {                // a stencil cell holds atoms the packed array has no room for: walk the stencil
      if (lane < nChunks) w.count[iBox * w.wavesMax + lane] = make_int2(0, -1);
      return;
   }
   const float4* __restrict__ P = w.posF;
   const unsigned capP = (unsigned)w.capP;

   for (int c0 = 0; c0 < nChunks; c0 += LJ_LIST_CHUNKS) {
      float cx[LJ_LIST_CHUNKS], cy[LJ_LIST_CHUNKS], cz[LJ_LIST_CHUNKS], hx[LJ_LIST_CHUNKS], hy[LJ_LIST_CHUNKS], hz[LJ_LIST_CHUNKS];
      int n[LJ_LIST_CHUNKS], nSelf[LJ_LIST_CHUNKS];
#pragma unroll
      for (int c = 0; c < LJ_LIST_CHUNKS; ++c) {
         n[c] = 0; nSelf[c] = 0;
         cx[c] = cy[c] = cz[c] = hx[c] = hy[c] = hz[c] = 0.0f;
         if (c0 + c < nChunks) {
            const int iSlot = (c0 + c) * 64 + lane;
            const float4 q = P[(unsigned)iBox * capP + (unsigned)(iSlot < ni ? iSlot : ni - 1)];      // idle lanes shadow the cell's last atom (it is in the last chunk)
            const float xlo = waveMinR(q.x), xhi = waveMaxR(q.x), ylo = waveMinR(q.y), yhi = waveMaxR(q.y), zlo = waveMinR(q.z), zhi = waveMaxR(q.z);
            // half widths rounded up by a factor: the box may only grow
            cx[c] = uniformR(0.5f * (xlo + xhi)); hx[c] = uniformR(0.5f * (xhi - xlo) * w.growF);
            cy[c] = uniformR(0.5f * (ylo + yhi)); hy[c] = uniformR(0.5f * (yhi - ylo) * w.growF);
            cz[c] = uniformR(0.5f * (zlo + zhi)); hz[c] = uniformR(0.5f * (zhi - zlo) * w.growF);
         }
      }
      const int nC = nChunks - c0 < LJ_LIST_CHUNKS ? nChunks - c0 : LJ_LIST_CHUNKS;
      // The stencil in batches of LJ_LIST_BATCH cells: the loads of a batch (up to 64 * LJ_LIST_TRIPS atoms per cell) are issued together, behind
      // the list stores of the batch before.  The stores sit in masked blocks, so hipcc can only wait for the loads with vmcnt(0), which waits
      // for those stores as well: one such stop per BATCH (nine per wave), where a cell-by-cell sweep with the next cell prefetched made 27.
      for (int k0 = 0; k0 < 27; k0 += LJ_LIST_BATCH) {
         float4 q[LJ_LIST_BATCH][LJ_LIST_TRIPS];
         int jb[LJ_LIST_BATCH], njb[LJ_LIST_BATCH];
#pragma unroll
         for (int b = 0; b < LJ_LIST_BATCH; ++b) {
            jb[b] = __builtin_amdgcn_readlane(myBox, k0 + b); njb[b] = __builtin_amdgcn_readlane(myCount, k0 + b);
#pragma unroll
            for (int t = 0; t < LJ_LIST_TRIPS; ++t) {
               const int j = 64 * t + lane;
               if (64 * t < njb[b]) q[b][t] = P[(unsigned)jb[b] * capP + (unsigned)(j < njb[b] ? j : 0)];
            }
         }
#pragma unroll
         for (int b = 0; b < LJ_LIST_BATCH; ++b) {
            const int njNow = njb[b];
            const unsigned base = (unsigned)jb[b] * capP;
            for (int j0 = 0; j0 < njNow; j0 += 64 * LJ_LIST_TRIPS) {
               if (j0 > 0) {                               // fuller cells: the later rounds are loaded on the spot
#pragma unroll
                  for (int t = 0; t < LJ_LIST_TRIPS; ++t) {
                     const int j = j0 + 64 * t + lane;
                     q[b][t] = P[base + (unsigned)(j < njNow ? j : 0)];
                  }
               }
#pragma unroll
               for (int t = 0; t < LJ_LIST_TRIPS; ++t) {
                  const int rem = njNow - (j0 + 64 * t);          // stencil atoms this trip still has (wave-uniform)
                  if (rem <= 0) break;
                  const unsigned long long valid = rem >= 64 ? ~0ull : (1ull << rem) - 1ull;
                  const unsigned entry = (base + (unsigned)(j0 + 64 * t + lane)) * (unsigned)sizeof(LjPos4);
#pragma unroll
                  for (int c = 0; c < LJ_LIST_CHUNKS; ++c) {
                     if (c >= nC) break;                          // (wave-uniform)
                     const float dx = __builtin_fmaxf(0.0f, __builtin_fabsf(q[b][t].x - cx[c]) - hx[c]);
                     const float dy = __builtin_fmaxf(0.0f, __builtin_fabsf(q[b][t].y - cy[c]) - hy[c]);
                     const float dz = __builtin_fmaxf(0.0f, __builtin_fabsf(q[b][t].z - cz[c]) - hz[c]);
                     // the compare's mask IS the ballot; lanes past the cell's last atom (they re-read its slot 0) are cleared on the scalar side
                     const unsigned long long mask = __ballot(dx*dx + dy*dy + dz*dz <= w.rc2BoxF) & valid;
                     // no early-out on a full row: the count keeps running (a row that ends up too long is marked at the end), the store is
                     // what is guarded
                     const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)n[c]));
                     char* __restrict__ L = (char*)(w.cand + (size_t)(iBox * w.wavesMax + c0 + c) * w.candCap);
                     if (__builtin_amdgcn_inverse_ballot_w64(mask) && pos < (unsigned)w.candCap) *(unsigned*)(L + (pos << 2)) = entry;
                     n[c] += __popcll(mask);
                  }
               }
            }
            if (k0 + b == 0) {
#pragma unroll
               for (int c = 0; c < LJ_LIST_CHUNKS; ++c) nSelf[c] = n[c];
            }
         }
      }
#pragma unroll
      for (int c = 0; c < LJ_LIST_CHUNKS; ++c)
         if (c0 + c < nChunks && lane == 0) w.count[iBox * w.wavesMax + c0 + c] = n[c] <= w.candCap ? make_int2(nSelf[c], n[c]) : make_int2(0, -1);
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
      const real_t fs = LJ_FORCE_SCALE(a);
      a.fx[iOff] = tx * fs; a.fy[iOff] = ty * fs; a.fz[iOff] = tz * fs;
      if (ENERGY) a.e[iOff] = te * R(2.0) * a.eps;
   }
}

// The generic chunk with a candidate list: replica g takes the g-th part of the list (parts of a multiple of four entries, so that a lane
// fetches four offsets with one 16-byte load), candidates arrive through per-lane loads of the packed records.
template <bool ENERGY>
__device__ __forceinline__ void ljChunkListed(const LjArgs& a, const LjPos4* __restrict__ pos, const unsigned* __restrict__ L, int nAll,
                                              int iBox, int ni, int chunk, int lane)
{
   const int m = ni - chunk * 64 < 64 ? ni - chunk * 64 : 64;
   const int G = 64 / m < 4 ? 64 / m : 4;
   const int g = lane / m, ai = lane - g * m;
   const bool valid = g < G;
   const size_t iOff = (size_t)iBox * a.cap + chunk * 64 + (valid ? ai : 0);
   const real_t xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
   real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0);
   const int len = (((nAll + G - 1) / G) + 3) & ~3;          // entries per replica
   const int first = valid ? g * len : nAll;
   const int mine = nAll - first < len ? nAll - first : len;  // may be <= 0
   const uint4* __restrict__ L4 = reinterpret_cast<const uint4*>(L + (valid ? first : 0));
   for (int it = 0; it < len; it += 4) {
      const bool on = it < mine;
      const uint4 id = on ? L4[it >> 2] : make_uint4(0u, 0u, 0u, 0u);
      const unsigned ids[4] = { id.x, id.y, id.z, id.w };
      LjPos4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = atByte(pos, it + u < mine ? ids[u] : 0u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const real_t dx = xi - q[u].x, dy = yi - q[u].y, dz = zi - q[u].z;
         const real_t r2 = dx*dx + dy*dy + dz*dz;
         if (it + u < mine && r2 <= a.rc2 && r2 > R(0.0)) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
      }
   }
   real_t tx = fx, ty = fy, tz = fz, te = e;
   for (int r = 1; r < G; ++r) {                       // all lanes take part; only lanes < m keep the result
      const int src = (ai + r * m) & 63;
      tx += bpermuteR(fx, src); ty += bpermuteR(fy, src); tz += bpermuteR(fz, src);
      if (ENERGY) te += bpermuteR(e, src);
   }
   if (lane < m) {
      const real_t fs = LJ_FORCE_SCALE(a);
      a.fx[iOff] = tx * fs; a.fy[iOff] = ty * fs; a.fz[iOff] = tz * fs;
      if (ENERGY) a.e[iOff] = te * R(2.0) * a.eps;
   }
}

// ---------------------------------------------------------------------------------------------------
// thread per atom, wave per 64-slot chunk of a cell.  Requires cap % 64 == 0.
// grid: one workgroup of `wavesPerCell` waves per cell, where wavesPerCell = ceil((largest occupancy + slack) / 64) as
// last seen by the host (SimGpu.max_atoms_cell) -- 3 waves for 5-sigma LJ Cu instead of cap/64 = 4, so no wave is born dead.
// A cell that outgrew that estimate is still complete: its waves take the extra chunks through the generic path.
// LISTED: the full waves read the candidates LJ_WaveCandidates left for them instead of walking the 27 cells.
template <bool ENERGY, bool LISTED>
__global__ __launch_bounds__(256)
void LJ_Force_thread_atom(LjArgs a, int wavesPerCell, LjWaveLists w)
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

   int nSelf = 0, nAll = -1;                                                // this wave's candidate list, if it has one
   if (LISTED && chunk < w.wavesMax) {
      const int2 c = w.count[iBox * w.wavesMax + chunk];
      nSelf = uniform(c.x); nAll = uniform(c.y);
   }
   const unsigned* __restrict__ L = LISTED ? w.cand + (size_t)(iBox * w.wavesMax + chunk) * w.candCap : nullptr;
   if (m <= 32) {
      if (LISTED && nAll >= 0) ljChunkListed<ENERGY>(a, w.pos, L, nAll, iBox, ni, chunk, lane);
      else                     ljChunkGeneric<ENERGY>(a, iBox, ni, chunk, lane);
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
      if (LISTED && nAll >= 0) {
         ljListLoop<true, ENERGY>(a, w.pos, L, 0, nSelf, xi, yi, zi, fx, fy, fz, e);
         ljListLoop<false, ENERGY>(a, w.pos, L, nSelf, nAll, xi, yi, zi, fx, fy, fz, e);
      } else {
         ljCellLoop<true, ENERGY>(a, iBox, xi, yi, zi, fx, fy, fz, e);
         for (int k = 1; k < 27; ++k) ljCellLoop<false, ENERGY>(a, uniform(nb[k]), xi, yi, zi, fx, fy, fz, e);
      }
      if (active) {
         const real_t fs = LJ_FORCE_SCALE(a);
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
                                         unsigned* __restrict__ plWords, real_t plCut2, const bool (&own)[2], const int jStart = 0, const int jStride = 8)
{
   // eight neighbours per trip: the twelve 16-byte LDS reads are issued together, then evaluated (the slab is padded to a multiple of 8).
   // jStart / jStride: the lanes of a replicated wave take every jStride / 8-th trip (then nSlab is a multiple of jStride and the reads are
   // per-lane instead of broadcasts); the default is the wave-wide walk.
   unsigned word = 0;
   for (int j = jStart; j < nSlab; j += jStride) {
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
   const real_t fs = LJ_FORCE_SCALE(a);
#pragma unroll
   for (int u = 0; u < 2; ++u)
      if (own[u]) {
         const size_t io = (size_t)iBox * a.cap + threadIdx.x + u * nThreads;
         a.fx[io] = fx[u] * fs; a.fy[io] = fy[u] * fs; a.fz[io] = fz[u] * fs;
         if (ENERGY) a.e[io] = e[u] * R(2.0) * a.eps;
      }
}

// ---------------------------------------------------------------------------------------------------
// CTA per link cell, the default form (no -L): every WAVE of the workgroup stages its own candidates.  A wave's 64 atoms only interact with
// the stencil atoms within the cutoff of their bounding box (about 59 % of the 27 cells for a slab of gid-ordered atoms, see
// LJ_WaveCandidates), so the wave streams the stencil through a point-to-box test -- 64 stencil atoms per trip, one per lane -- and appends
// the survivors (ballot + mbcnt, stencil order) to a private LDS region; whenever the region is full, and at the end of the own cell (its
// atoms need the r2 > 0 guard), the region is consumed with the broadcast loop of the slab kernel (eight neighbours per trip).  No barrier:
// the waves of a workgroup never wait for each other.  (The slab kernel below staged all 4000 stencil atoms for all waves, behind two
// barriers per slab: 4.77 ms at 80^3.)
#define LJ_CTA_WAVE_RECORDS 440            // candidates a wave can hold: 3 waves x 3 x 440 x 8 B = 31 KB per workgroup, five workgroups per CU
static inline size_t ljCtaBoxesLdsBytes(int threads) { return (size_t)(threads / 64) * 3 * LJ_CTA_WAVE_RECORDS * sizeof(real_t); }

template <bool ENERGY>
__global__ __launch_bounds__(256)
void LJ_Force_cta_cell_boxes(LjArgs a, real_t rc2Box, real_t grow)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   real_t* wx = (real_t*)ldsRaw + (size_t)wave * 3 * LJ_CTA_WAVE_RECORDS;
   real_t* wy = wx + LJ_CTA_WAVE_RECORDS;
   real_t* wz = wy + LJ_CTA_WAVE_RECORDS;

   const int ci = xcdRemap(blockIdx.x, gridDim.x);
   const int iBox = a.cells ? a.cells[ci] : ci;
   const int ni = uniform(a.nAtoms[iBox]);
   const int nThreads = blockDim.x;
   if (wave * 64 >= ni) return;                              // (cells of more than blockDim atoms: the wave's second atoms are wave * 64 + nThreads ...)

   // each thread owns up to two atoms (cells of up to 2 * blockDim atoms): t and t + blockDim
   real_t xi[2], yi[2], zi[2], fx[2] = {R(0.0), R(0.0)}, fy[2] = {R(0.0), R(0.0)}, fz[2] = {R(0.0), R(0.0)}, e[2] = {R(0.0), R(0.0)};
   bool own[2];
   const int iLast = wave * 64 + 63 < ni ? wave * 64 + 63 : ni - 1;          // idle lanes shadow an atom of the wave: the box only sees real atoms
#pragma unroll
   for (int u = 0; u < 2; ++u) {
      const int i = threadIdx.x + u * nThreads;
      own[u] = i < ni;
      const size_t io = (size_t)iBox * a.cap + (own[u] ? i : iLast);
      xi[u] = a.rx[io]; yi[u] = a.ry[io]; zi[u] = a.rz[io];
   }
   const bool second = uniform(ni > nThreads + wave * 64 ? 1 : 0) != 0;      // this wave has second atoms
   // an under-filled wave (the tail of a cell: 20 of 64 lanes at 148 atoms) replicates its atoms G times across the lanes and replica g
   // takes every G-th trip of the staged candidates; the replicas are added with ds_bpermute at the end
   const int m = ni - wave * 64 < 64 ? ni - wave * 64 : 64;
   const bool replicated = !second && m <= 32;
   const int G = replicated ? (64 / m < 4 ? 64 / m : 4) : 1;
   const int g = replicated ? lane / m : 0, ai = replicated ? lane - g * m : lane;
   const bool validRep = g < G;
   if (replicated) {
      const size_t io = (size_t)iBox * a.cap + wave * 64 + (validRep ? ai : 0);
      xi[0] = a.rx[io]; yi[0] = a.ry[io]; zi[0] = a.rz[io];
      xi[1] = xi[0]; yi[1] = yi[0]; zi[1] = zi[0];
      own[0] = validRep; own[1] = false;
   }
   const int jStart = 8 * (validRep ? g : 0), jStride = 8 * G;
   // the wave's bounding box as centre + half width (grown by a rounding margin)
   const real_t xlo = waveMinR(minR(xi[0], xi[1])), xhi = waveMaxR(maxR(xi[0], xi[1]));
   const real_t ylo = waveMinR(minR(yi[0], yi[1])), yhi = waveMaxR(maxR(yi[0], yi[1]));
   const real_t zlo = waveMinR(minR(zi[0], zi[1])), zhi = waveMaxR(maxR(zi[0], zi[1]));
   const real_t cx = uniformR(R(0.5) * (xlo + xhi)), hx = uniformR(R(0.5) * (xhi - xlo) * grow);
   const real_t cy = uniformR(R(0.5) * (ylo + yhi)), hy = uniformR(R(0.5) * (yhi - ylo) * grow);
   const real_t cz = uniformR(R(0.5) * (zlo + zhi)), hz = uniformR(R(0.5) * (zhi - zlo) * grow);

   LjPairlist noPl; noPl.words = nullptr; noPl.wavesMax = 0; noPl.plCut2 = R(0.0);
   int n = 0;
   auto consume = [&](const bool self) {
      if (n == 0) return;
      const int nPad = ((n + jStride - 1) / jStride) * jStride;                 // whole trips for every replica (jStride <= 32)
      if (lane < 32 && n + lane < nPad) { wx[n + lane] = FAR_AWAY; wy[n + lane] = FAR_AWAY; wz[n + lane] = FAR_AWAY; }
      __builtin_amdgcn_wave_barrier();
      if (self) {
         if (!second) slabLoop<1, 0, true, ENERGY>(wx, wy, wz, nPad, a, xi, yi, zi, fx, fy, fz, e, nullptr, noPl.plCut2, own, jStart, jStride);
         else         slabLoop<2, 0, true, ENERGY>(wx, wy, wz, nPad, a, xi, yi, zi, fx, fy, fz, e, nullptr, noPl.plCut2, own, jStart, jStride);
      } else {
         if (!second) slabLoop<1, 0, false, ENERGY>(wx, wy, wz, nPad, a, xi, yi, zi, fx, fy, fz, e, nullptr, noPl.plCut2, own, jStart, jStride);
         else         slabLoop<2, 0, false, ENERGY>(wx, wy, wz, nPad, a, xi, yi, zi, fx, fy, fz, e, nullptr, noPl.plCut2, own, jStart, jStride);
      }
      __builtin_amdgcn_wave_barrier();
      n = 0;
   };

   const int myBox = lane < 27 ? a.nbr[(size_t)iBox * 27 + lane] : 0;
   const int myCount = lane < 27 ? a.nAtoms[myBox] : 0;
   for (int k = 0; k < 27; ++k) {
      const int jBox = __builtin_amdgcn_readlane(myBox, k), nj = __builtin_amdgcn_readlane(myCount, k);
      const size_t base = (size_t)jBox * a.cap;
      for (int j0 = 0; j0 < nj; j0 += 256) {
         real_t x[4], y[4], z[4];
#pragma unroll
         for (int t = 0; t < 4; ++t) {
            const int j = j0 + 64 * t + lane;
            if (j0 + 64 * t < nj) { const size_t o = base + (j < nj ? j : 0); x[t] = a.rx[o]; y[t] = a.ry[o]; z[t] = a.rz[o]; }
         }
#pragma unroll
         for (int t = 0; t < 4; ++t) {
            const int j = j0 + 64 * t + lane;
            if (j0 + 64 * t >= nj) break;
            const real_t dx = maxR(R(0.0), absR(x[t] - cx) - hx);
            const real_t dy = maxR(R(0.0), absR(y[t] - cy) - hy);
            const real_t dz = maxR(R(0.0), absR(z[t] - cz) - hz);
            const bool keep = j < nj && dx*dx + dy*dy + dz*dz <= rc2Box;
            const unsigned long long mask = __ballot(keep);
            const int add = __popcll(mask);
            if (n + add > LJ_CTA_WAVE_RECORDS - 32) consume(k == 0);     // (room for the padding to whole trips of every replica)
            const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (keep) { wx[pos] = x[t]; wy[pos] = y[t]; wz[pos] = z[t]; }
            n += add;
         }
      }
      if (k == 0) consume(true);                             // the own cell is done: everything after it needs no r2 > 0 guard
   }
   consume(false);

   const real_t fs = LJ_FORCE_SCALE(a);
   if (replicated) {
      real_t tx = fx[0], ty = fy[0], tz = fz[0], te = e[0];
      for (int r = 1; r < G; ++r) {                       // all lanes take part; only lanes < m keep the result
         const int src = (ai + r * m) & 63;
         tx += bpermuteR(fx[0], src); ty += bpermuteR(fy[0], src); tz += bpermuteR(fz[0], src);
         if (ENERGY) te += bpermuteR(e[0], src);
      }
      if (lane < m) {
         const size_t io = (size_t)iBox * a.cap + wave * 64 + lane;
         a.fx[io] = tx * fs; a.fy[io] = ty * fs; a.fz[io] = tz * fs;
         if (ENERGY) a.e[io] = te * R(2.0) * a.eps;
      }
      return;
   }
#pragma unroll
   for (int u = 0; u < 2; ++u)
      if (own[u]) {
         const size_t io = (size_t)iBox * a.cap + threadIdx.x + u * nThreads;
         a.fx[io] = fx[u] * fs; a.fy[io] = fy[u] * fs; a.fz[io] = fz[u] * fs;
         if (ENERGY) a.e[io] = e[u] * R(2.0) * a.eps;
      }
}
