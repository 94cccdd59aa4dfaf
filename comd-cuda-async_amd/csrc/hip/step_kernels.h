// step_kernels.h -- integrator, energy reduction, link-cell redistribution and halo pack/unpack kernels.
//
// Behaviour of the reference's gpu_timestep.h:31-62 (AdvanceVelocity/AdvancePosition), gpu_reduce.h:32-98
// (ReduceEnergy), gpu_redistribute.h:135-268 (UpdateLinkCells/CompactAtoms), :376-402 (LoadAtomsBufferPacked),
// :499-620 (computeBoxIds/computeOffsets/UnloadAtomsBufferPacked), :638-672 (Load/UnloadForceBuffer) and
// :682-848 (gid sort), restructured for slot-addressed cells (cell*cap + i) on wave64 hardware:
//  - no a_list indirection: kernels run over slots and mask by nAtoms[cell];
//  - moved atoms are appended to their new cell with one atomic, the vacated slot is marked gid = -1, and one
//    workgroup per changed cell squeezes the holes AND restores ascending-gid order in a single pass
//    (rank sort in LDS) -- the reference compacts with one thread per cell and sorts <= 32 atoms per cell;
//  - energy is reduced in two deterministic stages (no fp64 atomics).
#pragma once
#include "device_common.h"

// ---- integrator -------------------------------------------------------------------------------------
// [round 4] 2^laneBits threads of a 256-thread workgroup serve one cell and loop over its atoms (slot = lane, lane + 2^laneBits, ...).  Round 3 ran a thread per
// SLOT: at 14 atoms in 44 slots (EAM with lists) or 10 in 24 two threads in three found nothing to do.  The host picks the lanes from the capacity
// (comd_device.hip integratorLaneBits): a cell fuller than its lanes takes a second trip, no atom is ever skipped.
#define COMD_CELL_SLOTS(laneBits) \
   const int c = (int)blockIdx.x * (256 >> (laneBits)) + ((int)threadIdx.x >> (laneBits)); \
   if (c >= nLocalBoxes) return; \
   for (long tid = (long)c * cap + ((int)threadIdx.x & ((1 << (laneBits)) - 1)), tidEnd = (long)c * cap + nAtoms[c]; tid < tidEnd; tid += (1 << (laneBits)))
__global__ __launch_bounds__(256)
void AdvanceVelocity(real_t* __restrict__ px, real_t* __restrict__ py, real_t* __restrict__ pz,
                     const real_t* __restrict__ fx, const real_t* __restrict__ fy, const real_t* __restrict__ fz,
                     const int* __restrict__ nAtoms, int nLocalBoxes, int cap, real_t dt, int laneBits)
{
   COMD_CELL_SLOTS(laneBits) { px[tid] += dt * fx[tid]; py[tid] += dt * fy[tid]; pz[tid] += dt * fz[tid]; }
}

__global__ __launch_bounds__(256)
void AdvancePosition(real_t* __restrict__ rx, real_t* __restrict__ ry, real_t* __restrict__ rz,
                     const real_t* __restrict__ px, const real_t* __restrict__ py, const real_t* __restrict__ pz,
                     const int* __restrict__ iSpecies, const real_t* __restrict__ speciesMass,
                     const int* __restrict__ nAtoms, int nLocalBoxes, int cap, real_t dt, int laneBits)
{
   COMD_CELL_SLOTS(laneBits) {
      const real_t invMass = R(1.0) / speciesMass[iSpecies[tid]];  // same expression order as timestep.c:168-173
      rx[tid] += dt * px[tid] * invMass; ry[tid] += dt * py[tid] * invMass; rz[tid] += dt * pz[tid] * invMass;
   }
}

// [round 4] Verlet lists: the drift kernels also answer "has an atom moved more than skin/2 since the list build?" (gpu_kernels.cu:1087-1110) for the
// positions they have just written -- the separate pass over r and lastR (40 us at EAM 80^3) is gone.  lastX == NULL: no lists, nothing is checked.
struct SkinCheck { const real_t* lastX; const real_t* lastY; const real_t* lastZ; real_t skinHalf2, softHalf2; int* hard; int* soft; int* progress; int stamp;
                   const int* status; int* statusMirror; };      // (and, always: the device status words mirrored into pinned memory by the first thread -- comdPollStatus reads them there)
// The flags are not booleans: a drift kernel that finds an atom beyond a threshold writes ITS OWN number (drift kernels of a simulation are numbered from 1 and never
// overlap on their stream), so a flag holds the number of the last drift that saw a violation and never has to be cleared -- the host compares it with the number the
// last list build was made at.  hard: (skin / 2)^2, the reference's rule; soft: a little less, for the host that decides two drifts late (comd_device.hip).
// progress: drift kernel G says "drift G - 1 and everything before it is done" as it starts (kernels of a stream do not overlap): how the host knows how far behind
// the device is without an event or a synchronisation (an hipEventRecord behind every drift kernel cost 0.1 ms per step at EAM 80^3).
__device__ __forceinline__ void skinProgress(const SkinCheck& sk)
{
   if (blockIdx.x != 0 || threadIdx.x != 0) return;
   if (sk.lastX) *sk.progress = sk.stamp - 1;
   if (sk.statusMirror) { sk.statusMirror[0] = sk.status[0]; sk.statusMirror[1] = sk.status[1]; sk.statusMirror[2] = sk.status[2]; sk.statusMirror[3] = sk.status[3]; }
}
__device__ __forceinline__ void skinCheck(const SkinCheck& sk, long s, real_t x, real_t y, real_t z)
{
   if (!sk.lastX) return;
   const real_t dx = x - sk.lastX[s], dy = y - sk.lastY[s], dz = z - sk.lastZ[s];
   const real_t d2 = dx*dx + dy*dy + dz*dz;
   if (d2 > sk.softHalf2) { *sk.soft = sk.stamp; if (d2 > sk.skinHalf2) *sk.hard = sk.stamp; }
}

// half kick followed by the drift in one pass (same operations, same order, as AdvanceVelocity then AdvancePosition)
__global__ __launch_bounds__(256)
void AdvanceVelocityPosition(real_t* __restrict__ rx, real_t* __restrict__ ry, real_t* __restrict__ rz,
                             real_t* __restrict__ px, real_t* __restrict__ py, real_t* __restrict__ pz,
                             const real_t* __restrict__ fx, const real_t* __restrict__ fy, const real_t* __restrict__ fz,
                             const int* __restrict__ iSpecies, const real_t* __restrict__ speciesMass,
                             const int* __restrict__ nAtoms, int nLocalBoxes, int cap, real_t dtKick, real_t dtDrift, SkinCheck sk, int laneBits)
{
   skinProgress(sk);
   COMD_CELL_SLOTS(laneBits) {
      const real_t invMass = R(1.0) / speciesMass[iSpecies[tid]];
      const real_t x = px[tid] + dtKick * fx[tid], y = py[tid] + dtKick * fy[tid], z = pz[tid] + dtKick * fz[tid];
      px[tid] = x; py[tid] = y; pz[tid] = z;
      const real_t nx = rx[tid] + dtDrift * x * invMass, ny = ry[tid] + dtDrift * y * invMass, nz = rz[tid] + dtDrift * z * invMass;
      rx[tid] = nx; ry[tid] = ny; rz[tid] = nz;
      skinCheck(sk, tid, nx, ny, nz);
   }
}

// second half kick of one step + first half kick and drift of the next, one pass (same operations, same order, as AdvanceVelocity followed
// by AdvanceVelocityPosition: the two kicks stay two roundings)
__global__ __launch_bounds__(256)
void AdvanceVelocityVelocityPosition(real_t* __restrict__ rx, real_t* __restrict__ ry, real_t* __restrict__ rz,
                                     real_t* __restrict__ px, real_t* __restrict__ py, real_t* __restrict__ pz,
                                     const real_t* __restrict__ fx, const real_t* __restrict__ fy, const real_t* __restrict__ fz,
                                     const int* __restrict__ iSpecies, const real_t* __restrict__ speciesMass,
                                     const int* __restrict__ nAtoms, int nLocalBoxes, int cap, real_t dtKick1, real_t dtKick2, real_t dtDrift, SkinCheck sk, int laneBits)
{
   skinProgress(sk);
   COMD_CELL_SLOTS(laneBits) {
      const real_t invMass = R(1.0) / speciesMass[iSpecies[tid]];
      const real_t gx = fx[tid], gy = fy[tid], gz = fz[tid];
      real_t x = px[tid] + dtKick1 * gx, y = py[tid] + dtKick1 * gy, z = pz[tid] + dtKick1 * gz;
      x += dtKick2 * gx; y += dtKick2 * gy; z += dtKick2 * gz;
      px[tid] = x; py[tid] = y; pz[tid] = z;
      const real_t nx = rx[tid] + dtDrift * x * invMass, ny = ry[tid] + dtDrift * y * invMass, nz = rz[tid] + dtDrift * z * invMass;
      rx[tid] = nx; ry[tid] = ny; rz[tid] = nz;
      skinCheck(sk, tid, nx, ny, nz);
   }
}

// ---- energy: stage 1 = per-block partial sums in a fixed order, stage 2 = one block adds the partials --------
__global__ __launch_bounds__(256)
void ReduceEnergyPartial(const real_t* __restrict__ e, const real_t* __restrict__ px, const real_t* __restrict__ py,
                         const real_t* __restrict__ pz, const int* __restrict__ iSpecies,
                         const real_t* __restrict__ speciesMass, const int* __restrict__ nAtoms,
                         int nLocalBoxes, int cap, real_t* __restrict__ partial)
{
   __shared__ real_t sE[4], sK[4];
   real_t accE = R(0.0), accK = R(0.0);
   const long nSlots = (long)nLocalBoxes * cap;
   for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < nSlots; s += (long)gridDim.x * blockDim.x) {
      const int c = (int)(s / cap);
      if ((int)(s - (long)c * cap) < nAtoms[c]) {
         accE += e[s];
         accK += (px[s]*px[s] + py[s]*py[s] + pz[s]*pz[s]) * (R(0.5) / speciesMass[iSpecies[s]]);
      }
   }
   accE = waveSum(accE); accK = waveSum(accK);
   const int wave = threadIdx.x >> 6;
   if ((threadIdx.x & 63) == 0) { sE[wave] = accE; sK[wave] = accK; }
   __syncthreads();
   if (threadIdx.x == 0) {
      partial[2 * blockIdx.x]     = (sE[0] + sE[1]) + (sE[2] + sE[3]);
      partial[2 * blockIdx.x + 1] = (sK[0] + sK[1]) + (sK[2] + sK[3]);
   }
}

__global__ __launch_bounds__(256)
void ReduceEnergyFinal(const real_t* __restrict__ partial, int nPartial, real_t* __restrict__ out)
{
   __shared__ real_t sE[4], sK[4];
   real_t accE = R(0.0), accK = R(0.0);
   for (int i = threadIdx.x; i < nPartial; i += blockDim.x) { accE += partial[2*i]; accK += partial[2*i + 1]; }
   accE = waveSum(accE); accK = waveSum(accK);
   const int wave = threadIdx.x >> 6;
   if ((threadIdx.x & 63) == 0) { sE[wave] = accE; sK[wave] = accK; }
   __syncthreads();
   if (threadIdx.x == 0) { out[0] = (sE[0] + sE[1]) + (sE[2] + sE[3]); out[1] = (sK[0] + sK[1]) + (sK[2] + sK[3]); }
}

// ---- redistribution -------------------------------------------------------------------------------------
// snapshot local occupancies, empty the halo cells (timestep.c:224), clear dirty flags
__global__ __launch_bounds__(256)
void SnapshotCells(int* __restrict__ nAtoms, int* __restrict__ nAtomsPrev, int* __restrict__ dirty, int* __restrict__ arrivals, int nLocal, int nTotal)
{
   const int c = blockIdx.x * blockDim.x + threadIdx.x;
   if (c >= nTotal) return;
   if (c < nLocal) nAtomsPrev[c] = nAtoms[c]; else { nAtoms[c] = 0; nAtomsPrev[c] = 0; }
   dirty[c] = 0;
   arrivals[c] = 0; arrivals[nTotal + c] = 0; arrivals[2 * (size_t)nTotal + c] = 0;      // (MirrorAtomCells below)
}

struct AtomArrays {
   real_t *rx, *ry, *rz, *px, *py, *pz;
   int *gid, *spec;
};

// gpu_redistribute.h:135-180 UpdateLinkCells: every local atom re-derives its cell from its coordinates; movers
// are appended to the destination cell (local or halo) and leave a hole (gid = -1) behind.
__global__ __launch_bounds__(256)
void UpdateLinkCells(AtomArrays at, int* __restrict__ nAtoms, const int* __restrict__ nAtomsPrev,
                     int* __restrict__ dirty, int* __restrict__ status, LinkCellGpu boxes, int cap)
{
   const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const int c = (int)(tid / cap);
   if (c >= boxes.nLocalBoxes || (int)(tid - (long)c * cap) >= nAtomsPrev[c]) return;
   const CellGeom g = makeGeom(boxes);
   const real_t x = at.rx[tid], y = at.ry[tid], z = at.rz[tid];
   const int nb = comdBoxFromCoord(&g, x, y, z);
   if (nb == c) return;
   if (!comdCoordInHalo(&g, x, y, z)) { atomicOr(&status[1], 1); return; }     // flew past the halo: lost
   const int slot = atomicAdd(&nAtoms[nb], 1);
   if (slot >= cap) { atomicOr(&status[0], 1); return; }
   const size_t d = (size_t)nb * cap + slot;
   at.rx[d] = x; at.ry[d] = y; at.rz[d] = z;
   at.px[d] = at.px[tid]; at.py[d] = at.py[tid]; at.pz[d] = at.pz[tid];
   at.gid[d] = at.gid[tid]; at.spec[d] = at.spec[tid];
   at.gid[tid] = -1;
   dirty[c] = 1; dirty[nb] = 1;
}

// Cells [first, first + nCells), `run` (<= COMPACT_RUN) consecutive cells per workgroup: if a cell is dirty, drop holes and rewrite the
// survivors in ascending-gid order (rank sort: rank = number of smaller keys; gids are unique).  blockDim.x >= cap.
// The host picks the run: long for the local cells (almost all clean: one flag read each), 1 for the halo cells (every one of them was
// just refilled; eight in a row per workgroup were 40 of the 60 us this kernel took at LJ 80^3).
#define COMPACT_RUN 8
__global__
void CompactSortCells(AtomArrays at, int* __restrict__ nAtoms, int* __restrict__ dirty, int* __restrict__ status,
                      int first, int nCells, int cap, int run, int* __restrict__ arrivals, int nTotal)
{
   extern __shared__ int sKey[];
   __shared__ int sFlag[COMPACT_RUN];
   const int t = threadIdx.x;
   // the run's flags with one round trip (read one after the other they are COMPACT_RUN dependent latencies per workgroup, and almost
   // every workgroup finds nothing to do)
   if (t < run) sFlag[t] = blockIdx.x * run + t < nCells ? dirty[first + blockIdx.x * run + t] : 0;
   __syncthreads();
   for (int k = 0; k < run; ++k) {
      const int idx = blockIdx.x * run + k;
      if (idx >= nCells) return;
      const int c = first + idx;
      if (!sFlag[k]) continue;                              // workgroup-uniform
      int n = nAtoms[c] + arrivals[c] + arrivals[nTotal + c] + arrivals[2 * (size_t)nTotal + c];      // (atoms mirrored in on self-neighbour axes are counted beside the cell)
      if (n > cap) n = cap;                                  // overflow already flagged by the writer
      const size_t o = (size_t)c * cap + t;
      int key = 0x7fffffff, spec = 0;
      real_t x = 0, y = 0, z = 0, px = 0, py = 0, pz = 0;
      if (t < n) {
         int g = at.gid[o];
         if (g >= 0) { key = g; spec = at.spec[o]; x = at.rx[o]; y = at.ry[o]; z = at.rz[o]; px = at.px[o]; py = at.py[o]; pz = at.pz[o]; }
      }
      __syncthreads();                                       // the previous cell's keys are no longer read
      if (t < cap) sKey[t] = key;
      __syncthreads();
      int rank = 0, live = 0;
      for (int j = 0; j < n; ++j) { int kj = sKey[j]; rank += (kj < key); live += (kj != 0x7fffffff); }
      if (key != 0x7fffffff) {
         const size_t d = (size_t)c * cap + rank;
         at.gid[d] = key; at.spec[d] = spec;
         at.rx[d] = x; at.ry[d] = y; at.rz[d] = z; at.px[d] = px; at.py[d] = py; at.pz[d] = pz;
      }
      if (t == 0) { nAtoms[c] = live; dirty[c] = 0; arrivals[c] = 0; arrivals[nTotal + c] = 0; arrivals[2 * (size_t)nTotal + c] = 0; }
   }
}

// Same for cap <= 64: one WAVE per run of COMPACT_RUN_WAVE cells, four waves per workgroup, keys exchanged with v_readlane (no LDS, no barrier).
#define COMPACT_RUN_WAVE 16
__global__ __launch_bounds__(256)
void CompactSortCellsWave(AtomArrays at, int* __restrict__ nAtoms, int* __restrict__ dirty, int first, int nCells, int cap, int run,
                          int* __restrict__ arrivals, int nTotal)
{
   const int lane = threadIdx.x & 63;
   const int wrun = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
   // a flag per lane, one round trip for the run of `run` (<= COMPACT_RUN_WAVE) cells (read one after the other they are as many dependent
   // latencies, and almost every cell is clean); only the dirty ones are visited
   const int mine = wrun * run + lane;
   const bool look = lane < run && mine < nCells;
   unsigned long long todo = __ballot(look && dirty[first + (look ? mine : 0)] != 0);
   while (todo) {
      const int k = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int c = first + wrun * run + k;
      int n = uniform(nAtoms[c] + arrivals[c] + arrivals[nTotal + c] + arrivals[2 * (size_t)nTotal + c]);
      if (n > cap) n = cap;
      const size_t o = (size_t)c * cap + lane;
      int key = 0x7fffffff, spec = 0;
      real_t x = 0, y = 0, z = 0, px = 0, py = 0, pz = 0;
      if (lane < n) {
         const int g = at.gid[o];
         if (g >= 0) { key = g; spec = at.spec[o]; x = at.rx[o]; y = at.ry[o]; z = at.rz[o]; px = at.px[o]; py = at.py[o]; pz = at.pz[o]; }
      }
      int rank = 0, live = 0;
      for (int j = 0; j < n; ++j) {
         const int kj = __builtin_amdgcn_readlane(key, j);
         rank += (kj < key); live += (kj != 0x7fffffff);
      }
      if (key != 0x7fffffff) {
         const size_t d = (size_t)c * cap + rank;
         at.gid[d] = key; at.spec[d] = spec;
         at.rx[d] = x; at.ry[d] = y; at.rz[d] = z; at.px[d] = px; at.py[d] = py; at.pz[d] = pz;
      }
      if (lane == 0) { nAtoms[c] = live; dirty[c] = 0; arrivals[c] = 0; arrivals[nTotal + c] = 0; arrivals[2 * (size_t)nTotal + c] = 0; }
   }
}

// ---- exclusive scan of nAtoms over cell lists (gpu_kernels.cu:357-407 fill + scan) -----------------------------
// Workgroup b serves job b: out[i] = sum_{k<i} nAtoms[list[k]], out[n] = total (also copied to *total when given: the count header of
// an atom message).  The occupancies are fetched into the LDS first -- list entry, then the cell it names, every load independent and
// coalesced -- and scanned from there 1024 at a time; a scan that chases list[i] -> nAtoms[...] inside its loop pays two dependent
// global round trips per 1024 cells (15 us for a face of 7200 cells).
struct ScanJobs { const int* list[12]; int n[12]; int* out[12]; int* total[12]; };

#define SCAN_LDS_CELLS 12288              // occupancies staged per pass (48 KB); longer lists take several passes with a carry
__global__ __launch_bounds__(1024)
void ScanCellCountsBatch(const int* __restrict__ nAtoms, ScanJobs jobs)
{
   __shared__ int sCnt[SCAN_LDS_CELLS];
   __shared__ int sWave[16];
   const int* __restrict__ list = jobs.list[blockIdx.x];
   const int n = jobs.n[blockIdx.x];
   int* __restrict__ out = jobs.out[blockIdx.x];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   int carry = 0;                                             // the same in every thread
   for (int p0 = 0; p0 < n; p0 += SCAN_LDS_CELLS) {
      const int m = n - p0 < SCAN_LDS_CELLS ? n - p0 : SCAN_LDS_CELLS;
      // every occupancy of the pass with independent, coalesced loads (the list, then the cells it names): one memory round trip
      {
         constexpr int PER = SCAN_LDS_CELLS / 1024;           // all list loads of the pass in flight, then all occupancy loads
         int cell[PER], cnt[PER];
#pragma unroll
         for (int u = 0; u < PER; ++u) { const int i = threadIdx.x + 1024 * u; cell[u] = i < m ? (list ? list[p0 + i] : p0 + i) : -1; }
#pragma unroll
         for (int u = 0; u < PER; ++u) cnt[u] = cell[u] >= 0 ? nAtoms[cell[u]] : 0;
#pragma unroll
         for (int u = 0; u < PER; ++u) { const int i = threadIdx.x + 1024 * u; if (i < m) sCnt[i] = cnt[u]; }
      }
      __syncthreads();
      // one block-wide scan per pass: a thread sums its E consecutive occupancies, the 1024 sums are scanned (wave shuffles + 16 wave
      // totals), and the thread walks its E again (1024 occupancies per block scan took four scans and eight barriers for a face of 3.8 k cells)
      {
         constexpr int PER = SCAN_LDS_CELLS / 1024;
         const int E = (m + 1023) >> 10;
         int v[PER], sum = 0;
#pragma unroll
         for (int u = 0; u < PER; ++u) { const int i = threadIdx.x * E + u; v[u] = (u < E && i < m) ? sCnt[i] : 0; sum += v[u]; }
         int incl = sum;
#pragma unroll
         for (int d = 1; d < 64; d <<= 1) { int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
         if (lane == 63) sWave[wave] = incl;
         __syncthreads();
         int before = carry, all = 0;
         for (int w = 0; w < 16; ++w) { const int t = sWave[w]; if (w < wave) before += t; all += t; }
         int run = before + incl - sum;
#pragma unroll
         for (int u = 0; u < PER; ++u) { const int i = threadIdx.x * E + u; if (u < E && i < m) { out[p0 + i] = run; run += v[u]; } }
         carry += all;
         __syncthreads();
      }
   }
   if (threadIdx.x == 0) { out[n] = carry; if (jobs.total[blockIdx.x]) *jobs.total[blockIdx.x] = carry; }
}

// four message counts of an axis phase -> pinned host memory (read by the host one step later, behind an event)
__global__ void MirrorCounts(const int* __restrict__ d0, const int* __restrict__ d1, const int* __restrict__ d2, const int* __restrict__ d3,
                             int* __restrict__ hostDst)
{
   if (threadIdx.x < 4) {
      const int* src = threadIdx.x == 0 ? d0 : threadIdx.x == 1 ? d1 : threadIdx.x == 2 ? d2 : d3;
      hostDst[threadIdx.x] = *src;
   }
}

// ---- atom halo message -----------------------------------------------------------------------------------
// Both faces of an axis phase are packed by ONE launch and unpacked by one (blockIdx.y = face of the pair): half the launches of a
// step that is, on one rank, a string of 5-microsecond kernels.
struct AtomPackJob { char* msg; const int* list; const int* offsets; int nCells; real_t sx, sy, sz; int capacityAtoms; };

// gpu_redistribute.h:376-402 LoadAtomsBufferPacked: one workgroup per listed cell gathers its atoms into the SoA
// message at offsets[cell]; positions are shifted across the periodic boundary.  blockDim.x >= cap.
__global__
void LoadAtomsBufferPacked(AtomPackJob j0, AtomPackJob j1, AtomArrays at, const int* __restrict__ nAtoms, int cap, int* __restrict__ status)
{
   const AtomPackJob& jb = blockIdx.y ? j1 : j0;
   if ((int)blockIdx.x >= jb.nCells) return;
   const int c = jb.list[blockIdx.x];
   const int t = threadIdx.x;
   const int n = jb.offsets[jb.nCells];
   if (n > jb.capacityAtoms) { if (blockIdx.x == 0 && t == 0) atomicOr(&status[2], 1); return; }
   if (t >= nAtoms[c]) return;
   const size_t o = (size_t)c * cap + t;
   const int d = jb.offsets[blockIdx.x] + t;
   int* mg = (int*)(jb.msg + COMD_ATOM_MSG_HEADER);
   int* mt = mg + n;
   real_t* m = (real_t*)(mt + n);
   mg[d] = at.gid[o]; mt[d] = at.spec[o];
   m[d] = at.rx[o] + jb.sx; m[n + d] = at.ry[o] + jb.sy; m[2*(size_t)n + d] = at.rz[o] + jb.sz;
   m[3*(size_t)n + d] = at.px[o]; m[4*(size_t)n + d] = at.py[o]; m[5*(size_t)n + d] = at.pz[o];
}

struct AtomUnpackJob { const char* msg; int nBuf, capacityAtoms; };

// gpu_redistribute.h:499-620: every received atom finds its cell from its coordinates and is appended there.
__global__ __launch_bounds__(256)
void UnloadAtomsBufferPacked(AtomUnpackJob j0, AtomUnpackJob j1, AtomArrays at, int* __restrict__ nAtoms,
                             int* __restrict__ dirty, int* __restrict__ status, LinkCellGpu boxes, int cap)
{
   const AtomUnpackJob& jb = blockIdx.y ? j1 : j0;
   const char* __restrict__ msg = jb.msg;
   const int n = jb.nBuf >= 0 ? jb.nBuf : ((const int*)msg)[0];
   if (n < 0 || n > jb.capacityAtoms) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&status[2], 1); return; }
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const int* mg = (const int*)(msg + COMD_ATOM_MSG_HEADER);
   const int* mt = mg + n;
   const real_t* m = (const real_t*)(mt + n);
   const real_t x = m[i], y = m[n + i], z = m[2*(size_t)n + i];
   const CellGeom g = makeGeom(boxes);
   if (!comdCoordInHalo(&g, x, y, z)) { atomicOr(&status[1], 1); return; }
   const int c = comdBoxFromCoord(&g, x, y, z);
   // The message is cell ordered, so most lanes of a wave target the same cell: one atomic per distinct cell per wave
   // (leader adds the group's size, members take consecutive slots) instead of 64 serialized same-address atomics.
   int slot = 0;
   for (bool pending = true; pending; ) {
      const int lead = __builtin_amdgcn_readfirstlane(c);
      if (c == lead) {
         const unsigned long long m = __ballot(1);
         const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
         int base = 0;
         if (rank == 0) base = atomicAdd(&nAtoms[lead], __popcll(m));
         slot = __builtin_amdgcn_readfirstlane(base) + rank;
         pending = false;
      }
   }
   if (slot >= cap) { atomicOr(&status[0], 1); return; }
   const size_t d = (size_t)c * cap + slot;
   at.gid[d] = mg[i]; at.spec[d] = mt[i];
   at.rx[d] = x; at.ry[d] = y; at.rz[d] = z;
   at.px[d] = m[3*(size_t)n + i]; at.py[d] = m[4*(size_t)n + i]; at.pz[d] = m[5*(size_t)n + i];
   dirty[c] = 1;
}

// ---- EAM force (dfEmbed) and position halo messages: gpu_redistribute.h:638-672 ------------------------------
// blockDim.x >= cap; one workgroup per listed cell; positional (both sides hold the cell in gid order); blockIdx.y = face of the pair.
// boundAtoms > 0: the size both ends of the message agreed on beforehand; more atoms than that cannot be sent -> status[2]
struct SlotJob { real_t* buf; const int* list; const int* offsets; int nCells; int boundAtoms; real_t sx, sy, sz; };

__global__
void LoadForceBuffer(SlotJob j0, SlotJob j1, const real_t* __restrict__ dfEmbed, const int* __restrict__ nAtoms, int cap, int* __restrict__ status)
{
   const SlotJob& jb = blockIdx.y ? j1 : j0;
   if ((int)blockIdx.x >= jb.nCells) return;
   if (jb.boundAtoms > 0 && jb.offsets[jb.nCells] > jb.boundAtoms) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&status[2], 1); return; }
   const int c = jb.list[blockIdx.x];
   if ((int)threadIdx.x < nAtoms[c]) jb.buf[jb.offsets[blockIdx.x] + threadIdx.x] = dfEmbed[(size_t)c * cap + threadIdx.x];
}

__global__
void UnloadForceBuffer(SlotJob j0, SlotJob j1, real_t* __restrict__ dfEmbed, const int* __restrict__ nAtoms, int cap)
{
   const SlotJob& jb = blockIdx.y ? j1 : j0;
   if ((int)blockIdx.x >= jb.nCells) return;
   const int c = jb.list[blockIdx.x];
   if ((int)threadIdx.x < nAtoms[c]) dfEmbed[(size_t)c * cap + threadIdx.x] = jb.buf[jb.offsets[blockIdx.x] + threadIdx.x];
}

// positional refresh of the halo copies between list builds (slot order == the sender's slot order); x, y, z triples in send-cell-list order
__global__
void LoadPositionBuffer(SlotJob j0, SlotJob j1, const real_t* __restrict__ rx, const real_t* __restrict__ ry, const real_t* __restrict__ rz,
                        const int* __restrict__ nAtoms, int cap, int* __restrict__ status)
{
   const SlotJob& jb = blockIdx.y ? j1 : j0;
   if ((int)blockIdx.x >= jb.nCells) return;
   if (jb.boundAtoms > 0 && jb.offsets[jb.nCells] > jb.boundAtoms) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&status[2], 1); return; }
   const int c = jb.list[blockIdx.x];
   if ((int)threadIdx.x < nAtoms[c]) {
      const size_t s = (size_t)c * cap + threadIdx.x;
      real_t* o = jb.buf + 3 * (size_t)(jb.offsets[blockIdx.x] + threadIdx.x);
      o[0] = rx[s] + jb.sx; o[1] = ry[s] + jb.sy; o[2] = rz[s] + jb.sz;
   }
}

__global__
void UnloadPositionBuffer(SlotJob j0, SlotJob j1, real_t* __restrict__ rx, real_t* __restrict__ ry, real_t* __restrict__ rz,
                          const int* __restrict__ nAtoms, int cap)
{
   const SlotJob& jb = blockIdx.y ? j1 : j0;
   if ((int)blockIdx.x >= jb.nCells) return;
   const int c = jb.list[blockIdx.x];
   if ((int)threadIdx.x < nAtoms[c]) {
      const size_t s = (size_t)c * cap + threadIdx.x;
      const real_t* o = jb.buf + 3 * (size_t)(jb.offsets[blockIdx.x] + threadIdx.x);
      rx[s] = o[0]; ry[s] = o[1]; rz[s] = o[2];
   }
}

// ---- [round 4] self-neighbour axes: halo cells filled straight from the cells they are periodic images of ------------------------------------------
// On an axis where a rank is its own minus and plus neighbour the destination of every halo cell is known: dst[k] is the image of src[k], displaced by
// shift[3k..3k+2].  The host resolves the x -> y -> z ordering of the exchange (haloExchange.c:1504-1520: an edge or corner cell is the image of an image)
// into ONE source per halo cell -- each coordinate is shifted by at most one axis, so the sum of the shifts is the same arithmetic as the chain -- and one
// launch takes the place of pack + unpack per axis.  Positional, like the messages it replaces: slot i of dst = slot i of src.  blockDim.x >= cap.
__global__
void MirrorSlotCells(int kind, int nPairs, const int* __restrict__ dst, const int* __restrict__ src, const real_t* __restrict__ shift,
                     real_t* __restrict__ rx, real_t* __restrict__ ry, real_t* __restrict__ rz, real_t* __restrict__ dfEmbed,
                     const int* __restrict__ nAtoms, int cap)
{
   const int k = blockIdx.x;
   if (k >= nPairs) return;
   const int d = dst[k], c = src[k];
   if ((int)threadIdx.x >= nAtoms[d]) return;
   const size_t to = (size_t)d * cap + threadIdx.x, from = (size_t)c * cap + threadIdx.x;
   if (kind == 0) dfEmbed[to] = dfEmbed[from];
   else { rx[to] = rx[from] + shift[3 * k]; ry[to] = ry[from] + shift[3 * k + 1]; rz[to] = rz[from] + shift[3 * k + 2]; }
}

// Atoms on a self-neighbour axis: pack + unpack of both faces of an axis phase as ONE kernel, no message, no scan.  Every atom of the send cells (the halo
// layer and the first local layer of either face, mkAtomCellList) is displaced by the shift of its face, finds its cell from the displaced coordinates and is
// appended there -- the arithmetic and the membership of the message path (LoadAtomsBufferPacked + UnloadAtomsBufferPacked), so the cells end up holding the
// same atoms and CompactSortCells the same order.  The message path packs both faces BEFORE it unpacks either, and the cells an atom arrives in are send
// cells of the opposite face: arrivals are therefore counted beside the cell (arrivals[axis][cell]), not in nAtoms -- a cell's content at the start of the
// phase is nAtoms + the arrivals of the EARLIER mirrored axes, stable for the whole launch.  CompactSortCells folds the counts back into nAtoms.
struct MirrorAtomsJob { const int* list[2]; int nCells[2]; real_t sx[2], sy[2], sz[2]; };

__global__ __launch_bounds__(256)
void MirrorAtomCells(MirrorAtomsJob jb, AtomArrays at, const int* __restrict__ nAtoms, int* __restrict__ arrivals, int nTotal, int firstAxis, int axis,
                     int* __restrict__ dirty, int* __restrict__ status, LinkCellGpu boxes, int cap)
{
   const int face = blockIdx.y;
   const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
   const int i = (int)(tid / cap), slot = (int)(tid - (long)i * cap);
   bool have = i < jb.nCells[face];
   int c = 0;
   if (have) {
      c = jb.list[face][i];
      int cnt = nAtoms[c];
      for (int a = firstAxis; a < axis; ++a) cnt += arrivals[(size_t)a * nTotal + c];
      have = slot < (cnt < cap ? cnt : cap);
   }
   const size_t o = (size_t)c * cap + slot;
   real_t x = R(0.0), y = R(0.0), z = R(0.0);
   int d = -1;
   if (have && at.gid[o] < 0) have = false;                  // a hole a mover left (the cells need not have been compacted since UpdateLinkCells: comd_hip.h skipSortAfterUpdate)
   if (have) {
      x = at.rx[o] + jb.sx[face]; y = at.ry[o] + jb.sy[face]; z = at.rz[o] + jb.sz[face];
      const CellGeom g = makeGeom(boxes);
      if (!comdCoordInHalo(&g, x, y, z)) { atomicOr(&status[1], 1); have = false; }
      else d = comdBoxFromCoord(&g, x, y, z);
   }
   // one atomic per distinct destination cell per wave (the atoms of a send cell nearly all go to one cell)
   int at0 = 0;
   for (bool pending = have; __builtin_amdgcn_ballot_w64(pending) != 0ull; ) {
      const unsigned long long todo = __builtin_amdgcn_ballot_w64(pending);
      const int lead = __builtin_amdgcn_readlane(d, __ffsll((long long)todo) - 1);
      if (pending && d == lead) {
         const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
         const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
         int base = 0;
         if (rank == 0) base = atomicAdd(&arrivals[(size_t)axis * nTotal + lead], __popcll(m));
         at0 = __builtin_amdgcn_readfirstlane(base) + rank;
         pending = false;
      }
   }
   if (!have) return;
   int start = nAtoms[d];
   for (int a = firstAxis; a < axis; ++a) start += arrivals[(size_t)a * nTotal + d];
   const int k = start + at0;
   if (k >= cap) { atomicOr(&status[0], 1); return; }
   const size_t to = (size_t)d * cap + k;
   at.gid[to] = at.gid[o]; at.spec[to] = at.spec[o];
   at.rx[to] = x; at.ry[to] = y; at.rz[to] = z;
   at.px[to] = at.px[o]; at.py[to] = at.py[o]; at.pz[to] = at.pz[o];
   dirty[d] = 1;
}
