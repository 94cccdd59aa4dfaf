// eam_brick_kernels.h -- EAM_Force_cta_brick: the cta_cell EAM kernel of round 3 (method cta_cell, passes 1 and 3).
//
// Same physics and per-cell work as EAM_Force_cta_cell (nl_kernels.h; reference gpu_eam_cta_cell.h:34-278 in results only), but the
// stencil is staged ONCE PER WORKGROUP for a brick of cells instead of once per cell:
//   * a workgroup owns a brick of 1 x BY x BZ link cells (default 1 x 4 x 2) and stages the 3 x (BY+2) x (BZ+2) cells around it
//     (72 instead of 8 x 27 = 216) densely packed into ONE LDS image -- round 2 fetched 3.06 GiB per force evaluation for 0.34 GiB of
//     algorithmic bytes because every wave fetched its own 27 cells (L2 hit rate 29 % in pass 3);
//   * the image is ordered z, y, x (x fastest) and the brick is ONE cell wide in x, so the 27-cell stencil of a brick cell is THREE
//     contiguous runs of records (one per z plane: 3 y rows x 3 x cells lie back to back) -- the lane -> record map of the build is two
//     compares, where round 2 walked a binary search through the LDS for each of the 6 records a lane staged per cell;
//   * no register staging across cells (round 2 kept 36 VGPRs of records in flight for the next cell): the waves of a workgroup pick
//     the brick's cells from an LDS counter, latency is hidden by the other workgroups of the CU;
//   * pass 1 leaves each atom's neighbours as 16-bit image numbers in the order pass 3's lanes consume them ([lane of the atom][trip] words of
//     two numbers, written with 16-byte stores; round 2 wrote them word by word) and pass 3 asks for a lane's words, the row length and the
//     pass-1 forces of the NEXT cell before it evaluates the current one.
// The image holds only the cells that lie in the stencil of a SELECTED cell of the brick, so its composition -- and with it the record
// numbers in the rows -- depends on nothing but the occupancies of those stencils: pass 1 and pass 3 of one force evaluation agree even
// when the launch covers a cell list (-a 1: interior cells while the halo cells are still being filled).
#pragma once
#include "nl_kernels.h"

//
// [round 4] LISTED = true: the Verlet-list method (-m thread_atom_nl -e) on the same machinery.  Between two list builds no atom changes its
// slot and no cell its occupancy (nl_kernels.h), so the record numbers of a brick's image stay valid for as long as the lists do: the rows
// are built ONCE per list build (STEP 0: the build sweeps of pass 1 with the cutoff inflated by the skin, nothing evaluated) and both
// passes of every force evaluation until the next build read them back -- no distance sweep over the 283 stencil candidates, 57 listed
// neighbours instead, the cutoff decided per pair inside the evaluation (a listed pair outside the cutoff is evaluated at the cutoff with
// weight 0: branch-free).  Listed rows live per (cell, round of 16 atoms) as [quad of 4 words][64 lanes] 16-byte elements: a wave reads the
// rows of a round with dense 1 KB loads.  The image of a listed launch always holds the whole block (the numbers must not depend on
// which cells a launch selects) plus one record FAR_AWAY that pads odd rows.
#define EAM_BRICK_MAX_CELLS 128           // cells of the staged block: 3 * (BY + 2) * (BZ + 2) <= 128
#define EAM_BRICK_SR 6                    // records a lane keeps in registers during the build (384 per wave; larger stencils read the LDS)
#ifndef EAM_BRICK_SR_BUILD
#define EAM_BRICK_SR_BUILD 6              // ... during a Verlet-row build (cells of cutoff + skin: 371 records in a stencil on average at 80^3)
#endif
#define EAM_BRICK_STAGE 8                 // staging iterations with all loads in flight (256 threads x 8 = 128 cells x 16 slots)
#define EAM_BRICK_STAGE_LISTED 10         // listed launches stage 32 slots per cell (cells of cutoff + skin hold 14 atoms on average): 256 x 10 = 80 cells x 32 slots
#define EAM_LIST_WORDS 12                 // words (two 16-bit numbers each) a lane holds of a listed row: 3 quads

// Scans and reductions over the 64 lanes on the DPP path (row shifts inside the rows of 16, then the row ends broadcast: six VALU instructions) -- as
// __shfl_up loops they are six ds_bpermute round trips through the LDS each.  Every lane must be active.
__device__ __forceinline__ int waveInclusiveScan(int v)
{
   v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);       // row_shr:1 (a lane without a source adds 0)
   v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);       // row_shr:2
   v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);       // row_shr:4
   v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);       // row_shr:8: inclusive inside each row of 16
   v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
   v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
   return v;
}
__device__ __forceinline__ int waveMin(int v)
{
   constexpr int BIG = 0x7fffffff;
   int o;
   o = __builtin_amdgcn_update_dpp(BIG, v, 0x111, 0xf, 0xf, false); v = o < v ? o : v;
   o = __builtin_amdgcn_update_dpp(BIG, v, 0x112, 0xf, 0xf, false); v = o < v ? o : v;
   o = __builtin_amdgcn_update_dpp(BIG, v, 0x114, 0xf, 0xf, false); v = o < v ? o : v;
   o = __builtin_amdgcn_update_dpp(BIG, v, 0x118, 0xf, 0xf, false); v = o < v ? o : v;      // lane 15 of a row: the row's minimum
   o = __builtin_amdgcn_update_dpp(BIG, v, 0x142, 0xa, 0xf, false); v = o < v ? o : v;
   o = __builtin_amdgcn_update_dpp(BIG, v, 0x143, 0xc, 0xf, false); v = o < v ? o : v;
   return __builtin_amdgcn_readlane(v, 63);
}

struct EamBrickArgs {
   CellGeom geom;                         // local grid (+ -H lookup tables, device pointers)
   int by, bz;                            // brick extent in y and z (cells); x extent is 1
   int nby, nbz;                          // bricks along y and z
   int imageCap;                          // records the LDS image holds
   int rows;                              // neighbours a row holds (multiple of 8, <= 256)
   unsigned* rowsG;                       // [cell][listRounds][listQuads][64 lanes] uint4: words of two image numbers, a lane's words 4 j .. 4 j + 3 in quad j (pass 1 -> pass 3; Verlet rows)
   unsigned short* rowCountG;             // [local slots]
   const int* sel; int tag;               // cell selection: NULL = every local cell, else the cells with sel[c] == tag
   const int* brickList;                  // NULL: workgroup w takes brick w; else brick brickList[w] (the bricks of one group of the overlap mode, every cell selected)
   int fuseEmbed;
   int* status;
   int debug;                             // experiments (COMD_EAM_ABLATE): 1 no build sweeps, 2 no pair evaluation
   int listRounds, listQuads;             // rounds of roundAtoms atoms a cell's capacity makes; 16-byte quads per lane and round (2 without lists, <= 3 with)
   real_t rBuild2;                        // STEP 0: (cutoff + skin)^2
   int* stats;                            // [0] longest list (STEP 0), [1] bricks whose block outgrew the image, counted by STEP 0 / pass 1 (NULL: not counted)
   unsigned long long* brickSel;          // not listed: [local cells] the selection of its brick pass 1 wrote the cell's rows under; pass 3 must stage for the same (status[3] |= 4)
};

__host__ __device__ static inline int eamBrickRowStrideL(int rows) { return rows + 2; }      // LDS row stride (entries): one dword of padding walks the atoms' rows over the banks
// per wave, only where rows are built (pass 1 without lists, STEP 0 with): [16][stride] rows + [16] counts
__host__ __device__ static inline bool eamBrickBuilds(int step, bool listed) { return listed ? step == 0 : step == 1; }
__host__ __device__ static inline size_t eamBrickWaveBytes(int step, bool listed, int rows)
{
   return eamBrickBuilds(step, listed) ? (((size_t)16 * eamBrickRowStrideL(rows) * 2 + 64 + 15) & ~(size_t)15) : 0;
}
__host__ __device__ static inline size_t eamBrickSharedBytes(int step, int imageCap)
{
   return (((size_t)(step == 3 ? 4 : 3) * imageCap * sizeof(real_t) + 15) & ~(size_t)15)      // records (+ F' in pass 3)
          + (size_t)(EAM_BRICK_MAX_CELLS + 4) * 4 + (size_t)EAM_BRICK_MAX_CELLS * 4 + 64 + 64;   // offsets, cell ids, scalars, list of selected cells
}
static inline size_t eamBrickLdsBytes(int step, bool listed, size_t tableDoubles, int imageCap, int rows, int waves)
{
   return eamTableBytesAligned(tableDoubles) + eamBrickSharedBytes(step, imageCap) + (size_t)waves * eamBrickWaveBytes(step, listed, rows);
}

__global__ __launch_bounds__(256)
void MarkCells(const int* __restrict__ list, int n, int* __restrict__ sel, int tag)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) sel[list[i]] = tag;
}

// Brick groups for the overlap mode (-a 1).  The host splits the local cells into boundary cells (two rings) and interior cells and launches every
// pass once per list; a brick that holds cells of both lists would be staged twice per pass, for a few cells each time.  The groups make the split at
// brick granularity: group[c] = 1 for every cell of a brick that holds a boundary cell (marks[c] == tag), 2 for the cells of the other bricks.
__global__ __launch_bounds__(256)
void ClassifyBrickCells(EamBrickArgs b, const int* __restrict__ marks, int tag, int* __restrict__ group, int* __restrict__ brickClass)
{
   const int bid = blockIdx.x * blockDim.x + threadIdx.x;
   const int gx = b.geom.g[0], gy = b.geom.g[1], gz = b.geom.g[2];
   if (bid >= gx * b.nby * b.nbz) return;
   const int bx = bid % gx, by0 = ((bid / gx) % b.nby) * b.by, bz0 = (bid / (gx * b.nby)) * b.bz;
   bool any = false;
   for (int dz = 0; dz < b.bz; ++dz)
      for (int dy = 0; dy < b.by; ++dy)
         if (by0 + dy < gy && bz0 + dz < gz) any = any || marks[comdBoxFromTuple(&b.geom, bx, by0 + dy, bz0 + dz)] == tag;
   for (int dz = 0; dz < b.bz; ++dz)
      for (int dy = 0; dy < b.by; ++dy)
         if (by0 + dy < gy && bz0 + dz < gz) group[comdBoxFromTuple(&b.geom, bx, by0 + dy, bz0 + dz)] = any ? 1 : 2;
   brickClass[bid] = any ? 1 : 2;
}

template <int STEP, bool LDS_TABLES, bool SPLINE, bool LISTED, bool CLAMP>
__global__ __launch_bounds__(256, 4)
void EAM_Force_cta_brick(EamArgs a, EamBrickArgs b)
{
   static_assert(STEP != 0 || LISTED, "STEP 0 builds Verlet rows");
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   constexpr int REC = 3;                                    // doubles per staged atom: x, y, z (24-byte stride); pass 3 keeps F' in an array of its own
   constexpr int SR = (STEP == 0) ? EAM_BRICK_SR_BUILD : EAM_BRICK_SR;
   constexpr bool BUILDS = LISTED ? STEP == 0 : STEP == 1;   // this launch sweeps the stencil for rows
   constexpr int SLOT_BITS = LISTED ? 5 : 4, SLOTS = 1 << SLOT_BITS;      // slots of a cell requested before its occupancy is known
   constexpr int STAGE = LISTED ? EAM_BRICK_STAGE_LISTED : EAM_BRICK_STAGE;
   constexpr int W = LISTED ? EAM_LIST_WORDS : 8;            // words of a row a lane holds
   const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), nThreads = blockDim.x, nWaves = nThreads >> 6;
   const int nRhoPad = a.rho.n + 3;
   const bool sameGrid = (STEP == 1) && LDS_TABLES && a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   real_t* sRho = (real_t*)ldsRaw;
   real_t* sPhi = sRho + nRhoPad;
   int tableDoubles = 0;
   if (LDS_TABLES && STEP != 0) tableDoubles = (STEP == 1) ? 2 * nRhoPad + (sameGrid ? 0 : (a.phi.n + 3 - nRhoPad)) : nRhoPad;

   real_t* __restrict__ sp = (real_t*)(ldsRaw + eamTableBytesAligned(tableDoubles));
   real_t* __restrict__ sd = sp + REC * b.imageCap;          // [imageCap] F' (pass 3)
   int* sOff = (int*)(ldsRaw + eamTableBytesAligned(tableDoubles) + ((((size_t)(STEP == 3 ? 4 : 3) * b.imageCap * sizeof(real_t)) + 15) & ~(size_t)15));
   int* sBox = sOff + EAM_BRICK_MAX_CELLS + 4;               // [128] cell ids of the block
   int* sMisc = sBox + EAM_BRICK_MAX_CELLS;                  // [16]: 0/1 selection mask, 4 records in the image
   unsigned char* sList = (unsigned char*)(sMisc + 16);      // [64] selected cells of the brick, compacted
   unsigned short* sHit = (unsigned short*)(sList + 64 + (size_t)wave * eamBrickWaveBytes(STEP, LISTED, b.rows));      // builds: [16][stride] rows under construction
   const int strideL = eamBrickRowStrideL(b.rows);
   int* sCnt = (int*)(sHit + 16 * strideL);                  // builds: [16] neighbours of the round's atoms
   const real_t rTest2 = (STEP == 0) ? b.rBuild2 : a.rc2;     // what a build sweep keeps

   // ---- the brick and its selected cells -------------------------------------------------------------------------------------------
   const int gx = b.geom.g[0], gy = b.geom.g[1], gz = b.geom.g[2];
   const int entry = b.brickList ? b.brickList[xcdRemap(blockIdx.x, gridDim.x)] : xcdRemap(blockIdx.x, gridDim.x);      // x fastest: consecutive bricks share two thirds of their block
   // Listed launches take their bricks from a list the host makes at every list build, when the occupancies are known and frozen: a brick whose block would
   // outgrow the image is listed as its two z halves (bits 28-29: 1 lower, 2 upper half), each staged and numbered on its own -- in every pass until the next build.
   const int half = LISTED ? (entry >> 28) & 3 : 0, bid = LISTED ? entry & 0x0fffffff : entry;
   const int bzN = half ? b.bz >> 1 : b.bz;                  // cells of this (half) brick along z
   const int bx = bid % gx, by0 = ((bid / gx) % b.nby) * b.by, bz0 = (bid / (gx * b.nby)) * b.bz + (half == 2 ? b.bz >> 1 : 0);
   const int HY = b.by + 2, HZ = bzN + 2, NH = 3 * HY * HZ, NC = b.by * bzN;
   unsigned long long selMask;
   if (b.sel) {      // a launch over a cell list: one more round trip, for the marks of the brick's cells
      if (wave == 0) {
         bool s = false;
         if (lane < NC) {
            const int iy = by0 + lane % b.by, iz = bz0 + lane / b.by;
            if (iy < gy && iz < gz) s = b.sel[comdBoxFromTuple(&b.geom, bx, iy, iz)] == b.tag;
         }
         const unsigned long long m = __builtin_amdgcn_ballot_w64(s);
         if (lane == 0) { sMisc[0] = (int)(unsigned)m; sMisc[1] = (int)(unsigned)(m >> 32); }
      }
      __syncthreads();
      selMask = ((unsigned long long)(unsigned)uniform(sMisc[1]) << 32) | (unsigned)uniform(sMisc[0]);
      if (selMask == 0ull) return;                           // (a list launch visits every brick: most of a boundary launch ends here)
   } else {
      bool s = false;
      if (lane < NC) s = by0 + lane % b.by < gy && bz0 + lane / b.by < gz;
      selMask = __builtin_amdgcn_ballot_w64(s);
   }
   const int nSel = __popcll(selMask);
   if (wave == 0) {                                          // the selected cells, compacted (every wave walks this list with a stride of the waves)
      const bool s = (selMask >> lane) & 1ull;
      if (s) sList[__builtin_amdgcn_mbcnt_hi((unsigned)(selMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)selMask, 0u))] = (unsigned char)lane;
   }
   // Without lists the image -- and with it every number in the rows -- depends on the selection: pass 3 must stage a cell's brick for the selection
   // pass 1 staged it for when it wrote the cell's rows.
   if (!LISTED && b.brickSel && wave == 0 && ((selMask >> lane) & 1ull)) {
      const int c = comdBoxFromTuple(&b.geom, bx, by0 + lane % b.by, bz0 + lane / b.by);
      if (STEP == 1) b.brickSel[c] = selMask;
      else if (b.brickSel[c] != selMask) atomicOr(&b.status[3], 4);
   }

   // ---- ONE round trip for the block: occupancies and records are requested together ------------------------------------------------------
   // A thread asks for slot s of block cell h (task = SLOTS h + s) without waiting for the occupancies: the 16 slots of a cell are one 128-byte line
   // per array whether they hold atoms or not.  What is not an atom is dropped when the occupancies have arrived and been scanned.
   real_t lx[STAGE], ly[STAGE], lz[STAGE], ld[STAGE];
   bool lok[STAGE];
   int myBox = -1;
   auto request = [&](const int k, const int bb, const int s) {      // (32-bit byte offsets: the launcher sends arrays of 4 GiB or more to the other kernel)
      lok[k] = bb >= 0 && s < a.cap;
      lx[k] = ly[k] = lz[k] = ld[k] = R(0.0);
      if (lok[k]) {
         const unsigned o = ((unsigned)bb * (unsigned)a.cap + (unsigned)s) * (unsigned)sizeof(real_t);
         lx[k] = *reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(a.rx) + o);
         ly[k] = *reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(a.ry) + o);
         lz[k] = *reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(a.rz) + o);
         if (STEP == 3) ld[k] = *reinterpret_cast<const real_t*>(reinterpret_cast<const char*>(a.dfEmbed) + o);
      }
   };
   // A brick away from the faces of the local grid with every cell selected -- nearly all of them -- has a block of local cells, numbered
   // x + gx (y + gy z), all of them needed.
   const bool plain = !b.geom.lookup && selMask == (NC >= 64 ? ~0ull : (1ull << NC) - 1ull) && bx >= 1 && bx <= gx - 2 && by0 >= 1 && by0 + b.by <= gy - 1
                      && bz0 >= 1 && bz0 + bzN <= gz - 1;
   if (plain) {
      const int hyMagic = (65536 + HY - 1) / HY;             // t / HY = (t * hyMagic) >> 16 for t < 128
      const int base = (bx - 1) + gx * ((by0 - 1) + gy * (bz0 - 1)), gxy = gx * gy;
      auto plainBox = [&](const int h) {                     // h = xh + 3 (yh + HY zh)
         const int t = (h * 171) >> 9, xh = h - 3 * t, zh = (t * hyMagic) >> 16, yh = t - zh * HY;
         return base + xh + gx * yh + gxy * zh;
      };
      if (tid < NH) myBox = plainBox(tid);
#pragma unroll
      for (int k = 0; k < STAGE; ++k) {
         const int task = k * nThreads + tid, h = task >> SLOT_BITS;
         request(k, h < NH ? plainBox(h) : -1, task & (SLOTS - 1));
      }
   } else {
      // every wave works out the cell ids of all 128 block cells (two per lane; -1: outside the grid or in no selected cell's stencil)
      auto blockBox = [&](const int h) {
         int box = -1;
         if (h < NH) {
            const int xh = h % 3, yh = (h / 3) % HY, zh = h / (3 * HY);
            const int iy = by0 + yh - 1, iz = bz0 + zh - 1;
            if (iy <= gy && iz <= gz) {
               bool need = LISTED;                            // in the stencil of a selected cell of the brick?  (listed rows: the whole block, always)
               if (!LISTED) {
#pragma unroll
                  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
                     for (int dy = -1; dy <= 1; ++dy) {
                        const int cy = yh - 1 + dy, cz = zh - 1 + dz;
                        if (cy >= 0 && cy < b.by && cz >= 0 && cz < bzN) need = need || ((selMask >> (cy + b.by * cz)) & 1ull);
                     }
               }
               if (need) box = comdBoxFromTuple(&b.geom, bx + xh - 1, iy, iz);
            }
         }
         return box;
      };
      const int boxLo = blockBox(lane), boxHi = blockBox(64 + lane);
      if (wave < 2) myBox = wave == 0 ? boxLo : boxHi;
#pragma unroll
      for (int k = 0; k < STAGE; ++k) {
         const int task = k * nThreads + tid, h = task >> SLOT_BITS;
         const int fromLo = __builtin_amdgcn_ds_bpermute((h & 63) << 2, boxLo), fromHi = __builtin_amdgcn_ds_bpermute((h & 63) << 2, boxHi);
         request(k, h < NH ? (h < 64 ? fromLo : fromHi) : -1, task & (SLOTS - 1));
      }
   }
   int myCnt = 0;
   if (myBox >= 0) myCnt = a.nAtoms[myBox];
   if (LDS_TABLES && STEP != 0) {
      if (sameGrid) {
         for (int t = tid; t < nRhoPad; t += nThreads) { sRho[2 * t] = a.phi.values[t]; sRho[2 * t + 1] = a.rho.values[t]; }
      } else {
         for (int t = tid; t < nRhoPad; t += nThreads) sRho[t] = a.rho.values[t];
         if (STEP == 1) for (int t = tid; t < a.phi.n + 3; t += nThreads) sPhi[t] = a.phi.values[t];
      }
   }
   const TableView rhoT = makeTable(a.rho, LDS_TABLES ? sRho : a.rho.values), phiT = makeTable(a.phi, LDS_TABLES ? sPhi : a.phi.values);
   if (tid < EAM_BRICK_MAX_CELLS) { sBox[tid] = myBox >= 0 ? myBox : 0; sOff[tid] = myCnt; }
   __syncthreads();
   if (wave == 0) {                                          // exclusive scan of the 128 counts: lane l takes entries 2l and 2l + 1
      const int c0 = sOff[2 * lane], c1 = sOff[2 * lane + 1];
      const int incl = waveInclusiveScan(c0 + c1);
      const int excl = incl - c0 - c1;
      sOff[2 * lane] = excl; sOff[2 * lane + 1] = excl + c0;
      if (lane == 63) { sOff[EAM_BRICK_MAX_CELLS] = incl; sMisc[4] = incl; }
   }
   __syncthreads();
   const int imageTotal = uniform(sMisc[4]);
   const bool fits = imageTotal + (LISTED ? 1 : 0) <= b.imageCap;
   if (fits) {
#pragma unroll
      for (int k = 0; k < STAGE; ++k) {
         const int task = k * nThreads + tid, h = task >> SLOT_BITS, s = task & (SLOTS - 1);
         if (lok[k]) {
            const int off = sOff[h], n = sOff[h + 1] - off;
            if (s < n) {
               real_t* r = sp + REC * (off + s);
               r[0] = lx[k]; r[1] = ly[k]; r[2] = lz[k];
               if (STEP == 3) sd[off + s] = ld[k];
            }
         }
      }
      // cells of more than SLOTS atoms: blocking copies
      for (int task = tid; task < NH * SLOTS; task += nThreads) {
         const int h = task >> SLOT_BITS, off = sOff[h], n = sOff[h + 1] - off;
         for (int s = (task & (SLOTS - 1)) + SLOTS; s < n; s += SLOTS) {
            const size_t o = (size_t)sBox[h] * a.cap + s;
            real_t* r = sp + REC * (off + s);
            r[0] = a.rx[o]; r[1] = a.ry[o]; r[2] = a.rz[o];
            if (STEP == 3) sd[off + s] = a.dfEmbed[o];
         }
      }
      if (LISTED && tid == 0) {                              // the record that pads odd rows: never inside a cutoff, F' = 0
         sp[REC * imageTotal] = FAR_AWAY; sp[REC * imageTotal + 1] = FAR_AWAY; sp[REC * imageTotal + 2] = FAR_AWAY;
         if (STEP == 3) sd[imageTotal] = R(0.0);
      }
   } else if (STEP != 3 && b.stats && tid == 0) atomicAdd(&b.stats[1], 1);      // bricks that take the thread-per-atom form (comdEamBrickStats)
   __syncthreads();
   if (b.debug & 4) return;

   // atoms per round: a lane holds at most W words = 2 W entries of a row, so an atom needs ceil(rows / 2 W) lanes
   const int lanesMin = (b.rows + 2 * W - 1) / (2 * W);
   const int roundAtoms = 64 / lanesMin < 16 ? 64 / lanesMin : 16;
   bool over = false;
   int longest = 0;

   // geometry of a round: nRound atoms, L lanes each; lane = L * ia + q
   auto roundOf = [&](const int ni, const int i0, int& nRound, int& L, int& ia, int& q) {
      nRound = ni - i0 < roundAtoms ? ni - i0 : roundAtoms;
      if (nRound < 1) nRound = 1;
      L = 64 / nRound; if (L > 16) L = 16;
      ia = (lane * ((65536 + L - 1) / L)) >> 16;             // lane / L (exact for lane < 64)
      q = lane - ia * L;
   };
   // A ROW of n entries is dealt to the atom's L lanes in pairs: pair p = entries 2p, 2p+1 goes to lane p % L as its word p / L (trip).
   // In memory, per (cell, round): [quad][64 lanes] 16-byte elements, a lane's words 4 j .. 4 j + 3 in quad j; pass 3 asks for its quads and the row length
   // together, before it knows the length.
   struct Pre3 { uint4 lo, hi, ex; int n; real_t f0x, f0y, f0z; };
   auto fetch3 = [&](const int iBox, const int ni, const int i0, Pre3& p) {
      int nRound, L, ia, q; roundOf(ni, i0, nRound, L, ia, q);
      p.lo = make_uint4(0u, 0u, 0u, 0u); p.hi = p.lo; p.ex = p.lo; p.n = 0; p.f0x = p.f0y = p.f0z = R(0.0);
      if (i0 + ia < ni && ia < nRound) {
         const size_t cellSlot = (size_t)iBox * a.cap;
         const unsigned ii = (unsigned)(i0 + ia);
         p.n = (b.rowCountG + cellSlot)[ii];
         const uint4* __restrict__ src = reinterpret_cast<const uint4*>(b.rowsG) + ((size_t)iBox * b.listRounds + (unsigned)(i0 / roundAtoms)) * b.listQuads * 64 + lane;
         p.lo = src[0];
         if (LISTED) {
            if (b.listQuads > 1) p.hi = src[64];      // (the third quad -- a lane's words 8 .. 11: rows beyond 16 L entries -- is fetched later, by the lanes that have such words)
         } else if (L <= 5) p.hi = src[64];      // rows of pass 1: the second quad is asked for with the first when the atoms of the round have few lanes each (a lane's fifth
                                                 // word exists from 8 L + 1 neighbours on); with six lanes or more it is fetched later, if a row turns out that long
         // pass 3 adds to the forces of pass 1: ask for them now, a whole cell of arithmetic before they are needed
         if (STEP == 3 && q == 0) { p.f0x = (a.fx + cellSlot)[ii]; p.f0y = (a.fy + cellSlot)[ii]; p.f0z = (a.fz + cellSlot)[ii]; }
      }
   };
   constexpr bool FETCHES = STEP == 3 || (LISTED && STEP == 1);      // rows come from memory
   // ---- the waves take the brick's selected cells in turn: wave w the cells w, w + nWaves, ... -----------------------------------------------
   auto cellHeader = [&](const int pick, int& iBox, int& ownStart, int& ni, int& yh, int& zh) {
      const int cl = sList[pick];
      yh = cl % b.by + 1; zh = cl / b.by + 1;
      const int hc = 1 + 3 * (yh + HY * zh);
      iBox = uniform(sBox[hc]); ownStart = uniform(sOff[hc]); ni = uniform(sOff[hc + 1]) - ownStart;
   };
   Pre3 pre;
   pre.lo = make_uint4(0u, 0u, 0u, 0u); pre.hi = pre.lo; pre.ex = pre.lo; pre.n = 0; pre.f0x = pre.f0y = pre.f0z = R(0.0);
   if (FETCHES && fits && wave < nSel) { int iBox, os, ni, yh, zh; cellHeader(wave, iBox, os, ni, yh, zh); fetch3(iBox, ni, 0, pre); }
   for (int pick = wave; pick < nSel; pick += nWaves) {
      int iBox, ownStart, ni, yh, zh;
      cellHeader(pick, iBox, ownStart, ni, yh, zh);
      if (!fits) {      // a block larger than the LDS image (small boxes have larger cells): thread-per-atom form, same tables (needs no rows: STEP 0 leaves none)
         if (STEP != 0) eamCellDirect<STEP == 0 ? 1 : STEP, SPLINE>(a, iBox, lane, rhoT, phiT, sameGrid, b.fuseEmbed);
         continue;
      }
      Pre3 cur = pre;
      if (FETCHES && pick + nWaves < nSel) { int nb, nos, nni, nyh, nzh; cellHeader(pick + nWaves, nb, nos, nni, nyh, nzh); fetch3(nb, nni, 0, pre); }
      if (ni == 0) continue;
      const size_t cellSlot = (size_t)iBox * a.cap;          // wave-uniform: the cell's arrays are addressed base + 32-bit lane offset
      // the stencil: three runs of records, one per z plane (rows yh-1 .. yh+1 of a plane lie back to back)
      const int rS0 = uniform(sOff[3 * ((yh - 1) + HY * (zh - 1))]), rE0 = uniform(sOff[3 * ((yh + 1) + HY * (zh - 1)) + 3]);
      const int rS1 = uniform(sOff[3 * ((yh - 1) + HY * zh)]),       rE1 = uniform(sOff[3 * ((yh + 1) + HY * zh) + 3]);
      const int rS2 = uniform(sOff[3 * ((yh - 1) + HY * (zh + 1))]), rE2 = uniform(sOff[3 * ((yh + 1) + HY * (zh + 1)) + 3]);
      const int l0 = rE0 - rS0, l01 = l0 + (rE1 - rS1), total = l01 + (rE2 - rS2);
      auto recOf = [&](const int t) { return t < l0 ? rS0 + t : t < l01 ? rS1 + (t - l0) : rS2 + (t - l01); };

      // build of a round's rows: two atoms at a time against ALL the cell's stencil records -- lane = record; for round 0 the
      // records sit in registers (read from the image once per cell), the atoms' positions are broadcast reads -- ballot + mbcnt append
      // the hits, in record order, to the atoms' rows
      auto buildRound = [&](const int i0, const bool fromRegs, const int (&rec)[SR], const real_t (&vx)[SR], const real_t (&vy)[SR], const real_t (&vz)[SR]) {
         const int nRound = ni - i0 < roundAtoms ? ni - i0 : roundAtoms;
         for (int pa = 0; pa < nRound; pa += 2) {
            const int recA = ownStart + i0 + pa, recB = pa + 1 < nRound ? recA + 1 : recA;
            const real_t xA = sp[REC * recA], yA = sp[REC * recA + 1], zA = sp[REC * recA + 2];
            const real_t xB = sp[REC * recB], yB = sp[REC * recB + 1], zB = sp[REC * recB + 2];
            unsigned short* __restrict__ rowA = sHit + pa * strideL;
            unsigned short* __restrict__ rowB = rowA + strideL;      // (with an odd atom count the last B repeats A and lands in an unused row)
            int nA = 0, nB = 0;
            auto sweep = [&](const int r, const real_t px, const real_t py, const real_t pz) {
               const real_t ax = xA - px, ay = yA - py, az = zA - pz;
               const real_t bx_ = xB - px, by_ = yB - py, bz_ = zB - pz;
               const real_t r2A = ax * ax + ay * ay + az * az, r2B = bx_ * bx_ + by_ * by_ + bz_ * bz_;
               // Lanes past the stencil hold a record at FAR_AWAY.  The ballot of a compare IS the compare's mask (the ballot of an and-ed bool costs
               // a cndmask and a second compare), and inverse_ballot turns the and-ed mask into the exec mask of the append without a VALU instruction.
               const unsigned long long mA = __builtin_amdgcn_ballot_w64(r2A <= rTest2) & __builtin_amdgcn_ballot_w64(r != recA);
               const unsigned long long mB = __builtin_amdgcn_ballot_w64(r2B <= rTest2) & __builtin_amdgcn_ballot_w64(r != recB);
               const bool hitA = __builtin_amdgcn_inverse_ballot_w64(mA), hitB = __builtin_amdgcn_inverse_ballot_w64(mB);
               const int kA = nA + __builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u));
               const int kB = nB + __builtin_amdgcn_mbcnt_hi((unsigned)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB, 0u));
               if (hitA && kA < b.rows) rowA[kA] = (unsigned short)r;
               if (hitB && kB < b.rows) rowB[kB] = (unsigned short)r;
               nA += __popcll(mA); nB += __popcll(mB);
            };
            int tDone = 0;
            if (b.debug & 1) tDone = total;
            else if (fromRegs) {
#pragma unroll
               for (int g = 0; g < SR; ++g)
                  if (g * 64 < total) sweep(rec[g], vx[g], vy[g], vz[g]);      // wave-uniform condition
               tDone = SR * 64;
            }
            for (int t0 = tDone; t0 < total; t0 += 64) {
               const int t = t0 + lane;
               const int r = recOf(t < total ? t : 0);
               sweep(r, t < total ? sp[REC * r] : FAR_AWAY, sp[REC * r + 1], sp[REC * r + 2]);
            }
            if (lane == 0) { sCnt[pa] = nA; sCnt[pa + 1] = nB; }
         }
      };
      if (BUILDS) {
         int rec[SR]; real_t vx[SR], vy[SR], vz[SR];
#pragma unroll
         for (int g = 0; g < SR; ++g) {
            const int t = g * 64 + lane;
            const int r = recOf(t < total ? t : 0);
            rec[g] = r; vx[g] = t < total ? sp[REC * r] : FAR_AWAY; vy[g] = sp[REC * r + 1]; vz[g] = sp[REC * r + 2];
         }
         buildRound(0, true, rec, vx, vy, vz);
      }

      // up to 16 atoms per round: evaluate their rows
      for (int i0 = 0; i0 < ni; i0 += roundAtoms) {
         int nRound, L, ia, q; roundOf(ni, i0, nRound, L, ia, q);
         const bool have = ia < nRound;
         const unsigned ii = have ? (unsigned)(i0 + ia) : 0u;
         const int recI = ownStart + (int)ii;
         const real_t xi = sp[REC * recI], yi = sp[REC * recI + 1], zi = sp[REC * recI + 2];
         const real_t dfi = (STEP == 3) ? sd[recI] : R(0.0);
         if (FETCHES && i0 != 0) fetch3(iBox, ni, i0, cur);          // later rounds of a cell of more than 16 atoms: blocking
         int n = 0;
         if (BUILDS) {
            if (i0 != 0) {                                   // (round 0 was built above)
               const int rec[SR] = { 0 }; const real_t v0[SR] = { R(0.0) };
               __builtin_amdgcn_wave_barrier();
               buildRound(i0, false, rec, v0, v0, v0);
            }
            __builtin_amdgcn_wave_barrier();
            n = have ? sCnt[ia] : 0;
         } else {
            n = have ? cur.n : 0;
         }
         if (n > b.rows) { over = true; n = b.rows; }
         if (b.debug & 2) n = 0;
         const int nPairs = (n + 1) >> 1;
         if (LISTED && STEP != 0 && b.listQuads > 2 && __builtin_amdgcn_ballot_w64(8 * L + q < nPairs) != 0ull) {      // a row of more than 16 L entries: blocking
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(b.rowsG) + ((size_t)iBox * b.listRounds + (unsigned)(i0 / roundAtoms)) * b.listQuads * 64 + lane;
            if (8 * L + q < nPairs) cur.ex = src[128];
         }
         if (!LISTED && STEP == 3 && L > 5 && __builtin_amdgcn_ballot_w64(4 * L + q < nPairs) != 0ull) {      // a long row in a round of few atoms: blocking
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(b.rowsG) + ((size_t)iBox * b.listRounds + (unsigned)(i0 / roundAtoms)) * b.listQuads * 64 + lane;
            if (4 * L + q < nPairs) cur.hi = src[64];
         }
         const unsigned short* __restrict__ myRow = sHit + (have ? ia : 0) * strideL;
         unsigned wReg[12] = { cur.lo.x, cur.lo.y, cur.lo.z, cur.lo.w, cur.hi.x, cur.hi.y, cur.hi.z, cur.hi.w, cur.ex.x, cur.ex.y, cur.ex.z, cur.ex.w };

         if (STEP == 0) {
            // ---- list build: hand the round's rows to the force passes.  Pair p of an atom's row is word p / L of its lane p % L; what lies beyond the row
            // names the far-away record (imageTotal), so the evaluation needs no lengths inside a word.
            longest = n > longest ? n : longest;
            const unsigned pad = (unsigned)imageTotal;
#pragma unroll
            for (int u = 0; u < W; ++u) {
               const int pr = u * L + q;
               unsigned w = pad | (pad << 16);
               if (have && pr < nPairs) {
                  const unsigned e0 = myRow[2 * pr], e1 = 2 * pr + 1 < n ? (unsigned)myRow[2 * pr + 1] : pad;
                  w = e0 | (e1 << 16);
               }
               wReg[u] = w;
            }
            uint4* __restrict__ dst = reinterpret_cast<uint4*>(b.rowsG) + ((size_t)iBox * b.listRounds + (unsigned)(i0 / roundAtoms)) * b.listQuads * 64 + lane;
            dst[0] = make_uint4(wReg[0], wReg[1], wReg[2], wReg[3]);
            if (b.listQuads > 1) dst[64] = make_uint4(wReg[4], wReg[5], wReg[6], wReg[7]);
            if (b.listQuads > 2) dst[128] = make_uint4(wReg[8], wReg[9], wReg[10], wReg[11]);
            if (lane < nRound) (b.rowCountG + cellSlot)[(unsigned)(i0 + lane)] = (unsigned short)(sCnt[lane] < b.rows ? sCnt[lane] : b.rows);
            __builtin_amdgcn_wave_barrier();
            continue;
         }

         real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0), rb = R(0.0);
         // two pairs per trip, branch-free.  Not listed: a missing second pair is evaluated at r = cutoff and weighted 0.  Listed: every number names a
         // record (the far-away one past the end of a row); a pair outside the cutoff is evaluated AT the cutoff and enters with weight 0.
         auto evalTrip = [&](const int j0, const int j1, const bool h1) {
            const real_t* r0 = sp + REC * j0; const real_t* r1 = sp + REC * j1;
            const real_t dx0 = xi - r0[0], dy0 = yi - r0[1], dz0 = zi - r0[2];
            const real_t dx1 = xi - r1[0], dy1 = yi - r1[1], dz1 = zi - r1[2];
            real_t s0 = dx0*dx0 + dy0*dy0 + dz0*dz0;
            real_t s1 = (LISTED || h1) ? dx1*dx1 + dy1*dy1 + dz1*dz1 : a.rc2;
            real_t w0 = R(1.0), w1 = R(1.0);
            if (LISTED) {
               w0 = s0 <= a.rc2 ? R(1.0) : R(0.0); w1 = s1 <= a.rc2 ? R(1.0) : R(0.0);
               s0 = minR(s0, a.rc2); s1 = minR(s1, a.rc2);
            }
            real_t rho0, drho0, dphi0, rho1, drho1, dphi1;
            if (SPLINE) {                                    // -P: cubic splines in r^2 give (1/r) d/dr directly, no square root
               interpolateSpline(a.rhoS, s0, rho0, drho0); interpolateSpline(a.rhoS, s1, rho1, drho1);
               if (STEP == 1) {
                  real_t phi0, phi1;
                  interpolateSpline(a.phiS, s0, phi0, dphi0); interpolateSpline(a.phiS, s1, phi1, dphi1);
                  if (LISTED) { e = fmaR(phi0, w0, e); e = fmaR(phi1, w1, e); rb = fmaR(rho0, w0, rb); rb = fmaR(rho1, w1, rb); }
                  else { e += phi0 + (h1 ? phi1 : R(0.0)); rb += rho0 + (h1 ? rho1 : R(0.0)); }
               } else {
                  dphi0 = (dfi + sd[STEP == 3 ? j0 : 0]) * drho0; dphi1 = (dfi + sd[STEP == 3 ? j1 : 0]) * drho1;
               }
               if (LISTED) { dphi0 *= w0; dphi1 *= w1; }
               else dphi1 = h1 ? dphi1 : R(0.0);
            } else {
               const real_t ir0 = rsqrtR(s0), ir1 = rsqrtR(s1);
               const real_t d0 = s0 * ir0, d1 = s1 * ir1;
               if (STEP == 1) {
                  real_t phi0, phi1;
                  if (sameGrid) { interpolatePair<CLAMP>(sRho, rhoT, d0, phi0, dphi0, rho0, drho0); interpolatePair<CLAMP>(sRho, rhoT, d1, phi1, dphi1, rho1, drho1); }
                  else { interpolate<CLAMP>(rhoT, d0, rho0, drho0); interpolate<CLAMP>(phiT, d0, phi0, dphi0); interpolate<CLAMP>(rhoT, d1, rho1, drho1); interpolate<CLAMP>(phiT, d1, phi1, dphi1); }
                  if (LISTED) { e = fmaR(phi0, w0, e); e = fmaR(phi1, w1, e); rb = fmaR(rho0, w0, rb); rb = fmaR(rho1, w1, rb); }
                  else { e += phi0 + (h1 ? phi1 : R(0.0)); rb += rho0 + (h1 ? rho1 : R(0.0)); }
               } else {
                  interpolate<CLAMP>(rhoT, d0, rho0, drho0); interpolate<CLAMP>(rhoT, d1, rho1, drho1);
                  dphi0 = (dfi + sd[STEP == 3 ? j0 : 0]) * drho0; dphi1 = (dfi + sd[STEP == 3 ? j1 : 0]) * drho1;
               }
               if (LISTED) { dphi0 = dphi0 * (ir0 * w0); dphi1 = dphi1 * (ir1 * w1); }
               else { dphi0 = dphi0 * ir0; dphi1 = h1 ? dphi1 * ir1 : R(0.0); }
            }
            fx -= dphi0 * dx0; fy -= dphi0 * dy0; fz -= dphi0 * dz0;
            fx -= dphi1 * dx1; fy -= dphi1 * dy1; fz -= dphi1 * dz1;
         };
         // the lane's pairs: q, q + L, q + 2L, ... below nPairs, one per trip (rows <= 2 W L: at most W trips)
#pragma unroll
         for (int u = 0; u < W; ++u) {                       // (unrolled: the words sit in, or go to, registers)
            const int pr = u * L + q;
            if (pr < nPairs) {
               if (BUILDS) wReg[u] = *reinterpret_cast<const unsigned*>(myRow + 2 * pr);      // entries 2 pr, 2 pr + 1
               const bool h1 = 2 * pr + 1 < n;
               evalTrip((int)(wReg[u] & 0xffffu), (LISTED || h1) ? (int)(wReg[u] >> 16) : recI, h1);
            }
         }
         // The L lanes of an atom are consecutive: a shift-down tree adds them into the first (quad-permute DPP when L is 4).  One ds_bpermute address
         // and one weight per step serve all the sums: lane q takes lane q + d with weight 1 while q + d < L, else weight 0 (what it reads there is
         // another atom's finite partial sum) -- an fma where round 2's tree spent two selects and an add per value; steps with d >= L are skipped.
         if (L == 4) {
            fx = quadSum(fx); fy = quadSum(fy); fz = quadSum(fz);
            if (STEP == 1) { e = quadSum(e); rb = quadSum(rb); }
         } else {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
               if (d < L) {                                   // (wave-uniform)
                  const int from = ((lane + d) & 63) << 2;
                  const real_t w = q + d < L ? R(1.0) : R(0.0);
                  fx = fmaR(bpermuteAddrR(fx, from), w, fx); fy = fmaR(bpermuteAddrR(fy, from), w, fy); fz = fmaR(bpermuteAddrR(fz, from), w, fz);
                  if (STEP == 1) { e = fmaR(bpermuteAddrR(e, from), w, e); rb = fmaR(bpermuteAddrR(rb, from), w, rb); }
               }
            }
         }
         // Order of the memory operations at the end of a round: the table samples of the embedding (pass 2 for these atoms: EAM_Force_embed needs
         // only the atom's own rhobar) are REQUESTED FIRST and every store comes after them -- loads and stores retire in issue order, and a load
         // queued behind this round's stores would wait for their acknowledgements; the stores of the previous cell are a cell of arithmetic old.
         const bool sumLane = have && q == 0;
         real_t pR = R(0.0), pv0 = R(0.0), pv1 = R(0.0), pv2 = R(0.0), pv3 = R(0.0);
         if (STEP == 1 && b.fuseEmbed && sumLane) {           // interpolate() split in two
            real_t r = maxR(rb, a.f.x0);
            r = minR(r, a.f.xn);
            r = r * a.f.invDx - a.f.invDxXx0;
            const real_t ri = floorR(r);
            const int it = (int)ri;
            pR = r - ri;
            pv0 = a.f.values[it]; pv1 = a.f.values[it + 1]; pv2 = a.f.values[it + 2]; pv3 = a.f.values[it + 3];
         }
         if (BUILDS) {
            // hand the lane's words to pass 3 (the second 16 bytes only when a trip beyond the fourth was made), and the row lengths
            // ([round 4] per (cell, round) [quad][64 lanes] 16-byte elements, as the Verlet rows: a round's first quads are one dense KB -- whole lines written
            // and read -- where a [slot][16 lanes][8 words] block per atom left 32-byte sectors half used and 2.4 GB allocated at 80^3)
            if (have) {
               uint4* __restrict__ dst = reinterpret_cast<uint4*>(b.rowsG) + ((size_t)iBox * b.listRounds + (unsigned)(i0 / roundAtoms)) * b.listQuads * 64 + lane;
               if (q < nPairs) dst[0] = make_uint4(wReg[0], wReg[1], wReg[2], wReg[3]);
               if (4 * L + q < nPairs) dst[64] = make_uint4(wReg[4], wReg[5], wReg[6], wReg[7]);
            }
            if (lane < nRound) (b.rowCountG + cellSlot)[(unsigned)(i0 + lane)] = (unsigned short)(sCnt[lane] < b.rows ? sCnt[lane] : b.rows);
         }
         if (sumLane) {
            if (STEP == 1) { (a.fx + cellSlot)[ii] = fx; (a.fy + cellSlot)[ii] = fy; (a.fz + cellSlot)[ii] = fz; (a.rhobar + cellSlot)[ii] = rb; }
            else { (a.fx + cellSlot)[ii] = cur.f0x + fx; (a.fy + cellSlot)[ii] = cur.f0y + fy; (a.fz + cellSlot)[ii] = cur.f0z + fz; }
         }
         if (STEP == 1 && sumLane) {
            real_t ei = R(0.5) * e;
            if (b.fuseEmbed) {
               const real_t g1 = pv2 - pv0, g2 = pv3 - pv1;
               ei += pv1 + R(0.5) * pR * (g1 + pR * (pv2 + pv0 - R(2.0) * pv1));
               (a.dfEmbed + cellSlot)[ii] = (g1 + pR * (g2 - g1)) * a.f.invDxHalf;
            }
            (a.e + cellSlot)[ii] = ei;
         }
         __builtin_amdgcn_wave_barrier();
      }
   }
   if (__builtin_amdgcn_ballot_w64(over) != 0ull && lane == 0) atomicOr(&b.status[3], LISTED ? 2 : 1);
   if (STEP == 0) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(longest, m); longest = o > longest ? o : longest; }
      if (lane == 0 && longest > 0) atomicMax(&b.stats[0], longest);
   }
}
