// eam_atom_brick_kernels.h -- EAM_Force_atom_brick: the thread-per-atom EAM kernel (method thread_atom, passes 1 and 3) with the brick image as its staging.
//
// Physics and results of the reference's EAM_Force_thread_atom<step> (gpu_eam_thread_atom.h:32-140): one THREAD owns one atom from the first candidate to the
// stores.  What round 2's kernel of that shape (eam_kernels.h) paid for was the walk: every thread streamed the 283 atoms of its 27-cell stencil from L2 and
// evaluated under divergence with one lane in seven inside the cutoff.  Here
//   * a workgroup owns a brick of 1 x BY x BZ link cells (comd_device.hip picks the shape) and stages the 3 x (BY + 2) x (BZ + 2) cells around it ONCE into the LDS, densely
//     packed, z-y-x order (the staging of eam_brick_kernels.h: the stencil of a cell is three contiguous runs of records);
//   * thread t takes the brick's t-th atom and works in two phases.  TEST: walk the three runs of its cell's stencil -- lanes of one cell read the same record
//     (an LDS broadcast) -- and append the records inside the cutoff to a row of its own in the LDS: one BYTE per neighbour, its offset inside its run (a
//     run is 9 cells, ~95 records; an atom with a run of more than 256 walks without a row), the three runs' parts back to back, each on an even byte; 60-byte
//     stride, an odd count of dwords: the rows of a wave's lanes start on distinct banks.  EVALUATE: walk that row, two pairs per trip, branch-free inside
//     the trip.  The divergence that is left is the row LENGTH (42 +- a few at 80^3), not the acceptance rate;
//   * pass 1 leaves the rows in memory, [brick][16-byte chunk][atom of the brick] (a wave's stores are dense KBs; the atom's index counts through ALL cells of
//     the brick, selected or not, so launches over different cell lists never write the same element), the three counts beside them, and the selection it
//     staged the brick under beside every cell.  Pass 3 asks for an atom's row (64 bytes: four registers of four) while the block is being staged and, when
//     every atom of the wave has a row numbered against this image (the cell was staged under the same selection: the offsets index the image, whose
//     composition depends on it), evaluates STRAIGHT FROM THE REGISTERS -- no TEST phase, and no rows in pass 3's LDS at all.  A wave that finds anything
//     else walks its stencils again, testing and evaluating as it goes (slow, rare: a launch over any cell list is complete in itself, the hand-over is an
//     optimisation the two passes need not agree on); when the launcher knows there is nothing to read (COMD_EAM_ATOM_HANDOVER=0, rows beyond 64 bytes) pass 3
//     gets rows in the LDS and works as pass 1 does;
//   * the LDS decides the occupancy, and the occupancy the speed (the kernel issues 180 VALU instructions per atom in the two passes where cta_cell's brick
//     kernel issues 235, and waits for the LDS): byte rows, rows only for the waves that take atoms (192 threads for a brick's 168 atoms; the fourth wave helps
//     to stage), none in pass 3 -- 52 KB and 47 KB per workgroup at 80^3, THREE workgroups per CU in both passes (16-bit rows and rows in pass 3: two, 1.50 ms
//     per force evaluation instead of 1.20).
// A brick whose block outgrows the image takes the streaming form (eamCellDirect, a wave per cell) -- counted in stats[1], the host re-sizes the image.
#pragma once
#include "eam_brick_kernels.h"

#define EAM_ATOM_BRICK_THREADS 384       // the largest workgroup (six waves); a launch uses 256 or 384 (comd_device.hip: enough for the block's slots in EAM_BRICK_STAGE rounds)
#define EAM_ATOM_MAX_CELLS 192           // cells of the staged block: 3 * (BY + 2) * (BZ + 2) <= 192 (three per lane of the wave that scans their occupancies)

#define EAM_ATOM_ROW_CHUNKS 4            // 16-byte chunks of a row handed from pass 1 to pass 3: rows of up to 64 bytes
// a thread's row: `rows` bytes + 4 of padding (a group of four candidates is appended before the row's end is looked at; an odd count of dwords for rows % 8 == 0)
__host__ __device__ static inline int eamAtomBrickRowStride(int rows) { return rows + 4; }
// LDS of one workgroup: tables | image (+ F' in pass 3) | offsets, cell ids, scalars, selected cells (eamBrickSharedBytes) | own-atom prefix | rows
// (rowThreads: the threads that take atoms -- whole waves, enough for the brick's atoms; EamBrickArgs.listRounds carries the number to the kernel.  The
// other waves of the 256 only help to stage: rows for threads that never have an atom would cost the second workgroup of a CU.)
// (ldsRows: pass 1 always; pass 3 only when it cannot read the rows of pass 1)
static inline size_t eamAtomBrickLdsBytes(int step, size_t tableDoubles, int imageCap, int rows, int rowThreads, bool ldsRows)
{
   return eamTableBytesAligned(tableDoubles) + ((((size_t)(step == 3 ? 4 : 3) * imageCap * sizeof(real_t)) + 15) & ~(size_t)15)      // records (+ F' in pass 3)
          + (size_t)(EAM_ATOM_MAX_CELLS + 4) * 4 + (size_t)EAM_ATOM_MAX_CELLS * 4 + 64 + 64       // offsets, cell ids, scalars, list of selected cells
          + (68 + 64) * sizeof(int) + (ldsRows ? (size_t)rowThreads * eamAtomBrickRowStride(rows) + 16 : 0);
}

// (the scans and the per-run wave minimum run on the DPP path: waveInclusiveScan, waveMin of eam_brick_kernels.h)
template <int STEP, bool LDS_TABLES, bool SPLINE, bool CLAMP>
__global__ __launch_bounds__(EAM_ATOM_BRICK_THREADS, 2)
void EAM_Force_atom_brick(EamArgs a, EamBrickArgs b)
{
   static_assert(STEP == 1 || STEP == 3, "passes 1 and 3");
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   constexpr int REC = 3, SLOT_BITS = 4, SLOTS = 1 << SLOT_BITS, STAGE = EAM_BRICK_STAGE;
   const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), nThreads = blockDim.x, nWaves = nThreads >> 6;
   const int nRhoPad = a.rho.n + 3;
   const bool sameGrid = (STEP == 1) && LDS_TABLES && a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   real_t* sRho = (real_t*)ldsRaw;
   real_t* sPhi = sRho + nRhoPad;
   int tableDoubles = 0;
   if (LDS_TABLES) tableDoubles = (STEP == 1) ? 2 * nRhoPad + (sameGrid ? 0 : (a.phi.n + 3 - nRhoPad)) : nRhoPad;

   real_t* __restrict__ sp = (real_t*)(ldsRaw + eamTableBytesAligned(tableDoubles));
   real_t* __restrict__ sd = sp + REC * b.imageCap;          // [imageCap] F' (pass 3)
   int* sOff = (int*)(ldsRaw + eamTableBytesAligned(tableDoubles) + ((((size_t)(STEP == 3 ? 4 : 3) * b.imageCap * sizeof(real_t)) + 15) & ~(size_t)15));
   int* sBox = sOff + EAM_ATOM_MAX_CELLS + 4;                // [192] cell ids of the block
   int* sMisc = sBox + EAM_ATOM_MAX_CELLS;                   // [16]: 0/1 selection mask, 4 records in the image
   unsigned char* sList = (unsigned char*)(sMisc + 16);      // [64] selected cells of the brick, compacted
   int* sOwn = (int*)(sList + 64);                           // [65] atoms of the selected cells before cell k of that list
   int* sFull = sOwn + 68;                                   // [64] atoms of the brick before brick cell c, selected or not (the hand-over's atom index)
   const int strideL = eamAtomBrickRowStride(b.rows);
   unsigned char* __restrict__ myRow = (unsigned char*)(sFull + 64) + (size_t)tid * strideL;

   // ---- the brick and its selected cells (as EAM_Force_cta_brick) ---------------------------------------------------------------------------
   const int gx = b.geom.g[0], gy = b.geom.g[1], gz = b.geom.g[2];
   const int bid = b.brickList ? b.brickList[xcdRemap(blockIdx.x, gridDim.x)] : xcdRemap(blockIdx.x, gridDim.x);      // x fastest: consecutive bricks share two thirds of their block
   const int bx = bid % gx, by0 = ((bid / gx) % b.nby) * b.by, bz0 = (bid / (gx * b.nby)) * b.bz;
   const int HY = b.by + 2, HZ = b.bz + 2, NH = 3 * HY * HZ, NC = b.by * b.bz;
   unsigned long long selMask;
   if (b.sel) {      // a launch over a cell list: the marks of the brick's cells
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
      if (selMask == 0ull) return;
   } else {
      bool s = false;
      if (lane < NC) s = by0 + lane % b.by < gy && bz0 + lane / b.by < gz;
      selMask = __builtin_amdgcn_ballot_w64(s);
   }
   const int nSel = __popcll(selMask);
   if (wave == 0) {
      const bool s = (selMask >> lane) & 1ull;
      if (s) sList[__builtin_amdgcn_mbcnt_hi((unsigned)(selMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)selMask, 0u))] = (unsigned char)lane;
   }

   int brickCnt = 0;                                         // wave 0, lane c: atoms of brick cell c (local cells: their occupancies stand during a force evaluation)
   if (b.rowsG && wave == 0 && lane < NC && by0 + lane % b.by < gy && bz0 + lane / b.by < gz) brickCnt = a.nAtoms[comdBoxFromTuple(&b.geom, bx, by0 + lane % b.by, bz0 + lane / b.by)];

   // ---- ONE round trip for the block: occupancies and records are requested together (slot s of block cell h = task SLOTS h + s) ---------------
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
                      && bz0 >= 1 && bz0 + b.bz <= gz - 1;
   if (plain) {
      const int hyMagic = (65536 + HY - 1) / HY;             // t / HY = (t * hyMagic) >> 16 for t < 128 (t = h / 3 < 64)
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
      // every wave works out the cell ids of all 192 block cells (three per lane; -1: outside the grid or in no selected cell's stencil -- a launch over the
      // interior cells runs while the halo cells are being filled and must not look at them)
      auto blockBox = [&](const int h) {
         int box = -1;
         if (h < NH) {
            const int xh = h % 3, yh = (h / 3) % HY, zh = h / (3 * HY);
            const int iy = by0 + yh - 1, iz = bz0 + zh - 1;
            if (iy <= gy && iz <= gz) {
               bool need = false;
#pragma unroll
               for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
                  for (int dy = -1; dy <= 1; ++dy) {
                     const int cy = yh - 1 + dy, cz = zh - 1 + dz;
                     if (cy >= 0 && cy < b.by && cz >= 0 && cz < b.bz) need = need || ((selMask >> (cy + b.by * cz)) & 1ull);
                  }
               if (need) box = comdBoxFromTuple(&b.geom, bx + xh - 1, iy, iz);
            }
         }
         return box;
      };
      const int boxLo = blockBox(lane), boxHi = blockBox(64 + lane), boxTop = blockBox(128 + lane);
      if (wave < 3) myBox = wave == 0 ? boxLo : wave == 1 ? boxHi : boxTop;
#pragma unroll
      for (int k = 0; k < STAGE; ++k) {
         const int task = k * nThreads + tid, h = task >> SLOT_BITS;
         const int fromLo = __builtin_amdgcn_ds_bpermute((h & 63) << 2, boxLo), fromHi = __builtin_amdgcn_ds_bpermute((h & 63) << 2, boxHi);
         const int fromTop = __builtin_amdgcn_ds_bpermute((h & 63) << 2, boxTop);
         request(k, h < NH ? (h < 64 ? fromLo : h < 128 ? fromHi : fromTop) : -1, task & (SLOTS - 1));
      }
   }
   int myCnt = 0;
   if (myBox >= 0) myCnt = a.nAtoms[myBox];
   if (LDS_TABLES) {
      if (sameGrid) {
         for (int t = tid; t < nRhoPad; t += nThreads) { sRho[2 * t] = a.phi.values[t]; sRho[2 * t + 1] = a.rho.values[t]; }
      } else {
         for (int t = tid; t < nRhoPad; t += nThreads) sRho[t] = a.rho.values[t];
         if (STEP == 1) for (int t = tid; t < a.phi.n + 3; t += nThreads) sPhi[t] = a.phi.values[t];
      }
   }
   const TableView rhoT = makeTable(a.rho, LDS_TABLES ? sRho : a.rho.values), phiT = makeTable(a.phi, LDS_TABLES ? sPhi : a.phi.values);
   if (tid < EAM_ATOM_MAX_CELLS) { sBox[tid] = myBox >= 0 ? myBox : 0; sOff[tid] = myCnt; }
   __syncthreads();
   if (wave == 0) {                                          // exclusive scan of the 192 counts: lane l takes entries 3l, 3l + 1, 3l + 2
      const int c0 = sOff[3 * lane], c1 = sOff[3 * lane + 1], c2 = sOff[3 * lane + 2];
      const int incl = waveInclusiveScan(c0 + c1 + c2);
      const int excl = incl - c0 - c1 - c2;
      sOff[3 * lane] = excl; sOff[3 * lane + 1] = excl + c0; sOff[3 * lane + 2] = excl + c0 + c1;
      if (lane == 63) { sOff[EAM_ATOM_MAX_CELLS] = incl; sMisc[4] = incl; }
      // the brick's own atoms, cell after cell of the selection: lane k = the k-th selected cell
      __builtin_amdgcn_wave_barrier();
      int own = 0;
      if (lane < nSel) {
         const int cl = sList[lane], hc = 1 + 3 * ((cl % b.by + 1) + HY * (cl / b.by + 1));
         own = sOff[hc + 1] - sOff[hc];
      }
      const int inclOwn = waveInclusiveScan(own);
      sOwn[lane + 1] = inclOwn;
      if (lane == 0) sOwn[0] = 0;
      sFull[lane] = waveInclusiveScan(brickCnt) - brickCnt;
   }
   __syncthreads();
   const int imageTotal = uniform(sMisc[4]);
   const bool fits = imageTotal <= b.imageCap;
   if (!fits) {      // a block larger than the LDS image: streaming form, a wave per cell, same tables
      if (STEP == 1 && b.stats && tid == 0) atomicAdd(&b.stats[1], 1);
      for (int pick = wave; pick < nSel; pick += nWaves) {
         const int cl = sList[pick], hc = 1 + 3 * ((cl % b.by + 1) + HY * (cl / b.by + 1));
         if (STEP == 1 && b.brickSel && lane == 0) b.brickSel[sBox[hc]] = 0ull;      // no rows for this cell
         eamCellDirect<STEP, SPLINE>(a, uniform(sBox[hc]), lane, rhoT, phiT, sameGrid, b.fuseEmbed);
      }
      return;
   }
   // own atom t of the brick -> its cell (k-th selected, brick cell cl) and its number in the cell
   const int nOwn = uniform(sOwn[nSel]);
   const int rowThreads = b.listRounds;                      // threads that take atoms (whole waves)
   const int rowCap = b.listQuads >> 8;                      // atoms of a brick that can leave a row for pass 3 (the launcher: the row threads rounded up to 256 or 512)
   const bool ldsRows = STEP == 1 || (b.listQuads & 1) != 0;       // pass 3 has rows in the LDS only where the launcher found no hand-over to read (else the LDS they would take buys a workgroup per CU)
   auto ownAtom = [&](const int t, int& cl, int& ia) {
      int k = 0;
      for (int s = 1; s < nSel; ++s) k += t >= sOwn[s] ? 1 : 0;
      cl = sList[k]; ia = t - sOwn[k];
   };
   // the selection is written beside the cells whose rows this launch leaves (pass 1); pass 3 asks for the first round's rows now, a staging away from their use
   constexpr int CH = EAM_ATOM_ROW_CHUNKS;
   uint4 pre[CH];
   unsigned preN = 0xffffffffu;
   unsigned long long preSel = ~selMask;
   const uint4* __restrict__ rowsG4 = reinterpret_cast<const uint4*>(b.rowsG) + (size_t)bid * CH * rowCap;      // [brick][chunk][atom of the brick]
   unsigned* __restrict__ rowCountG = reinterpret_cast<unsigned*>(b.rowCountG) + (size_t)bid * rowCap;        // [brick][atom of the brick] n0 | n1 << 8 | n2 << 16
#pragma unroll
   for (int c = 0; c < CH; ++c) pre[c] = make_uint4(0u, 0u, 0u, 0u);
   if (STEP == 3 && b.rowsG && (wave << 6) < rowThreads && (wave << 6) < nOwn) {
      int cl, ia; ownAtom((wave << 6) + lane < nOwn ? (wave << 6) + lane : nOwn - 1, cl, ia);
      const int full = sFull[cl] + ia;
      if (full < rowCap) {
         preN = rowCountG[full];
         preSel = b.brickSel[sBox[1 + 3 * ((cl % b.by + 1) + HY * (cl / b.by + 1))]];
#pragma unroll
         for (int c = 0; c < CH; ++c) pre[c] = rowsG4[c * rowCap + full];
      }
   }
   if (STEP == 1 && b.rowsG && wave == 0 && lane < nSel) {
      const int cl = sList[lane];
      b.brickSel[sBox[1 + 3 * ((cl % b.by + 1) + HY * (cl / b.by + 1))]] = selMask;
   }
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
   __syncthreads();

   if (b.debug & 4) return;
   // ---- thread t takes own atom t, t + rowThreads, ... ---------------------------------------------------------------------------------------------
   if ((wave << 6) >= rowThreads) return;
   for (int base = wave << 6; base < nOwn; base += rowThreads) {      // (wave-uniform: a wave whose 64 atoms do not exist has nothing to do)
      const bool have = base + lane < nOwn;
      const int t = have ? base + lane : nOwn - 1;           // lanes past the last atom repeat it and store nothing: the walk below has wave-uniform parts
      int cl, ia; ownAtom(t, cl, ia);
      const int yh = cl % b.by + 1, zh = cl / b.by + 1, hc = 1 + 3 * (yh + HY * zh);
      const int iBox = sBox[hc], recI = sOff[hc] + ia;
      const bool firstRound = base == (wave << 6);
      const real_t xi = sp[REC * recI], yi = sp[REC * recI + 1], zi = sp[REC * recI + 2];
      const real_t dfi = (STEP == 3) ? sd[recI] : R(0.0);
      // the three runs of the cell's stencil (rows yh-1 .. yh+1 of a z plane lie back to back)
      const int rs0 = sOff[3 * ((yh - 1) + HY * (zh - 1))], len0 = sOff[3 * ((yh + 1) + HY * (zh - 1)) + 3] - rs0;
      const int rs1 = sOff[3 * ((yh - 1) + HY * zh)],       len1 = sOff[3 * ((yh + 1) + HY * zh) + 3] - rs1;
      const int rs2 = sOff[3 * ((yh - 1) + HY * (zh + 1))], len2 = sOff[3 * ((yh + 1) + HY * (zh + 1)) + 3] - rs2;

      real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0), rb = R(0.0);
      // two pairs per trip, branch-free; a missing second pair is evaluated at r = cutoff and weighted 0
      auto evalTrip = [&](const int j0, const int j1, const bool h1) __attribute__((always_inline)) {
         const real_t* r0 = sp + REC * j0; const real_t* r1 = sp + REC * j1;
         const real_t dx0 = xi - r0[0], dy0 = yi - r0[1], dz0 = zi - r0[2];
         const real_t dx1 = xi - r1[0], dy1 = yi - r1[1], dz1 = zi - r1[2];
         const real_t s0 = dx0*dx0 + dy0*dy0 + dz0*dz0;
         const real_t s1 = h1 ? dx1*dx1 + dy1*dy1 + dz1*dz1 : a.rc2;
         real_t rho0, drho0, dphi0, rho1, drho1, dphi1;
         if (SPLINE) {                                       // -P: cubic splines in r^2 give (1/r) d/dr directly, no square root
            interpolateSpline(a.rhoS, s0, rho0, drho0); interpolateSpline(a.rhoS, s1, rho1, drho1);
            if (STEP == 1) {
               real_t phi0, phi1;
               interpolateSpline(a.phiS, s0, phi0, dphi0); interpolateSpline(a.phiS, s1, phi1, dphi1);
               e += phi0 + (h1 ? phi1 : R(0.0)); rb += rho0 + (h1 ? rho1 : R(0.0));
            } else {
               dphi0 = (dfi + sd[STEP == 3 ? j0 : 0]) * drho0; dphi1 = (dfi + sd[STEP == 3 ? j1 : 0]) * drho1;
            }
            dphi1 = h1 ? dphi1 : R(0.0);
         } else {
            const real_t ir0 = rsqrtR(s0), ir1 = rsqrtR(s1);
            const real_t d0 = s0 * ir0, d1 = s1 * ir1;
            if (STEP == 1) {
               real_t phi0, phi1;
               if (sameGrid) { interpolatePair<CLAMP>(sRho, rhoT, d0, phi0, dphi0, rho0, drho0); interpolatePair<CLAMP>(sRho, rhoT, d1, phi1, dphi1, rho1, drho1); }
               else { interpolate<CLAMP>(rhoT, d0, rho0, drho0); interpolate<CLAMP>(phiT, d0, phi0, dphi0); interpolate<CLAMP>(rhoT, d1, rho1, drho1); interpolate<CLAMP>(phiT, d1, phi1, dphi1); }
               e += phi0 + (h1 ? phi1 : R(0.0)); rb += rho0 + (h1 ? rho1 : R(0.0));
            } else {
               interpolate<CLAMP>(rhoT, d0, rho0, drho0); interpolate<CLAMP>(rhoT, d1, rho1, drho1);
               dphi0 = (dfi + sd[STEP == 3 ? j0 : 0]) * drho0; dphi1 = (dfi + sd[STEP == 3 ? j1 : 0]) * drho1;
            }
            dphi0 = dphi0 * ir0; dphi1 = h1 ? dphi1 * ir1 : R(0.0);
         }
         fx -= dphi0 * dx0; fy -= dphi0 * dy0; fz -= dphi0 * dz0;
         fx -= dphi1 * dx1; fy -= dphi1 * dy1; fz -= dphi1 * dz1;
      };
      // A ROW is bytes: the offsets of the hits inside their run, run after run, each run's part starting on an even byte -- a trip (two bytes) lies in one run:
      // bytes [0, n0) run 0, [S1, S1 + n1) run 1, [S2, S2 + n2) run 2 with S1 = n0 rounded up to even, S2 = S1 + n1 rounded up
      auto tripOf = [&](const int u, const unsigned pair, const int S1, const int S2, const int end0, const int end1, const int end2) __attribute__((always_inline)) {
         const int rsP = u >= S2 ? rs2 : u >= S1 ? rs1 : rs0, endP = u >= S2 ? end2 : u >= S1 ? end1 : end0;
         const bool h1 = u + 1 < endP;
         evalTrip(rsP + (int)(pair & 0xffu), h1 ? rsP + (int)((pair >> 8) & 0xffu) : recI, h1);
      };
      // the walk without rows: test and evaluate as it goes (an atom whose row overflowed; pass 3 without a row to read)
      auto directWalk = [&]() __attribute__((always_inline)) {
         const int l01 = len0 + len1, total = l01 + len2;
#pragma unroll 1
         for (int t = 0; t < total; ++t) {
            const int r = t < len0 ? rs0 + t : t < l01 ? rs1 + (t - len0) : rs2 + (t - l01);
            const real_t ax = xi - sp[REC * r], ay = yi - sp[REC * r + 1], az = zi - sp[REC * r + 2];
            if (ax * ax + ay * ay + az * az <= a.rc2 && r != recI) evalTrip(r, recI, false);
         }
      };
      // pass 3: the row of pass 1, when the atom has one that was numbered against this image.  Without rows in the LDS (the launcher found rows to read) the lanes
      // choose for themselves: an atom without a valid row -- its brick staged under another selection, its row overflowed, a second-round atom -- walks its stencil
      // again while the others evaluate from their registers; with rows in the LDS a wave in which any atom has none takes the TEST phase as pass 1 does.
      const bool rowValid = STEP == 3 && firstRound && preN != 0xffffffffu && preSel == selMask;
      const bool handed = STEP == 3 && (ldsRows ? __builtin_amdgcn_ballot_w64(rowValid) == ~0ull : true);
      if (handed) {
         if (rowValid) {
            const int n0 = (int)(preN & 0xffu), n1 = (int)((preN >> 8) & 0xffu), n2 = (int)((preN >> 16) & 0xffu);
            const int S1 = (n0 + 1) & ~1, S2 = S1 + ((n1 + 1) & ~1), total = (b.debug & 2) ? 0 : S2 + n2;
            const unsigned wr[4 * CH] = { pre[0].x, pre[0].y, pre[0].z, pre[0].w, pre[1].x, pre[1].y, pre[1].z, pre[1].w,
                                          pre[2].x, pre[2].y, pre[2].z, pre[2].w, pre[3].x, pre[3].y, pre[3].z, pre[3].w };
#pragma unroll
            for (int tr = 0; tr < 8 * CH; ++tr)            // (unrolled: the words sit in registers; a trip no lane has is a skipped branch)
               if (2 * tr < total) tripOf(2 * tr, (tr & 1) ? wr[tr >> 1] >> 16 : wr[tr >> 1], S1, S2, n0, S1 + n1, total);
         } else directWalk();
      } else {
         // TEST.  Four candidates per group, the records of the NEXT group requested before the current one is tested (two register sets, ping-pong): the walk
         // lives on LDS latency, and a brick workgroup's waves are few.  The groups every lane of the wave has whole are walked without masks, addresses as
         // immediate offsets from one running pointer; what is left of the longer runs takes the masked form (a candidate past the end names the atom itself: no
         // hit).  Every candidate's offset is written at the row's end and the end advanced past a hit (no exec-mask round trip per candidate); the capacity is
         // looked at once per group (the row's padding takes the group).  Only the middle plane holds the atom itself.
         unsigned char* w = myRow;
         unsigned char* const wEnd = myRow + b.rows;
         const int runMax = (b.debug & 16) ? 64 : 256;           // (COMD_EAM_ABLATE=16, tests: the walk without a row for every atom of a normal lattice)
         bool over = len0 > runMax || len1 > runMax || len2 > runMax;      // an offset would not fit a byte: no row
         int cnt[3] = { 0, 0, 0 };
         constexpr int G = 4;
#pragma unroll
         for (int p = 0; p < 3; ++p) {
            const int rs = p == 0 ? rs0 : p == 1 ? rs1 : rs2, len = (b.debug & 1) ? 0 : p == 0 ? len0 : p == 1 ? len1 : len2;
            unsigned char* const wStart = w;
            real_t ax[G], ay[G], az[G], cx[G], cy[G], cz[G];
            const int whole = waveMin(len / (2 * G));
            int u = 0;
            if (whole > 0) {
               const real_t* __restrict__ q = sp + REC * rs;
               auto askU = [&](const real_t* __restrict__ src, real_t (&X)[G], real_t (&Y)[G], real_t (&Z)[G]) {
#pragma unroll
                  for (int g = 0; g < G; ++g) { X[g] = src[REC * g]; Y[g] = src[REC * g + 1]; Z[g] = src[REC * g + 2]; }
               };
               auto testU = [&](const int u0, const real_t (&X)[G], const real_t (&Y)[G], const real_t (&Z)[G]) {
#pragma unroll
                  for (int g = 0; g < G; ++g) {
                     const real_t dx = xi - X[g], dy = yi - Y[g], dz = zi - Z[g];
                     bool hit = dx * dx + dy * dy + dz * dz <= a.rc2;
                     if (p == 1) hit = hit && rs + u0 + g != recI;
                     *w = (unsigned char)(u0 + g);
                     w += hit ? 1 : 0;
                  }
                  over = over || w > wEnd; w = w > wEnd ? wEnd : w;
               };
               askU(q, ax, ay, az);
               for (int it = 0; it < whole; ++it) {
                  askU(q + REC * G, cx, cy, cz); testU(u, ax, ay, az);
                  askU(q + 2 * REC * G, ax, ay, az); testU(u + G, cx, cy, cz);      // (the last request reads past the whole groups, inside the image or its neighbours in the LDS: dropped)
                  q += 2 * REC * G; u += 2 * G;
               }
            }
            for (; u < len; u += G) {
#pragma unroll
               for (int g = 0; g < G; ++g) {
                  const int r = u + g < len ? rs + u + g : recI;
                  ax[g] = sp[REC * r]; ay[g] = sp[REC * r + 1]; az[g] = sp[REC * r + 2];
               }
#pragma unroll
               for (int g = 0; g < G; ++g) {
                  const int r = u + g < len ? rs + u + g : recI;
                  const real_t dx = xi - ax[g], dy = yi - ay[g], dz = zi - az[g];
                  const bool hit = dx * dx + dy * dy + dz * dz <= a.rc2 && r != recI;
                  *w = (unsigned char)(u + g);
                  w += hit ? 1 : 0;
               }
               over = over || w > wEnd; w = w > wEnd ? wEnd : w;
            }
            cnt[p] = (int)(w - wStart);
            if (p < 2) w += cnt[p] & 1;                       // the next run's part starts on an even byte
         }
         over = over || w > wEnd;
         const int n0 = cnt[0], n1 = cnt[1], n2 = cnt[2];
         const int S1 = (n0 + 1) & ~1, S2 = S1 + ((n1 + 1) & ~1), total = S2 + n2;
         if (STEP == 1 && b.rowsG && firstRound && have) {      // leave the row for pass 3 (a row that outgrew its capacity: marked, pass 3 walks again)
            const int full = sFull[cl] + ia;
            if (full < rowCap) {
               rowCountG[full] = over ? 0xffffffffu : (unsigned)n0 | ((unsigned)n1 << 8) | ((unsigned)n2 << 16);
               uint4* __restrict__ dst = reinterpret_cast<uint4*>(b.rowsG) + (size_t)bid * CH * rowCap + full;
               const unsigned* __restrict__ rowR = reinterpret_cast<const unsigned*>(myRow);
               for (int c = 0; c < (over ? 0 : (total + 15) >> 4); ++c) dst[c * rowCap] = make_uint4(rowR[4 * c], rowR[4 * c + 1], rowR[4 * c + 2], rowR[4 * c + 3]);
            }
         }
         // EVALUATE
         if (over) directWalk();      // more neighbours than a row holds (a density far above the lattice's): thread_atom is the method without a limit
         else
            for (int u = 0; u < ((b.debug & 2) ? 0 : total); u += 2)
               tripOf(u, *reinterpret_cast<const unsigned short*>(myRow + u), S1, S2, n0, S1 + n1, total);
      }
      if (!have) continue;
      const size_t iOff = (size_t)iBox * a.cap + ia;
      if (STEP == 1) {
         a.fx[iOff] = fx; a.fy[iOff] = fy; a.fz[iOff] = fz; a.rhobar[iOff] = rb;
         real_t ei = R(0.5) * e;
         if (b.fuseEmbed) { real_t F, dF; interpolate(makeTable(a.f, a.f.values), rb, F, dF); a.dfEmbed[iOff] = dF; ei += F; }
         a.e[iOff] = ei;
      } else {
         a.fx[iOff] += fx; a.fy[iOff] += fy; a.fz[iOff] += fz;
      }
   }
}
