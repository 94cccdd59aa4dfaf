// nl_kernels.h -- Verlet neighbour lists on the device (methods thread_atom_nl / warp_atom_nl).
//
// What the reference does with gpu_neighborList.c + gpu_kernels.cu:1087-1110 (updateNeighborListRequiredKernel),
// :1486-2029 (buildNeighborListGpu and the *_nl force kernels) and its gid hash table (hashTable.c): lists hold every
// atom within cutoff + skin, link cells are sized cutoff + skin, and the lists stay valid until some atom has moved
// more than skin/2 since the build.  Between builds no atom changes its slot (no re-binning, CoMD.c:257-268 /
// timestep.c:278-352), so here a list entry is simply the neighbour's global slot and the halo copies are refreshed
// in place by a positional exchange (the slot-ordered path of the dF/drho exchange) -- no hash table.
//
// Layout: list[(cell * maxNbr + k) * cap + i] = slot of the k-th neighbour of atom i of `cell`: a wave reading entry k
// of its 64 atoms reads 256 consecutive bytes.
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
void BuildNeighborList(const double* __restrict__ rx, const double* __restrict__ ry, const double* __restrict__ rz,
                       const int* __restrict__ nAtoms, const int* __restrict__ nbr, const int* __restrict__ cells, int nCells, int cap,
                       NlView nl, double rBuild2, double* __restrict__ lastX, double* __restrict__ lastY, double* __restrict__ lastZ,
                       int* __restrict__ status)
{
   int iBox, i;
   if (!nlSlot(cells, nAtoms, nCells, cap, iBox, i)) return;
   const size_t iSlot = (size_t)iBox * cap + i;
   const double xi = rx[iSlot], yi = ry[iSlot], zi = rz[iSlot];
   lastX[iSlot] = xi; lastY[iSlot] = yi; lastZ[iSlot] = zi;
   int* __restrict__ row = nl.list + (size_t)iBox * nl.maxNbr * cap + i;
   int n = 0;
   for (int k = 0; k < 27; ++k) {
      const int jBox = nbr[(size_t)iBox * 27 + k];
      const int nj = nAtoms[jBox];
      const size_t base = (size_t)jBox * cap;
      for (int j = 0; j < nj; ++j) {
         const double dx = xi - rx[base + j], dy = yi - ry[base + j], dz = zi - rz[base + j];
         const double r2 = dx*dx + dy*dy + dz*dz;
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
void NeighborListUpdateRequired(const double* __restrict__ rx, const double* __restrict__ ry, const double* __restrict__ rz,
                                const double* __restrict__ lastX, const double* __restrict__ lastY, const double* __restrict__ lastZ,
                                const int* __restrict__ nAtoms, int nLocalBoxes, int cap, double skinHalf2, int* __restrict__ flag)
{
   int iBox, i;
   if (!nlSlot(nullptr, nAtoms, nLocalBoxes, cap, iBox, i)) return;
   const size_t s = (size_t)iBox * cap + i;
   const double dx = rx[s] - lastX[s], dy = ry[s] - lastY[s], dz = rz[s] - lastZ[s];
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
   const double xi = a.rx[iSlot], yi = a.ry[iSlot], zi = a.rz[iSlot];
   const int n = nl.count[iSlot];
   const int* __restrict__ row = nl.list + (size_t)iBox * nl.maxNbr * a.cap + i;
   double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
   int k = 0;
   for (; k + 4 <= n; k += 4) {                 // four gathers in flight
      int j[4]; double dx[4], dy[4], dz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) j[u] = row[(size_t)(k + u) * a.cap];
#pragma unroll
      for (int u = 0; u < 4; ++u) { dx[u] = xi - a.rx[j[u]]; dy[u] = yi - a.ry[j[u]]; dz[u] = zi - a.rz[j[u]]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const double r2 = dx[u]*dx[u] + dy[u]*dy[u] + dz[u]*dz[u];
         if (r2 <= a.rc2) ljPair<ENERGY>(dx[u], dy[u], dz[u], r2, a, fx, fy, fz, e);
      }
   }
   for (; k < n; ++k) {
      const int j = row[(size_t)k * a.cap];
      const double dx = xi - a.rx[j], dy = yi - a.ry[j], dz = zi - a.rz[j];
      const double r2 = dx*dx + dy*dy + dz*dz;
      if (r2 <= a.rc2) ljPair<ENERGY>(dx, dy, dz, r2, a, fx, fy, fz, e);
   }
   const double fs = 24.0 * a.eps;
   a.fx[iSlot] = fx * fs; a.fy[iSlot] = fy * fs; a.fz[iSlot] = fz * fs;
   if (ENERGY) a.e[iSlot] = e * 2.0 * a.eps;
}

// ---- EAM passes 1 and 3 over the list; tables in LDS when they fit (funcfl), else through L2 (setfl) ----------------------------
template <int STEP, bool LDS_TABLES>
__global__ __launch_bounds__(256)
void EAM_Force_thread_atom_nl(EamArgs a, NlView nl)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   double* sRho = (double*)ldsRaw;
   const int nRhoPad = a.rho.n + 3, nPhiPad = (STEP == 1) ? a.phi.n + 3 : 0;
   double* sPhi = sRho + nRhoPad;
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
   const double xi = a.rx[iSlot], yi = a.ry[iSlot], zi = a.rz[iSlot];
   const int n = nl.count[iSlot];
   const int* __restrict__ row = nl.list + (size_t)iBox * nl.maxNbr * a.cap + i;
   double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0, rb = 0.0, dfi = 0.0;
   if (STEP == 3) { fx = a.fx[iSlot]; fy = a.fy[iSlot]; fz = a.fz[iSlot]; dfi = a.dfEmbed[iSlot]; }
   for (int k0 = 0; k0 < n; k0 += 4) {
      int j[4]; double dx[4], dy[4], dz[4], dfj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) j[u] = (k0 + u < n) ? row[(size_t)(k0 + u) * a.cap] : (int)iSlot;     // padding pairs have r2 = 0 and are rejected
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         dx[u] = xi - a.rx[j[u]]; dy[u] = yi - a.ry[j[u]]; dz[u] = zi - a.rz[j[u]];
         if (STEP == 3) dfj[u] = a.dfEmbed[j[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const double r2 = dx[u]*dx[u] + dy[u]*dy[u] + dz[u]*dz[u];
         if (r2 <= a.rc2 && r2 > 0.0) {
            const double ir = rsqrt64(r2), r = r2 * ir;
            double rho, drho, dphi;
            if (STEP == 1) {
               double phi;
               if (sameGrid) interpolatePair(sRho, rhoT, r, phi, dphi, rho, drho);
               else { interpolate(rhoT, r, rho, drho); interpolate(phiT, r, phi, dphi); }
               e += phi; rb += rho;
            } else {
               interpolate(rhoT, r, rho, drho);
               dphi = (dfi + dfj[u]) * drho;
            }
            dphi *= ir;
            fx -= dphi * dx[u]; fy -= dphi * dy[u]; fz -= dphi * dz[u];
         }
      }
   }
   a.fx[iSlot] = fx; a.fy[iSlot] = fy; a.fz[iSlot] = fz;
   if (STEP == 1) { a.e[iSlot] = 0.5 * e; a.rhobar[iSlot] = rb; }
}

// ---- positional refresh of the halo copies between list builds (slot order == the sender's slot order) ----------------------------
// blockDim.x >= cap; one workgroup per listed cell; buffer holds x, y, z triples in send-cell-list order
__global__
void LoadPositionBuffer(double* __restrict__ buf, const int* __restrict__ list, const int* __restrict__ offsets,
                        const double* __restrict__ rx, const double* __restrict__ ry, const double* __restrict__ rz,
                        const int* __restrict__ nAtoms, int cap, double sx, double sy, double sz)
{
   const int c = list[blockIdx.x];
   if ((int)threadIdx.x < nAtoms[c]) {
      const size_t s = (size_t)c * cap + threadIdx.x;
      double* o = buf + 3 * (size_t)(offsets[blockIdx.x] + threadIdx.x);
      o[0] = rx[s] + sx; o[1] = ry[s] + sy; o[2] = rz[s] + sz;
   }
}

__global__
void UnloadPositionBuffer(const double* __restrict__ buf, const int* __restrict__ list, const int* __restrict__ offsets,
                          double* __restrict__ rx, double* __restrict__ ry, double* __restrict__ rz,
                          const int* __restrict__ nAtoms, int cap)
{
   const int c = list[blockIdx.x];
   if ((int)threadIdx.x < nAtoms[c]) {
      const size_t s = (size_t)c * cap + threadIdx.x;
      const double* o = buf + 3 * (size_t)(offsets[blockIdx.x] + threadIdx.x);
      rx[s] = o[0]; ry[s] = o[1]; rz[s] = o[2];
   }
}
