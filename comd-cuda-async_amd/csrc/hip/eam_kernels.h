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
//  thread_atom : one thread per atom, a power-of-two share of a wave per cell, tables in the LDS when they fit.
//  cta_cell    : EAM_Force_cta_cell in nl_kernels.h (it shares the staging, the table layout and the row evaluation of the list kernel).
#pragma once
#include "device_common.h"

struct EamArgs {
   const real_t* __restrict__ rx; const real_t* __restrict__ ry; const real_t* __restrict__ rz;
   real_t* __restrict__ fx; real_t* __restrict__ fy; real_t* __restrict__ fz; real_t* __restrict__ e;
   real_t* __restrict__ rhobar; real_t* __restrict__ dfEmbed;
   const int* __restrict__ nAtoms;
   const int* __restrict__ nbr;
   const int* __restrict__ cells;
   int nCells, cap;
   real_t rc2;
   InterpolationObjectGpu phi, rho, f;
   InterpolationSplineObjectGpu phiS, rhoS;
   const int* __restrict__ sel; int tag;      // EAM_Force_embed: when sel is given, only the cells with sel[cell] == tag (the brick groups of cta_cell, comd_device.hip)
};

// ---------------------------------------------------------------------------------------------------
// thread per atom (the reference's EAM_Force_thread_atom shape): every thread walks the 27-cell stencil of its atom.
// `lanesPerCell` (a power of two <= 256, at least the fullest cell the host has seen) threads serve one cell, so a wave holds the atoms
// of 64 / lanesPerCell neighbouring cells instead of one cell's 10 atoms in 32 slots; a cell that outgrew the estimate is still
// complete (its threads take a second atom).  Tables: in the LDS when they fit (funcfl: one interleaved {phi, rho} table in pass 1),
// else through L2; the workgroups are persistent so that the table load is paid once per workgroup, not once per 16 cells.
template <int STEP, bool SPLINE, bool LDS_TABLES>
__global__ __launch_bounds__(256)
void EAM_Force_thread_atom(EamArgs a, int lanesPerCell)
{
   extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
   real_t* sRho = (real_t*)ldsRaw;
   const int nRhoPad = a.rho.n + 3;
   real_t* sPhi = sRho + nRhoPad;
   const bool sameGrid = (STEP == 1) && LDS_TABLES && a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   if (LDS_TABLES) {
      if (sameGrid) {
         for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) { sRho[2 * t] = a.phi.values[t]; sRho[2 * t + 1] = a.rho.values[t]; }
      } else {
         for (int t = threadIdx.x; t < nRhoPad; t += blockDim.x) sRho[t] = a.rho.values[t];
         if (STEP == 1) for (int t = threadIdx.x; t < a.phi.n + 3; t += blockDim.x) sPhi[t] = a.phi.values[t];
      }
      __syncthreads();
   }
   const TableView rhoT = makeTable(a.rho, LDS_TABLES ? sRho : a.rho.values), phiT = makeTable(a.phi, LDS_TABLES ? sPhi : a.phi.values);

   const int cellsPerBlock = 256 / lanesPerCell;
   const int cellInBlock = threadIdx.x / lanesPerCell, l = threadIdx.x - cellInBlock * lanesPerCell;
   const int nGroups = (a.nCells + cellsPerBlock - 1) / cellsPerBlock;
   // groups are dealt in contiguous runs (XCD-contiguous): neighbouring cells, whose stencils overlap, share an L2
   const int first = xcdRemap(blockIdx.x, gridDim.x);
   const int per = (nGroups + gridDim.x - 1) / gridDim.x;
   for (int grp = first * per; grp < (first + 1) * per && grp < nGroups; ++grp) {
      const int ci = grp * cellsPerBlock + cellInBlock;
      if (ci >= a.nCells) continue;
      const int iBox = a.cells ? a.cells[ci] : ci;
      const int ni = a.nAtoms[iBox];
      const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
      for (int ia = l; ia < ni; ia += lanesPerCell) {
         const size_t iOff = (size_t)iBox * a.cap + ia;
         const real_t xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
         real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0), rb = R(0.0), dfi = R(0.0);
         if (STEP == 3) { fx = a.fx[iOff]; fy = a.fy[iOff]; fz = a.fz[iOff]; dfi = a.dfEmbed[iOff]; }
         for (int k = 0; k < 27; ++k) {
            const int jBox = nb[k];
            const int nj = a.nAtoms[jBox];
            const size_t base = (size_t)jBox * a.cap;
            for (int j = 0; j < nj; ++j) {
               const real_t dx = xi - a.rx[base + j], dy = yi - a.ry[base + j], dz = zi - a.rz[base + j];
               const real_t r2 = dx*dx + dy*dy + dz*dz;
               if (r2 <= a.rc2 && r2 > R(0.0)) {
                  real_t rho, drho, dphi;
                  if (SPLINE) {                                  // drho, dphi are (1/r) d/dr already
                     interpolateSpline(a.rhoS, r2, rho, drho);
                     if (STEP == 1) { real_t phi; interpolateSpline(a.phiS, r2, phi, dphi); e += phi; rb += rho; }
                     else           { dphi = (dfi + a.dfEmbed[base + j]) * drho; }
                  } else {
                     const real_t ir = rsqrtR(r2), r = r2 * ir;
                     if (STEP == 1) {
                        real_t phi;
                        if (sameGrid) interpolatePair(sRho, rhoT, r, phi, dphi, rho, drho);
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
         if (STEP == 1) { a.e[iOff] = R(0.5) * e; a.rhobar[iOff] = rb; }
      }
   }
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
   if (a.sel && a.sel[iBox] != a.tag) return;
   const size_t iOff = (size_t)iBox * a.cap + ia;
   const TableView fT = makeTable(a.f, a.f.values);
   real_t F, dF;
   interpolate(fT, a.rhobar[iOff], F, dF);
   a.dfEmbed[iOff] = dF;
   a.e[iOff] += F;
}

// ---------------------------------------------------------------------------------------------------
// A cell whose 27-cell stencil does not fit a wave's LDS slice (small boxes have larger cells) is handled by the same wave of
// EAM_Force_cta_cell (nl_kernels.h) in the thread-per-atom form: lane = i atom, neighbours streamed from global memory, same tables.
template <int STEP, bool SPLINE>
__device__ __forceinline__ void eamCellDirect(const EamArgs& a, int iBox, int lane, const TableView& rhoT, const TableView& phiT, bool sameGrid, int fuseEmbed)
{
   const int ni = a.nAtoms[iBox];
   const int* __restrict__ nb = a.nbr + (size_t)iBox * 27;
   for (int i = lane; i < ni; i += 64) {
      const size_t iOff = (size_t)iBox * a.cap + i;
      const real_t xi = a.rx[iOff], yi = a.ry[iOff], zi = a.rz[iOff];
      real_t fx = R(0.0), fy = R(0.0), fz = R(0.0), e = R(0.0), rb = R(0.0), dfi = R(0.0);
      if (STEP == 3) { fx = a.fx[iOff]; fy = a.fy[iOff]; fz = a.fz[iOff]; dfi = a.dfEmbed[iOff]; }
      for (int k = 0; k < 27; ++k) {
         const int jBox = nb[k];
         const int nj = a.nAtoms[jBox];
         const size_t base = (size_t)jBox * a.cap;
         for (int j = 0; j < nj; ++j) {
            const real_t dx = xi - a.rx[base + j], dy = yi - a.ry[base + j], dz = zi - a.rz[base + j];
            const real_t r2 = dx*dx + dy*dy + dz*dz;
            if (r2 <= a.rc2 && r2 > R(0.0)) {
               real_t rho, drho, dphi;
               if (SPLINE) {
                  interpolateSpline(a.rhoS, r2, rho, drho);
                  if (STEP == 1) { real_t phi; interpolateSpline(a.phiS, r2, phi, dphi); e += phi; rb += rho; }
                  else           { dphi = (dfi + a.dfEmbed[base + j]) * drho; }
               } else {
                  const real_t ir = rsqrtR(r2), r = r2 * ir;
                  if (STEP == 1) {
                     real_t phi;
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
      if (STEP == 1) {
         real_t ei = R(0.5) * e;
         if (fuseEmbed) { real_t F, dF; interpolate(makeTable(a.f, a.f.values), rb, F, dF); a.dfEmbed[iOff] = dF; ei += F; }
         a.e[iOff] = ei; a.rhobar[iOff] = rb;
      }
   }
}

static inline size_t eamCtaTableBytes(int step, int nRho, int nPhi) { return (size_t)(nRho + 3 + (step == 1 ? nPhi + 3 : 0)) * sizeof(real_t); }
