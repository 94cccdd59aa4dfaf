// device_common.h -- wave64 helpers shared by the gfx950 kernels.
//
// Stands where the reference's gpu_common.h does (interpolate :48-86, warp_reduce :252-278,
// PTX laneid/bfi :231-236/:316-321), re-expressed for CDNA4: 64-lane wavefronts, DPP/bpermute
// cross-lane moves, v_rcp_f64 + Newton instead of rsqrt.approx PTX.
#pragma once

#include <hip/hip_runtime.h>
#include "comd_hip.h"
#include "comd_geometry.h"

#define WAVE 64

__device__ __forceinline__ int laneId() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 1/x to fp64 round-off: v_rcp_f64 is good to ~2^-24 relative, two Newton steps square that twice.
__device__ __forceinline__ double rcp64(double x)
{
   double y = __builtin_amdgcn_rcp(x);
   double e = __builtin_fma(-x, y, 1.0);
   y = __builtin_fma(y, e, y);
   e = __builtin_fma(-x, y, 1.0);
   y = __builtin_fma(y, e, y);
   return y;
}

// 1/sqrt(x) to fp64 round-off from v_rsq_f64 (~2^-24) with two Newton steps.
__device__ __forceinline__ double rsqrt64(double x)
{
   double y = __builtin_amdgcn_rsq(x);
   double h = 0.5 * x;
   y = y * __builtin_fma(-h * y, y, 1.5);
   y = y * __builtin_fma(-h * y, y, 1.5);
   return y;
}

// 64-bit cross-lane read through ds_bpermute (two dword permutes); srcLane in [0,63].
__device__ __forceinline__ double bpermute64(double v, int srcLane)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_ds_bpermute(srcLane << 2, lo);
   hi = __builtin_amdgcn_ds_bpermute(srcLane << 2, hi);
   return __hiloint2double(hi, lo);
}

// butterfly sum over the 64 lanes; every lane ends with the total, in a fixed order
__device__ __forceinline__ double waveSum(double v)
{
#pragma unroll
   for (int m = 32; m >= 1; m >>= 1) v += bpermute64(v, laneId() ^ m);
   return v;
}

// ---- DPP / permlane reductions: pure VALU, no trip through the LDS crossbar --------------------------------------
// dpp_ctrl encodings (gfx9): quad_perm 0x00-0xFF, row_ror:n 0x120+n, row_bcast:15 0x142, row_bcast:31 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dppMove64(double v)       // lanes not written by the DPP pattern receive 0.0
{
   int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
   int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
   return __hiloint2double(hi, lo);
}

// sum within each row of 16 lanes; every lane of the row ends with the row total
__device__ __forceinline__ double rowSum(double v)
{
   v += dppMove64<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
   v += dppMove64<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
   v += dppMove64<0x124, 0xF>(v);      // row_ror:4
   v += dppMove64<0x128, 0xF>(v);      // row_ror:8
   return v;
}

// Two atoms at once: pa / pb are per-lane partial sums of atoms A and B.  v_permlane32_swap exchanges the upper half of
// pa with the lower half of pb, so one add leaves A's sums in lanes 0-31 and B's in lanes 32-63; rows are then summed
// with DPP and row_bcast:15 folds row 0 into row 1 and row 2 into row 3.  Lane 31 holds A's total, lane 63 B's.
__device__ __forceinline__ double pairSum(double pa, double pb)
{
   auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(pa), (unsigned)__double2loint(pb), false, false);
   auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(pa), (unsigned)__double2hiint(pb), false, false);
   double v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
   v = rowSum(v);
   v += dppMove64<0x142, 0xA>(v);      // row_bcast:15 into rows 1 and 3
   return v;
}

__device__ __forceinline__ int waveSumInt(int v)
{
#pragma unroll
   for (int m = 32; m >= 1; m >>= 1) v += __builtin_amdgcn_ds_bpermute((laneId() ^ m) << 2, v);
   return v;
}

// Blocks are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8).  Give each XCD one contiguous run
// of the logical grid so that blocks sharing neighbour cells also share an L2 (bijective for any grid size).
__device__ __forceinline__ int xcdRemap(int bid, int nBlocks)
{
   const int nx = 8;
   int q = nBlocks / nx, r = nBlocks % nx, xcd = bid % nx, k = bid / nx;
   int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
   return start + k;
}

// Quadratic table interpolation, value + derivative (reference gpu_common.h:48-86; host eam.c:557-579).
// `v` points at the padded table: v[0] is the leading pad, v[i+1] is sample i.
struct TableView { const double* v; double x0, xn, invDx, invDxHalf, invDxXx0; };

__device__ __forceinline__ TableView makeTable(const InterpolationObjectGpu& t, const double* values)
{
   TableView tv; tv.v = values; tv.x0 = t.x0; tv.xn = t.xn; tv.invDx = t.invDx; tv.invDxHalf = t.invDxHalf; tv.invDxXx0 = t.invDxXx0;
   return tv;
}

__device__ __forceinline__ void interpolate(const TableView& t, double r, double& f, double& df)
{
   r = fmax(r, t.x0);
   r = fmin(r, t.xn);
   r = r * t.invDx - t.invDxXx0;
   double ri = floor(r);
   int ii = (int)ri;
   r -= ri;
   double v0 = t.v[ii], v1 = t.v[ii + 1], v2 = t.v[ii + 2], v3 = t.v[ii + 3];
   double g1 = v2 - v0, g2 = v3 - v1;
   f  = v1 + 0.5 * r * (g1 + r * (v2 + v0 - 2.0 * v1));
   df = (g1 + r * (g2 - g1)) * t.invDxHalf;
}

// phi(r) and rho(r) tabulated on the SAME grid (funcfl files are): values interleaved {phi_i, rho_i} so one index
// computation and four 16-byte LDS reads serve both interpolations.  v[2*i], v[2*i+1]; i = 0 is the leading pad.
__device__ __forceinline__ void interpolatePair(const double* __restrict__ v, const TableView& t, double r,
                                                double& phi, double& dphi, double& rho, double& drho)
{
   r = fmax(r, t.x0);
   r = fmin(r, t.xn);
   r = r * t.invDx - t.invDxXx0;
   const double ri = floor(r);
   const int ii = (int)ri;
   r -= ri;
   const double2* __restrict__ q = reinterpret_cast<const double2*>(v) + ii;
   const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
   {
      const double g1 = a2.x - a0.x, g2 = a3.x - a1.x;
      phi  = a1.x + 0.5 * r * (g1 + r * (a2.x + a0.x - 2.0 * a1.x));
      dphi = (g1 + r * (g2 - g1)) * t.invDxHalf;
   }
   {
      const double g1 = a2.y - a0.y, g2 = a3.y - a1.y;
      rho  = a1.y + 0.5 * r * (g1 + r * (a2.y + a0.y - 2.0 * a1.y));
      drho = (g1 + r * (g2 - g1)) * t.invDxHalf;
   }
}

// Cubic spline in r^2 (reference gpu_common.h:95-129): value f(r) and (1/r) df/dr -- no square root on the path except the
// single-precision one that picks the table interval.
__device__ __forceinline__ void interpolateSpline(const InterpolationSplineObjectGpu& t, double r2, double& f, double& df)
{
   float r = __builtin_sqrtf((float)r2);
   r = fmaxf(r, t.x0);
   r = fminf(r, t.xn);
   r = r * t.invDx - t.invDxXx0;
   int ii = (int)floorf(r);
   ii = ii < t.n - 1 ? ii : t.n - 1;                          // r == xn lands on the last interval
   const double* __restrict__ c = t.coefficients + 4 * ii;
   const double a = c[0], b = c[1], cc = c[2], d = c[3];
   const double tmp = a * r2 + b;
   f = (tmp * r2 + cc) * r2 + d;
   df = 2.0 * ((3.0 * tmp - b) * r2 + cc);
}

__device__ __forceinline__ CellGeom makeGeom(const LinkCellGpu& b)
{
   CellGeom c;
   for (int a = 0; a < 3; ++a) { c.g[a] = b.gridSize[a]; c.lmin[a] = b.localMin[a]; c.lmax[a] = b.localMax[a]; c.inv[a] = b.invBoxSize[a]; }
   c.nLocal = b.nLocalBoxes; c.nTotal = b.nTotalBoxes;
   c.lookup = b.boxIDLookUp; c.reverse = b.boxIDLookUpReverse;
   return c;
}
