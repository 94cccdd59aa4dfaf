// device_common.h -- wave64 helpers shared by the gfx950 kernels, in the precision of the build (real_t).
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

// Literals and vector types of the build's precision (comd_hip.h real_t): every floating constant in the kernels goes through R(),
// so the single-precision build does pure fp32 arithmetic (a bare 0.5 would drag the expression into fp64).
#define R(x) ((real_t)(x))
#ifdef COMD_SINGLE
typedef float2 real2;
#else
typedef double2 real2;
#endif
#define FAR_AWAY R(1.0e15)                  // padding coordinate: never inside a cutoff, its square still finite in fp32

// 1/x to round-off.  fp64: v_rcp_f64 is good to ~2^-23 relative, two Newton steps square that twice; fp32: v_rcp_f32 (1 ulp) + one step.
__device__ __forceinline__ double rcpR(double x)
{
   // v_rcp_f64 is good to ~2^-23; with e = 1 - x y0 exactly (fma), 1/x = y0 (1 + e + e^2 + ...): two terms leave e^3 ~ 2^-69
   const double y = __builtin_amdgcn_rcp(x);
   const double e = __builtin_fma(-x, y, 1.0);
   return __builtin_fma(y, __builtin_fma(e, e, e), y);
}
__device__ __forceinline__ float rcpR(float x)
{
   float y = __builtin_amdgcn_rcpf(x);
   const float e = __builtin_fmaf(-x, y, 1.0f);
   return __builtin_fmaf(y, e, y);
}

// 1/sqrt(x) to round-off from v_rsq_f64 (~2^-23) with two Newton steps / v_rsq_f32 with one.
__device__ __forceinline__ double rsqrtR(double x)
{
   // v_rsq_f64 is good to ~2^-23; with e = 1 - x y0^2 (the product x y0 rounded once, the rest in an fma), 1/sqrt(x) = y0 (1 - e)^(-1/2)
   // = y0 (1 + e/2 + 3 e^2/8 + ...): two terms leave (5/16) e^3 ~ 2^-70.  Five operations after the seed (two Newton steps: seven).
   const double y = __builtin_amdgcn_rsq(x);
   const double e = __builtin_fma(-(x * y), y, 1.0);
   return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
}
__device__ __forceinline__ float rsqrtR(float x)
{
   float y = __builtin_amdgcn_rsqf(x);
   const float h = 0.5f * x;
   return y * __builtin_fmaf(-h * y, y, 1.5f);
}

__device__ __forceinline__ double absR(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float  absR(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double fmaR(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float  fmaR(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double floorR(double x) { return __builtin_floor(x); }
__device__ __forceinline__ float  floorR(float x) { return __builtin_floorf(x); }
__device__ __forceinline__ double fractR(double x) { return __builtin_amdgcn_fract(x); }
__device__ __forceinline__ float  fractR(float x) { return __builtin_amdgcn_fractf(x); }
__device__ __forceinline__ double minR(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float  minR(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double maxR(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float  maxR(float a, float b) { return __builtin_fmaxf(a, b); }

// cross-lane read through ds_bpermute (one dword permute per 32 bits); srcLane in [0,63].
__device__ __forceinline__ double bpermuteR(double v, int srcLane)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_ds_bpermute(srcLane << 2, lo);
   hi = __builtin_amdgcn_ds_bpermute(srcLane << 2, hi);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bpermuteR(float v, int srcLane)
{
   return __int_as_float(__builtin_amdgcn_ds_bpermute(srcLane << 2, __float_as_int(v)));
}

// the same with the byte address (4 * source lane) already formed: one address serves several values
__device__ __forceinline__ double bpermuteAddrR(double v, int addr)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_ds_bpermute(addr, lo);
   hi = __builtin_amdgcn_ds_bpermute(addr, hi);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bpermuteAddrR(float v, int addr)
{
   return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}

// butterfly sum over the 64 lanes; every lane ends with the total, in a fixed order
__device__ __forceinline__ real_t waveSum(real_t v)
{
#pragma unroll
   for (int m = 32; m >= 1; m >>= 1) v += bpermuteR(v, laneId() ^ m);
   return v;
}

// ---- DPP / permlane reductions: pure VALU, no trip through the LDS crossbar --------------------------------------
// dpp_ctrl encodings (gfx9): quad_perm 0x00-0xFF, row_ror:n 0x120+n, row_bcast:15 0x142, row_bcast:31 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dppMoveR(double v)        // lanes not written by the DPP pattern receive 0.0
{
   int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
   int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
   return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dppMoveR(float v)
{
   return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
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
struct TableView { const real_t* v; real_t x0, xn, invDx, invDxHalf, invDxXx0; };

__device__ __forceinline__ TableView makeTable(const InterpolationObjectGpu& t, const real_t* values)
{
   TableView tv; tv.v = values; tv.x0 = t.x0; tv.xn = t.xn; tv.invDx = t.invDx; tv.invDxHalf = t.invDxHalf; tv.invDxXx0 = t.invDxXx0;
   return tv;
}

// CLAMP = false: the caller guarantees x0 <= r <= xn (the brick kernel's pairs: 0 < r <= cutoff, with tables that start at or below 0 and reach the cutoff --
// the two clamps and the NaN-quieting moves hipcc puts in front of them are then dead weight: 4 of ~45 instructions per pair)
template <bool CLAMP = true>
__device__ __forceinline__ void interpolate(const TableView& t, real_t r, real_t& f, real_t& df)
{
   if (CLAMP) { r = maxR(r, t.x0); r = minR(r, t.xn); }
   r = r * t.invDx - t.invDxXx0;
   int ii;
   if (CLAMP) { const real_t ri = floorR(r); ii = (int)ri; r -= ri; }
   else { ii = (int)r; r = fractR(r); }      // r >= 0 here: the conversion truncates to the floor, v_fract is r - floor(r) -- the same bits, one instruction fewer
   real_t v0 = t.v[ii], v1 = t.v[ii + 1], v2 = t.v[ii + 2], v3 = t.v[ii + 3];
   real_t g1 = v2 - v0, g2 = v3 - v1;
   f  = v1 + R(0.5) * r * (g1 + r * (v2 + v0 - R(2.0) * v1));
   df = (g1 + r * (g2 - g1)) * t.invDxHalf;
}

// phi(r) and rho(r) tabulated on the SAME grid (funcfl files are): values interleaved {phi_i, rho_i} so one index
// computation and four 16-byte LDS reads serve both interpolations.  v[2*i], v[2*i+1]; i = 0 is the leading pad.
template <bool CLAMP = true>
__device__ __forceinline__ void interpolatePair(const real_t* __restrict__ v, const TableView& t, real_t r,
                                                real_t& phi, real_t& dphi, real_t& rho, real_t& drho)
{
   if (CLAMP) { r = maxR(r, t.x0); r = minR(r, t.xn); }
   r = r * t.invDx - t.invDxXx0;
   int ii;
   if (CLAMP) { const real_t ri = floorR(r); ii = (int)ri; r -= ri; }
   else { ii = (int)r; r = fractR(r); }
   const real2* __restrict__ q = reinterpret_cast<const real2*>(v) + ii;
   const real2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
   {
      const real_t g1 = a2.x - a0.x, g2 = a3.x - a1.x;
      phi  = a1.x + R(0.5) * r * (g1 + r * (a2.x + a0.x - R(2.0) * a1.x));
      dphi = (g1 + r * (g2 - g1)) * t.invDxHalf;
   }
   {
      const real_t g1 = a2.y - a0.y, g2 = a3.y - a1.y;
      rho  = a1.y + R(0.5) * r * (g1 + r * (a2.y + a0.y - R(2.0) * a1.y));
      drho = (g1 + r * (g2 - g1)) * t.invDxHalf;
   }
}

// Cubic spline in r^2 (reference gpu_common.h:95-129): value f(r) and (1/r) df/dr -- no square root on the path except the
// single-precision one that picks the table interval.
__device__ __forceinline__ void interpolateSpline(const InterpolationSplineObjectGpu& t, real_t r2, real_t& f, real_t& df)
{
   float r = __builtin_sqrtf((float)r2);
   r = fmaxf(r, t.x0);
   r = fminf(r, t.xn);
   r = r * t.invDx - t.invDxXx0;
   int ii = (int)floorf(r);
   ii = ii < t.n - 1 ? ii : t.n - 1;                          // r == xn lands on the last interval
   const real_t* __restrict__ c = t.coefficients + 4 * ii;
   const real_t a = c[0], b = c[1], cc = c[2], d = c[3];
   const real_t tmp = a * r2 + b;
   f = (tmp * r2 + cc) * r2 + d;
   df = R(2.0) * ((R(3.0) * tmp - b) * r2 + cc);
}

__device__ __forceinline__ CellGeom makeGeom(const LinkCellGpu& b)
{
   CellGeom c;
   for (int a = 0; a < 3; ++a) { c.g[a] = b.gridSize[a]; c.lmin[a] = b.localMin[a]; c.lmax[a] = b.localMax[a]; c.inv[a] = b.invBoxSize[a]; }
   c.nLocal = b.nLocalBoxes; c.nTotal = b.nTotalBoxes;
   c.lookup = b.boxIDLookUp; c.reverse = b.boxIDLookUpReverse;
   return c;
}
