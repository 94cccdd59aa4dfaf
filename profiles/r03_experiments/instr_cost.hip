// instr_cost.hip -- issue cost of the instructions an accepted LJ pair is made of, on saturated SIMDs (8 waves each, 8 independent streams per wave):
// v_fma_f64, v_rcp_f64, v_rsq_f64, v_cvt_f32_f64, v_cvt_f64_f32, v_rcp_f32, and the two ways to a 2^-23 reciprocal seed.
// hipcc --offload-arch=gfx950 -O2 instr_cost.hip -o instr_cost && ./instr_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void K(double* out, int iters)
{
   double a[8];
#pragma unroll
   for (int u = 0; u < 8; ++u) a[u] = 1.5 + 0.01 * (threadIdx.x & 63) + u;
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
         if (OP == 0) a[u] = __builtin_fma(a[u], 0.999999, 1e-9);
         if (OP == 1) a[u] = __builtin_amdgcn_rcp(a[u]);
         if (OP == 2) a[u] = __builtin_amdgcn_rsq(a[u]);
         if (OP == 3) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a[u])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[u]) : "v"(f)); }
         if (OP == 4) { float f; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(a[u])); asm volatile("v_rcp_f32 %0, %1" : "=v"(f) : "v"(f)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[u]) : "v"(f)); }
         if (OP == 5) { float f = (float)a[u]; asm volatile("v_rcp_f32 %0, %1" : "=v"(f) : "v"(f)); a[u] += f; }
      }
   }
   double s = 0; for (int u = 0; u < 8; ++u) s += a[u];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
   double* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const char* name[6] = {"v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_cvt_f32_f64 + v_cvt_f64_f32", "cvt + v_rcp_f32 + cvt", "cvt + v_rcp_f32 + v_cvt + v_add_f64 (compiler's)"};
   float base = 0;
   for (int op = 0; op < 6; ++op) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
         hipEventRecord(e0, 0);
         const int it = 2000;
         if (op == 0) hipLaunchKernelGGL(K<0>, dim3(256 * 8), dim3(256), 0, 0, d, it);
         if (op == 1) hipLaunchKernelGGL(K<1>, dim3(256 * 8), dim3(256), 0, 0, d, it);
         if (op == 2) hipLaunchKernelGGL(K<2>, dim3(256 * 8), dim3(256), 0, 0, d, it);
         if (op == 3) hipLaunchKernelGGL(K<3>, dim3(256 * 8), dim3(256), 0, 0, d, it);
         if (op == 4) hipLaunchKernelGGL(K<4>, dim3(256 * 8), dim3(256), 0, 0, d, it);
         if (op == 5) hipLaunchKernelGGL(K<5>, dim3(256 * 8), dim3(256), 0, 0, d, it);
         hipEventRecord(e1, 0); hipEventSynchronize(e1);
         hipEventElapsedTime(&ms, e0, e1);
      }
      if (op == 0) base = ms;
      printf("%-50s %.3f ms = %.2f x v_fma_f64\n", name[op], ms, ms / base);
   }
   return 0;
}
