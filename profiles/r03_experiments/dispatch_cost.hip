// dispatch_cost.hip -- what do 24389 workgroups of the EAM brick kernel's shape (256 threads, 40 KB of LDS, 128 VGPRs) cost before they do anything?
// hipcc --offload-arch=gfx950 -O2 dispatch_cost.hip -o dispatch_cost && ./dispatch_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>      // 0: nothing; 1: 8 KB table -> LDS (as the brick kernel's prologue does); 2: the same + a barrier and a dependent second load
__global__ __launch_bounds__(256, 4) void K(const double* __restrict__ table, double* out, int n)
{
   extern __shared__ double lds[];
   const int tid = threadIdx.x;
   if (MODE >= 1) {
      for (int t = tid; t < 1006; t += 256) lds[t] = table[t];
      __syncthreads();
      if (MODE == 2) {
         const int idx = (int)lds[tid] & 1023;
         lds[1024 + tid] = table[idx];
         __syncthreads();
      }
      if (lds[(tid * 7) & 1023] == 12345.0) out[blockIdx.x] = lds[1024 + tid];
   } else if (n < 0) out[blockIdx.x] = 1.0;
}

int main()
{
   double *t, *o; hipMalloc(&t, 2048 * 8); hipMalloc(&o, 30000 * 8); hipMemset(t, 0, 2048 * 8);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const int grid = 24389, lds = 40 * 1024;
   for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
         hipEventRecord(e0, 0);
         if (mode == 0) hipLaunchKernelGGL(K<0>, dim3(grid), dim3(256), lds, 0, t, o, 1);
         if (mode == 1) hipLaunchKernelGGL(K<1>, dim3(grid), dim3(256), lds, 0, t, o, 1);
         if (mode == 2) hipLaunchKernelGGL(K<2>, dim3(grid), dim3(256), lds, 0, t, o, 1);
         hipEventRecord(e1, 0); hipEventSynchronize(e1);
         float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("mode %d: %.1f us for %d workgroups (%.2f us per workgroup slot: 1024 slots)\n", mode, best * 1e3, grid, best * 1e3 / (grid / 1024.0));
   }
   return 0;
}
