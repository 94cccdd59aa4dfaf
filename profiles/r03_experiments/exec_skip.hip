// exec_skip.hip -- does gfx950 skip the 16-lane passes of a wave64 VALU instruction whose lanes are all masked off?
// hipcc --offload-arch=gfx950 -O2 exec_skip.hip -o exec_skip && ./exec_skip
// A chain-free stream of v_fma_f64 (and v_fma_f32) under `lane < K` for K = 16, 32, 48, 64, all SIMDs saturated (8 waves each): time per K.
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void K(T* out, int active, int iters)
{
   const int lane = threadIdx.x & 63;
   T a0 = (T)1.0 + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
   const T m = (T)0.999999, c = (T)1e-9;
   if (lane < active) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
         for (int u = 0; u < 8; ++u) {
            a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c; a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
         }
      }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename T> void run(const char* name)
{
   T* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(T));
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   for (int active : {64, 48, 32, 16, 8}) {
      hipLaunchKernelGGL(K<T>, dim3(256 * 8), dim3(256), 0, 0, d, active, 100);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(K<T>, dim3(256 * 8), dim3(256), 0, 0, d, active, 4000);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr = 256.0 * 8 * 4 * 4000.0 * 64;          // wave-level fma instructions
      printf("%s lanes < %2d: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, active, ms, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
   }
   hipFree(d);
}

int main() { run<double>("v_fma_f64"); run<float>("v_fma_f32"); return 0; }
