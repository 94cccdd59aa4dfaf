// rcp_accuracy.hip -- how good are v_rcp_f64 / v_rsq_f64 on gfx950, and what do one or two correction terms leave?
// hipcc --offload-arch=gfx950 -O2 rcp_accuracy.hip -o rcp_accuracy && ./rcp_accuracy      (max relative error in units of 2^-53, against long double on the host)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

__global__ void K(const double* x, double* o, int n)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const double v = x[i];
   const double y = __builtin_amdgcn_rcp(v);
   const double e = __builtin_fma(-v, y, 1.0);
   o[i] = y;                                                   // raw
   o[n + i] = __builtin_fma(y, e, y);                          // one quadratic step
   o[2 * n + i] = __builtin_fma(y, __builtin_fma(e, e, e), y); // the cubic step the kernels use
   const double s = __builtin_amdgcn_rsq(v);
   const double f = __builtin_fma(-(v * s), s, 1.0);
   o[3 * n + i] = s;
   o[4 * n + i] = __builtin_fma(s * f, 0.5, s);                // one term
   o[5 * n + i] = __builtin_fma(s * f, __builtin_fma(f, 0.375, 0.5), s);   // two terms (the kernels')
}

int main()
{
   const int n = 1 << 22;
   std::vector<double> x(n), o(6 * (size_t)n);
   std::mt19937_64 g(7);
   for (int i = 0; i < n; ++i) {
      const double u = (double)(g() >> 11) / 9007199254740992.0;
      x[i] = i < n / 2 ? 1.0 + 200.0 * u : std::ldexp(1.0 + u, (int)(g() % 120) - 60);      // the r^2 range of the kernels; then all binades around 1
   }
   double *dx, *dout;
   hipMalloc(&dx, n * sizeof(double)); hipMalloc(&dout, 6 * (size_t)n * sizeof(double));
   hipMemcpy(dx, x.data(), n * sizeof(double), hipMemcpyHostToDevice);
   hipLaunchKernelGGL(K, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
   hipMemcpy(o.data(), dout, 6 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
   const char* name[6] = {"v_rcp_f64 raw", "rcp + fma(y,e,y)", "rcp + fma(y,fma(e,e,e),y)", "v_rsq_f64 raw", "rsq + one term", "rsq + two terms"};
   for (int k = 0; k < 6; ++k) {
      long double worst = 0;
      for (int i = 0; i < n; ++i) {
         const long double ref = k < 3 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
         const long double err = fabsl(((long double)o[(size_t)k * n + i] - ref) / ref);
         if (err > worst) worst = err;
      }
      printf("%-28s max relative error %.3Le = 2^%.1Lf = %.1Lf ulp(2^-53)\n", name[k], worst, log2l(worst), worst * 9007199254740992.0L);
   }
   return 0;
}
