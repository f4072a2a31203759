// Phase clocks of the LDS-resident LDL^T solver (dense_ldlt.h) on a random SPD system (developer probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../morb_slam_amd/csrc/dense_ldlt.h"
__global__ __launch_bounds__(morbdense::LT) void k(const double* H, double* x, int n, unsigned long long* dbg) {
  extern __shared__ double sm[];
  __shared__ int sOk;
  morbdense::ldlt_solve<true>(H, x, x, n, sm, &sOk, dbg);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 120;
  const bool clocks = argc > 2 && atoi(argv[2]) != 0;   // phase clocks on: the time per solve then includes them
  std::vector<double> A((size_t)n * n), H((size_t)n * n, 0.0), b(n), x(n);
  for (auto& v : A) v = (rand() % 2001 - 1000) / 1000.0;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = i == j ? n : 0; for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * A[(size_t)j * n + k]; H[(size_t)i * n + j] = s; }
  for (auto& v : b) v = (rand() % 2001 - 1000) / 1000.0;
  double *dH, *dx; unsigned long long* dd;
  (void)hipMalloc(&dH, H.size() * 8); (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dd, 64);
  (void)hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice);
  const size_t lds = morbdense::lds_doubles(n) * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 20; ++it) {
    (void)hipMemcpy(dx, b.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemset(dd, 0, 64);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1), dim3(morbdense::LT), lds, 0, dH, dx, n, clocks ? dd : nullptr);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  (void)hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost);
  double res = 0; for (int i = 0; i < n; ++i) { double s = -b[i]; for (int j = 0; j < n; ++j) s += H[(size_t)i * n + j] * x[j]; res = fmax(res, fabs(s)); }
  unsigned long long h[8]; (void)hipMemcpy(h, dd, 64, hipMemcpyDeviceToHost);
  printf("n = %d: %.1f us per solve, residual %.2e; cycles of wave 0: load %llu, first diagonal block %llu, panel stages %llu, trailing updates: barrier wait %llu + own tile %llu + next diagonal block %llu, back substitution %llu; of the diagonal blocks, in-register stages %llu\n", n, best * 1e3, res, h[0], h[1], h[2], h[3], h[5], h[6], h[4], h[7]);
  return 0;
}
