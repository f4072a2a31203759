// Does a wave whose EXEC mask has only a few lanes set issue FP64 instructions faster?  (PoseOptimization's 6 x 6 solve is uniform code on one wave.)
//   dependent v_fma_f64 chain and eight independent chains, with 64 / 16 / 1 active lanes.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_exec tools/micro/f64_exec.hip && /tmp/f64_exec
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int n, int lanes, int indep, double* out, long long* cyc) {
  const int tid = threadIdx.x;
  double a = 1.0 + tid * 1e-9, b = 0.999999;
  double t[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  long long t0 = 0, t1 = 0;
  if (tid < lanes) {
    t0 = clock64();
    if (!indep) {
      for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t[0]) : "v"(b), "v"(a));
      }
    } else {
      for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t[k]) : "v"(b), "v"(a));
      }
    }
    t1 = clock64();
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += t[k];
  out[tid] = s;
  if (tid == 0) cyc[0] = t1 - t0;
}
int main() {
  double* dOut; long long* dCyc; long long h;
  (void)hipMalloc(&dOut, 64 * 8); (void)hipMalloc(&dCyc, 8);
  const int lanesv[4] = {64, 32, 16, 1};
  for (int indep = 0; indep < 2; ++indep)
    for (int li = 0; li < 4; ++li)
      for (int rep = 0; rep < 2; ++rep) {
        const int n = 4096;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, n, lanesv[li], indep, dOut, dCyc);
        (void)hipMemcpy(&h, dCyc, 8, hipMemcpyDeviceToHost);
        if (rep) printf("%s v_fma_f64, %2d active lanes: %.2f cycles per instruction\n", indep ? "independent" : "dependent  ", lanesv[li], (double)h / n);
      }
  return 0;
}
