// v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) operand layout probe: which lanes of which operand end up where
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
  const int l = threadIdx.x;
  const unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[l] = r[0]; o[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main() {
  int* d; int h[256];
  (void)hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 first ", "swap32 second", "swap16 first ", "swap16 second"};
  for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%d", l, h[64 * r + l]); printf("\n"); }
  return 0;
}
