// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, checked against a host product (developer probe; the layout was
// found with one-hot operands):  A[i][k] in lane i + 16 k,  B[k][j] in lane j + 16 k,  D[i][j] in lane j + 16 (i % 4), register i / 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(4 * v + (l >> 4)) * 16 + (l & 15)] = acc[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (double)(rand() % 17 - 8); hB[i] = (double)(rand() % 13 - 6); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("v_mfma_f64_16x16x4_f64 layout: %s (%d mismatches)\n", bad ? "WRONG" : "as documented", bad);
  return bad != 0;
}
