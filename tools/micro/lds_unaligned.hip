// Does gfx950 LDS serve unaligned ds_read_b32 / ds_read_b64, and at what cost?  (developer probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_probe(unsigned long long* out, int shift) {
  __shared__ __align__(16) unsigned char buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = (unsigned char)(i * 7 + 3);
  __syncthreads();
  unsigned addr = (unsigned)(size_t)(buf) + threadIdx.x * 8 + shift;
  unsigned long long v; unsigned w;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr) : "memory");
  out[threadIdx.x * 2] = v; out[threadIdx.x * 2 + 1] = w;
}
// throughput: 17 random byte reads vs 7 random unaligned b64 reads per lane
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(unsigned* sink, int iters, unsigned seed) {
  __shared__ __align__(16) unsigned char buf[6144];
  for (int i = threadIdx.x; i < 6144; i += blockDim.x) buf[i] = (unsigned char)(i * 7 + 3);
  __syncthreads();
  unsigned base = (unsigned)(size_t)(buf);
  unsigned acc = 0, r = seed + threadIdx.x * 2654435761u;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    r = r * 1664525u + 1013904223u;
    const unsigned off = 400 + ((r >> 8) % 5000u);   // random pixel
    if (MODE == 0) {
      const unsigned char* p = buf + off;
#pragma unroll
      for (int k = 0; k < 17; ++k) acc += p[(k % 7 - 3) * 128 + (k % 5) - 2];
    } else {
      unsigned long long v[7]; const unsigned a = base + off - 3 * 128 - 3;
      asm volatile("ds_read_b64 %0, %7\n\tds_read_b64 %1, %7 offset:128\n\tds_read_b64 %2, %7 offset:256\n\tds_read_b64 %3, %7 offset:384\n\t"
                   "ds_read_b64 %4, %7 offset:512\n\tds_read_b64 %5, %7 offset:640\n\tds_read_b64 %6, %7 offset:768\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]) : "v"(a) : "memory");
#pragma unroll
      for (int k = 0; k < 7; ++k) acc += (unsigned)v[k] + (unsigned)(v[k] >> 32);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { sink[1] = (unsigned)(t1 - t0); }
  if (acc == 0x12345) sink[0] = acc;
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 64 * 16));
  for (int shift = 0; shift < 8; ++shift) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, shift);
    unsigned long long h[128]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int ok64 = 1, ok32 = 1;
    for (int t = 0; t < 64; ++t) {
      unsigned long long e = 0; for (int b = 7; b >= 0; --b) e = (e << 8) | (unsigned char)((t * 8 + shift + b) * 7 + 3);
      if (h[2 * t] != e) ok64 = 0;
      if ((unsigned)h[2 * t + 1] != (unsigned)e) ok32 = 0;
    }
    printf("shift %d: ds_read_b64 %s, ds_read_b32 %s\n", shift, ok64 ? "exact" : "WRONG", ok32 ? "exact" : "WRONG");
  }
  unsigned* s; CK(hipMalloc(&s, 64));
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256 * 8), dim3(256), 0, 0, s, 10, 1u); else hipLaunchKernelGGL(k_rate<1>, dim3(256 * 8), dim3(256), 0, 0, s, 10, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256 * 8), dim3(256), 0, 0, s, 2000, 1u); else hipLaunchKernelGGL(k_rate<1>, dim3(256 * 8), dim3(256), 0, 0, s, 2000, 1u);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d (%s): %.3f ms for 2048 workgroups x 2000 random pixels per lane\n", mode, mode ? "7 unaligned ds_read_b64" : "17 ds_read_u8", ms);
  }
  return 0;
}
