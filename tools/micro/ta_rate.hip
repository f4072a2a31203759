// Vector-memory issue cost by access shape on gfx950 (developer tool): cycles per wave-instruction per CU for loads that hit the
// vector L1 / L2 (a 48-KiB window per workgroup, re-read many times), as a function of bytes per lane, lane stride and alignment.
// Answers "what does the memory pipe charge for a wave's load when the lanes' addresses overlap or straddle" — the shapes the image
// kernels use (k_resize: 8 bytes per lane every 4.8 bytes; k_blur: 16 bytes per lane every 8 bytes; k_describe: byte gathers).
//   hipcc --offload-arch=gfx950 -O3 -o ta_rate tools/micro/ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> __device__ __forceinline__ unsigned fold(T v);
template <> __device__ __forceinline__ unsigned fold(uint8_t v) { return v; }
template <> __device__ __forceinline__ unsigned fold(uint32_t v) { return v; }
template <> __device__ __forceinline__ unsigned fold(uint2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
struct u3 { unsigned x, y, z; };
template <> __device__ __forceinline__ unsigned fold(u3 v) { return v.x ^ v.y ^ v.z; }

// every wave issues ITER x 8 independent loads of T; lane address = wave window + lane * strideBytes10 / 10 + misalign + row * 1024
template <typename T> __global__ __launch_bounds__(256) void k_load(const uint8_t* __restrict__ buf, int stride10, int misalign, int iters, unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint8_t* p = buf + (size_t)(blockIdx.x & 63) * 49152 + wave * 8192 + (lane * stride10) / 10 + misalign;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_memcpy(&v[k], p + ((it & 3) * 8 + k) * 1024, sizeof(T));
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= fold(v[k]);
  }
  if (acc == 0x5Au) sink[blockIdx.x] = acc;
}
template <typename T> __global__ __launch_bounds__(256) void k_store(uint8_t* __restrict__ buf, int stride10, int misalign, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* p = buf + (size_t)(blockIdx.x & 63) * 49152 + wave * 8192 + (lane * stride10) / 10 + misalign;
  T v{};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_memcpy(p + ((it & 3) * 8 + k) * 1024, &v, sizeof(T));
  }
}

template <typename T> static void run(const char* name, bool store, uint8_t* buf, unsigned* sink, int stride10, int misalign) {
  const int grid = 256 * 8, iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    if (store) hipLaunchKernelGGL(k_store<T>, dim3(grid), dim3(256), 0, 0, buf, stride10, misalign, iters);
    else hipLaunchKernelGGL(k_load<T>, dim3(grid), dim3(256), 0, 0, buf, stride10, misalign, iters, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double instr = (double)grid * 4 * iters * 8;           // wave-instructions
  const double clkPerCu = ms * 1e-3 * 2.4e9 * 256 / instr;     // at a nominal 2.4 GHz
  printf("%-6s %-10s stride %5.1f B misalign %d : %7.2f clk per wave-instruction per CU  (%6.1f B/clk/CU requested)\n", store ? "store" : "load", name, stride10 / 10.0, misalign, clkPerCu,
         64.0 * sizeof(T) / clkPerCu);
}

int main() {
  uint8_t* buf; unsigned* sink;
  CHECK(hipMalloc(&buf, 64 * 49152 + 65536)); CHECK(hipMalloc(&sink, 1 << 20));
  CHECK(hipMemset(buf, 1, 64 * 49152 + 65536));
  run<uint32_t>("dword", false, buf, sink, 40, 0);
  run<uint32_t>("dword", false, buf, sink, 40, 1);
  run<uint32_t>("dword", false, buf, sink, 48, 0);
  run<uint2>("dwordx2", false, buf, sink, 80, 0);
  run<uint2>("dwordx2", false, buf, sink, 80, 4);
  run<uint2>("dwordx2", false, buf, sink, 80, 3);
  run<uint2>("dwordx2", false, buf, sink, 48, 0);
  run<uint2>("dwordx2", false, buf, sink, 48, 3);
  run<uint2>("dwordx2", false, buf, sink, 40, 0);
  run<u3>("dwordx3", false, buf, sink, 120, 0);
  run<u3>("dwordx3", false, buf, sink, 48, 0);
  run<u3>("dwordx3", false, buf, sink, 40, 0);
  run<uint4>("dwordx4", false, buf, sink, 160, 0);
  run<uint4>("dwordx4", false, buf, sink, 160, 4);
  run<uint4>("dwordx4", false, buf, sink, 160, 3);
  run<uint4>("dwordx4", false, buf, sink, 80, 0);
  run<uint4>("dwordx4", false, buf, sink, 48, 0);
  run<uint4>("dwordx4", false, buf, sink, 40, 0);
  run<uint8_t>("ubyte", false, buf, sink, 10, 0);
  run<uint8_t>("ubyte", false, buf, sink, 40, 0);
  run<uint8_t>("ubyte", false, buf, sink, 370, 0);
  run<uint32_t>("dword", true, buf, sink, 40, 0);
  run<uint2>("dwordx2", true, buf, sink, 80, 0);
  run<uint4>("dwordx4", true, buf, sink, 160, 0);
  run<uint4>("dwordx4", true, buf, sink, 160, 3);
  return 0;
}
