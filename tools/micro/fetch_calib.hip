// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (developer tool): streaming kernels that move a KNOWN number of bytes
// with 1 / 2 / 4 / 8 / 16 bytes per lane, coalesced, plus the two access shapes the image kernels use (16-byte loads at an
// unaligned address; byte gathers through a wide window).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
// passes, tools/fetch_calib.sh); tools/fetch_calib.py divides the counters by the bytes printed here.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/micro/fetch_calib.hip && ./fetch_calib [MiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> __device__ __forceinline__ unsigned fold(T v);
template <> __device__ __forceinline__ unsigned fold(uint8_t v) { return v; }
template <> __device__ __forceinline__ unsigned fold(uint16_t v) { return v; }
template <> __device__ __forceinline__ unsigned fold(uint32_t v) { return v; }
template <> __device__ __forceinline__ unsigned fold(uint2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// read-only: every lane loads consecutive elements of width sizeof(T); one dword per workgroup is written at the end
template <typename T> __global__ __launch_bounds__(256) void k_calib_read(const T* __restrict__ src, size_t n, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= fold(src[i]);
  if (acc == 0x5Au) sink[blockIdx.x] = acc;   // (never true for the test pattern, but not provably so: keeps the loads alive without a store stream)
}
// write-only
template <typename T> __global__ __launch_bounds__(256) void k_calib_write(T* __restrict__ dst, size_t n) {
  T v{};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
// 16-byte loads at addresses that are 1 mod 16 (the window loads of k_fastw / k_blur are unaligned like this)
__global__ __launch_bounds__(256) void k_calib_read_unaligned16(const uint8_t* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 1 < n16; i += (size_t)gridDim.x * 256) {
    uint4 v;
    __builtin_memcpy(&v, src + i * 16 + 1, 16);
    acc ^= fold(v);
  }
  if (acc == 0x5Au) sink[blockIdx.x] = acc;
}
// rows of 48 bytes out of a pitch of 832 (a k_fastw cell window on a 752-px level): 3 x 16-byte loads per row, rows 832 bytes apart
__global__ __launch_bounds__(256) void k_calib_read_window48(const uint8_t* __restrict__ src, size_t rows, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rows * 3; i += (size_t)gridDim.x * 256) {
    const size_t r = i / 3, c = i - r * 3;
    uint4 v;
    __builtin_memcpy(&v, src + r * 832 + 35 + c * 16, 16);
    acc ^= fold(v);
  }
  if (acc == 0x5Au) sink[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;   // > the 256-MiB Infinity Cache
  const size_t bytes = mib << 20;
  uint8_t *a, *b; unsigned* sink;
  CHECK(hipMalloc(&a, bytes + 64)); CHECK(hipMalloc(&b, bytes + 64)); CHECK(hipMalloc(&sink, 1 << 20));
  CHECK(hipMemset(a, 1, bytes + 64)); CHECK(hipMemset(b, 1, bytes + 64));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 16;
  printf("bytes_per_kernel %zu\n", bytes);
#define RD(T, name) hipLaunchKernelGGL(k_calib_read<T>, dim3(grid), dim3(256), 0, 0, (const T*)a, bytes / sizeof(T), sink); CHECK(hipDeviceSynchronize()); \
                    hipLaunchKernelGGL(k_calib_read<T>, dim3(grid), dim3(256), 0, 0, (const T*)b, bytes / sizeof(T), sink); CHECK(hipDeviceSynchronize());
#define WR(T, name) hipLaunchKernelGGL(k_calib_write<T>, dim3(grid), dim3(256), 0, 0, (T*)a, bytes / sizeof(T)); CHECK(hipDeviceSynchronize()); \
                    hipLaunchKernelGGL(k_calib_write<T>, dim3(grid), dim3(256), 0, 0, (T*)b, bytes / sizeof(T)); CHECK(hipDeviceSynchronize());
  RD(uint8_t, r1) RD(uint16_t, r2) RD(uint32_t, r4) RD(uint2, r8) RD(uint4, r16)
  WR(uint8_t, w1) WR(uint16_t, w2) WR(uint32_t, w4) WR(uint2, w8) WR(uint4, w16)
  for (uint8_t* p : {a, b}) { hipLaunchKernelGGL(k_calib_read_unaligned16, dim3(grid), dim3(256), 0, 0, p, bytes / 16, sink); CHECK(hipDeviceSynchronize()); }
  for (uint8_t* p : {a, b}) { hipLaunchKernelGGL(k_calib_read_window48, dim3(grid), dim3(256), 0, 0, p, bytes / 832, sink); CHECK(hipDeviceSynchronize()); }
  printf("window48_bytes %zu (requested) %zu (64-byte lines touched)\n", (bytes / 832) * 48, (bytes / 832) * 128);
  return 0;
}
