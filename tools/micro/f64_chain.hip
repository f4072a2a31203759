// Dependent FP64 addition chains on gfx950: what one term of PoseOptimization's edge-order sums costs a wave that is (nearly) alone on its SIMD.
//   (a) N dependent v_add_f64 on registers (the floor: the instruction's dependent-issue latency)
//   (b) the same with a ds_read2_b64 per two terms issued eight rows ahead (k_pose_opt2's software-pipelined form)
//   (c) (a) with a second wave on the SIMD doing FP64 work of its own (the workers of the next stage)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_chain tools/micro/f64_chain.hip && /tmp/f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(512) void k_chain(int n, int mode, double* out, long long* cyc) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, wv = tid >> 6;
  for (int i = tid; i < 29 * 448; i += 512) lds[i] = 1e-3 * (i % 97);
  __syncthreads();
  double t = 0.0;
  if (wv == 0) {
    const long long t0 = clock64();
    if (mode >= 3) {} else if (mode == 0 || mode == 2) {
      double a = 1.0 + tid * 1e-9;
      for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(t) : "v"(a));
      }
    } else {
      const double* p = lds + (tid & 31);
      double v[8], w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[k * 29];
      for (int e = 0; e + 16 <= n; e += 16) {
        const int r1 = (e + 8) % 440, r2 = (e + 16) % 440;
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = p[(r1 + k) * 29];
        asm volatile("" : "+v"(t) : : "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
        asm volatile("" : "+v"(t) : : "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(r2 + k) * 29];
        asm volatile("" : "+v"(t) : : "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) t += w[k];
        asm volatile("" : "+v"(t) : : "memory");
      }
    }
    if (mode == 3) {   // loads interleaved with the additions: one ds_read2_b64 (two rows of the next batch) in the shadow of every second addition
      const double* p = lds + (tid & 31);
      double v[8], w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[k * 29];
      asm volatile("" : "+v"(t) : : "memory");
      for (int e = 0; e + 16 <= n; e += 16) {
        const int r1 = (e + 8) % 440, r2 = (e + 16) % 440;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t += v[2 * j]; t += v[2 * j + 1];
          w[2 * j] = p[(r1 + 2 * j) * 29]; w[2 * j + 1] = p[(r1 + 2 * j + 1) * 29];
          asm volatile("" : "+v"(t) : : "memory");
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t += w[2 * j]; t += w[2 * j + 1];
          v[2 * j] = p[(r2 + 2 * j) * 29]; v[2 * j + 1] = p[(r2 + 2 * j + 1) * 29];
          asm volatile("" : "+v"(t) : : "memory");
        }
      }
    }
    if (mode == 4) {   // as (b), but ONE wait per batch: an empty asm that "rewrites" the batch's eight registers makes the compiler wait for all of them there
      const double* p = lds + (tid & 31);
      double v[8], w[8];
#define LD8(dst, row) _Pragma("unroll") for (int k = 0; k < 8; ++k) dst[k] = p[((row) + k) * 29]; asm volatile("" : "+v"(t) : : "memory")
#define AD8(src) asm volatile("" : "+v"(src[0]), "+v"(src[1]), "+v"(src[2]), "+v"(src[3]), "+v"(src[4]), "+v"(src[5]), "+v"(src[6]), "+v"(src[7]), "+v"(t) : : "memory"); \
                 _Pragma("unroll") for (int k = 0; k < 8; ++k) t += src[k]; asm volatile("" : "+v"(t) : : "memory")
      LD8(v, 0);
      for (int e = 0; e + 32 <= n; e += 32) {
        const int r0 = e % 400;
        LD8(w, r0 + 8); AD8(v);
        LD8(v, r0 + 16); AD8(w);
        LD8(w, r0 + 24); AD8(v);
        LD8(v, r0 + 32); AD8(w);
      }
    }
    const long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
  } else if (mode == 2 && wv == 4) {   // a second wave on SIMD 0 (waves are dealt round-robin over the four SIMDs) busy with FP64 multiply-adds
    double a = 1.0 + tid * 1e-9, b = 0.5, c = 0.25, d = 0.125;
    for (int i = 0; i < 4 * n; ++i) { a = a * 1.0000001 + 1e-9; b = b * 1.0000002 + 1e-9; c = c * 1.0000003 + 1e-9; d = d * 1.0000004 + 1e-9; }
    t = a + b + c + d;
  }
  out[blockIdx.x * 512 + tid] = t;
}
int main() {
  double* d_out; long long* d_cyc; long long h;
  CK(hipMalloc(&d_out, 512 * 8)); CK(hipMalloc(&d_cyc, 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 29 * 448 * 8));
  const char* name[5] = {"dependent v_add_f64 on registers, wave alone", "with ds_read2_b64 eight rows ahead (k_pose_opt2's loop)", "registers, a second wave on the SIMD doing FP64 fma", "ds_read2_b64 interleaved: one per two additions", "eight rows ahead, one wait per batch, 32 terms per trip"};
  for (int mode = 0; mode < 5; ++mode)
    for (int n : {512, 2048}) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_chain, dim3(1), dim3(512), 29 * 448 * 8, 0, n, mode, d_out, d_cyc);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(&h, d_cyc, 8, hipMemcpyDeviceToHost));
      printf("%-64s n = %4d: %6lld cycles = %.2f per term\n", name[mode], n, h, (double)h / n);
    }
  return 0;
}
