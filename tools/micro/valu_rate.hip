// VALU throughput of gfx950 at SATURATION (every SIMD holds 8 waves): cycles of a SIMD per wave64 instruction, by instruction kind.
// tools/alu_issue.hip timed single workgroups per CU with s_memtime and had to flag its 4- and 8-wave columns (the dispatcher does not
// place one workgroup per CU); here the chip is simply over-subscribed — 4 generations of 8 waves per SIMD — and the rate comes from the
// kernel's wall time:  cycles = time x clock x 1024 SIMDs / (waves x instructions per wave); the clock is measured by a dependent
// s_memtime-stamped chain in the same run (s_memtime counts shader cycles here: alu_issue's "shader clock" columns).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/micro/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
enum { AND, ADD, SUB, XOR, OR, LSHR, LSHL, MOV, MIN_I32, MAX_U32, MIN3, MAX3, PK_MIN, PK_ADD, PK_SUBSAT, PERM, MAD24, MUL24, XAD, ALIGNBIT, AND_OR, OR3, ADD3, LSHL_ADD, BFE, CNDMASK, DOT4, SAD, FMA32, MUL32F,
       PKFMA32, ADD_DPP, FFBL, BCNT, CMP_VCC, MIX_MIN3_AND, MIX_PK_AND, GRP_XOR_MIN3, GRP_XAD_MIN3, SALU_MIX, NOPS };
static const char* kName[NOPS] = {"v_and_b32", "v_add_u32", "v_sub_u32", "v_xor_b32", "v_or_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_min_i32", "v_max_u32", "v_min3_i32", "v_max3_i32",
  "v_pk_min_u16", "v_pk_add_u16", "v_pk_sub_u16 clamp", "v_perm_b32", "v_mad_i32_i24", "v_mul_i32_i24", "v_xad_u32", "v_alignbit_b32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_lshl_add_u32", "v_bfe_u32",
  "v_cndmask_b32 (vcc)", "v_dot4_u32_u8", "v_sad_u8", "v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_add_u32_dpp row_shr:1", "v_ffbl_b32", "v_bcnt_u32_b32", "v_cmp_lt_u32 vcc",
  "4 v_min3_i32 + 4 v_and_b32 (per instr)", "4 v_pk_min_u16 + 4 v_and_b32 (per instr)", "16 v_xor_b32 then 16 v_min3_i32 (per instr)", "16 v_xad_u32 then 16 v_min3_i32 (per instr)", "8 v_and_b32 + 4 s_add_u32 (per VALU instr)"};
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned* sink, unsigned seed) {
  unsigned a[8], b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x01020304u;
  unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
  float fb = 1.0f + 1e-6f * threadIdx.x, fc = 1e-7f;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 97u + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#define X1(ins) REP8(ins)
      if constexpr (OP == AND) {
#define I(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == ADD) {
#define I(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == SUB) {
#define I(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == XOR) {
#define I(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == OR) {
#define I(k) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == LSHR) {
#define I(k) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[k]));
        X1(I)
#undef I
      } else if constexpr (OP == LSHL) {
#define I(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k]));
        X1(I)
#undef I
      } else if constexpr (OP == MOV) {
#define I(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == MIN_I32) {
#define I(k) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == MAX_U32) {
#define I(k) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == MIN3) {
#define I(k) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == MAX3) {
#define I(k) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == PK_MIN) {
#define I(k) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == PK_ADD) {
#define I(k) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == PK_SUBSAT) {
#define I(k) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == PERM) {
#define I(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == MAD24) {
#define I(k) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == MUL24) {
#define I(k) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == XAD) {
#define I(k) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == ALIGNBIT) {
#define I(k) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == AND_OR) {
#define I(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == OR3) {
#define I(k) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == ADD3) {
#define I(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == LSHL_ADD) {
#define I(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == BFE) {
#define I(k) asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(a[k]));
        X1(I)
#undef I
      } else if constexpr (OP == CNDMASK) {
#define I(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
        X1(I)
#undef I
      } else if constexpr (OP == DOT4) {
#define I(k) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == SAD) {
#define I(k) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        X1(I)
#undef I
      } else if constexpr (OP == FMA32) {
#define I(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(fb), "v"(fc));
        X1(I)
#undef I
      } else if constexpr (OP == MUL32F) {
#define I(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(fb));
        X1(I)
#undef I
      } else if constexpr (OP == PKFMA32) {
        // four register pairs
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[0])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[2])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[4])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[0])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[2])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[4])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[0])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<unsigned long long*>(&a[2])) : "v"(*reinterpret_cast<unsigned long long*>(&a[6])));
      } else if constexpr (OP == ADD_DPP) {
#define I(k) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        X1(I)
#undef I
      } else if constexpr (OP == FFBL) {
#define I(k) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[k]));
        X1(I)
#undef I
      } else if constexpr (OP == BCNT) {
#define I(k) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        X1(I)
#undef I
      } else if constexpr (OP == CMP_VCC) {
#define I(k) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
        X1(I)
#undef I
      } else if constexpr (OP == MIX_MIN3_AND) {
        asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[1]) : "v"(b));
        asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[3]) : "v"(b));
        asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(b), "v"(c)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[5]) : "v"(b));
        asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[6]) : "v"(b), "v"(c)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[7]) : "v"(b));
      } else if constexpr (OP == MIX_PK_AND) {
        asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[1]) : "v"(b));
        asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[3]) : "v"(b));
        asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[4]) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[5]) : "v"(b));
        asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[6]) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[7]) : "v"(b));
      } else if constexpr (OP == GRP_XOR_MIN3) {
        if (rep & 1) {
#define I(k) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
          X1(I)
#undef I
        } else {
#define I(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
          X1(I)
#undef I
        }
      } else if constexpr (OP == GRP_XAD_MIN3) {
        if (rep & 1) {
#define I(k) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
          X1(I)
#undef I
        } else {
#define I(k) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
          X1(I)
#undef I
        }
      } else if constexpr (OP == SALU_MIX) {
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0));
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[1]) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1));
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[3]) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[4]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s2));
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[5]) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[6]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s3));
        asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[7]) : "v"(b));
      }
    }
  }
  unsigned r = s0 ^ s1 ^ s2 ^ s3;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i];
  if (r == 0x12345u && iters < 0) sink[threadIdx.x] = r;
}
__global__ void k_clock(unsigned long long* out, int n) {   // shader cycles (s_memtime) against the wall clock (s_memrealtime, 100 MHz)
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  unsigned a = threadIdx.x;
  for (int i = 0; i < n; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(a));
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a; }
}
template <int OP>
static void run(double ghz, unsigned* sink) {
  const int iters = 2048, perIter = 32;   // 32 VALU instructions per iteration (MIX rows: 32 of the named pattern's VALU + extras)
  const int wgs = 256 * 8 * 4;            // 4 generations of 8 waves per SIMD (4-wave workgroups: 8 per CU)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, iters, sink, 7u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, iters, sink, 7u);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double waves = (double)wgs * 4, instr = (double)iters * perIter;
  const double cyc = ms * 1e-3 * ghz * 1e9 * 1024.0 / (waves * instr);
  printf("%-44s %6.2f cycles per wave64 instruction per SIMD   (%.3f ms)\n", kName[OP], cyc, ms);
}
template <int OP>
static void run_all(double ghz, unsigned* sink) { run<OP>(ghz, sink); if constexpr (OP + 1 < NOPS) run_all<OP + 1>(ghz, sink); }
int main() {
  unsigned long long* d; unsigned* sink;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 4096));
  // clock under load: measured while a VALU kernel runs beside it would be best; a lone wave reports the boost clock
  hipLaunchKernelGGL(k<AND>, dim3(256 * 8 * 2), dim3(256), 0, 0, 4096, sink, 7u);
  hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, s2, d, 200000);
  CK(hipDeviceSynchronize());
  unsigned long long h[3];
  CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // s_memrealtime ticks at 100 MHz
  printf("# shader clock under load: %.3f GHz (s_memtime / s_memrealtime)\n", ghz);
  run_all<0>(ghz, sink);
  return 0;
}
