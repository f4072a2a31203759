// VALU / MFMA issue-rate microbenchmark for gfx950 (VERDICT r01 task 1a): how many wave64 instructions per cycle does ONE
// SIMD issue for the integer / byte / FP64 operations the extractor and optimiser kernels are made of, as a function
// of the number of waves resident on the SIMD?  Every wave runs `iters` iterations of 16 independent instructions of
// one kind (eight accumulator chains, so a chain's dependent distance is 8 instructions), stamps s_memtime (shader
// cycles) around the loop, and the host reports
//     cycles per wave-instruction per SIMD = (elapsed cycles) / (waves per SIMD x instructions per wave).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/alu_issue tools/alu_issue.hip && /tmp/alu_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

enum Op { PK_MIN_U16, AND_B32, SUB_U32, DOT4_U8, MIN3_I32, MAD_I24, PERM_B32, ALIGNBYTE, LSHL_OR, FMA_F32, ADD_F64, FMA_F64, MUL_F64,
          MFMA_F64_16, MFMA_F64_4, SAD_U8, PK_SUB_U16, BFE_U32, CNDMASK, CMP_BALLOT, DPP_ADD, MIN_I32, ADD3_U32, CNDMASK_E64, LSHRREV, S_ADD, MIX_V_S, MIX_PK_S, MIX_V_2S, NOPS };
static const char* kNames[NOPS] = {"v_pk_min_u16", "v_and_b32", "v_sub_u32", "v_dot4_u32_u8", "v_min3_i32", "v_mad_i32_i24",
                                   "v_perm_b32", "v_alignbyte_b32", "v_lshl_or_b32", "v_fma_f32", "v_add_f64", "v_fma_f64", "v_mul_f64",
                                   "v_mfma_f64_16x16x4_f64", "v_mfma_f64_4x4x4_4b_f64", "v_sad_u8", "v_pk_sub_u16", "v_bfe_u32", "v_cndmask_b32",
                                   "v_cmp_lt_u32 (sgpr pair dst)", "v_add_u32_dpp row_shr:1", "v_min_i32 (VOP2)", "v_add3_u32 (VOP3)",
                                   "v_cndmask_b32 (VOP3, sgpr pair)", "v_lshrrev_b32 (VOP2)", "s_add_u32 (scalar ALU)",
                                   "8 x (v_and_b32 + s_add_u32), per instruction", "8 x (v_pk_min_u16 + s_add_u32), per instruction",
                                   "8 v_and + 4 v_pk_min + 4 s_add, per instruction"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void k_issue(int iters, unsigned long long* out, unsigned* sink, unsigned seed) {
  unsigned a[8], b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x01020304u;
  unsigned sa[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sa[k] = seed + k;
  double fa[8], fb = 1.0 + 1e-9 * threadIdx.x, fc = 1e-12;
  d4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 97u + k; fa[k] = 1.0 + k; acc[k] = d4{0, 0, 0, 0}; }
  extern __shared__ unsigned lds_pad[];   // dynamic LDS only fixes how many workgroups a CU can hold
  if (iters < 0) lds_pad[threadIdx.x] = seed;
  const unsigned long long selmask = 0x5555555555555555ull * (seed & 1);
  __builtin_amdgcn_s_barrier();
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      if constexpr (OP == PK_MIN_U16) {
#define X(k) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (OP == AND_B32) {
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (OP == SUB_U32) {
#define X(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (OP == DOT4_U8) {
#define X(k) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (OP == MIN3_I32) {
#define X(k) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (OP == MAD_I24) {
#define X(k) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (OP == PERM_B32) {
#define X(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (OP == ALIGNBYTE) {
#define X(k) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (OP == LSHL_OR) {
#define X(k) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (OP == FMA_F32) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (OP == ADD_F64) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(fa[k]) : "v"(fc));
        REP8(X)
#undef X
      } else if constexpr (OP == FMA_F64) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(fa[k]) : "v"(fb), "v"(fc));
        REP8(X)
#undef X
      } else if constexpr (OP == MUL_F64) {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(fa[k]) : "v"(fb));
        REP8(X)
#undef X
      } else if constexpr (OP == MFMA_F64_16) {
#define X(k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, fc, acc[k], 0, 0, 0);
        REP8(X)
#undef X
      } else if constexpr (OP == MFMA_F64_4) {
#define X(k) fa[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(fb, fc, fa[k], 0, 0, 0);
        REP8(X)
#undef X
      } else if constexpr (OP == SAD_U8) {
#define X(k) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (OP == PK_SUB_U16) {
#define X(k) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (OP == BFE_U32) {
#define X(k) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[k]));
        REP8(X)
#undef X
      } else if constexpr (OP == CNDMASK) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if constexpr (OP == CMP_BALLOT) {
        unsigned long long m;
#define X(k) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b)); a[k] += (unsigned)__builtin_amdgcn_readfirstlane((int)m) & 0u;
        REP8(X)
#undef X
      } else if constexpr (OP == MIN_I32) {
#define X(k) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
      } else if constexpr (OP == ADD3_U32) {
#define X(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if constexpr (OP == CNDMASK_E64) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(selmask));
        REP8(X)
#undef X
      } else if constexpr (OP == LSHRREV) {
#define X(k) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[k]));
        REP8(X)
#undef X
      } else if constexpr (OP == S_ADD) {
#define X(k) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa[k]) : "s"(seed) : "scc");
        REP8(X)
#undef X
      } else if constexpr (OP == MIX_V_S) {
        if (rep == 0) {
#define X(k) asm volatile("v_and_b32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(a[k]), "+s"(sa[k]) : "v"(b), "s"(seed) : "scc");
        REP8(X)
#undef X
        }
      } else if constexpr (OP == MIX_PK_S) {
        if (rep == 0) {
#define X(k) asm volatile("v_pk_min_u16 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(a[k]), "+s"(sa[k]) : "v"(b), "s"(seed) : "scc");
        REP8(X)
#undef X
        }
      } else if constexpr (OP == MIX_V_2S) {   // the k_fast mix: 0.38 scalar instructions per vector one
        if (rep == 0) {
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        REP8(X)
#undef X
        } else {
#define X(k) asm volatile("v_pk_min_u16 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(a[k]), "+s"(sa[k]) : "v"(b), "s"(seed) : "scc");
        X(0) X(1) X(2) X(3)
#undef X
        }
      } else if constexpr (OP == DPP_ADD) {
#define X(k) asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        REP8(X)
#undef X
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  unsigned s = 0;
  double fs = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { s ^= a[k] ^ sa[k]; fs += fa[k] + acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3]; }
  if (iters < 0) sink[threadIdx.x & 15] = s + (unsigned)(long long)fs;   // never true at run time: keeps the chains alive
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[4 * w] = t0; out[4 * w + 1] = t1; out[4 * w + 2] = r0; out[4 * w + 3] = r1;
  }
}

template <int OP>
static void run(int wavesPerSimd, int iters, unsigned long long* d_out, unsigned* d_sink, double* cyc, double* ghz) {
  // wavesPerSimd <= 4: ONE workgroup of 4 * wavesPerSimd waves per CU (the waves of a workgroup are dealt over the four SIMDs), and
  // a dynamic-LDS request of more than half the CU's 160 KiB so that no CU can take two; 8 waves / SIMD: two 1024-thread
  // workgroups per CU (a third of the LDS each would admit three, but 2048 threads per CU is the cap).
  const int wgWaves = wavesPerSimd <= 4 ? 4 * wavesPerSimd : 16;
  const int wgPerCu = wavesPerSimd <= 4 ? 1 : 2;
  const int grid = 256 * wgPerCu;
  const size_t lds = wgPerCu == 1 ? 96 * 1024 : 64 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_issue<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(wgWaves * 64), lds, 0, 16, d_out, d_sink, 1u);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(wgWaves * 64), lds, 0, iters, d_out, d_sink, 1u);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(4 * (size_t)grid * wgWaves);
  CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
  // chip-wide span on the constant 100 MHz clock (s_memrealtime): first wave's start to last wave's end; if every wave was
  // resident at once this equals one wave's own elapsed time
  unsigned long long rmin = ~0ull, rmax = 0;
  std::vector<double> d, clk;
  for (size_t w = 0; w < h.size() / 4; ++w) {
    d.push_back((double)(h[4 * w + 1] - h[4 * w]));
    clk.push_back((double)(h[4 * w + 1] - h[4 * w]) / (double)(h[4 * w + 3] - h[4 * w + 2]) * 0.1);
    rmin = std::min(rmin, h[4 * w + 2]); rmax = std::max(rmax, h[4 * w + 3]);
  }
  std::sort(d.begin(), d.end()); std::sort(clk.begin(), clk.end());
  const double med = d[d.size() / 2];
  *ghz = clk[clk.size() / 2];
  const double spanCycles = (double)(rmax - rmin) * 10.0 * *ghz;   // 10 ns per tick x shader GHz
  // co-residency check: the chip-wide span must not exceed the median wave's own time by more than a few per cent
  *cyc = (spanCycles > 1.1 * med ? -1.0 : 1.0) * med / ((double)wavesPerSimd * iters * 16);
}

int main() {
  unsigned long long* d_out; unsigned* d_sink;
  CK(hipMalloc(&d_out, 16 * 8192 * 4));
  CK(hipMalloc(&d_sink, 64));
  const int iters = 4096;
  printf("# gfx950 issue cost: shader cycles (s_memtime) per wave64 instruction per SIMD, median wave; %d x 16 independent-chain\n"
         "# instructions per wave; w = waves resident per SIMD (one workgroup of 4w waves per CU; 8 = two 1024-thread workgroups).\n"
         "# A negative entry means the waves were NOT all resident together (chip-wide span > 1.1 x a wave's own time).\n", iters);
  printf("%-34s %8s %8s %8s %8s   %s\n", "instruction", "1w", "2w", "4w", "8w", "shader clock GHz @1/8w");
  auto row = [&](auto tag) {
    constexpr int OP = decltype(tag)::value;
    double c[4], g[4];
    const int ws[4] = {1, 2, 4, 8};
    for (int i = 0; i < 4; ++i) run<OP>(ws[i], iters, d_out, d_sink, &c[i], &g[i]);
    printf("%-34s %8.2f %8.2f %8.2f %8.2f   %.2f %.2f\n", kNames[OP], c[0], c[1], c[2], c[3], g[0], g[3]);
  };
#define ROW(op) row(std::integral_constant<int, op>{});
  ROW(AND_B32) ROW(SUB_U32) ROW(MIN_I32) ROW(LSHRREV) ROW(CNDMASK) ROW(CNDMASK_E64) ROW(FMA_F32)
  ROW(PK_MIN_U16) ROW(PK_SUB_U16) ROW(DOT4_U8) ROW(MIN3_I32) ROW(ADD3_U32) ROW(MAD_I24) ROW(PERM_B32) ROW(ALIGNBYTE) ROW(LSHL_OR)
  ROW(SAD_U8) ROW(BFE_U32) ROW(CMP_BALLOT) ROW(DPP_ADD)
  ROW(S_ADD) ROW(MIX_V_S) ROW(MIX_PK_S) ROW(MIX_V_2S)
  ROW(ADD_F64) ROW(FMA_F64) ROW(MUL_F64) ROW(MFMA_F64_16) ROW(MFMA_F64_4)
  return 0;
}
