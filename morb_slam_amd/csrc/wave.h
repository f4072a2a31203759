// Wave64 reductions on the DPP data path (gfx9 row_shr / row_bcast), result broadcast to all lanes.
// __shfl_xor butterflies compile to ds_bpermute (an LDS-crossbar round trip per step, six dependent steps); in the
// greedy / sequential kernels of this library (one wave replaying an order-dependent loop) a reduction sits on the
// critical path of every iteration, so it is done with six DPP VALU ops + one readlane instead.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace morbwave {

// dpp_ctrl: row_shr:n = 0x110 + n, row_bcast:15 = 0x142 (row_mask 0xa), row_bcast:31 = 0x143 (row_mask 0xc)
#define MORB_DPP_SCAN(v, ident, OP)                                                              \
  do {                                                                                           \
    v = OP(v, (decltype(v))__builtin_amdgcn_update_dpp((int)(ident), (int)(v), 0x111, 0xf, 0xf, false)); \
    v = OP(v, (decltype(v))__builtin_amdgcn_update_dpp((int)(ident), (int)(v), 0x112, 0xf, 0xf, false)); \
    v = OP(v, (decltype(v))__builtin_amdgcn_update_dpp((int)(ident), (int)(v), 0x114, 0xf, 0xf, false)); \
    v = OP(v, (decltype(v))__builtin_amdgcn_update_dpp((int)(ident), (int)(v), 0x118, 0xf, 0xf, false)); \
    v = OP(v, (decltype(v))__builtin_amdgcn_update_dpp((int)(ident), (int)(v), 0x142, 0xa, 0xf, false)); \
    v = OP(v, (decltype(v))__builtin_amdgcn_update_dpp((int)(ident), (int)(v), 0x143, 0xc, 0xf, false)); \
  } while (0)

__device__ __forceinline__ uint32_t op_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ int op_add(int a, int b) { return a + b; }

// min over the 64 lanes (all lanes must be active)
__device__ __forceinline__ uint32_t min_u32(uint32_t v) {
  MORB_DPP_SCAN(v, 0xFFFFFFFFu, op_umin);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// sum over the 64 lanes (all lanes must be active)
__device__ __forceinline__ int sum_i32(int v) {
  MORB_DPP_SCAN(v, 0, op_add);
  return __builtin_amdgcn_readlane(v, 63);
}
// min of a 64-bit key: high words first, then the low words of the lanes that hold the winning high word
__device__ __forceinline__ unsigned long long min_u64(unsigned long long v) {
  const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
  const uint32_t mh = min_u32(hi);
  const uint32_t ml = min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return ((unsigned long long)mh << 32) | ml;
}

// sum of a double over the 64 lanes (all lanes must be active): the two halves travel through DPP separately
__device__ __forceinline__ double sum_f64(double v) {
#define MORB_DPP_ADD_F64(ctrl, rowmask)                                                                             \
  do {                                                                                                              \
    const unsigned long long u_ = (unsigned long long)__double_as_longlong(v);                                      \
    const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u_, ctrl, rowmask, 0xf, false);                   \
    const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u_ >> 32), ctrl, rowmask, 0xf, false);           \
    v += __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi_ << 32) | (uint32_t)lo_));             \
  } while (0)
  MORB_DPP_ADD_F64(0x111, 0xf);
  MORB_DPP_ADD_F64(0x112, 0xf);
  MORB_DPP_ADD_F64(0x114, 0xf);
  MORB_DPP_ADD_F64(0x118, 0xf);
  MORB_DPP_ADD_F64(0x142, 0xa);
  MORB_DPP_ADD_F64(0x143, 0xc);
#undef MORB_DPP_ADD_F64
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum of a double over each 16-lane row (all lanes active), broadcast to the row's lanes: four row_shr steps put the total in the
// row's lane 15, row_newbcast:15 (gfx90a+) hands it to the other lanes.  Fixed order, no LDS.
__device__ __forceinline__ double row_sum_f64(double v) {
#define MORB_DPP_ROW_F64(ctrl, OPASSIGN)                                                                            \
  do {                                                                                                              \
    const unsigned long long u_ = (unsigned long long)__double_as_longlong(v);                                      \
    const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u_, ctrl, 0xf, 0xf, false);                       \
    const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u_ >> 32), ctrl, 0xf, 0xf, false);               \
    const double o_ = __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi_ << 32) | (uint32_t)lo_)); \
    OPASSIGN;                                                                                                       \
  } while (0)
  MORB_DPP_ROW_F64(0x111, v += o_);
  MORB_DPP_ROW_F64(0x112, v += o_);
  MORB_DPP_ROW_F64(0x114, v += o_);
  MORB_DPP_ROW_F64(0x118, v += o_);
#undef MORB_DPP_ROW_F64
  // (row_newbcast is the one DPP control with a 64-bit form: v_mov_b64_dpp)
  return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x15F, 0xf, 0xf, false));
}
// broadcast of lane `src`'s double (src wave-uniform; a compile-time constant compiles to two v_readlane_b32)
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

}  // namespace morbwave
