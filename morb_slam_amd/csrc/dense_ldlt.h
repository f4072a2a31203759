// Dense symmetric solve H x = b by LDL^T for the reduced camera systems (n <= ~160: LocalBundleAdjustment's 6 nFree, LocalInertialBA's
// 15 nOpt), ONE workgroup with the whole lower triangle resident in LDS.
//
// Round 1 factorised in global memory (LocalInertialBA: ~230 us at n = 150) or with a one-thread diagonal block and
// column-at-a-time substitutions (LocalBA: 98 us at n = 120); with a single workgroup nothing hides a global round trip.  Here:
//   * the lower triangle is packed in LDS (row r at r (r + 1) / 2), the right-hand side is carried as an extra ROW n of the matrix:
//     the panel step then performs the forward substitution and the division by D on the way (row n ends up holding D^-1 L^-1 b),
//     so only the back substitution L^T x = w remains;
//   * right-looking, 16-column panels.  Diagonal block: one wave, row r in lane r, the pivot row reaches the other lanes through
//     the row_newbcast DPP control (no SGPR round trip).  Rows below it: SIXTEEN lanes per row, one per column — the 16 dependent
//     stages of a row are one v_fmac_f64_dpp each, the lane's row of L sits in registers (a thread-per-row version with
//     the 120 LDS operands hoisted by the compiler spilled to scratch and took 7x longer).  Trailing triangle: rank-16 update on
//     the FP64 matrix cores, one wave per 16 x 16 tile.  3 barriers per panel;
//   * the back substitution is one wave's column-oriented sweep with x in registers (no barrier).
// Round 4 (tools/micro/ldlt_phases.hip, n = 120): 51 us -> see profiles/r04/README.md; the critical path is the one-wave diagonal block, whose
// broadcast + multiply-subtract went from four instructions to one.
// Rounding differs from Eigen's / the oracle's column order only in the order of the updates (all FP64).
#pragma once
#include <hip/hip_runtime.h>

#include "wave.h"

namespace morbdense {
namespace {

constexpr int NB = 16, NBP = 17, LT = 512;   // 512 threads: two waves per SIMD
typedef double ld_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tri(int r, int c) { return (int)(__umul24((unsigned)r, (unsigned)(r + 1)) >> 1) + c; }   // c <= r; r < 4096: a 24-bit multiply (v_mul_lo_u32 is quarter rate)

// lane SRC (0..15, compile-time) of every 16-lane row -> all lanes of the row: ONE v_mov_b64_dpp row_newbcast (the only DPP control gfx90a+
// offers on 64-bit operands; as two v_mov_b32_dpp halves every broadcast was two of the ~6-cycle issue slots a lone wave gets)
template <int SRC>
__device__ __forceinline__ double row_bcast_f64(double v) {
  return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + SRC, 0xf, 0xf, false));
}
// The same broadcast as inline assembly, for values an inline-assembly instruction may just have written (the compiler's hazard recogniser
// does not see those writes: a VALU write needs two wait states before a DPP read of the register).  All lanes of the result are written.
template <int SRC>
__device__ __forceinline__ double row_bcast_asm_f64(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(SRC));
  return r;
}
// acc -= (lane SRC of b's 16-lane row) * c as ONE instruction: gfx90a+ let the 64-bit VALU operations take the row_newbcast DPP control, and
// v_fmac_f64 has a VOP2 encoding (DPP applies to src0).  The compiler never forms it (it emits v_mov_b64 0 / s_nop / v_mov_b64_dpp / v_fma_f64
// for the intrinsic form): four issue slots of the ~6-cycle kind a lone wave gets, where this takes one plus the wait states.
// WAIT: the register read through DPP may have been written by the instruction before (two wait states needed; the compiler's hazard
// recogniser does not look into inline assembly)
template <int SRC, bool WAIT = true>
__device__ __forceinline__ void fnma_row_bcast_f64(double& acc, double b, double c) {
  if constexpr (WAIT) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(c), "n"(SRC));
  else asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(c), "n"(SRC));
}
// Stage J of the in-register LDL^T of a 16 x 16 block, row r in lane r of each 16-lane row, LOWER triangle only (a[c] is meaningful for
// c <= r; what sits in the other registers is never read): with u = column J below the pivot, l = u / d,
//     A[r][C] -= l_r u_C   (J < C <= r),   u_C = lane C's u: one v_fmac_f64_dpp row_newbcast:C per column.
// u was written a whole stage earlier, so these need no wait states; the pivot broadcast does.  A lone wave issues an instruction every
// ~6 cycles whatever it is, so the stage is priced in instructions: ~20 on average (it was ~31 with the compiler's four-instruction
// broadcast-multiply-subtract and a symmetric block).
template <int J>
struct DiagStage {
  template <bool PIVOT_POSITIVE>
  static __device__ __forceinline__ void run(double (&a)[16], bool& ok, double (&rinv)[16]) {
    const double d = row_bcast_asm_f64<J>(a[J]);
    ok = ok && (PIVOT_POSITIVE ? (d > 0) : !(d == 0 || d != d));
    // 1 / d: v_rcp_f64 + two Newton steps (5 dependent instructions; the IEEE division sequence is 12).  <= 1 ulp from the rounded quotient.
    double inv = __builtin_amdgcn_rcp(d);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    rinv[J] = inv;   // (the same in every lane of the 16-lane row)
    const double u = a[J];
    const double l = u * inv;
    bc<J + 1>(a, u, l);
    a[J] = l;   // (rows <= J: not meaningful, not read)
    if constexpr (J + 1 < 16) DiagStage<J + 1>::template run<PIVOT_POSITIVE>(a, ok, rinv);
  }
  template <int C>
  static __device__ __forceinline__ void bc(double (&a)[16], double u, double l) {
    if constexpr (C < 16) { fnma_row_bcast_f64<C, false>(a[C], u, l); bc<C + 1>(a, u, l); }
  }
};
// stages of a row below the block, one column per lane: v_c -= u_K L[c][K] with u_K = the (final) value of lane K
template <int K>
struct PanelStage {
  static __device__ __forceinline__ void run(double& v, const double (&lrow)[16]) {
    fnma_row_bcast_f64<K>(v, v, lrow[K]);   // lrow[K] = 0 for K >= c: lanes at or left of the pivot column keep their value
    if constexpr (K + 1 < 16) PanelStage<K + 1>::run(v, lrow);
  }
};

// LDS doubles the solver needs for an n x n system (panel copies padded to whole 16-row tiles)
__host__ __device__ inline size_t lds_doubles(int n) {
  const size_t prow = (size_t)((n + 1 + NB - 1) / NB * NB);
  return (size_t)(n + 1) * (n + 2) / 2 + 2 * prow * NBP + NB * NBP + NB + (n > 64 ? n : 64);
}

// PIVOT_POSITIVE: a pivot must be > 0 (LocalInertialBA's rule); otherwise it must be non-zero and not NaN (LocalBA's).
// Hs: symmetric n x n, row-major, pitch n (only the lower triangle is read); b, x: n.  Returns false if a pivot failed (x untouched).
template <bool PIVOT_POSITIVE>
__device__ bool ldlt_solve(const double* __restrict__ Hs, const double* b, double* x /* may alias b */, int n, double* sm, int* sOk,
                           unsigned long long* dbg = nullptr /* developer hook: phase clocks */) {
  // (phase clocks: kept in scalar registers and written once at the end — a read-modify-write of dbg[] per mark cost a global round trip each)
#define LD_MARK(k) do { if (dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc_[k] += now_ - t0_; t0_ = now_; } } while (0)
  unsigned long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0_ = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform for the compiler too: tile loops on the scalar unit)
  const int prow = (n + 1 + NB - 1) / NB * NB;
  double* L = sm;                                        // packed lower triangle of the (n + 1) x (n + 1) augmented matrix
  double* pnlU = L + (size_t)(n + 1) * (n + 2) / 2;      // [prow][NBP]: u = l d of the rows below the current diagonal block
  double* pnlL = pnlU + (size_t)prow * NBP;              // [prow][NBP]: l of the same rows
  double* dblk = pnlL + (size_t)prow * NBP;              // [NB][NBP]: the factorised diagonal block: strictly lower part L, rest 0
  double* rdiag = dblk + NB * NBP;                       // [NB]: 1 / D
  double* dump = rdiag + NB;                             // [max(n, 64)]: a spare word per lane for the branch-free masked stores
  {   // one wave per row, coalesced reads of the row's lower part — EIGHT rows requested before the first is stored: a request per
      // iteration (load, wait, store) made this phase 4.2 of the solver's 51 us at n = 120 (30 dependent L2 round trips per wave).
      // The zero fill of the panel copies and the right-hand side row go between the first batch's requests and its stores.
    constexpr int RB = 8, NP = 3;   // rows in flight, 64-column pieces per row (n <= 192)
    auto fill = [&]() {
      for (int i = tid; i < 2 * prow * NBP + NB * NBP; i += LT) pnlU[i] = 0.0;   // pnlU | pnlL (rows beyond m feed MFMA lanes whose results are dropped: keep them finite) | dblk
    };
    const int cb = tid <= n ? tid : n;   // (n + 1 <= LT)
    const double bv = b[cb < n ? cb : 0];
    bool filled = false;
    for (int r0 = wv; r0 < n; r0 += RB * (LT / 64)) {
      double v[RB][NP];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k * (LT / 64), rc = r < n ? r : n - 1;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const int c = lane + 64 * q;
          if (64 * q <= rc) v[k][q] = Hs[(size_t)rc * n + (c <= rc ? c : rc)];   // (piece-uniform test; clamped column: always a valid address)
        }
      }
      if (!filled) { fill(); filled = true; }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k * (LT / 64);
        if (r >= n) break;   // wave-uniform
        double* rowp = L + tri(r, 0) + lane;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          if (64 * q + 63 <= r) rowp[64 * q] = v[k][q];                 // a whole piece: no lane test (wave-uniform branch)
          else if (64 * q <= r && lane + 64 * q <= r) rowp[64 * q] = v[k][q];
        }
      }
    }
    if (!filled) fill();
    if (tid <= n) L[tri(n, tid)] = tid < n ? bv : 0.0;
  }
  if (tid == 0) *sOk = 1;
  __syncthreads();
  LD_MARK(0);
  // The diagonal block at j0: wave 0 only, row r in lane r (identity padding beyond nb).  It is the critical path of the whole
  // factorisation (one wave issues an instruction every ~6 cycles), so from the second block on it runs UNDER the previous panel's
  // trailing update: wave 0 updates the tile that is the next diagonal block first, factorises it, and only the other seven waves
  // work through the remaining tiles (look-ahead).
  auto factor_diag = [&](int j0) {
    const int nb = n - j0 < NB ? n - j0 : NB;
      // row r of the block's lower triangle -> lane r (lanes 16..63 repeat the block in their own DPP rows; only lanes 0..15 store): sixteen
      // independent LDS reads per lane from the row's start; the entries right of the diagonal are whatever follows in the packed triangle
      // (in bounds, never used).  A short last block is padded with identity rows and exact zeros (the panel stage multiplies by them).
      const int row = lane & 15;
      double* rp = L + tri(j0 + (row < nb ? row : 0), j0);
      double a[NB];
      if (nb == NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = rp[c];
      } else {
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          const bool in = row < nb && c <= row;
          const double v = rp[in ? c : 0];
          a[c] = in ? v : (row == c ? 1.0 : 0.0);
        }
      }
      bool ok = true;
      double rinv[NB];
      unsigned long long td_ = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
      DiagStage<0>::template run<PIVOT_POSITIVE>(a, ok, rinv);
      if (dbg) tacc_[7] += __builtin_amdgcn_s_memtime() - td_;
      ok = __ballot(!ok && lane < NB) == 0;
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) rdiag[c] = rinv[c];
      }
      // the strictly lower part = L goes to dblk (zeros from the diagonal on: the panel stage multiplies by them) and back into the packed
      // triangle (the back substitution reads it there).  Branch-free: one compare per column, the lanes that must not write (c >= row:
      // a zero of dblk / a word of another row of the triangle; padding rows) write to a spare word of their own instead.
      // D is not kept: the panel stage takes 1 / D from rdiag, nothing else reads it.
      if (lane < NB) {
        double* spare = dump + row;
        double* drow = dblk + row * NBP;
        const bool real = row < nb;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          const bool lo = c < row;
          double* d1 = lo ? drow + c : spare;            // (dblk is zero from the diagonal on since the start of the solve and stays so)
          double* d2 = (lo && real) ? rp + c : spare;
          *d1 = a[c];
          *d2 = a[c];
        }
      }
      if (lane == 0 && !ok) *sOk = 0;
  };
  if (wv == 0) factor_diag(0);
  __syncthreads();
  LD_MARK(1);
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int nb = n - j0 < NB ? n - j0 : NB, m = n + 1 - j0 - nb;   // m rows below the block (the last one is the right-hand side)
    if (*sOk == 0) break;   // uniform
    // (2) rows below the block: u = a - sum_k u_k L[c][k], l = u / d — sixteen lanes per row, one per column
    {
      const int c = tid & 15;
      double lrow[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) lrow[k] = dblk[c * NBP + k];   // strictly lower part of row c, zero from the diagonal on
      const double rdc = rdiag[c];
      for (int rr = tid >> 4; rr < m; rr += LT / 16) {
        const int R = j0 + nb + rr;
        const double t = L[tri(R, j0 + (c < nb ? c : 0))];
        double v = c < nb ? t : 0.0;
        PanelStage<0>::run(v, lrow);
        const double l = v * rdc;
        pnlU[rr * NBP + c] = v;
        pnlL[rr * NBP + c] = l;
        if (c < nb) L[tri(R, j0 + c)] = l;
      }
    }
    __syncthreads();
    LD_MARK(2);
    // (3) trailing update A[r][c] -= sum_k u[r][k] l[c][k] on the FP64 matrix cores: one wave per 16 x 16 tile of the lower
    // triangle (4 v_mfma_f64_16x16x4_f64; A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D[i][j] in lane j + 16 (i % 4),
    // register i / 4 — tools/micro/mfma_f64_layout.hip)
    {
      const int mt = (m + NB - 1) / NB, ntile = mt * (mt + 1) / 2;
      const int li = lane & 15, lk = lane >> 4;
      const bool ahead = j0 + nb < n;   // another diagonal block follows: wave 0 = tile 0 + that block, waves 1.. = the other tiles
      for (int t = ahead ? (wv == 0 ? 0 : wv) : wv; t < ntile; t += ahead ? (wv == 0 ? ntile : LT / 64 - 1) : LT / 64) {
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;   // tile (ti, tj), tj <= ti
        const int tj = t - ti * (ti + 1) / 2;
        ld_d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < NB / 4; ++s4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pnlU[(ti * NB + li) * NBP + 4 * s4 + lk], pnlL[(tj * NB + li) * NBP + 4 * s4 + lk], acc, 0, 0, 0);
        // Branch-free read-modify-write of the tile's four 4-row groups: a lane whose element is outside the triangle (above the
        // diagonal of a diagonal tile, beyond the matrix) works on a spare word of its own; the four old values are requested before
        // the matrix cores' result is waited for.  (As `if (in) L[..] -= acc[v4]` this was four branches and four dependent LDS round trips.)
        const int cc = tj * NB + li;
        double* dst[4];
        double old[4];
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          const int rr = ti * NB + 4 * v4 + lk;
          const bool in = rr < m && cc <= rr && j0 + nb + cc < n;
          double* p = L + tri(j0 + nb + rr, j0 + nb + cc);
          dst[v4] = in ? p : dump + lane;
          old[v4] = *dst[v4];
        }
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) *dst[v4] = old[v4] - acc[v4];
      }
    }
    LD_MARK(5);   // (wave 0: its tile)
    if (wv == 0 && j0 + nb < n) factor_diag(j0 + nb);
    LD_MARK(6);   // (wave 0: the next diagonal block)
    __syncthreads();
    LD_MARK(3);   // (wave 0: waiting for the other waves' tiles)
  }
  __syncthreads();
  if (*sOk == 0) return false;
  // Back substitution L^T x = w (w = row n of the factorised matrix) by ONE wave, column-oriented: lane l keeps x[l], x[l + 64], x[l + 128] in
  // registers; from the last unknown down, x_i is final, is read into scalar registers (v_readlane), filed into the output registers
  // (a select on the lane number) and leaves every x_j (j < i) through one FMA with L[i][j] — row i of the packed triangle, consecutive lanes = consecutive
  // LDS words, fetched eight steps ahead of its use.  No lane masks: lanes j >= i are multiplied by whatever follows row i in LDS (in bounds), but
  // their values have been filed already.  A lone wave pays ~6 cycles per instruction, so the step is priced in instructions: 2 + 3 + 1 + 2 per
  // 64 unknowns below i.  The blocked form this replaces (a 15-stage in-register solve per 16 x 16 block, then an update of the rows above by
  // the whole workgroup, two barriers per block) took 13 of the solver's 51 us at n = 120.
  if (wv == 0) {
    constexpr int NV = 3, U = 8;   // n <= 192 (the LDS-resident solver stops earlier)
    double xr[NV], xo[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) { const int j = 64 * q + lane; xr[q] = j < n ? L[tri(n, j)] : 0.0; xo[q] = 0.0; }
    const double* lp = L + lane;
#pragma unroll
    for (int q = NV - 1; q >= 0; --q) {
      const int base = 64 * q;
      if (base < n) {
        const int top = n - 1 < base + 63 ? n - 1 : base + 63;
        auto step = [&](int i, const double (&lrow)[NV]) {   // x_i is final: file it, take it out of the unknowns below
          const int li = __builtin_amdgcn_readfirstlane(i - base);
          const double xi = morbwave::readlane_f64(xr[q], li);
          xo[q] = lane == li ? xi : xo[q];
#pragma unroll
          for (int qq = 0; qq <= q; ++qq) xr[qq] = __builtin_fma(-lrow[qq], xi, xr[qq]);
        };
        int i0 = top;
        int rowOff = tri(top, 0);
        for (; i0 - (U - 1) >= base; i0 -= U) {   // eight rows of L requested, then the eight dependent steps
          double lv[U][NV];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const double* r = lp + rowOff;
#pragma unroll
            for (int qq = 0; qq <= q; ++qq) lv[u][qq] = r[64 * qq];
            rowOff -= i0 - u;   // tri(i - 1, 0) = tri(i, 0) - i
          }
          asm volatile("" ::: "memory");   // (keeps the requests ahead of the steps: the compiler otherwise sinks each one to its use)
#pragma unroll
          for (int u = 0; u < U; ++u) step(i0 - u, lv[u]);
        }
        for (; i0 >= base; --i0) {
          double lv[NV];
          const double* r = lp + rowOff;
#pragma unroll
          for (int qq = 0; qq <= q; ++qq) lv[qq] = r[64 * qq];
          rowOff -= i0;
          step(i0, lv);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) { const int j = 64 * q + lane; if (j < n) x[j] = xo[q]; }
  }
  __syncthreads();   // (callers read x from every thread)
  LD_MARK(4);
  if (dbg && threadIdx.x == 0) for (int k = 0; k < 8; ++k) dbg[k] += tacc_[k];
#undef LD_MARK
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same factorisation for systems whose packed triangle does not fit LDS (n > ~176: LocalBundleAdjustment beyond 29 free keyframes —
// the reference takes every covisible keyframe, Optimizer.cc:1058-1070 — and LocalInertialBA beyond 11): the matrix stays in global
// memory (L2), only the current 16-column panel lives in LDS.  ONE workgroup of 1024 threads, blocked right-looking LDL^T:
// the 16 x 16 diagonal block is factorised by one wave in registers, every row below it by sixteen lanes (PanelStage), then one
// trailing update of the matrix per panel on the FP64 matrix cores, TWO tiles per wave in flight so that eight global loads per lane
// hide the L2 round trip (a wave per row with the row requested 64 columns at a time took 1.0 ms per solve at n = 375; whole rows
// at once 0.74; this form 0.5).  3 barriers per panel.  Substitutions are blocked the same way.
constexpr int GT = 1024;   // threads of the global-memory solver
// LDS doubles: dblk[NB * NBP] | y[n]; and the panel copies pnlL | pnlU, (n + NB) * NBP doubles each, in LDS when they fit or in global scratch
__host__ __device__ inline size_t global_lds_doubles(int n) { return (size_t)NB * NBP + (size_t)n; }
__host__ __device__ inline size_t global_panel_doubles(int n) { return 2 * (size_t)(n + NB) * NBP; }

// LDL^T of a 16 x 16 block held in LDS (pitch 17; rows / columns >= nb are identity padding) by ONE wave: lane r keeps row r in
// registers, other rows' entries arrive through v_readlane.  Leaves L below the diagonal and D on it.
template <bool PIVOT_POSITIVE>
__device__ inline bool wave_ldl_factor16(double* blk, int lane) {
  const int row = lane < NB ? lane : NB - 1;
  double a[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) a[c] = blk[row * NBP + c];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double d = morbwave::readlane_f64(a[j], j);
    ok = ok && (PIVOT_POSITIVE ? (d > 0) : !(d == 0 || d != d));
    const double l = a[j] / d;
#pragma unroll
    for (int c = j + 1; c < NB; ++c) a[c] -= l * morbwave::readlane_f64(a[j], c);
    if (row > j) a[j] = l;
  }
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < NB; ++c) if (c <= row) blk[row * NBP + c] = a[c];
  }
  return ok;
}

// A: symmetric n x n, row-major, pitch n; its lower triangle is overwritten by L (unit, strictly lower) and D (diagonal).
// b, x: n (x may alias b).  pnl: global_panel_doubles(n) doubles (LDS or global), sm: global_lds_doubles(n) doubles of LDS.
// Returns false if a pivot failed (x untouched).  All GT threads must call.
template <bool PIVOT_POSITIVE>
__device__ bool ldlt_solve_global(double* __restrict__ A, const double* b, double* x, int n, double* pnl, double* sm, int* sOk) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int np = n + NB;   // (the panel copies are read in whole 16-row tiles by the matrix-core trailing update)
  double* pnlL = pnl; double* pnlU = pnl + (size_t)np * NBP; double* dblk = sm; double* y = sm + NB * NBP;
  if (tid == 0) *sOk = 1;
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int nb = n - j0 < NB ? n - j0 : NB, m = n - j0 - nb;   // m rows below the diagonal block
    // (1) diagonal block -> LDS (identity padding), factorised by wave 0
    if (tid < NB * NB) {
      const int r = tid / NB, c = tid - r * NB;
      dblk[r * NBP + c] = (r < nb && c < nb) ? (c <= r ? A[(size_t)(j0 + r) * n + j0 + c] : A[(size_t)(j0 + c) * n + j0 + r]) : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (wv == 0) { const bool ok = wave_ldl_factor16<PIVOT_POSITIVE>(dblk, lane); if (lane == 0 && !ok) *sOk = 0; }
    __syncthreads();
    if (*sOk == 0) break;   // uniform
    if (tid < NB * NB) {   // write L / D of the block back
      const int r = tid / NB, c = tid - r * NB;
      if (r < nb && c <= r) A[(size_t)(j0 + r) * n + j0 + c] = dblk[r * NBP + c];
    }
    // (2) panel rows below the block: u = a - sum_k u_k L[c][k], l = u / d — sixteen lanes per row, one per column, the 16 dependent
    // stages are one DPP row broadcast + one FMA each.  (The thread-per-row form, fully unrolled, had the compiler hoist its 120 LDS
    // operands into registers and spill 748 bytes per lane.)
    {
      const int c = tid & 15;
      double lrow[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) { const double t = dblk[c * NBP + k]; lrow[k] = k < c ? t : 0.0; }
      const double dc = dblk[c * NBP + c];
      for (int rr = tid >> 4; rr < m; rr += GT / 16) {
        const size_t g = (size_t)(j0 + nb + rr) * n + j0;
        const double t = A[g + (c < nb ? c : 0)];
        double v = c < nb ? t : 0.0;
        PanelStage<0>::run(v, lrow);
        const double l = v / dc;
        pnlU[rr * NBP + c] = v;
        pnlL[rr * NBP + c] = l;
        if (c < nb) A[g + c] = l;
      }
      // rows m .. next multiple of 16 of the panel copies feed matrix-core lanes whose results are dropped: keep them zero
      for (int i = m * NBP + tid; i < ((m + NB - 1) / NB * NB) * NBP; i += GT) { pnlU[i] = 0.0; pnlL[i] = 0.0; }
    }
    __syncthreads();
    // (3) trailing update A[r][cc] -= sum_k u[r][k] l[cc][k] on the FP64 matrix cores: one wave per 16 x 16 tile of the lower triangle
    {
      const int mt = (m + NB - 1) / NB, ntile = mt * (mt + 1) / 2;
      const int li = lane & 15, lk = lane >> 4;
      double* Abase = A + (size_t)(j0 + nb) * n + j0 + nb;
      constexpr int NW = GT / 64;
      for (int t0 = wv; t0 < ntile; t0 += 2 * NW) {
        int ti[2], tj[2];
        bool live[2];
        double c[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = t0 + u * NW;
          live[u] = t < ntile;
          int a = 0;
          while ((a + 1) * (a + 2) / 2 <= (live[u] ? t : 0)) ++a;
          ti[u] = a; tj[u] = (live[u] ? t : 0) - a * (a + 1) / 2;
#pragma unroll
          for (int v4 = 0; v4 < 4; ++v4) {
            const int rr = ti[u] * NB + 4 * v4 + lk, cc = tj[u] * NB + li;
            c[u][v4] = (live[u] && rr < m && cc <= rr) ? Abase[(size_t)rr * n + cc] : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (!live[u]) continue;   // wave-uniform
          ld_d4 acc = {0, 0, 0, 0};
#pragma unroll
          for (int s4 = 0; s4 < NB / 4; ++s4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pnlU[(ti[u] * NB + li) * NBP + 4 * s4 + lk], pnlL[(tj[u] * NB + li) * NBP + 4 * s4 + lk], acc, 0, 0, 0);
#pragma unroll
          for (int v4 = 0; v4 < 4; ++v4) {
            const int rr = ti[u] * NB + 4 * v4 + lk, cc = tj[u] * NB + li;
            if (rr < m && cc <= rr) Abase[(size_t)rr * n + cc] = c[u][v4] - acc[v4];
          }
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (*sOk == 0) return false;
  // blocked substitutions: per 16-column panel the 16 x 16 triangular block is solved by wave 0 in registers (v_readlane), the
  // rest of the panel is one 16-term dot product per row
  for (int r = tid; r < n; r += GT) y[r] = b[r];
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += NB) {   // L y = b
    const int nb = n - j0 < NB ? n - j0 : NB;
    if (tid < NB * NB) {
      const int r = tid / NB, c = tid - r * NB;
      dblk[r * NBP + c] = (r < nb && c < r) ? A[(size_t)(j0 + r) * n + j0 + c] : 0.0;
    }
    __syncthreads();
    if (wv == 0) {
      const int row = lane < NB ? lane : NB - 1;
      double yr = row < nb ? y[j0 + row] : 0.0;
#pragma unroll
      for (int c = 0; c < NB - 1; ++c) {
        const double yc = morbwave::readlane_f64(yr, c);
        if (row > c) yr -= dblk[row * NBP + c] * yc;
      }
      if (lane < nb) y[j0 + lane] = yr;
    }
    __syncthreads();
    for (int r = j0 + nb + tid; r < n; r += GT) {
      const double* lr = A + (size_t)r * n + j0;
      double acc = 0;
      for (int k = 0; k < nb; ++k) acc += lr[k] * y[j0 + k];
      y[r] -= acc;
    }
    __syncthreads();
  }
  for (int r = tid; r < n; r += GT) y[r] /= A[(size_t)r * n + r];
  __syncthreads();
  for (int j0 = ((n - 1) / NB) * NB; j0 >= 0; j0 -= NB) {   // L^T x = y
    const int nb = n - j0 < NB ? n - j0 : NB;
    if (tid < NB * NB) {
      const int r = tid / NB, c = tid - r * NB;
      dblk[r * NBP + c] = (r < nb && c < r) ? A[(size_t)(j0 + r) * n + j0 + c] : 0.0;
    }
    __syncthreads();
    if (wv == 0) {
      const int row = lane < NB ? lane : NB - 1;
      double xr = row < nb ? y[j0 + row] : 0.0;
#pragma unroll
      for (int c = NB - 1; c > 0; --c) {
        const double xc = morbwave::readlane_f64(xr, c);
        if (row < c) xr -= dblk[c * NBP + row] * xc;   // L^T[row][c] = L[c][row]
      }
      if (lane < nb) y[j0 + lane] = xr;
    }
    __syncthreads();
    for (int r = tid; r < j0; r += GT) {
      double acc = 0;
      for (int k = 0; k < nb; ++k) acc += A[(size_t)(j0 + k) * n + r] * y[j0 + k];
      y[r] -= acc;
    }
    __syncthreads();
  }
  for (int r = tid; r < n; r += GT) x[r] = y[r];
  return true;
}

}  // namespace
}  // namespace morbdense
