// Dense symmetric solve H x = b by LDL^T for the reduced camera systems (n <= ~160: LocalBundleAdjustment's 6 nFree, LocalInertialBA's
// 15 nOpt), ONE workgroup with the whole lower triangle resident in LDS.
//
// Round 1 factorised in global memory (LocalInertialBA: ~230 us at n = 150) or with a one-thread diagonal block and
// column-at-a-time substitutions (LocalBA: 98 us at n = 120); with a single workgroup nothing hides a global round trip.  Here:
//   * the lower triangle is packed in LDS (row r at r (r + 1) / 2), the right-hand side is carried as an extra ROW n of the matrix:
//     the panel step then performs the forward substitution and the division by D on the way (row n ends up holding D^-1 L^-1 b),
//     so only the back substitution L^T x = w remains;
//   * right-looking, 16-column panels.  Diagonal block: one wave, row r in lane r, the pivot row reaches the other lanes through
//     v_mov_b32_dpp row_newbcast (no SGPR round trip).  Rows below it: SIXTEEN lanes per row, one per column — the 16 dependent
//     stages of a row are one DPP broadcast + one FMA each, the lane's row of L sits in registers (a thread-per-row version with
//     the 120 LDS operands hoisted by the compiler spilled to scratch and took 7x longer).  Trailing triangle: rank-16 update on
//     the FP64 matrix cores, one wave per 16 x 16 tile.  3 barriers per panel; the back substitution takes 2 more per panel.
// Rounding differs from Eigen's / the oracle's column order only in the order of the updates (all FP64).
#pragma once
#include <hip/hip_runtime.h>

#include "wave.h"

namespace morbdense {
namespace {

constexpr int NB = 16, NBP = 17, LT = 512;   // 512 threads: two waves per SIMD
typedef double ld_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tri(int r, int c) { return ((r * (r + 1)) >> 1) + c; }   // c <= r

// lane SRC (0..15, compile-time) of every 16-lane row -> all lanes of the row: ONE v_mov_b64_dpp row_newbcast (the only DPP control gfx90a+
// offers on 64-bit operands; as two v_mov_b32_dpp halves every broadcast was two of the ~6-cycle issue slots a lone wave gets)
template <int SRC>
__device__ __forceinline__ double row_bcast_f64(double v) {
  return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + SRC, 0xf, 0xf, false));
}
// stage J of the in-register LDL^T of a 16 x 16 block (row r in lane r of each 16-lane row)
template <int J>
struct DiagStage {
  template <bool PIVOT_POSITIVE>
  static __device__ __forceinline__ void run(double (&a)[16], int row, bool& ok, double& rd) {
    const double d = row_bcast_f64<J>(a[J]);
    ok = ok && (PIVOT_POSITIVE ? (d > 0) : !(d == 0 || d != d));
    // 1 / d: v_rcp_f64 + two Newton steps (5 dependent instructions; the IEEE division sequence is 12, and this chain is what one wave
    // alone on its SIMD — ~6 cycles per instruction — spends the diagonal block on).  <= 1 ulp from the rounded quotient.
    double inv = __builtin_amdgcn_rcp(d);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    if (row == J) rd = inv;
    const double l = a[J] * inv;
    bc<J + 1>(a, l);
    if (row > J) a[J] = l;
    if constexpr (J + 1 < 16) DiagStage<J + 1>::template run<PIVOT_POSITIVE>(a, row, ok, rd);
  }
  template <int C>
  static __device__ __forceinline__ void bc(double (&a)[16], double l) {
    if constexpr (C < 16) { a[C] -= l * row_bcast_f64<J>(a[C]); bc<C + 1>(a, l); }
  }
};
// stages of a row below the block, one column per lane: v_c -= u_K L[c][K] with u_K = the (final) value of lane K
template <int K>
struct PanelStage {
  static __device__ __forceinline__ void run(double& v, const double (&lrow)[16]) {
    v -= row_bcast_f64<K>(v) * lrow[K];   // lrow[K] = 0 for K >= c: lanes at or left of the pivot column keep their value
    if constexpr (K + 1 < 16) PanelStage<K + 1>::run(v, lrow);
  }
};

// back substitution inside a block, row per lane: x_row -= L[C][row] x_C for C = 15 .. 1 (x_C final when its turn comes)
template <int C>
struct BackStage {
  static __device__ __forceinline__ void run(double& xr, const double (&lcol)[16]) {
    xr -= lcol[C] * row_bcast_f64<C>(xr);
    if constexpr (C > 1) BackStage<C - 1>::run(xr, lcol);
  }
};

// LDS doubles the solver needs for an n x n system (panel copies padded to whole 16-row tiles)
__host__ __device__ inline size_t lds_doubles(int n) {
  const size_t prow = (size_t)((n + 1 + NB - 1) / NB * NB);
  return (size_t)(n + 1) * (n + 2) / 2 + 2 * prow * NBP + NB * NBP + NB + n;
}

// PIVOT_POSITIVE: a pivot must be > 0 (LocalInertialBA's rule); otherwise it must be non-zero and not NaN (LocalBA's).
// Hs: symmetric n x n, row-major, pitch n (only the lower triangle is read); b, x: n.  Returns false if a pivot failed (x untouched).
template <bool PIVOT_POSITIVE>
__device__ bool ldlt_solve(const double* __restrict__ Hs, const double* b, double* x /* may alias b */, int n, double* sm, int* sOk,
                           unsigned long long* dbg = nullptr /* developer hook: phase clocks */) {
#define LD_MARK(k) do { if (dbg && threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg[k] += now_ - t0_; t0_ = now_; } } while (0)
  unsigned long long t0_ = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int prow = (n + 1 + NB - 1) / NB * NB;
  double* L = sm;                                        // packed lower triangle of the (n + 1) x (n + 1) augmented matrix
  double* pnlU = L + (size_t)(n + 1) * (n + 2) / 2;      // [prow][NBP]: u = l d of the rows below the current diagonal block
  double* pnlL = pnlU + (size_t)prow * NBP;              // [prow][NBP]: l of the same rows
  double* dblk = pnlL + (size_t)prow * NBP;              // [NB][NBP]: the factorised diagonal block: strictly lower part L, rest 0
  double* rdiag = dblk + NB * NBP;                       // [NB]: 1 / D
  double* xs = rdiag + NB;                               // [n]
  for (int r = wv; r < n; r += LT / 64)   // one wave per row: coalesced reads of the row's lower part
    for (int c = lane; c <= r; c += 64) L[tri(r, c)] = Hs[(size_t)r * n + c];
  for (int c = tid; c <= n; c += LT) L[tri(n, c)] = c < n ? b[c] : 0.0;
  for (int i = tid; i < 2 * prow * NBP; i += LT) pnlU[i] = 0.0;   // (rows beyond m feed MFMA lanes whose results are dropped: keep them finite)
  if (tid == 0) *sOk = 1;
  __syncthreads();
  LD_MARK(0);
  // The diagonal block at j0: wave 0 only, row r in lane r (identity padding beyond nb).  It is the critical path of the whole
  // factorisation (one wave issues an instruction every ~6 cycles), so from the second block on it runs UNDER the previous panel's
  // trailing update: wave 0 updates the tile that is the next diagonal block first, factorises it, and only the other seven waves
  // work through the remaining tiles (look-ahead).
  auto factor_diag = [&](int j0) {
    const int nb = n - j0 < NB ? n - j0 : NB;
      // the symmetric block -> dblk (four passes of 64 lanes, branch-free), then row r -> lane r (same wave: LDS keeps the order)
#pragma unroll
      for (int q = 0; q < NB * NB / 64; ++q) {
        const int idx = lane + 64 * q, r = idx >> 4, c = idx & 15;
        const bool in = r < nb && c < nb;
        const int hi = r > c ? r : c, lo = r > c ? c : r;
        const double v = L[in ? tri(j0 + hi, j0 + lo) : 0];
        dblk[r * NBP + c] = in ? v : (r == c ? 1.0 : 0.0);
      }
      const int row = lane & 15;   // (lanes 16..63 repeat the block in their own DPP rows; only lanes 0..15 store)
      double a[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) a[c] = dblk[row * NBP + c];
      bool ok = true;
      double rd = 1.0;
      DiagStage<0>::template run<PIVOT_POSITIVE>(a, row, ok, rd);
      ok = __ballot(!ok && lane < NB) == 0;
      if (lane < NB) {
        rdiag[row] = rd;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          dblk[row * NBP + c] = c < row ? a[c] : 0.0;
          if (c <= row && row < nb) L[tri(j0 + row, j0 + c)] = a[c];
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
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
          const int rr = ti * NB + 4 * v4 + lk, cc = tj * NB + li;
          if (rr < m && cc <= rr && j0 + nb + cc < n) L[tri(j0 + nb + rr, j0 + nb + cc)] -= acc[v4];
        }
      }
    }
    if (wv == 0 && j0 + nb < n) factor_diag(j0 + nb);
    __syncthreads();
    LD_MARK(3);
  }
  __syncthreads();
  if (*sOk == 0) return false;
  // back substitution L^T x = w, w = row n of the factorised matrix
  for (int r = tid; r < n; r += LT) xs[r] = L[tri(n, r)];
  __syncthreads();
  for (int j0 = ((n - 1) / NB) * NB; j0 >= 0; j0 -= NB) {
    const int nb = n - j0 < NB ? n - j0 : NB;
    if (wv == 0) {   // the block's unit upper triangle, lane = row
      const int row = lane & 15;
      double lcol[NB];
#pragma unroll
      for (int c = 1; c < NB; ++c) {   // L^T[row][c] = L[c][row]
        const bool in = row < c && c < nb;
        const double t = L[in ? tri(j0 + c, j0 + row) : 0];
        lcol[c] = in ? t : 0.0;
      }
      lcol[0] = 0.0;
      double xr = row < nb ? xs[j0 + row] : 0.0;
      BackStage<NB - 1>::run(xr, lcol);
      if (lane < nb) xs[j0 + lane] = xr;
    }
    __syncthreads();
    for (int r = tid; r < j0; r += LT) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < NB; ++k) if (k < nb) acc += L[tri(j0 + k, r)] * xs[j0 + k];
      xs[r] -= acc;
    }
    __syncthreads();
  }
  for (int r = tid; r < n; r += LT) x[r] = xs[r];
  LD_MARK(4);
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
