// Schur complement of the landmarks on the FP64 matrix cores (north_star: "MFMA used only for the dense per-camera 6x6 / 9x9
// Schur reduction"; g2o: BlockSolver::buildSystem / solve, Thirdparty/g2o/g2o/core/block_solver.hpp:354-480).
//
// g2o accumulates  Hschur(i1, i2) -= sum_m  Hpl(i1, m) (Hll(m) + lambda I)^-1 Hpl(i2, m)^T  block pair by block pair over the
// landmarks two cameras share.  Here the same sum is ONE dense product over ALL landmarks,
//     C = WD^T W,   W [3 nMP][Mp] : row 3 m + c holds column c of every camera's Hpl(., m) (zero where camera and landmark do not
//                                   meet), plus one extra column (index nc) with the landmark's right-hand side b_l(m)[c],
//                   WD[3 nMP][Mp] : the same with Hpl(., m) D(m)^-1,
// so C[0:nc, 0:nc] is the matrix part and C[0:nc, nc] = W D^-1 b_l the right-hand-side part of the complement.  ~25 % of W's 6 x 3
// blocks are non-zero on the C5 graph, i.e. the dense product spends ~8x the sparse form's flops — and is still several times
// faster, because v_mfma_f64_16x16x4_f64 retires 2048 flops per instruction where the per-pair VALU form was bound by 36 DPP
// wave reductions per block pair (LocalBA, 51 us) or by FP64 LDS atomics (LocalInertialBA, 1.0 ms at 25 keyframes).  The
// summation order is fixed (K split into `nsplit` ranges, each summed in k order, partials added in split order): deterministic.
//
// Operand layout of the instruction (tools/micro/mfma_f64_layout.hip): A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k,
// D[i][j] in lane j + 16 (i % 4), register i / 4.  Both operands are read K-major, 16 consecutive doubles per k: coalesced.
#pragma once
#include <hip/hip_runtime.h>

namespace morbschur {
namespace {   // (a kernel per translation unit: the library is linked from separately compiled objects)

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int SB = 32;   // a wave owns a 32 x 32 block of C (2 x 2 MFMA tiles sharing their operand loads)
constexpr int SU = 8;    // k-steps per software-pipeline group

// Partial products: part[(split * nblk + blk) * 1024 + i * 32 + j] = sum over the split's k range of WD[k][32 bi + i] W[k][32 bj + j],
// for the upper-triangular blocks (bi <= bj) listed in `blocks`.  Grid (nblk, nsplit), 64 threads.
__global__ __launch_bounds__(64) void k_schur_mfma(const double* __restrict__ WD, const double* __restrict__ W, int Mp, int ksteps,
                                                   int stepsPerSplit, const int2* __restrict__ blocks, double* __restrict__ part,
                                                   const int* __restrict__ skip) {
  if (skip && *skip) return;   // (device-side LM control: the solve has finished, this launch was queued ahead)
  const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
  const int2 blk = blocks[blockIdx.x];
  const int ks0 = blockIdx.y * stepsPerSplit, ks1 = ks0 + stepsPerSplit;   // (rows beyond the last landmark are zero)
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = d4{0, 0, 0, 0};
  const double* pa = WD + (size_t)(4 * ks0 + kq) * Mp + SB * blk.x + r;
  const double* pb = W + (size_t)(4 * ks0 + kq) * Mp + SB * blk.y + r;
  const size_t step = (size_t)4 * Mp;
  // Software pipeline in groups of SU k-steps: the 4 SU operand loads of the next group are in flight while the 4 SU MFMAs of
  // the current one issue (~2 k cycles, about one L2 round trip).  stepsPerSplit is a multiple of SU and the arrays carry one
  // spare, zero group behind the last split.
  double ca[SU][2], cb[SU][2];
#pragma unroll
  for (int u = 0; u < SU; ++u) { ca[u][0] = pa[u * step]; ca[u][1] = pa[u * step + 16]; cb[u][0] = pb[u * step]; cb[u][1] = pb[u * step + 16]; }
  for (int ks = ks0; ks < ks1; ks += SU) {
    pa += SU * step; pb += SU * step;
    double na[SU][2], nb[SU][2];
#pragma unroll
    for (int u = 0; u < SU; ++u) { na[u][0] = pa[u * step]; na[u][1] = pa[u * step + 16]; nb[u][0] = pb[u * step]; nb[u][1] = pb[u * step + 16]; }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[u][0], cb[u][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[u][0], cb[u][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[u][1], cb[u][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[u][1], cb[u][1], acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) { ca[u][0] = na[u][0]; ca[u][1] = na[u][1]; cb[u][0] = nb[u][0]; cb[u][1] = nb[u][1]; }
  }
  double* out = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (SB * SB);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) out[(16 * a + 4 * v + kq) * SB + 16 * b + r] = acc[a][b][v];
}

// Sum of the partials of element (i, j), i <= j's block; blkIndex[bi * nb + bj] = position in `blocks` (bi <= bj).  FOUR
// consecutive lanes share an element: lane q adds the q-th quarter of the splits in split order (eight loads in flight), the
// quarters are then added in quarter order — a fixed order, so the result is reproducible.  All four lanes return the sum.
__device__ __forceinline__ double schur_sum4(const double* __restrict__ part, const int* __restrict__ blkIndex, int nb, int nblk, int nsplit,
                                             int i, int j, int q) {
  const int bi = i / SB, bj = j / SB;
  const double* p = part + (size_t)blkIndex[bi * nb + bj] * (SB * SB) + (i % SB) * SB + (j % SB);
  const size_t stride = (size_t)nblk * (SB * SB);
  const int per = (nsplit + 3) >> 2, s0 = q * per, s1 = min(s0 + per, nsplit);
  double s = 0;
  int sp = s0;
  for (; sp + 8 <= s1; sp += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(sp + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp < s1; ++sp) s += p[(size_t)sp * stride];
  const int base = (threadIdx.x & 63) & ~3;
  const double q0 = __shfl(s, base, 64), q1 = __shfl(s, base + 1, 64), q2 = __shfl(s, base + 2, 64), q3 = __shfl(s, base + 3, 64);
  return ((q0 + q1) + q2) + q3;
}

// Host-side plan of one product: block list, split count, workspace sizes.
struct Plan {
  int Mp = 0, nb = 0, nblk = 0, Kp = 0, ksteps = 0, nsplit = 0, stepsPerSplit = 0;
  size_t wElems() const { return (size_t)(Kp + 4 * SU) * Mp; }   // + one spare group (the pipelined loads run one group ahead)
  size_t partElems() const { return (size_t)nsplit * nblk * SB * SB; }
};
static inline Plan make_plan(int ncols /* matrix columns + 1 right-hand-side column */, int K) {
  Plan p;
  p.Mp = (ncols + SB - 1) / SB * SB;
  p.nb = p.Mp / SB;
  p.nblk = p.nb * (p.nb + 1) / 2;
  p.ksteps = (K + 3) / 4;
  // enough (block, split) waves to give every SIMD of the chip about one, at least 8 k-steps each
  int ns = (1024 + p.nblk - 1) / p.nblk;
  ns = ns < 1 ? 1 : ns;
  while (ns > 1 && (p.ksteps + ns - 1) / ns < 2 * SU) --ns;
  p.stepsPerSplit = ((p.ksteps + ns - 1) / ns + SU - 1) / SU * SU;
  p.nsplit = (p.ksteps + p.stepsPerSplit - 1) / p.stepsPerSplit;
  p.Kp = p.nsplit * p.stepsPerSplit * 4;   // rows beyond K stay zero
  return p;
}

}  // namespace
}  // namespace morbschur
