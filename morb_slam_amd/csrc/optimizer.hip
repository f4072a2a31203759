// Levenberg-Marquardt optimisers for MI355X (gfx950), device-resident: Optimizer::PoseOptimization
// (reference src/Optimizer.cc:762-1051) and Optimizer::LocalBundleAdjustment (:1053-1441), i.e. g2o's
// OptimizationAlgorithmLevenberg (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-194) over
// BlockSolver_6_3 (core/block_solver.hpp:354-590) with the reference's edges (src/OptimizableTypes.cpp,
// g2o/types/types_six_dof_expmap.cpp), restated as batched kernels:
//   * PoseOptimization: one 256-thread workgroup per frame, the whole LM schedule (4 robust / outlier rounds) inside ONE
//     launch per batch; residual + Jacobian + J^T W J per edge, reduced in a fixed order (27 doubles, DPP wave sums)
//     -> deterministic; the 6x6 system is solved in registers.
//   * LocalBundleAdjustment, grid mode (default): one launch per LM phase over the whole chip.  Hpp blocks per keyframe by a
//     wave per 64-edge chunk, Hll per map point by a thread; the Schur complement of the landmarks is ONE dense FP64 product
//     WD^T W on the matrix cores (schur_mfma.h: v_mfma_f64_16x16x4_f64, split-K with fixed-order partial sums); the reduced
//     camera system is factorised in LDS (dense_ldlt.h); back-substitution per map point.  The LM control flow
//     (optimization_algorithm_levenberg.cpp:61-169: rho, lambda schedule, <= 10 trials, the ORB-SLAM stop rule) runs ON THE
//     DEVICE in a one-workgroup decision kernel; every phase kernel reads the LM state and returns at once when the solve
//     is finished or the phase is not due, so the host only keeps the queue one trial ahead and watches a mapped flag.
//   * LocalBundleAdjustment, persistent mode: one 1024-thread workgroup runs the whole loop (many small problems side by side).
// All arithmetic is FP64 like g2o; the reference's float leaks (float camera parameters, `const float invz` in
// the stereo projection, float Huber deltas, float chi2 tests) are reproduced.
#include <hip/hip_runtime.h>

#include <sched.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "internal_abi.h"
#include "kb8.h"
#include "dense_ldlt.h"
#include "schur_mfma.h"
#include "wave.h"

using namespace morb;

namespace {

struct Cam { float fx, fy, cx, cy, bf; };

struct SE3 {
  double q[4];  // x y z w
  double t[3];
};

// ---- SE3Quat algebra (g2o/types/se3quat.h; Eigen quaternion formulas) ---------------------------------------
__device__ __forceinline__ void se3_normalize(SE3& s) {
  if (s.q[3] < 0) { s.q[0] = -s.q[0]; s.q[1] = -s.q[1]; s.q[2] = -s.q[2]; s.q[3] = -s.q[3]; }
  const double n = sqrt(s.q[0] * s.q[0] + s.q[1] * s.q[1] + s.q[2] * s.q[2] + s.q[3] * s.q[3]);
  s.q[0] /= n; s.q[1] /= n; s.q[2] /= n; s.q[3] /= n;
}
__device__ __forceinline__ void q_rotate(const double* q, const double* v, double* out) {
  const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
  double a = uy * v[2] - uz * v[1], b = uz * v[0] - ux * v[2], c = ux * v[1] - uy * v[0];
  a += a; b += b; c += c;
  out[0] = v[0] + w * a + (uy * c - uz * b);
  out[1] = v[1] + w * b + (uz * a - ux * c);
  out[2] = v[2] + w * c + (ux * b - uy * a);
}
__device__ __forceinline__ void se3_map(const SE3& T, const double* x, double* out) {
  q_rotate(T.q, x, out);
  out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
__device__ __forceinline__ void q_to_R(const double* q, double* R) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Eigen's matrix -> quaternion (Quaternion.h, QuaternionBase::operator=(MatrixBase)); the largest-diagonal branch indexes the matrix with
// i, j = (i + 1) % 3, k = (j + 1) % 3 — as run-time indices they put the matrix into scratch memory in every pose update (a store and nine
// dependent loads on the critical path of each LM trial), so the three cases are spelled out with constant indices.
template <int I, int J, int K>
__device__ __forceinline__ void R_to_q_case(const double* m, double* q) {
  double t = sqrt(m[I * 3 + I] - m[J * 3 + J] - m[K * 3 + K] + 1.0);
  q[I] = 0.5 * t;
  t = 0.5 / t;
  q[3] = (m[K * 3 + J] - m[J * 3 + K]) * t;
  q[J] = (m[J * 3 + I] + m[I * 3 + J]) * t;
  q[K] = (m[K * 3 + I] + m[I * 3 + K]) * t;
}
__device__ __forceinline__ void R_to_q(const double* m, double* q) {
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
  } else {
    const bool one = m[4] > m[0];
    const bool two = m[8] > (one ? m[4] : m[0]);
    if (two) R_to_q_case<2, 0, 1>(m, q);
    else if (one) R_to_q_case<1, 2, 0>(m, q);
    else R_to_q_case<0, 1, 2>(m, q);
  }
}
__device__ __forceinline__ SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r = a;
  double rt[3];
  q_rotate(a.q, b.t, rt);
  r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
  const double* p = a.q; const double* o = b.q;
  r.q[3] = p[3] * o[3] - p[0] * o[0] - p[1] * o[1] - p[2] * o[2];
  r.q[0] = p[3] * o[0] + p[0] * o[3] + p[1] * o[2] - p[2] * o[1];
  r.q[1] = p[3] * o[1] + p[1] * o[3] + p[2] * o[0] - p[0] * o[2];
  r.q[2] = p[3] * o[2] + p[2] * o[3] + p[0] * o[1] - p[1] * o[0];
  se3_normalize(r);
  return r;
}
// x^3 rounded once (up to a double rounding in rare cases): glibc's pow — what g2o's `pow(theta, 3)` and `pow(2 * rho - 1, 3)` call on the CPU — is
// accurate to ~0.52 ulp, x * x * x carries two roundings.  Error-free products through FMA, then one sum.
__device__ __forceinline__ double cube_rn(double x) {
  const double p = x * x, e = __builtin_fma(x, x, -p);      // x^2 = p + e
  const double q = p * x, f = __builtin_fma(p, x, -q);      // p x = q + f
  return q + (f + e * x);
}
// glibc's sin for |x| < 0.126 (sysdeps/ieee754/dbl-64/s_sin.c: TAYLOR_SIN, 0.501 ulp; |x| < 2^-26: x): the argument range of an LM update's rotation.
// Larger arguments fall back to the device library's sin (<= 1 ulp from it).
__device__ __forceinline__ double sin_glibc_small(double x) {
  const double ax = fabs(x);
  if (ax < 0x1p-26) return x;
  if (ax < 0.126) {
    const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ECEp-7, s3 = -0x1.A01A019DB08B8p-13, s4 = 0x1.71DE27B9A7ED9p-19, s5 = -0x1.ADDFFC2FCDF59p-26;
    const double xx = x * x;
    const double poly = ((((s5 * xx + s4) * xx + s3) * xx + s2) * xx) + s1;
    const double t = (poly * x - 0.5 * 0.0) * xx + 0.0;   // TAYLOR_SIN(xx, a, da) with da = 0
    return x + t;
  }
  return sin(x);
}
#ifdef XP_TRIG

__device__ __forceinline__ double xp_sin(double x) { const double xx = x * x; return x * (1.0 + xx * (-1.0 / 6 + xx * (1.0 / 120 + xx * (-1.0 / 5040 + xx * (1.0 / 362880 + xx * (-1.0 / 39916800)))))); }
__device__ __forceinline__ double xp_cos(double x) { const double xx = x * x; return 1.0 + xx * (-0.5 + xx * (1.0 / 24 + xx * (-1.0 / 720 + xx * (1.0 / 40320 + xx * (-1.0 / 3628800 + xx * (1.0 / 479001600)))))); }
#endif
__device__ __forceinline__ SE3 se3_exp(const double* u) {
  const double wx = u[0], wy = u[1], wz = u[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  const double Om[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double Om2[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Om2[i * 3 + j] = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
  double R[9], V[9];
  if (theta < 0.00001) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
#ifdef XP_TRIG
    const double s = xp_sin(theta), c = xp_cos(theta);
    const double a = s / theta, b = (1 - c) / (theta * theta), cc = (theta - s) / (theta * theta * theta);
#else
    const double s = sin_glibc_small(theta), c = cos(theta);
    const double a = s / theta, b = (1 - c) / (theta * theta), cc = (theta - s) / cube_rn(theta);
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
      V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + cc * Om2[i];
    }
  }
  SE3 r;
  R_to_q(R, r.q);
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
  se3_normalize(r);
  return r;
}
__device__ __forceinline__ SE3 se3_from_float(const float* p) {
  SE3 s;
  for (int i = 0; i < 4; ++i) s.q[i] = (double)p[i];
  for (int i = 0; i < 3; ++i) s.t[i] = (double)p[4 + i];
  se3_normalize(s);
  return s;
}

// ---- edges -------------------------------------------------------------------------------------------------
// error = obs - project(xc); stereo = (ur >= 0).  Returns chi2 = info * |err|^2 (information = info * I).
__device__ __forceinline__ double edge_error(const Cam& cam, bool stereo, const double* xc, const float* obs, double info,
                                             double* err) {
  if (!stereo) {  // Pinhole::project(Vector3d) (Pinhole.cpp:38-44)
    err[0] = (double)obs[0] - ((double)cam.fx * xc[0] / xc[2] + (double)cam.cx);
    err[1] = (double)obs[1] - ((double)cam.fy * xc[1] / xc[2] + (double)cam.cy);
    err[2] = 0;
    return err[0] * (info * err[0]) + err[1] * (info * err[1]);
  }
  const float invz = (float)(1.0 / xc[2]);  // cam_project: `const float invz` (types_six_dof_expmap.cpp:191,340)
  const double p0 = xc[0] * invz * (double)cam.fx + (double)cam.cx;
  const double p1 = xc[1] * invz * (double)cam.fy + (double)cam.cy;
  const double p2 = p0 - (double)cam.bf * invz;
  err[0] = (double)obs[0] - p0; err[1] = (double)obs[1] - p1; err[2] = (double)obs[2] - p2;
  return err[0] * (info * err[0]) + err[1] * (info * err[1]) + err[2] * (info * err[2]);
}
// Huber (robust_kernel_impl.cpp:78-91): returns rho[0], *w = rho[1]
__device__ __forceinline__ double huber(double delta, double e, double* w) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { *w = 1.0; return e; }
  const double sqrte = sqrt(e);
  *w = delta / sqrte;
  return 2 * sqrte * delta - dsqr;
}
// pose Jacobian (d x 6); unary = the "...OnlyPose" formulas
__device__ __forceinline__ void jac_pose(const Cam& cam, bool stereo, bool unary, const double* xc, double* Jp) {
  const double x = xc[0], y = xc[1], z = xc[2];
  const double fx = cam.fx, fy = cam.fy, bf = cam.bf;
  if (!stereo) {  // -projectJac * SE3deriv (OptimizableTypes.cpp:49-62 / :134-156)
    const double a = fx / z, b = -fx * x / (z * z), c = fy / z, d = -fy * y / (z * z);
    Jp[0] = -(b * y); Jp[1] = -(a * z + b * -x); Jp[2] = -(a * -y); Jp[3] = -a; Jp[4] = -0.0; Jp[5] = -b;
    Jp[6] = -(c * -z + d * y); Jp[7] = -(d * -x); Jp[8] = -(c * x); Jp[9] = -0.0; Jp[10] = -c; Jp[11] = -d;
    for (int i = 12; i < 18; ++i) Jp[i] = 0;
  } else if (unary) {  // EdgeStereoSE3ProjectXYZOnlyPose::linearizeOplus (:375-403)
    const double invz = 1.0 / z, invz_2 = invz * invz;
    Jp[0] = x * y * invz_2 * fx; Jp[1] = -(1 + (x * x * invz_2)) * fx; Jp[2] = y * invz * fx;
    Jp[3] = -invz * fx; Jp[4] = 0; Jp[5] = x * invz_2 * fx;
    Jp[6] = (1 + y * y * invz_2) * fy; Jp[7] = -x * y * invz_2 * fy; Jp[8] = -x * invz * fy;
    Jp[9] = 0; Jp[10] = -invz * fy; Jp[11] = y * invz_2 * fy;
    Jp[12] = Jp[0] - bf * y * invz_2; Jp[13] = Jp[1] + bf * x * invz_2; Jp[14] = Jp[2];
    Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - bf * invz_2;
  } else {  // EdgeStereoSE3ProjectXYZ::linearizeOplus (:228-270)
    const double z_2 = z * z;
    Jp[0] = x * y / z_2 * fx; Jp[1] = -(1 + (x * x / z_2)) * fx; Jp[2] = y / z * fx;
    Jp[3] = -1. / z * fx; Jp[4] = 0; Jp[5] = x / z_2 * fx;
    Jp[6] = (1 + y * y / z_2) * fy; Jp[7] = -x * y / z_2 * fy; Jp[8] = -x / z * fy;
    Jp[9] = 0; Jp[10] = -1. / z * fy; Jp[11] = y / z_2 * fy;
    Jp[12] = Jp[0] - bf * y / z_2; Jp[13] = Jp[1] + bf * x / z_2; Jp[14] = Jp[2];
    Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - bf / z_2;
  }
}
// point Jacobian (d x 3)
__device__ __forceinline__ void jac_point(const Cam& cam, bool stereo, const double* xc, const double* R, double* Jl) {
  const double x = xc[0], y = xc[1], z = xc[2];
  const double fx = cam.fx, fy = cam.fy, bf = cam.bf;
  if (!stereo) {  // -projectJac * R
    const double a = fx / z, b = -fx * x / (z * z), c = fy / z, d = -fy * y / (z * z);
    for (int k = 0; k < 3; ++k) { Jl[k] = -(a * R[k] + b * R[6 + k]); Jl[3 + k] = -(c * R[3 + k] + d * R[6 + k]); Jl[6 + k] = 0; }
  } else {
    const double z_2 = z * z;
    for (int k = 0; k < 3; ++k) {
      Jl[k] = -fx * R[k] / z + fx * x * R[6 + k] / z_2;
      Jl[3 + k] = -fy * R[3 + k] / z + fy * y * R[6 + k] / z_2;
      Jl[6 + k] = Jl[k] - bf * R[6 + k] / z_2;
    }
  }
}

// ---- block reductions (fixed order -> deterministic) --------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v) { return morbwave::sum_f64(v); }   // DPP (wave.h), all lanes active
// Wave totals of 28 doubles at once, written to out[0 .. 27] (LDS).  Two butterfly steps on gfx950's v_permlane32_swap / v_permlane16_swap fold the
// values four to a register — after them row r (16 lanes) of register g holds partial sums of value 4 g + {0, 2, 1, 3}[r]
// (tools/micro/permlane_swap.hip prints the operand layout) — then a row total is sixteen v_fmac_f64_dpp row_newbcast (dense_ldlt.h) for four values together.
// ~200 instruction slots instead of ~1200 for 28 six-step DPP reductions: a third of k_pose_opt's iteration, whose waves are alone on their SIMDs.
// Fixed order (deterministic); not the order of sum_f64 (the tree-sum mode of PoseOptimization is the only caller).
__device__ __forceinline__ double swap_add_f64(double a, double b, bool rows16) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  unsigned lo0, lo1, hi0, hi1;
  if (rows16) {
    auto l = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false); lo0 = l[0]; lo1 = l[1];
    auto h = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false); hi0 = h[0]; hi1 = h[1];
  } else {
    auto l = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false); lo0 = l[0]; lo1 = l[1];
    auto h = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false); hi0 = h[0]; hi1 = h[1];
  }
  return __longlong_as_double((long long)(((unsigned long long)hi0 << 32) | lo0)) + __longlong_as_double((long long)(((unsigned long long)hi1 << 32) | lo1));
}
template <int K>
__device__ __forceinline__ void row_total_step(double& acc, double v, double minusOne) {
  if constexpr (K < 16) { morbdense::fnma_row_bcast_f64<K, K == 0>(acc, v, minusOne); row_total_step<K + 1>(acc, v, minusOne); }
}
__device__ __forceinline__ void wave_sum28_to(const double (&v)[28], double* __restrict__ out, int lane) {
  double c[14], d[7];
#pragma unroll
  for (int p = 0; p < 14; ++p) c[p] = swap_add_f64(v[2 * p], v[2 * p + 1], false);   // lanes 0 .. 31: value 2 p, lanes 32 .. 63: value 2 p + 1
#pragma unroll
  for (int g = 0; g < 7; ++g) d[g] = swap_add_f64(c[2 * g], c[2 * g + 1], true);      // rows 0 .. 3: values 4 g, 4 g + 2, 4 g + 1, 4 g + 3
  const double minusOne = -1.0;
  const int row = lane >> 4, idx = ((row & 1) << 1) | (row >> 1);
#pragma unroll
  for (int g = 0; g < 7; ++g) {
    double tot = 0.0;
    row_total_step<0>(tot, d[g], minusOne);   // tot += lane k of the row, k = 0 .. 15, in that order: every lane of the row ends with the row's total
    if ((lane & 15) == 0) out[4 * g + idx] = tot;
  }
}
template <int NW>
__device__ __forceinline__ double block_sum_d(double v, double* red /*[NW]*/) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += red[i];
  return s;
}

// LDL^T solve of an n x n SPD system held in registers/local arrays (n = 6)
__device__ __forceinline__ bool ldlt6(const double* Hin, const double* rhs, double* x) {
  // (every loop fully unrolled: an index that is not a compile-time constant puts the arrays into scratch memory, and the solve sits on
  // the critical path of every LM iteration)
  double A[36], D[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = Hin[i];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k] * D[k];
    ok = ok && (d > 0);   // LinearSolverDense: _cholesky.isPositive() (the factorisation runs on; the caller discards x)
    D[j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k] * D[k];
      A[i * 6 + j] = s / d;
    }
  }
  if (!ok) return false;
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    y[i] = rhs[i];
#pragma unroll
    for (int k = 0; k < i; ++k) y[i] -= A[i * 6 + k] * y[k];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] /= D[i];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
#pragma unroll
    for (int k = i + 1; k < 6; ++k) y[i] -= A[k * 6 + i] * y[k];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = y[i];
  return true;
}

// ---- KannalaBrandt8 (fisheye) camera in the optimisers (KannalaBrandt8.cpp:48-66, :149-184) ------------------
struct Rig {            // fisheye stereo rig: left / right KB8 cameras and mTrl (left-camera frame -> right-camera frame)
  float kbL[8], kbR[8];
  SE3 Trl;
};
using morbkb8::kb8_project_d;
using morbkb8::kb8_project_jac;
// unary edge of PoseOptimization: residual (returns chi2, sets st = "3-D stereo residual") ...
template <bool FISH>
__device__ __forceinline__ double pose_edge_error(const Cam& cam, const Rig& rig, const SE3& P, const SE3& Pr, bool right,
                                                  const double* X, const float* o, double info, double* err, bool& st,
                                                  double* xc) {
  se3_map(P, X, xc);
  if (!FISH) {
    st = !(o[2] < 0);
    return edge_error(cam, st, xc, o, info, err);
  }
  st = false;
  double uv[2];
  if (!right) kb8_project_d(rig.kbL, xc, uv);                       // EdgeSE3ProjectXYZOnlyPose, pCamera = left KB8
  else { double xr[3]; se3_map(Pr, X, xr); kb8_project_d(rig.kbR, xr, uv); }   // ...ToBody: (mTrl * T).map(Xw)
  err[0] = (double)o[0] - uv[0]; err[1] = (double)o[1] - uv[1]; err[2] = 0;
  return err[0] * (info * err[0]) + err[1] * (info * err[1]);
}
// ... and its 2x6 / 3x6 Jacobian w.r.t. the pose
template <bool FISH>
__device__ __forceinline__ void pose_edge_jac(const Cam& cam, const Rig& rig, bool right, bool st, const double* xc, double* Jp) {
  if (!FISH) { jac_pose(cam, st, true, xc, Jp); return; }
  const double x = xc[0], y = xc[1], z = xc[2];
  double pj[6], pjM[6];
  if (!right) {
    kb8_project_jac(rig.kbL, xc, pj);
    for (int k = 0; k < 6; ++k) pjM[k] = pj[k];
  } else {  // -projectJac(X_r) * R_rl * SE3deriv(X_l), X_r = mTrl.map(T.map(Xw))  (OptimizableTypes.cpp:88-104)
    double xr[3], M[9];
    se3_map(rig.Trl, xc, xr);
    kb8_project_jac(rig.kbR, xr, pj);
    q_to_R(rig.Trl.q, M);
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 3; ++c) pjM[r * 3 + c] = pj[r * 3] * M[c] + pj[r * 3 + 1] * M[3 + c] + pj[r * 3 + 2] * M[6 + c];
  }
  for (int r = 0; r < 2; ++r) {
    const double a = pjM[r * 3], b = pjM[r * 3 + 1], c = pjM[r * 3 + 2];
    Jp[r * 6 + 0] = -(b * -z + c * y); Jp[r * 6 + 1] = -(a * z + c * -x); Jp[r * 6 + 2] = -(a * -y + b * x);
    Jp[r * 6 + 3] = -a; Jp[r * 6 + 4] = -b; Jp[r * 6 + 5] = -c;
  }
  for (int k = 12; k < 18; ++k) Jp[k] = 0;
}

// ---- binary edges of LocalBundleAdjustment, pinhole or fisheye rig ---------------------------------------------
// obs[2] >= 0: EdgeStereoSE3ProjectXYZ; -1: EdgeSE3ProjectXYZ with the pinhole camera; -2: EdgeSE3ProjectXYZ with the
// left KB8 camera; -3: EdgeSE3ProjectXYZToBody (right KB8 camera behind mTrl).  (Optimizer.cc:1244-1351)
__device__ __forceinline__ bool edge_is_kb8(const Rig* rig, const float* o) { return rig != nullptr && o[2] < -1.5f; }
__device__ __forceinline__ double ba_edge_error(const Cam& cam, const Rig* rig, bool st, const double* xc, const float* o,
                                                double info, double* err) {
  if (!edge_is_kb8(rig, o)) return edge_error(cam, st, xc, o, info, err);
  double uv[2];
  if (o[2] > -2.5f) kb8_project_d(rig->kbL, xc, uv);
  else { double xr[3]; se3_map(rig->Trl, xc, xr); kb8_project_d(rig->kbR, xr, uv); }   // (mTrl * T).map(Xw) = mTrl.map(T.map(Xw))
  err[0] = (double)o[0] - uv[0]; err[1] = (double)o[1] - uv[1]; err[2] = 0;
  return err[0] * (info * err[0]) + err[1] * (info * err[1]);
}
// Jp (d x 6) and / or Jl (d x 3); R = rotation of the keyframe pose.  OptimizableTypes.cpp:134-156 / :185-208
// RIG = false: the problem has no fisheye rig.  The KannalaBrandt8 branch (its float libm, the right camera's extrinsics) otherwise costs
// the pinhole build kernel 100 VGPRs (220 instead of 119) and puts the pose Jacobian into scratch memory — for a branch it never takes.
template <bool RIG>
__device__ __forceinline__ void ba_edge_jac(const Cam& cam, const Rig* rig, bool st, const double* xc, const float* o,
                                            const double* R, double* Jp, double* Jl) {
  // RIG = true: EVERY edge of the problem is a KannalaBrandt8 edge (morb_ba_problem_create_fisheye writes obs[2] = -2 / -3 for all of them),
  // so the choice is made at compile time: with a run-time `edge_is_kb8` both branches wrote Jp and the array went to scratch memory (160 B)
  if (!RIG) {
    if (Jp) jac_pose(cam, st, false, xc, Jp);
    if (Jl) jac_point(cam, st, xc, R, Jl);
    return;
  }
  const double x = xc[0], y = xc[1], z = xc[2];
  double pj[6], pjM[6];
  if (o[2] > -2.5f) {
    kb8_project_jac(rig->kbL, xc, pj);
    _Pragma("unroll") for (int k = 0; k < 6; ++k) pjM[k] = pj[k];
  } else {
    double xr[3], M[9];
    se3_map(rig->Trl, xc, xr);
    kb8_project_jac(rig->kbR, xr, pj);
    q_to_R(rig->Trl.q, M);
    _Pragma("unroll") for (int r = 0; r < 2; ++r)
      _Pragma("unroll") for (int c = 0; c < 3; ++c) pjM[r * 3 + c] = pj[r * 3] * M[c] + pj[r * 3 + 1] * M[3 + c] + pj[r * 3 + 2] * M[6 + c];
  }
  _Pragma("unroll") for (int r = 0; r < 2; ++r) {
    const double a = pjM[r * 3], b = pjM[r * 3 + 1], c = pjM[r * 3 + 2];
    if (Jp) {
      Jp[r * 6 + 0] = -(b * -z + c * y); Jp[r * 6 + 1] = -(a * z + c * -x); Jp[r * 6 + 2] = -(a * -y + b * x);
      Jp[r * 6 + 3] = -a; Jp[r * 6 + 4] = -b; Jp[r * 6 + 5] = -c;
    }
    if (Jl) _Pragma("unroll") for (int k = 0; k < 3; ++k) Jl[r * 3 + k] = -(a * R[k] + b * R[3 + k] + c * R[6 + k]);
  }
  if (Jp) _Pragma("unroll") for (int k = 12; k < 18; ++k) Jp[k] = 0;
  if (Jl) _Pragma("unroll") for (int k = 6; k < 9; ++k) Jl[k] = 0;
}
// isDepthPositive (OptimizableTypes.h:117-123 / :152-158)
__device__ __forceinline__ bool ba_depth_positive(const Rig* rig, const float* o, const double* xc) {
  if (edge_is_kb8(rig, o) && !(o[2] > -2.5f)) { double xr[3]; se3_map(rig->Trl, xc, xr); return xr[2] > 0.0; }
  return xc[2] > 0.0;
}

// =====================================================================================================
// PoseOptimization: one workgroup per frame
// =====================================================================================================
// ORDERED: every sum over the edges (the 21 + 6 entries of H and b, the robustified chi2) is taken in EDGE ORDER, one addition after the
// other, like g2o's sequential loop over its id-sorted active edges (sparse_optimizer.cpp:482-487, block_solver.hpp:502-560) and like the oracle:
// near convergence rho = dChi2 / scale is ~0 and its sign — an LM decision — follows the last bits of those sums.  A strided partial sum per
// thread + a tree gives other last bits (observed: one trial more or less in one of nine problems).  Edges are taken 256 at a time: every
// thread writes its edge's 28 contributions to LDS (an inactive edge: exact zeros, which leave a floating-point sum unchanged), lanes 0 .. 27
// of wave 0 add their entry's 256 values in order.  A chain of dependent FP64 additions per sum: 0.68 ms instead of 0.43 ms per 256 frames of
// 600 edges (tools/pose_opt_modes.py; round 4's figures — round 5's k_pose_opt2 runs the ordered sums on the matrix core: 0.41 against 0.40 ms, and the
// edge-order mode became the default).
#ifndef MORB_PO_NT
#define MORB_PO_NT 256   // threads per frame of PoseOptimization's default (tree-sum) mode
#endif
constexpr int PO_PITCH = 29;   // doubles per edge row of the contribution buffer (28 used)
// tot + p[0] + p[STRIDE] + ... (m terms, m wave-uniform, added strictly in that order): sixteen LDS reads in flight, then their sixteen additions
// — one read per addition made the chain ~100 cycles per term — and no test inside the full batches (a `if (e + k < m)` per addition, although
// uniform, cost the lone wave four instruction slots per term instead of one); the reads of the last, partial batch stay inside the 256-row buffer
template <int STRIDE>
__device__ __forceinline__ double ordered_add(double tot, const double* __restrict__ p, int m) {
  int e = 0;
  for (; e + 16 <= m; e += 16) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = p[(e + k) * STRIDE];
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += v[k];
  }
  if (e < m) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = p[(e + k) * STRIDE];   // (rows beyond m hold an earlier chunk's values: not added)
#pragma unroll
    for (int k = 0; k < 16; ++k) if (e + k < m) tot += v[k];
  }
  return tot;
}
// The same sum, software-pipelined for a wave that is (nearly) alone on its SIMD.  Such a wave issues an instruction every ~6 cycles whatever it is and a
// dependent v_add_f64 every ~12 (tools/micro/f64_chain.hip -> profiles/r05/f64_chain_cycles_per_term.txt), so a term costs 12 cycles + 6 per other
// instruction: the next eight rows are REQUESTED before the current eight are added (the compiler had placed every batch's reads directly in front of
// their use: a full LDS round trip per batch, ~23 cycles per term); empty asm statements that carry the running sum and clobber memory pin that order
// (17); 32 terms per loop trip (two register sets, no copies).  Interleaving one read per two additions measured 20 in the micro-benchmark: reads do not
// hide in an addition's shadow.  ROWS = rows of the buffer (a multiple of 8): reads never leave it; rows >= m are read and not added.
template <int STRIDE, int ROWS>
__device__ __forceinline__ double ordered_add_pipe(double tot, const double* __restrict__ p, int m) {
  static_assert(ROWS % 8 == 0, "whole batches");
  double v[8], w[8];
#define MORB_LOAD8(dst, row)                                                   \
  do {                                                                         \
    const int r_ = (row) <= ROWS - 8 ? (row) : ROWS - 8;                       \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) dst[k] = p[(r_ + k) * STRIDE]; \
    asm volatile("" : "+v"(tot) : : "memory");                                 \
  } while (0)
#define MORB_ADD8(src)                                                                                                                       \
  do {                                                                                                                                       \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) tot += src[k];                                                                             \
    asm volatile("" : "+v"(tot) : : "memory");                                                                                               \
  } while (0)
  MORB_LOAD8(v, 0);
  int e = 0;
  for (; e + 32 <= m; e += 32) {
    MORB_LOAD8(w, e + 8);  MORB_ADD8(v);
    MORB_LOAD8(v, e + 16); MORB_ADD8(w);
    MORB_LOAD8(w, e + 24); MORB_ADD8(v);
    MORB_LOAD8(v, e + 32); MORB_ADD8(w);
  }
  for (; e + 8 <= m; e += 8) {   // (v holds rows e .. e + 7)
    MORB_LOAD8(w, e + 8);
    MORB_ADD8(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = w[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) if (e + k < m) tot += v[k];
#undef MORB_LOAD8
#undef MORB_ADD8
  return tot;
}
// The same sums on the FP64 matrix core.  v_mfma_f64_4x4x4_4b_f64 computes D[b][i][j] = C[b][i][j] + sum_k A[b][i][k] B[b][k][j] for four 4 x 4
// blocks b; on gfx950 the four products are added one after the other, k = 0 .. 3, each sum rounded to double — with B = 1 that is
// (((c + a0) + a1) + a2) + a3, bit for bit what four dependent v_add_f64 produce (tools/micro/mfma_chain.hip: 262 144 random accumulations with
// mixed signs, magnitudes 2^-30 .. 2^30 and near-total cancellation, all equal; k_mfma_order_selftest repeats the check when a handle is created and
// the VALU chain above stays as the form a device that fails it would run).  One instruction therefore advances 16 independent ordered sums
// (b, i) by FOUR edges in 4 passes; a dependent MFMA issues after ~25 cycles: ~7 cycles per edge with the 28 sums in two waves (16 + 12) where the
// v_add_f64 chain measured 14 - 17 inside this kernel (profiles/r05/README.md).
// Layout (found with one-hot operands): A lane = 16 k + 4 b + i, B lane = 16 k + 4 b + j, C / D lane = 16 i + 4 b + j.  Lane l of the summing wave
// reads entry (l & 15) [+ 16 in the second wave] of edge row e + (l >> 4); sum c (< 16) comes out in lanes 16 (c & 3) + 4 (c >> 2) + j.
// The rows [m, m16) have been zeroed by the workers (x + 0.0 is exact and the running sums are never -0.0).  ROWS is a multiple of 16.
// One accumulator (16 sums) per wave: a dependent MFMA every ~25 cycles, the reads of the rows 32 ahead in its shadow.  Two waves on two SIMDs carry the
// 28 sums at ~7 cycles per edge (sC16 = sC + 16 x the wave's index; ROWS is a multiple of 32).  (Both accumulators interleaved in one wave: 11 cycles per edge.)
template <int STRIDE, int ROWS>
__device__ __forceinline__ void ordered_add_mfma1(double& d, const double* __restrict__ sC16, int lane, int m16) {
  static_assert(ROWS % 32 == 0, "whole batches");
  const double* p = sC16 + (lane >> 4) * STRIDE + (lane & 15);
  double r[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) r[t] = p[4 * t * STRIDE];
  asm volatile("" : "+v"(d) : : "memory");
  int e = 0;
  for (; e + 32 <= m16; e += 32) {
    const double* q = p + (e + 32 <= ROWS - 32 ? e + 32 : ROWS - 32) * STRIDE;   // (reads never leave the buffer)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      d = __builtin_amdgcn_mfma_f64_4x4x4f64(r[t], 1.0, d, 0, 0, 0);
      r[t] = q[4 * t * STRIDE];
      asm volatile("" : "+v"(d) : : "memory");
    }
  }
  if (e < m16) {
#pragma unroll
    for (int t = 0; t < 4; ++t) d = __builtin_amdgcn_mfma_f64_4x4x4f64(r[t], 1.0, d, 0, 0, 0);
  }
}
// one wave: `rounds` accumulations of four pseudo-random terms per lane group through the matrix core and through dependent v_add_f64; *bad counts
// the results that differ in any bit
__global__ __launch_bounds__(64) void k_mfma_order_selftest(int rounds, int* __restrict__ bad) {
  const int lane = threadIdx.x;
  unsigned long long x = 0x9E3779B97F4A7C15ull * (unsigned long long)(lane + 1);
  auto next = [&]() {   // xorshift64* -> a double with a random sign, a full mantissa and an exponent in [-30, 30]
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    const unsigned long long r = x * 0x2545F4914F6CDD1Dull;
    const unsigned long long expo = 1023ull - 30ull + (r >> 52) % 61ull;
    return __longlong_as_double((long long)((r & 0x800FFFFFFFFFFFFFull) | (expo << 52)));
  };
  const int c16 = 4 * ((lane >> 2) & 3) + (lane >> 4);   // the sum this lane's C / D register holds
  const int s16 = lane & 15, dl = 16 * (s16 & 3) + 4 * (s16 >> 2);   // the sum this lane's A register feeds, and a D lane that holds it
  int nbad = 0;
  double c = 0.0;
  for (int r = 0; r < rounds; ++r) {
    double a = next();
    const double cs = __shfl(c, dl), a0s = __shfl(a, s16);
    if ((r & 3) == 1 && (lane >> 4) == 1) a = -(cs + a0s) * (1.0 + 0x1p-40 * (double)(x & 1023));   // the second term nearly cancels the running sum
    double ref = c;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double ak = __shfl(a, c16 + 16 * k); asm volatile("v_add_f64 %0, %0, %1" : "+v"(ref) : "v"(ak)); }
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, c, 0, 0, 0);
    nbad += __double_as_longlong(c) != __double_as_longlong(ref);
    if ((r & 15) == 15) c = next();   // a fresh running sum now and then
  }
  if (nbad) atomicAdd(bad, nbad);
}
#ifdef MORB_PO_TRACE
__device__ double g_poTrace[6 * 520];
__device__ int g_poTraceN;
extern "C" int morb_po_trace(double* out, int* n) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_poTrace), sizeof(double) * 6 * 520); (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_poTraceN), sizeof(int)); const int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_poTraceN), &z, sizeof z); return 0; }
#endif
template <bool FISH, bool ORDERED, int NT>
__global__ __launch_bounds__(NT) void k_pose_opt(int cap, const int* __restrict__ count, const uint8_t* __restrict__ hasMP,
                                                  const float* __restrict__ obs, const float* __restrict__ invSigma2,
                                                  const float* __restrict__ Xw, Cam cam, Rig rig, const int* __restrict__ nLeft,
                                                  float* __restrict__ poseIO, uint8_t* __restrict__ outlier,
                                                  int* __restrict__ nInliers, int* __restrict__ stats) {
  constexpr int NW = NT / 64;
  static_assert(!ORDERED || NT == 256, "the edge-order mode takes its edges 256 at a time");
  __shared__ double red[NW];
  __shared__ double sH[NW][28];
  __shared__ double sC[ORDERED ? 256 * PO_PITCH : 1];   // [edge of the chunk][entry]
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = count ? count[f] : cap;
  const size_t base = (size_t)f * cap;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);

  int nInit = 0;
  for (int i = tid; i < n; i += NT) {
    if (hasMP[base + i]) { ++nInit; outlier[base + i] = 0; }
  }
  nInit = (int)block_sum_d<NW>((double)nInit, red);
  if (nInit < 3) {  // Optimizer.cc:951
    if (tid == 0) { nInliers[f] = 0; if (stats) { stats[2 * f] = 0; stats[2 * f + 1] = 0; } }
    return;
  }
  const SE3 T0 = se3_from_float(poseIO + 7 * f);
  SE3 T = T0, Teval = T0;
  bool robust = true;
  int nBadEdges = 0, outerIts = 0, trials = 0;

  const int nL = FISH ? nLeft[f] : n;   // features >= nL are right-camera observations (fisheye rig)
  // robustified chi2 of the active edges at pose P
  auto chi2Active = [&](const SE3& P) -> double {
    const SE3 Pr = FISH ? se3_mul(rig.Trl, P) : P;
    double s = 0;
    if (ORDERED) {
      double tot = 0;
      for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        double c = 0;
        if (i < n && hasMP[base + i] && !outlier[base + i]) {
          const float* o = obs + (base + i) * 3;
          const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
          double xc[3], err[3], w;
          bool st;
          c = pose_edge_error<FISH>(cam, rig, P, Pr, FISH && i >= nL, X, o, (double)invSigma2[base + i], err, st, xc);
          if (robust) c = huber(st ? deltaStereo : deltaMono, c, &w);
        }
        sC[tid] = c;
        __syncthreads();
        if (tid == 0) tot = ordered_add<1>(tot, sC, n - c0 < 256 ? n - c0 : 256);
        __syncthreads();
      }
      if (tid == 0) red[0] = tot;
      __syncthreads();
      tot = red[0];
      __syncthreads();
      return tot;
    }
    for (int i = tid; i < n; i += NT) {
      if (!hasMP[base + i] || outlier[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      double xc[3], err[3], w;
      bool st;
      double c = pose_edge_error<FISH>(cam, rig, P, Pr, FISH && i >= nL, X, o, (double)invSigma2[base + i], err, st, xc);
      if (robust) c = huber(st ? deltaStereo : deltaMono, c, &w);
      s += c;
    }
    return block_sum_d<NW>(s, red);
  };

  for (int it = 0; it < 4; ++it) {
    T = T0;  // vSE3->setEstimate(pFrame->GetPose()) (:962-964)
    // ---- optimizer.optimize(10) ----
    double lambda = 0, ni = 2;
    int nBad = 0;
    for (int iter = 0; iter < 10; ++iter) {
      ++outerIts;
      // computeActiveErrors + activeRobustChi2 + buildSystem in one pass
      const SE3 Tr = FISH ? se3_mul(rig.Trl, T) : T;
      double acc[28];
#pragma unroll
      for (int k = 0; k < 28; ++k) acc[k] = 0;
      if (ORDERED) {
        double tot = 0;   // lanes 0 .. 27 of wave 0: entry `lane` (H upper triangle 0 .. 20, b 21 .. 26, chi2 27)
        for (int c0 = 0; c0 < n; c0 += 256) {
          const int i = c0 + tid;
          double con[28];
#pragma unroll
          for (int k = 0; k < 28; ++k) con[k] = 0;
          if (i < n && hasMP[base + i] && !outlier[base + i]) {
            const float* o = obs + (base + i) * 3;
            const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
            double xc[3], err[3], Jp[18], w = 1.0;
            bool st;
            const double info = (double)invSigma2[base + i];
            const bool right = FISH && i >= nL;
            double c = pose_edge_error<FISH>(cam, rig, T, Tr, right, X, o, info, err, st, xc);
            if (robust) c = huber(st ? deltaStereo : deltaMono, c, &w);
            con[27] = c;
            pose_edge_jac<FISH>(cam, rig, right, st, xc, Jp);
            // g2o's own expressions (base_unary_edge.hpp:54-66): omega_r = -Omega e (then * rho'), b += J^T omega_r, H += J^T (rho' Omega) J
            const double wr[3] = {-info * err[0] * w, -info * err[1] * w, -info * err[2] * w};
            const double wo = w * info;
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
              double bb = 0;
#pragma unroll
              for (int k = 0; k < 3; ++k) bb += Jp[k * 6 + r] * wr[k];   // (mono: row 2 and err[2] are zero)
              con[21 + r] = bb;
#pragma unroll
              for (int cc = 0; cc <= r; ++cc) {   // the LOWER triangle, (J_r w Omega) J_c as Eigen forms it: LinearSolverDense's LDLT reads that triangle
                double h = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) h += Jp[k * 6 + r] * wo * Jp[k * 6 + cc];
                con[q++] = h;
              }
            }
          }
#pragma unroll
          for (int k = 0; k < 28; ++k) sC[tid * PO_PITCH + k] = con[k];
          __syncthreads();
          if (tid < 28) tot = ordered_add<PO_PITCH>(tot, sC + tid, n - c0 < 256 ? n - c0 : 256);
          __syncthreads();
        }
        if (tid < 28) sH[0][tid] = tot;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 28; ++k) acc[k] = sH[0][k];
        __syncthreads();
      } else
      for (int i = tid; i < n; i += NT) {
        if (!hasMP[base + i] || outlier[base + i]) continue;
        const float* o = obs + (base + i) * 3;
        const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
        double xc[3], err[3], Jp[18], w = 1.0;
        bool st;
        const double info = (double)invSigma2[base + i];
        const bool right = FISH && i >= nL;
        double c = pose_edge_error<FISH>(cam, rig, T, Tr, right, X, o, info, err, st, xc);
        if (robust) c = huber(st ? deltaStereo : deltaMono, c, &w);
        acc[27] += c;
        pose_edge_jac<FISH>(cam, rig, right, st, xc, Jp);
        const double wo = w * info;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double bb = 0;
#pragma unroll
          for (int k = 0; k < 3; ++k) bb += Jp[k * 6 + r] * (info * err[k]);   // mono: row 2 and err[2] are zero; a runtime bound would push Jp into scratch memory
          acc[21 + r] -= w * bb;  // b -= rho' * J^T * Omega * e  (base_unary_edge.hpp:61)
#pragma unroll
          for (int cc = 0; cc <= r; ++cc) {   // the LOWER triangle, (J_r w Omega) J_c as Eigen forms it: LinearSolverDense's LDLT reads that triangle
            double h = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) h += Jp[k * 6 + r] * wo * Jp[k * 6 + cc];
            acc[q++] += h;
          }
        }
      }
      if (!ORDERED) {
        // (the chi2 entry keeps the reduction order of chi2Active's block sum: rho compares the two, and near convergence their difference is of the
        // size of a summation-order effect)
        const double chiW = wave_sum_d(acc[27]);
        acc[27] = 0;
        __syncthreads();
        wave_sum28_to(acc, sH[wv], lane);
        if (lane == 0) sH[wv][27] = chiW;   // (a later instruction than the zero lane 48 just stored there: LDS keeps a wave's instructions in order)
        __syncthreads();
      }
      double H[36], b[6];
      {
        double tot[28];
        for (int k = 0; k < 28; ++k) {
          if (ORDERED) tot[k] = acc[k];
          else { double t = sH[0][k]; for (int w = 1; w < NW; ++w) t += sH[w][k]; tot[k] = t; }   // ((s0 + s1) + s2) + s3 ...: the order of the 4-wave form
        }
        int q = 0;
        for (int r = 0; r < 6; ++r) for (int cc = 0; cc <= r; ++cc) { H[r * 6 + cc] = tot[q]; H[cc * 6 + r] = tot[q]; ++q; }   // (the solve reads the lower triangle)
        for (int r = 0; r < 6; ++r) b[r] = tot[21 + r];
        acc[27] = tot[27];
      }
#ifdef MORB_PO_TRACE
      if (f == 0 && tid == 0 && outerIts == 1) { double* t = g_poTrace + 6 * 504; for (int k = 0; k < 36; ++k) t[k] = H[k]; for (int k = 0; k < 6; ++k) t[36 + k] = b[k]; }
#endif
      double currentChi = acc[27];
      const double iniChi = currentChi;
      if (iter == 0) {  // computeLambdaInit (tau = 1e-5)
        double m = 0;
        for (int r = 0; r < 6; ++r) m = fmax(fabs(H[r * 6 + r]), m);
        lambda = 1e-5 * m; ni = 2; nBad = 0;
      }
      double rho = 0;
      int qmax = 0;
      do {
        const SE3 backup = T;
        double Hl[36], x[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 36; ++k) Hl[k] = H[k];
        for (int r = 0; r < 6; ++r) Hl[r * 6 + r] += lambda;
        const bool ok2 = ldlt6(Hl, b, x);
        T = se3_mul(se3_exp(x), T);
        Teval = T;
        double tempChi = chi2Active(T);
        if (!ok2) tempChi = 1.7976931348623157e308;
        rho = currentChi - tempChi;
        double scale = 0;
        for (int r = 0; r < 6; ++r) scale += x[r] * (lambda * x[r] + b[r]);
        scale += 1e-3;
        rho /= scale;
#ifdef MORB_PO_TRACE
        if (f == 0 && tid == 0 && g_poTraceN == 0) { double* t = g_poTrace + 6 * 500; for (int k = 0; k < 6; ++k) t[k] = x[k]; for (int k = 0; k < 4; ++k) t[6 + k] = Teval.q[k]; for (int k = 0; k < 3; ++k) t[10 + k] = Teval.t[k]; }
        if (f == 0 && tid == 0 && g_poTraceN < 500) { double* t = g_poTrace + 6 * g_poTraceN++; t[0] = currentChi; t[1] = tempChi; t[2] = lambda; t[3] = rho; t[4] = scale; t[5] = ok2; }
#endif
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - cube_rn(2 * rho - 1);
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          T = backup;
        }
        ++qmax; ++trials;
      } while (rho < 0 && qmax < 10);
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) break;
    }
    // ---- classify (:966-1037): inlier edges keep the error of the LAST evaluated state (Teval, which is a
    // rejected trial when the LM loop ended on a failure), current outliers are re-evaluated at the final pose
    int bad = 0;
    __syncthreads();
    const SE3 TrFin = FISH ? se3_mul(rig.Trl, T) : T, TrEval = FISH ? se3_mul(rig.Trl, Teval) : Teval;
    for (int i = tid; i < n; i += NT) {
      if (!hasMP[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      double xc[3], err[3];
      bool st;
      const SE3& Pc = outlier[base + i] ? T : Teval;
      const float chi2 = (float)pose_edge_error<FISH>(cam, rig, Pc, outlier[base + i] ? TrFin : TrEval, FISH && i >= nL, X, o,
                                                      (double)invSigma2[base + i], err, st, xc);
      const bool isOut = chi2 > (st ? 7.815f : 5.991f);
      outlier[base + i] = isOut ? 1 : 0;
      bad += isOut ? 1 : 0;
    }
    nBadEdges = (int)block_sum_d<NW>((double)bad, red);
    if (it == 2) robust = false;
    if (nInit < 10) break;  // optimizer.edges().size() < 10 (:1039)
  }
  if (tid == 0) {
    for (int k = 0; k < 4; ++k) poseIO[7 * f + k] = (float)T.q[k];
    for (int k = 0; k < 3; ++k) poseIO[7 * f + 4 + k] = (float)T.t[k];
    nInliers[f] = nInit - nBadEdges;
    if (stats) { stats[2 * f] = outerIts; stats[2 * f + 1] = trials; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: PoseOptimization rebuilt around what a lone frame's latency is made of (the launch is one frame's latency for 1 .. 256 frames).
//  * The active edges (map point present, not an outlier of the previous round) are COMPACTED, in feature order, at the start of each of
//    the four rounds, and every worker thread keeps its <= PO2_EPT edges in registers for the round: no global load inside the LM loop, and
//    the edge-order sums run over the active edges only (a skipped edge contributed an exact +0.0: same bits) — ~600 of 1200 features.
//  * ONE pass over the edges per LM trial instead of two.  g2o evaluates the errors twice at every accepted state: once for the trial's
//    chi2 (optimization_algorithm_levenberg.cpp:118-120) and once more, with the Jacobians, when the next iteration builds H and b
//    (:77-84).  Here the trial's pass speculatively produces H, b AND chi2 at the trial state; accepted (the common case), the next
//    iteration takes them as they are — the same expressions on the same inputs, so the same bits — and a rejected trial keeps the H, b of
//    the state it falls back to, as g2o does.
//  * In the edge-order mode wave 0 owns the 28 ordered sums and the 6 x 6 solve; waves 1 .. 7 compute the edges.  A stage is 448 edges:
//    while wave 0 adds stage s, the workers already compute stage s + 1.
//  * The solve, exp and pose update run on wave 0 only; the other waves pick the new pose up from LDS (they used to repeat all of it).
#ifndef MORB_PO2_EPT
#define MORB_PO2_EPT 4   // edges a worker thread keeps in registers per round
#endif
constexpr int PO2_NT = 512, PO2_NW = PO2_NT / 64, PO2_EPT = MORB_PO2_EPT;
// edges per stage of the edge-order mode: every wave but the summing one(s) computes edges
__host__ __device__ constexpr int po2_stage(bool mfma) { return PO2_NT - 64 * (mfma ? 2 : 1); }   // (matrix-core chain: waves 0 and 1 carry 16 + 12 sums)
// matrix-core chain: the FIRST stage is computed by all eight waves (the summing waves have nothing to add yet) and holds PO2_NT edges
__host__ __device__ constexpr int po2_rows(bool mfma) { return mfma ? PO2_NT : po2_stage(false); }   // rows of the contribution buffer
#ifndef MORB_PO2_SKIP_ROUNDS
#define MORB_PO2_SKIP_ROUNDS 1   // a round that would repeat the one before it bit for bit is not run
#endif
#ifndef MORB_PO2_FIRST_PREVIEW
#define MORB_PO2_FIRST_PREVIEW 2   // first trials are previewed after a rejection: 1 in this round, 2 in this call (measured best: 318 k against 312 / 317 k frames/s), 3 always
#endif
#ifndef MORB_PO2_SPEC
#define MORB_PO2_SPEC 1      // matrix-core chain: an iteration's first solve carries the nine trials that can follow it (one lambda per lane)
#endif
#ifndef MORB_PO2_PREVIEW
#define MORB_PO2_PREVIEW 1   // ... and a trial that follows a rejection is first judged by a tree sum of its chi2 (a rigorous "certainly rejected" test)
#endif
// frames beyond the registers' stages (PO2_NT + (PO2_EPT - 1) x stage edges) read the further edges again in every pass; the list of active features
// (two bytes per feature) has to fit in LDS beside the contribution buffer
constexpr int PO2_MAX_CAP = 8192;

struct PoEdge { float o[3], X[3], info; int right; };

// (OUT: double (&)[28] registers, or a pointer to the edge's row of the LDS buffer — the first stage writes every entry as soon as it exists, so the
// 28 x 8 bytes x 448 edges do not arrive at the LDS together at the end of the stage: the write port moves 128 bytes per cycle)
template <bool FISH, class OUT>
__device__ __forceinline__ void po2_contrib(const Cam& cam, const Rig& rig, const SE3& P, const SE3& Pr, const PoEdge& e, bool robust,
                                            double deltaMono, double deltaStereo, OUT&& con) {
  const double X[3] = {(double)e.X[0], (double)e.X[1], (double)e.X[2]};
  double xc[3], err[3], Jp[18], w = 1.0;
  bool st;
  const double info = (double)e.info;
  const bool right = FISH && e.right;
  double c = pose_edge_error<FISH>(cam, rig, P, Pr, right, X, e.o, info, err, st, xc);
  if (robust) c = huber(st ? deltaStereo : deltaMono, c, &w);
  con[27] = c;
  pose_edge_jac<FISH>(cam, rig, right, st, xc, Jp);
  // g2o's own expressions (base_unary_edge.hpp:54-66): omega_r = -Omega e (then * rho'), b += J^T omega_r, H += J^T (rho' Omega) J
  const double wr[3] = {-info * err[0] * w, -info * err[1] * w, -info * err[2] * w};
  const double wo = w * info;
  // Entries of Jp that are zero BY CONSTRUCTION are left out of the sums: pinhole (jac_pose, unary): Jp[0][4], Jp[1][3], Jp[2][4] (and a mono
  // edge's whole third row, which a stereo lane of the same wave needs); KB8: the third row.  Their products are exact zeros, x + 0 = x, and a
  // sum that starts with its first term instead of 0.0 + term differs at most in the SIGN of a zero — which the edge-order sums (they start at
  // +0.0 and can never reach -0.0) and the tree sums do not see.  63 -> 45 products for H, 18 -> 15 for b: a sixth of the edge's FP64 instructions.
  auto nz = [](int k, int r) { return FISH ? k < 2 : !((k == 0 && r == 4) || (k == 1 && r == 3) || (k == 2 && r == 4)); };
  int q = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double bb = 0;
    bool first = true;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (nz(k, r)) { const double t = Jp[k * 6 + r] * wr[k]; bb = first ? t : bb + t; first = false; }
    con[21 + r] = bb;
#pragma unroll
    for (int cc = 0; cc <= r; ++cc) {   // the LOWER triangle, (J_r w Omega) J_c as Eigen forms it: LinearSolverDense's LDLT reads that triangle
      double h = 0;
      bool firstH = true;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (nz(k, r) && nz(k, cc)) { const double t = Jp[k * 6 + r] * wo * Jp[k * 6 + cc]; h = firstH ? t : h + t; firstH = false; }
      con[q++] = h;
    }
  }
}

#ifdef MORB_PO_CYCLES   // developer build (tools/ab_build.py): thread 0 of frame 0 adds up where its cycles go
__device__ unsigned long long g_po2Cyc[8];
extern "C" int morb_po2_cycles(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_po2Cyc), sizeof(unsigned long long) * 8); const unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_po2Cyc), z, sizeof z); return 0; }
#define PO2_T0(v) const long long v = clock64()
#define PO2_ADD(slot, v) do { if (f == 0 && tid == 0) g_po2Cyc[slot] += (unsigned long long)(clock64() - v); } while (0)
#define PO2_CNT(slot) do { if (f == 0 && tid == 0) g_po2Cyc[slot] += 1; } while (0)
#define PO2_ADD_T(slot, v, t) do { if (f == 0 && tid == (t)) g_po2Cyc[slot] += (unsigned long long)(clock64() - v); } while (0)   // (timed by thread t)
#else
#define PO2_ADD_T(slot, v, t)
#define PO2_T0(v)
#define PO2_ADD(slot, v)
#define PO2_CNT(slot)
#endif
template <bool FISH, bool ORDERED, bool MFMA, bool BIG>
__global__ __launch_bounds__(PO2_NT) void k_pose_opt2(int cap, const int* __restrict__ count, const uint8_t* __restrict__ hasMP,
                                                      const float* __restrict__ obs, const float* __restrict__ invSigma2,
                                                      const float* __restrict__ Xw, Cam cam, Rig rig, const int* __restrict__ nLeft,
                                                      float* __restrict__ poseIO, uint8_t* __restrict__ outlier,
                                                      int* __restrict__ nInliers, int* __restrict__ stats) {
  constexpr int NT = PO2_NT, NW = PO2_NW;
  constexpr int NCW = MFMA ? 2 : 1;             // waves that carry the ordered sums (wave 0 also solves)
  constexpr int W0 = ORDERED ? 64 * NCW : 0;    // first worker thread
  constexpr int NWORK = NT - W0;                // edges per stage
  static_assert(!ORDERED || NWORK == po2_stage(MFMA), "stage size");
  extern __shared__ __align__(16) uint8_t po2Raw[];
  constexpr int ROWS = po2_rows(MFMA);          // rows of the contribution buffer
  constexpr int S0 = ORDERED && MFMA ? NT : NWORK;   // edges of the first stage (matrix-core chain: every wave computes, the summing ones included)
  double* sC = reinterpret_cast<double*>(po2Raw);                                          // ORDERED: [ROWS][PO_PITCH] contributions of a stage
  uint16_t* actList = reinterpret_cast<uint16_t*>(po2Raw + (ORDERED ? (size_t)ROWS * PO_PITCH * 8 + 32 : 0));   // [cap] active features, in order

  __shared__ double red[NW];
  __shared__ double sH[NW][28];
  __shared__ double sTot[2][28];     // H (lower triangle 0 .. 20), b (21 .. 26), robust chi2 (27) at the state last built / at the trial state
  __shared__ double sKeep[3][7];     // uniform poses that would otherwise sit in every thread's registers: T0, Teval, the trial's backup
  __shared__ double sSpec[10][8];    // trial poses + scales of an iteration: slot q = its trial q (if trials 0 .. q - 1 are rejected)
  __shared__ int sSpecFlag[10];
  __shared__ double sChiA[2][NW];    // the waves' partial sums of a trial's chi2 preview, by trial parity
  __shared__ int sWaveCnt[NW];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = min(count ? count[f] : cap, cap);
  const size_t base = (size_t)f * cap;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  // this thread's row in a stage (-1: not a worker)
  const int wrow = tid >= W0 ? tid - W0 : -1;
  // Speculation (matrix-core chain): a rejected trial is followed by a trial from the SAME state with lambda * ni, the next one with that * 2 ni, ... —
  // everything the solves of an iteration's up to ten trials need is known when its first trial is solved.  Wave 0's solve is uniform code, so its
  // lanes carry all ten at once: lane g the trial with lambda_g (the bookkeeping's own `lambda *= ni; ni *= 2`, g times: the same bits).  The ~5 k
  // cycles of LDL^T + exp + pose update are paid once per iteration instead of once per trial (more than half of all trials follow a rejection: near
  // convergence g2o's LM rejects its way up in lambda, up to ten trials per iteration).
  const bool spec = ORDERED && MFMA && MORB_PO2_SPEC;
  const int s0 = S0, nwork = NWORK;   // edges of the first / of a later stage
  // the edge thread `tid` computes in stage s (its row of the stage, the stage's first edge)
  auto stage_row = [&](int s) { return ORDERED && (s > 0 || !MFMA) ? (wrow < nwork ? wrow : -1) : (tid < s0 ? tid : -1); };
  auto stage_base = [&](int s) { return s == 0 ? 0 : s0 + (s - 1) * nwork; };

  PO2_T0(tAll);
#ifdef MORB_PO_FRAME_CYCLES
  const long long tFrame0 = clock64();
  int nCertain = 0;
#endif
  int nInit = 0;
  for (int i = tid; i < n; i += NT) {
    if (hasMP[base + i]) { ++nInit; outlier[base + i] = 0; }
  }
  nInit = (int)block_sum_d<NW>((double)nInit, red);
  if (nInit < 3) {  // Optimizer.cc:951
    if (tid == 0) { nInliers[f] = 0; if (stats) { stats[2 * f] = 0; stats[2 * f + 1] = 0; } }
    return;
  }
  SE3 T = se3_from_float(poseIO + 7 * f);
  auto put = [&](int slot, const SE3& P) { if (tid == 0) { for (int k = 0; k < 4; ++k) sKeep[slot][k] = P.q[k]; for (int k = 0; k < 3; ++k) sKeep[slot][4 + k] = P.t[k]; } };
  auto get = [&](int slot) { SE3 P; for (int k = 0; k < 4; ++k) P.q[k] = sKeep[slot][k]; for (int k = 0; k < 3; ++k) P.t[k] = sKeep[slot][4 + k]; return P; };
  put(0, T); put(1, T);     // T0, Teval (read after later barriers)
  bool robust = true;
  int nBadEdges = 0, outerIts = 0, trials = 0;
  const int nL = FISH ? nLeft[f] : n;   // features >= nL are right-camera observations (fisheye rig)
  PoEdge ed[PO2_EPT];
  int nAct = 0;
  bool sawReject = MORB_PO2_FIRST_PREVIEW == 3;

  auto load_edge = [&](int i, PoEdge& e) {
    e.o[0] = obs[(base + i) * 3]; e.o[1] = obs[(base + i) * 3 + 1]; e.o[2] = obs[(base + i) * 3 + 2];
    e.X[0] = Xw[(base + i) * 3]; e.X[1] = Xw[(base + i) * 3 + 1]; e.X[2] = Xw[(base + i) * 3 + 2];
    e.info = invSigma2[base + i];
    e.right = i >= nL;
  };
  // The solve of an iteration's trials, by wave 0: (H + lam I) x = b of sTot[src], Tn = exp(x) Tb, scale = x . (lam x + b) + 1e-3.  Lane g < nTrials solves
  // trial g — lam_g = lam after g rejections — and writes sSpec[g] / sSpecFlag[g]; the other lanes repeat lane nTrials - 1's arithmetic.
  auto solve_trials = [&](int src, double lam, double niv, const SE3& Tb, int nTrials) {
    const int g = min(lane, nTrials - 1);
    for (int j = 0; j < g; ++j) { lam *= niv; niv *= 2; }
    double H[36], b[6], x[6];
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = 0; cc <= r; ++cc) { const double v = sTot[src][q++]; H[r * 6 + cc] = v; H[cc * 6 + r] = v; }
#pragma unroll
    for (int r = 0; r < 6; ++r) { b[r] = sTot[src][21 + r]; H[r * 6 + r] += lam; x[r] = 0; }
    const bool ok2 = ldlt6(H, b, x);
    const SE3 Tn = se3_mul(se3_exp(x), Tb);
    double scale = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) scale += x[r] * (lam * x[r] + b[r]);
    scale += 1e-3;
    if (lane < nTrials) {
      double* o8 = sSpec[lane];
      for (int k = 0; k < 4; ++k) o8[k] = Tn.q[k];
      for (int k = 0; k < 3; ++k) o8[4 + k] = Tn.t[k];
      o8[7] = scale;
      sSpecFlag[lane] = ok2 ? 1 : 0;
    }
  };
  // H, b, chi2 of the active edges at pose P -> sTot[buf]
  auto pass = [&](const SE3& P, int buf) {
    PO2_T0(tp); PO2_CNT(4);
    const SE3 Pr = FISH ? se3_mul(rig.Trl, P) : P;
    if (ORDERED) {
      double tot = 0;   // VALU chain: lanes 0 .. 27 of wave 0 hold entry `lane`; matrix-core chain: the accumulator (D layout) of waves 0 and 1
      const int nAct16 = (nAct + 15) & ~15;
#pragma unroll
      for (int s = 0; s < PO2_EPT; ++s) {   // (unrolled: ed[s] must stay in registers)
        const int e0 = stage_base(s), row = stage_row(s);
        if (e0 >= nAct) break;
        double con[28];
        const int e = e0 + row;
        const bool mine = row >= 0 && e < nAct;
        // (The first stage needs no barrier in front of its LDS writes — the previous pass ended with one — and writes from inside the edge math.
        // KB8 rig: through registers in every stage — writing from inside its longer edge math costs that kernel 400 bytes of scratch memory.)
        if (mine) {
          if (s == 0 && !FISH) po2_contrib<FISH>(cam, rig, P, Pr, ed[s], robust, deltaMono, deltaStereo, sC + row * PO_PITCH);
          else po2_contrib<FISH>(cam, rig, P, Pr, ed[s], robust, deltaMono, deltaStereo, con);   // (beside the sums of stage s - 1)
        }
        if (s == 0) PO2_ADD_T(3, tp, NT - 64);   // (a worker's edge math of the first stage)
        // Stage s - 1 has been added.  (Measured and not kept: progress words published by the summing waves, so that a later stage's writers wait for
        // their own row instead of this barrier — 0.404 -> 0.435 ms per launch: writes that arrive while the sums run delay the sums' LDS reads.)
        if (s > 0) __syncthreads();
        if (mine && (s > 0 || FISH)) {
#pragma unroll
          for (int k = 0; k < 28; ++k) sC[row * PO_PITCH + k] = con[k];
        } else if (!mine && MFMA && row >= 0 && e < nAct16) {   // the matrix core takes four rows at a time: the last batch's missing rows add 0.0
#pragma unroll
          for (int k = 0; k < 28; ++k) sC[row * PO_PITCH + k] = 0.0;
        }
        __syncthreads();
        PO2_T0(tch);
        const int m = min(s == 0 ? s0 : nwork, (MFMA ? nAct16 : nAct) - e0);
        if (MFMA) { if (wv < 2) ordered_add_mfma1<PO_PITCH, ROWS>(tot, sC + 16 * wv, lane, m); }
        else if (tid < 28) tot = ordered_add_pipe<PO_PITCH, ROWS>(tot, sC + tid, m);
        PO2_ADD(7, tch);
      }
      // frames with more active edges than the threads' registers hold (PO2_EPT stages): the further stages read their edge again in every pass
      if (BIG) for (int s = PO2_EPT; stage_base(s) < nAct; ++s) {
        const int e0 = stage_base(s), row = stage_row(s);
        double con[28];
        const int e = e0 + row;
        const bool mine = row >= 0 && e < nAct;
        if (mine) {
          PoEdge ee;
          load_edge(actList[e], ee);
          po2_contrib<FISH>(cam, rig, P, Pr, ee, robust, deltaMono, deltaStereo, con);
        }
        __syncthreads();
        if (mine) {
#pragma unroll
          for (int k = 0; k < 28; ++k) sC[row * PO_PITCH + k] = con[k];
        } else if (MFMA && row >= 0 && e < nAct16) {
#pragma unroll
          for (int k = 0; k < 28; ++k) sC[row * PO_PITCH + k] = 0.0;
        }
        __syncthreads();
        const int m = min(nwork, (MFMA ? nAct16 : nAct) - e0);
        if (MFMA) { if (wv < 2) ordered_add_mfma1<PO_PITCH, ROWS>(tot, sC + 16 * wv, lane, m); }
        else if (tid < 28) tot = ordered_add_pipe<PO_PITCH, ROWS>(tot, sC + tid, m);
      }
      if (MFMA) {
        const int c16 = 4 * ((lane >> 2) & 3) + (lane >> 4);
        if (wv < 2 && (lane & 3) == 0 && 16 * wv + c16 < 28) sTot[buf][16 * wv + c16] = tot;
      } else if (tid < 28) sTot[buf][tid] = tot;
      __syncthreads();
      PO2_ADD(2, tp);
    } else {
      double acc[28];
#pragma unroll
      for (int k = 0; k < 28; ++k) acc[k] = 0;
#pragma unroll
      for (int s = 0; s < PO2_EPT; ++s) {
        if (s * NWORK + tid < nAct) {
          double con[28];
          po2_contrib<FISH>(cam, rig, P, Pr, ed[s], robust, deltaMono, deltaStereo, con);
#pragma unroll
          for (int k = 0; k < 28; ++k) acc[k] += con[k];
        }
      }
      if (BIG) for (int e = PO2_EPT * NWORK + tid; e < nAct; e += NWORK) {   // (larger frames: the edge is read again in every pass)
        PoEdge ee;
        load_edge(actList[e], ee);
        double con[28];
        po2_contrib<FISH>(cam, rig, P, Pr, ee, robust, deltaMono, deltaStereo, con);
#pragma unroll
        for (int k = 0; k < 28; ++k) acc[k] += con[k];
      }
      __syncthreads();                         // (sH / sTot[buf] of an earlier pass have been read)
      wave_sum28_to(acc, sH[wv], lane);
      __syncthreads();
      if (tid < 28) { double t = sH[0][tid]; for (int w = 1; w < NW; ++w) t += sH[w][tid]; sTot[buf][tid] = t; }
      __syncthreads();
      PO2_ADD(2, tp);
    }
  };

  // A round whose classification changes no flag is followed by an IDENTICAL round (same active edges, same start pose, same robust kernel: rounds 1 - 3 of
  // the reference's four all use the Huber kernel; the computation is deterministic) that would end in the same pose and the same flags: it is not run,
  // its iterations and trials are counted.  (Outlier sets usually settle after the first round or two; the fourth round, without the kernel, always runs.)
  // (decided at the END of a round — the loop counter jumps over the rounds that would repeat it: no second path around the round's body)
  for (int it = 0; it < 4; ++it) {
    const int its0 = outerIts, trials0 = trials;
    // ---- the round's active edges, in feature order; each worker thread takes its edges into registers
    PO2_T0(tc);
    __syncthreads();
    T = get(0);  // vSE3->setEstimate(pFrame->GetPose()) (:962-964)
    nAct = 0;
    for (int c0 = 0; c0 < n; c0 += NT) {
      const int i = c0 + tid;
      const bool a = i < n && hasMP[base + i] && !outlier[base + i];
      const unsigned long long m = __ballot(a);
      if (lane == 0) sWaveCnt[wv] = __popcll(m);
      __syncthreads();
      int off = nAct, totc = 0;
      for (int w = 0; w < NW; ++w) { const int c = sWaveCnt[w]; if (w < wv) off += c; totc += c; }
      if (a) actList[off + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
      nAct += totc;
      __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < PO2_EPT; ++s) {
      const int e = stage_base(s) + stage_row(s);
      if (stage_row(s) >= 0 && e < nAct) {
        load_edge(actList[e], ed[s]);
      }
    }
    PO2_ADD(5, tc);
    // ---- optimizer.optimize(10) ----
    int cur = 0;
    pass(T, cur);
    double lambda = 0, ni = 2;
#if MORB_PO2_FIRST_PREVIEW == 1
    sawReject = false;   // (per round: a round's first iterations accept their first trials; after its first rejection most first trials are rejected too)
#endif
    int nBad = 0;
    for (int iter = 0; iter < 10; ++iter) {
      ++outerIts;
      double currentChi = sTot[cur][27];
      const double iniChi = currentChi;
      if (iter == 0) {  // computeLambdaInit (tau = 1e-5)
        double m = 0;
        int q = 0;
        for (int r = 0; r < 6; ++r) { q += r; m = fmax(fabs(sTot[cur][q + r]), m); }   // diagonal entries of the packed lower triangle
        lambda = 1e-5 * m; ni = 2; nBad = 0;
      }
      double rho = 0;
      int qmax = 0;
      do {
        PO2_T0(ts);
        // the iteration's first trial (every trial without the speculation): the solve; a trial after a rejection finds its pose in its slot
        if (qmax == 0 || !spec) {
          if (qmax == 0) put(2, T);           // backup (thread 0 writes it, its own wave reads it back below: LDS operations of a wave execute in order)
          if (wv == 0) solve_trials(cur, lambda, ni, get(2), spec ? 10 : 1);   // (the slots' last readers are behind the barriers of the pass before)
          __syncthreads();
        }
        const int sl = spec ? qmax : 0;
        for (int k = 0; k < 4; ++k) T.q[k] = sSpec[sl][k];
        for (int k = 0; k < 3; ++k) T.t[k] = sSpec[sl][4 + k];
        const double scale = sSpec[sl][7];
        const bool ok2 = sSpecFlag[sl] != 0;
        PO2_ADD(1, ts);
        put(1, T);           // Teval
        // A trial that follows a rejection is usually rejected too (g2o's LM climbs in lambda near convergence), and a rejected trial leaves nothing
        // behind but the decision rho <= 0: its H, b and even its chi2 are dropped.  The decision is chi2(trial) > chi2(current state) — sums of
        // non-negative terms — so a PREVIEW decides most of them rigorously: every thread adds the robustified chi2 of its own edges (errors only: a
        // third of the edge math), a tree sum over the workgroup, and since two floating-point sums of the same n non-negative terms differ by less
        // than 2 n 2^-53 of their value (any order, any association), `preview (1 - 8 n 2^-53) > currentChi` implies that the edge-order sum is larger
        // too: rejected, for certain, without the edge-order sums, the Jacobians or the LDS hand-over (measured on the oracle's traces: 90 % of the
        // rejections, half of all trials).  Otherwise the trial runs its pass as before.  (Its pose has been waiting since the iteration's first trial.)
        bool certainlyRejected = false;
        if (spec && MORB_PO2_PREVIEW && (qmax > 0 || sawReject) && ok2 && scale > 0) {   // (an iteration's first trial too once this round has rejected one)
          const int par = trials & 1;
          double part = 0;
          {
            const SE3 Tr = FISH ? se3_mul(rig.Trl, T) : T;
#pragma unroll
            for (int s = 0; s < PO2_EPT; ++s) {
              const int row = stage_row(s), e = stage_base(s) + row;
              if (row >= 0 && e < nAct) {
                const double X[3] = {(double)ed[s].X[0], (double)ed[s].X[1], (double)ed[s].X[2]};
                double xc[3], err[3], w;
                bool st;
                double c = pose_edge_error<FISH>(cam, rig, T, Tr, FISH && ed[s].right, X, ed[s].o, (double)ed[s].info, err, st, xc);
                if (robust) c = huber(st ? deltaStereo : deltaMono, c, &w);
                part += c;
              }
            }
          }
          if (BIG) {
            const SE3 Tr = FISH ? se3_mul(rig.Trl, T) : T;
            for (int s = PO2_EPT; stage_base(s) < nAct; ++s) {   // (the stages whose edges are not in registers)
              const int row = stage_row(s), e = stage_base(s) + row;
              if (row >= 0 && e < nAct) {
                PoEdge ee;
                load_edge(actList[e], ee);
                const double X[3] = {(double)ee.X[0], (double)ee.X[1], (double)ee.X[2]};
                double xc[3], err[3], w;
                bool st;
                double c = pose_edge_error<FISH>(cam, rig, T, Tr, FISH && ee.right, X, ee.o, (double)ee.info, err, st, xc);
                if (robust) c = huber(st ? deltaStereo : deltaMono, c, &w);
                part += c;
              }
            }
          }
          part = wave_sum_d(part);
          if (lane == 0) sChiA[par][wv] = part;
          __syncthreads();
          double tt = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) tt += sChiA[par][w];
          certainlyRejected = isfinite(tt) && tt * (1.0 - 8.0 * (double)nAct * 0x1p-53) > currentChi;
        }
        double tempChi = 1.7976931348623157e308;
        if (!certainlyRejected) {
          pass(T, cur ^ 1);
          tempChi = sTot[cur ^ 1][27];
          if (!ok2) tempChi = 1.7976931348623157e308;
        }
#ifdef MORB_PO_FRAME_CYCLES
        nCertain += certainlyRejected ? 1 : 0;
#endif
        rho = certainlyRejected ? -1.0 : (currentChi - tempChi) / scale;   // (a certainly rejected trial: only the sign of rho is ever read)
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - cube_rn(2 * rho - 1);
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          cur ^= 1;           // H, b at the accepted state are already there
        } else {
          lambda *= ni;
          ni *= 2;
          T = get(2);         // sTot[cur] still holds H, b of this state
          sawReject = true;
        }
        ++qmax; ++trials;
      } while (rho < 0 && qmax < 10);
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) break;
    }
    // ---- classify (:966-1037): inlier edges keep the error of the LAST evaluated state (Teval, which is a
    // rejected trial when the LM loop ended on a failure), current outliers are re-evaluated at the final pose
    int bad = 0;
    PO2_T0(tk);
    __syncthreads();
    const SE3 Teval = get(1);
    const SE3 TrFin = FISH ? se3_mul(rig.Trl, T) : T, TrEval = FISH ? se3_mul(rig.Trl, Teval) : Teval;
    for (int i = tid; i < n; i += NT) {
      if (!hasMP[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      double xc[3], err[3];
      bool st;
      const bool wasOut = outlier[base + i] != 0;
      const SE3& Pc = wasOut ? T : Teval;
      const float chi2 = (float)pose_edge_error<FISH>(cam, rig, Pc, wasOut ? TrFin : TrEval, FISH && i >= nL, X, o,
                                                      (double)invSigma2[base + i], err, st, xc);
      const bool isOut = chi2 > (st ? 7.815f : 5.991f);
      outlier[base + i] = isOut ? 1 : 0;
      bad += (isOut ? 1 : 0) + (isOut != wasOut ? 65536 : 0);   // (outliers | flags that changed: two counts in one exact sum)
    }
    {
      const int packed = (int)block_sum_d<NW>((double)bad, red);
      nBadEdges = packed & 0xFFFF;
      if (MORB_PO2_SKIP_ROUNDS && (packed >> 16) == 0 && it < 2 && nInit >= 10) {   // rounds it + 1 .. 2 would repeat this one (fewer than 10 edges: one round only, :1039)
        const int k = 2 - it;
        outerIts += k * (outerIts - its0); trials += k * (trials - trials0);
        it = 2;
      }
    }
    PO2_ADD(6, tk);
    if (it == 2) robust = false;
    if (nInit < 10) break;  // optimizer.edges().size() < 10 (:1039)
  }
  if (tid == 0) {
    for (int k = 0; k < 4; ++k) poseIO[7 * f + k] = (float)T.q[k];
    for (int k = 0; k < 3; ++k) poseIO[7 * f + 4 + k] = (float)T.t[k];
    nInliers[f] = nInit - nBadEdges;
#ifdef MORB_PO_FRAME_CYCLES   // developer build: the frame's kernel time (kilocycles) and its certain rejections instead of iterations / trials
    if (stats) { stats[2 * f] = (int)((clock64() - tFrame0) / 1000); stats[2 * f + 1] = trials * 100 + nCertain; }
#else
    if (stats) { stats[2 * f] = outerIts; stats[2 * f + 1] = trials; }
#endif
  }
  PO2_ADD(0, tAll);
}
// registers hold the edges of PO2_EPT stages; larger frames run the BIG instantiation (further stages read their edge again in every pass: kept out of
// the common kernels, where the extra code costs registers — 3 % on the tracking chain)
__host__ __device__ constexpr int po2_reg_cap(bool ordered, bool mfma) {
  return !ordered ? PO2_EPT * PO2_NT : mfma ? PO2_NT + (PO2_EPT - 1) * po2_stage(true) : PO2_EPT * po2_stage(false);
}
template <bool FISH, bool ORDERED, bool MFMA, bool BIG>
static int launch_pose_opt2_as(int nframes, hipStream_t st, size_t lds, int cap, const int* d_count, const uint8_t* d_hasMP, const float* d_obs,
                               const float* d_invSigma2, const float* d_Xw, const Cam& cam, const Rig& rig, const int* d_nLeft, float* d_pose,
                               uint8_t* d_outlier, int* d_nInliers, int* d_stats) {
  if (lds > 48 * 1024)
    MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_opt2<FISH, ORDERED, MFMA, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_pose_opt2<FISH, ORDERED, MFMA, BIG>), dim3(nframes), dim3(PO2_NT), lds, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig,
                     d_nLeft, d_pose, d_outlier, d_nInliers, d_stats);
  return MORB_OK;
}
// false: this mode / size has no k_pose_opt2 form (the caller launches k_pose_opt)
static bool pose_opt2_covers(bool ordered, bool mfmaChain, int cap) {
  return cap <= PO2_MAX_CAP && (cap <= po2_reg_cap(ordered, mfmaChain) || !ordered || mfmaChain);   // (no BIG form of the vector-chain mode)
}
template <bool FISH>
static int launch_pose_opt2(bool ordered, bool mfmaChain, int nframes, hipStream_t st, int cap, const int* d_count, const uint8_t* d_hasMP, const float* d_obs,
                            const float* d_invSigma2, const float* d_Xw, const Cam& cam, const Rig& rig, const int* d_nLeft, float* d_pose,
                            uint8_t* d_outlier, int* d_nInliers, int* d_stats) {
  const size_t listBytes = ((size_t)cap * 2 + 15) & ~(size_t)15;
  const bool big = cap > po2_reg_cap(ordered, mfmaChain);
  const size_t lds = ordered ? (size_t)po2_rows(mfmaChain) * PO_PITCH * 8 + 32 + listBytes : listBytes;   // (+32: the second accumulator's lanes 12 .. 15 read past the last row)
#define MORB_PO2_GO(O, M, B) return launch_pose_opt2_as<FISH, O, M, B>(nframes, st, lds, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig, d_nLeft, d_pose, d_outlier, d_nInliers, d_stats)
  if (ordered && mfmaChain) { if (big) MORB_PO2_GO(true, true, true); else MORB_PO2_GO(true, true, false); }
  if (ordered) MORB_PO2_GO(true, false, false);
  if (big) MORB_PO2_GO(false, false, true);
  MORB_PO2_GO(false, false, false);
#undef MORB_PO2_GO
}

// =====================================================================================================
// LocalBundleAdjustment: one 1024-thread workgroup per problem
// =====================================================================================================
struct BaDev {
  int nKF, nMP, nE, nFree, P;           // P = 6 * nFree
  const int* kfCol;                     // [nKF] column of a free keyframe, -1 if fixed
  const int *eKF, *eMP;                 // [nE]
  const float *eObs, *eInfo;            // [nE][3], [nE]
  const int *mpStart, *mpEdges;         // CSR by map point
  const int *kfStart, *kfEdges;         // CSR by keyframe
  int nPairs;                           // upper-triangular block pairs (i1 <= i2) of the reduced system that occur
  const int *pairBlock;                 // [nPairs] i1 * nFree + i2
  const int *pairStart;                 // [nPairs + 1]
  const int2 *pairEntries;              // (e1, e2): two observations of one map point, col(e1) = i1, col(e2) = i2
  double *pose, *poseBk, *poseEval;     // [nKF][7]
  double *pt, *ptBk, *ptEval;           // [nMP][3]
  double *Hpp;                          // [nFree][36]
  double *Hll, *Dinv;                   // [nMP][9]
  double *Hpl;                          // [nE][18]
  double *b, *x;                        // [P + 3 nMP]
  double *HsG;                          // [P*P] global fallback for the reduced system
  float *poseIO, *ptIO;                 // results (float)
  uint8_t* erase;                       // [nE]
  int* stats;                           // [2]
  const int* stop;                      // device-visible abort flag (may be NULL)
  Cam cam;
  const struct Rig* rig;                // fisheye rig (KB8 cameras + mTrl) or NULL; its edges carry obs[2] = -2 (left camera) / -3 (right, "ToBody")
  double userLambda;
  // grid mode (one launch per LM phase across the whole chip)
  int nChunks;                          // keyframe edge lists cut into chunks of <= 64 edges (one wave each)
  const int *chunkKF, *chunkStart, *chunkEnd;   // [nChunks]
  const int *kfChunkStart;              // [nKF + 1]
  double *kfPart;                       // [nChunks][27]
  double *redPart;                      // [2][redBlocks] block partial sums (chi2, scale)
  double *scal;                         // [8] device scalars: chi2, scale, ok, maxdiag
  // device-side LM control (k_g_lm_*): the state of optimization_algorithm_levenberg.cpp's loop, and its mirror in mapped host memory
  double *lmd;                          // [4] LMD_*: currentChi, lambda, ni, iniChi
  int *lmi;                             // [16] LM_*
  int *lmHost;                          // [4] mapped host memory: decided trials, done, outer iterations, trials
  int *kfTicket;                        // [nKF] chunks of the keyframe that have delivered their partial blocks (k_g_build)
  int dupPairs;                         // some (keyframe, landmark) pair carries two edges (fisheye rig: both cameras)
  // Schur complement on the FP64 matrix cores (schur_mfma.h): dense K-major operands, partial products, block directory
  double *sW, *sWD, *sPart;
  const int2* sBlocks;
  const int* sBlkIndex;
  int sMp, sNb, sNblk, sNsplit;
};

__device__ __forceinline__ SE3 load_se3(const double* p) {
  SE3 s;
  for (int i = 0; i < 4; ++i) s.q[i] = p[i];
  for (int i = 0; i < 3; ++i) s.t[i] = p[4 + i];
  return s;
}
__device__ __forceinline__ void store_se3(double* p, const SE3& s) {
  for (int i = 0; i < 4; ++i) p[i] = s.q[i];
  for (int i = 0; i < 3; ++i) p[4 + i] = s.t[i];
}
__device__ __forceinline__ void inv3(const double* m, double* o) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

constexpr int BA_T = 512, BA_W = BA_T / 64;

__global__ __launch_bounds__(BA_T) void k_local_ba(const BaDev* __restrict__ probs, int useLds) {
  extern __shared__ double sHs[];  // reduced camera system when it fits
  __shared__ double red[BA_W];
  __shared__ int sFlag;
  const BaDev pb = probs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int P = pb.P, nMP = pb.nMP, nE = pb.nE, nKF = pb.nKF;
  double* Hs = useLds ? sHs : pb.HsG;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  const Cam cam = pb.cam;

  auto terminate = [&]() -> bool {
    if (!pb.stop) return false;
    __syncthreads();
    if (tid == 0) sFlag = __hip_atomic_load(pb.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-written, pinned
    __syncthreads();
    return sFlag != 0;
  };
  // errors at the current estimate -> robust chi2; remembers the evaluation state (for the final chi2 test)
  auto chi2All = [&]() -> double {
    for (int i = tid; i < nKF * 7; i += BA_T) pb.poseEval[i] = pb.pose[i];
    for (int i = tid; i < nMP * 3; i += BA_T) pb.ptEval[i] = pb.pt[i];
    double s = 0;
    for (int e = tid; e < nE; e += BA_T) {
      const SE3 T = load_se3(pb.pose + 7 * pb.eKF[e]);
      double xc[3], err[3], w;
      se3_map(T, pb.pt + 3 * pb.eMP[e], xc);
      const float* o = pb.eObs + 3 * e;
      const bool st = !(o[2] < 0);
      const double c = ba_edge_error(cam, pb.rig, st, xc, o, (double)pb.eInfo[e], err);
      s += huber(st ? deltaStereo : deltaMono, c, &w);
    }
    return block_sum_d<BA_W>(s, red);
  };

  if (terminate()) {  // Optimizer.cc:1355-1356
    if (tid == 0) { pb.stats[0] = 0; pb.stats[1] = 0; }
    return;
  }
  double lambda = 0, ni = 2;
  int nBad = 0, its = 0, trials = 0;
  for (int iter = 0; iter < 10; ++iter) {
    if (terminate()) break;
    ++its;
    double currentChi = chi2All();
    const double iniChi = currentChi;
    // ---- buildSystem ----
    // (1) per map point: Hll, bl
    for (int m = tid; m < nMP; m += BA_T) {
      double Hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
      const double* X = pb.pt + 3 * m;
      for (int k = pb.mpStart[m]; k < pb.mpStart[m + 1]; ++k) {
        const int e = pb.mpEdges[k];
        const SE3 T = load_se3(pb.pose + 7 * pb.eKF[e]);
        double xc[3], err[3], w, R[9], Jl[9];
        se3_map(T, X, xc);
        const float* o = pb.eObs + 3 * e;
        const bool st = !(o[2] < 0);
        const double info = (double)pb.eInfo[e];
        const double c = ba_edge_error(cam, pb.rig, st, xc, o, info, err);
        huber(st ? deltaStereo : deltaMono, c, &w);
        q_to_R(T.q, R);
        if (pb.rig) ba_edge_jac<true>(cam, pb.rig, st, xc, o, R, nullptr, Jl); else ba_edge_jac<false>(cam, pb.rig, st, xc, o, R, nullptr, Jl);   // (persistent mode: one kernel for both cameras)
        const double wo = w * info;
        for (int r = 0; r < 3; ++r) {
          double s = 0;
          _Pragma("unroll") for (int i = 0; i < 3; ++i) s += Jl[i * 3 + r] * (-info * err[i] * w);
          bl[r] += s;
          for (int cc = 0; cc < 3; ++cc) {
            double h = 0;
            _Pragma("unroll") for (int i = 0; i < 3; ++i) h += Jl[i * 3 + r] * wo * Jl[i * 3 + cc];
            Hl[r * 3 + cc] += h;
          }
        }
      }
      for (int k = 0; k < 9; ++k) pb.Hll[(size_t)m * 9 + k] = Hl[k];
      for (int k = 0; k < 3; ++k) pb.b[P + 3 * m + k] = bl[k];
    }
    // (2) per free keyframe (one wave each): Hpp, bp; per edge: Hpl
    for (int kf = wv; kf < nKF; kf += BA_W) {
      const int col = pb.kfCol[kf];
      if (col < 0) continue;
      const SE3 T = load_se3(pb.pose + 7 * kf);
      double R[9];
      q_to_R(T.q, R);
      double acc[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[k] = 0;
      for (int k = pb.kfStart[kf] + lane; k < pb.kfStart[kf + 1]; k += 64) {
        const int e = pb.kfEdges[k];
        double xc[3], err[3], w, Jp[18], Jl[9];
        se3_map(T, pb.pt + 3 * pb.eMP[e], xc);
        const float* o = pb.eObs + 3 * e;
        const bool st = !(o[2] < 0);
        const double info = (double)pb.eInfo[e];
        const double c = ba_edge_error(cam, pb.rig, st, xc, o, info, err);
        huber(st ? deltaStereo : deltaMono, c, &w);
        if (pb.rig) ba_edge_jac<true>(cam, pb.rig, st, xc, o, R, Jp, Jl); else ba_edge_jac<false>(cam, pb.rig, st, xc, o, R, Jp, Jl);
        const double wo = w * info;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double s = 0;
          _Pragma("unroll") for (int i = 0; i < 3; ++i) s += Jp[i * 6 + r] * (-info * err[i] * w);
          acc[21 + r] += s;
#pragma unroll
          for (int cc = r; cc < 6; ++cc) {
            double h = 0;
            _Pragma("unroll") for (int i = 0; i < 3; ++i) h += Jp[i * 6 + r] * wo * Jp[i * 6 + cc];
            acc[q++] += h;
          }
          for (int cc = 0; cc < 3; ++cc) {
            double h = 0;
            _Pragma("unroll") for (int i = 0; i < 3; ++i) h += Jp[i * 6 + r] * wo * Jl[i * 3 + cc];
            pb.Hpl[(size_t)e * 18 + r * 3 + cc] = h;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[k] = wave_sum_d(acc[k]);
      if (lane == 0) {
        int q = 0;
        for (int r = 0; r < 6; ++r)
          for (int cc = r; cc < 6; ++cc) { pb.Hpp[(size_t)col * 36 + r * 6 + cc] = acc[q]; pb.Hpp[(size_t)col * 36 + cc * 6 + r] = acc[q]; ++q; }
        for (int r = 0; r < 6; ++r) pb.b[6 * col + r] = acc[21 + r];
      }
    }
    __syncthreads();
    if (iter == 0) {  // computeLambdaInit
      if (pb.userLambda > 0) lambda = pb.userLambda;
      else {
        double m = 0;
        for (int i = tid; i < pb.nFree * 6; i += BA_T) m = fmax(m, fabs(pb.Hpp[(size_t)(i / 6) * 36 + (i % 6) * 7]));
        for (int i = tid; i < nMP * 3; i += BA_T) m = fmax(m, fabs(pb.Hll[(size_t)(i / 3) * 9 + (i % 3) * 4]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
        __syncthreads();
        if (lane == 0) red[wv] = m;
        __syncthreads();
        m = 0;
        for (int i = 0; i < BA_W; ++i) m = fmax(m, red[i]);
        lambda = 1e-5 * m;
      }
      ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      // push()
      for (int i = tid; i < nKF * 7; i += BA_T) pb.poseBk[i] = pb.pose[i];
      for (int i = tid; i < nMP * 3; i += BA_T) pb.ptBk[i] = pb.pt[i];
      // ---- BlockSolver::solve (Schur) with lambda on every diagonal (block_solver.hpp:354-480) ----
      // (a) per map point: Dinv = (Hll + lambda I)^-1
      for (int m = tid; m < nMP; m += BA_T) {
        double D[9], Di[9];
        for (int k = 0; k < 9; ++k) D[k] = pb.Hll[(size_t)m * 9 + k];
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3(D, Di);
        for (int k = 0; k < 9; ++k) pb.Dinv[(size_t)m * 9 + k] = Di[k];
      }
      for (int i = tid; i < P * P; i += BA_T) Hs[i] = 0;
      __syncthreads();
      // (b) one wave per block pair (i1 <= i2): Hschur(i1,i2) = [Hpp + lambda I] - sum_l (Hpl_i1 Dinv_l) Hpl_i2^T,
      //     entries summed in a fixed order (deterministic), mirrored into the lower triangle
      for (int bp = wv; bp < pb.nPairs; bp += BA_W) {
        const int i1 = pb.pairBlock[bp] / pb.nFree, i2 = pb.pairBlock[bp] % pb.nFree;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0;
        for (int k = pb.pairStart[bp] + lane; k < pb.pairStart[bp + 1]; k += 64) {
          const int2 en = pb.pairEntries[k];
          const double* B1 = pb.Hpl + (size_t)en.x * 18;
          const double* B2 = pb.Hpl + (size_t)en.y * 18;
          const double* Di = pb.Dinv + (size_t)pb.eMP[en.x] * 9;
          double BD[18];
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) BD[r * 3 + c] = B1[r * 3] * Di[c] + B1[r * 3 + 1] * Di[3 + c] + B1[r * 3 + 2] * Di[6 + c];
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[r * 6 + c] += BD[r * 3] * B2[c * 3] + BD[r * 3 + 1] * B2[c * 3 + 1] + BD[r * 3 + 2] * B2[c * 3 + 2];
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = wave_sum_d(acc[k]);
        if (lane < 36) {
          const int r = lane / 6, c = lane % 6;
          double a = 0;
#pragma unroll
          for (int k = 0; k < 36; ++k) if (k == lane) a = acc[k];
          double v = -a;
          if (i1 == i2) { v += pb.Hpp[(size_t)i1 * 36 + lane]; if (r == c) v += lambda; }
          Hs[(size_t)(6 * i1 + r) * P + 6 * i2 + c] = v;
          if (i1 != i2) Hs[(size_t)(6 * i2 + c) * P + 6 * i1 + r] = v;
        }
      }
      // (c) bschur = bp - sum Hpl Dinv bl, one wave per free keyframe over its CSR edge list
      for (int kf = wv; kf < nKF; kf += BA_W) {
        const int col = pb.kfCol[kf];
        if (col < 0) continue;
        double a6[6] = {0, 0, 0, 0, 0, 0};
        for (int k = pb.kfStart[kf] + lane; k < pb.kfStart[kf + 1]; k += 64) {
          const int e = pb.kfEdges[k];
          const int m = pb.eMP[e];
          const double* Di = pb.Dinv + (size_t)m * 9;
          const double* bl = pb.b + P + 3 * m;
          double db[3];
          for (int r = 0; r < 3; ++r) db[r] = Di[r * 3] * bl[0] + Di[r * 3 + 1] * bl[1] + Di[r * 3 + 2] * bl[2];
          const double* B1 = pb.Hpl + (size_t)e * 18;
          for (int r = 0; r < 6; ++r) a6[r] += B1[r * 3] * db[0] + B1[r * 3 + 1] * db[1] + B1[r * 3 + 2] * db[2];
        }
        for (int r = 0; r < 6; ++r) a6[r] = wave_sum_d(a6[r]);
        if (lane == 0) for (int r = 0; r < 6; ++r) pb.x[6 * col + r] = pb.b[6 * col + r] - a6[r];
      }
      __syncthreads();
      // (d) reduced system: right-looking LDL^T by the whole workgroup (LinearSolverEigen / SimplicialLDLT:
      //     fails only on a zero pivot), then the two triangular solves by wave 0
      if (tid == 0) sFlag = 1;
      __syncthreads();
      for (int j = 0; j < P; ++j) {
        const double d = Hs[(size_t)j * P + j];
        if (d == 0 || d != d) { if (tid == 0) sFlag = 0; break; }   // uniform: every thread reads the same d
        __syncthreads();
        // trailing update with the UNSCALED column: A_ik -= A_ij * A_kj / d  (i >= k > j), then scale column j
        const int nrem = P - j - 1;
        for (int t = tid; t < nrem * nrem; t += BA_T) {
          const int i = j + 1 + t / nrem, k = j + 1 + t % nrem;
          if (k <= i) Hs[(size_t)i * P + k] -= Hs[(size_t)i * P + j] * Hs[(size_t)k * P + j] / d;
        }
        __syncthreads();
        for (int i = j + 1 + tid; i < P; i += BA_T) Hs[(size_t)i * P + j] /= d;
      }
      __syncthreads();
      if (sFlag != 0 && wv == 0) {
        for (int j = 0; j < P; ++j) {           // forward: L y = b
          const double xj = pb.x[j];
          for (int i = j + 1 + lane; i < P; i += 64) pb.x[i] -= Hs[(size_t)i * P + j] * xj;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
        for (int i = lane; i < P; i += 64) pb.x[i] /= Hs[(size_t)i * P + i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int j = P - 1; j >= 0; --j) {      // backward: L^T x = y
          const double xj = pb.x[j];
          for (int i = lane; i < j; i += 64) pb.x[i] -= Hs[(size_t)j * P + i] * xj;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
      const bool ok2 = sFlag != 0;
      if (ok2) {
        // xl = Dinv * (bl - Hpl^T xp)
        for (int m = tid; m < nMP; m += BA_T) {
          double cl[3] = {pb.b[P + 3 * m], pb.b[P + 3 * m + 1], pb.b[P + 3 * m + 2]};
          for (int a = pb.mpStart[m]; a < pb.mpStart[m + 1]; ++a) {
            const int e = pb.mpEdges[a];
            const int i1 = pb.kfCol[pb.eKF[e]];
            if (i1 < 0) continue;
            const double* B = pb.Hpl + (size_t)e * 18;
            for (int c = 0; c < 3; ++c)
              for (int r = 0; r < 6; ++r) cl[c] -= B[r * 3 + c] * pb.x[6 * i1 + r];
          }
          const double* Di = pb.Dinv + (size_t)m * 9;
          for (int r = 0; r < 3; ++r) pb.x[P + 3 * m + r] = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
        }
      } else {
        for (int i = tid; i < P + 3 * nMP; i += BA_T) pb.x[i] = 0;
      }
      __syncthreads();
      // update (oplus)
      for (int kf = tid; kf < nKF; kf += BA_T) {
        const int col = pb.kfCol[kf];
        if (col < 0) continue;
        double u[6];
        for (int r = 0; r < 6; ++r) u[r] = pb.x[6 * col + r];
        store_se3(pb.pose + 7 * kf, se3_mul(se3_exp(u), load_se3(pb.pose + 7 * kf)));
      }
      for (int i = tid; i < nMP * 3; i += BA_T) pb.pt[i] += pb.x[P + i];
      __syncthreads();
      double tempChi = chi2All();
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = currentChi - tempChi;
      double part = 0;
      for (int i = tid; i < P + 3 * nMP; i += BA_T) part += pb.x[i] * (lambda * pb.x[i] + pb.b[i]);
      double scale = block_sum_d<BA_W>(part, red) + 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - cube_rn(2 * rho - 1);
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        __syncthreads();
        for (int i = tid; i < nKF * 7; i += BA_T) pb.pose[i] = pb.poseBk[i];
        for (int i = tid; i < nMP * 3; i += BA_T) pb.pt[i] = pb.ptBk[i];
        __syncthreads();
      }
      ++qmax; ++trials;
    } while (rho < 0 && qmax < 10 && !terminate());
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) break;
  }
  __syncthreads();
  // ---- post: chi2 / depth gates on the stored errors (state of the last evaluation), write-back as float ----
  for (int e = tid; e < nE; e += BA_T) {
    const float* o = pb.eObs + 3 * e;
    const bool st = !(o[2] < 0);
    double xc[3], err[3];
    se3_map(load_se3(pb.poseEval + 7 * pb.eKF[e]), pb.ptEval + 3 * pb.eMP[e], xc);
    const double c = ba_edge_error(cam, pb.rig, st, xc, o, (double)pb.eInfo[e], err);
    se3_map(load_se3(pb.pose + 7 * pb.eKF[e]), pb.pt + 3 * pb.eMP[e], xc);
    pb.erase[e] = (c > (st ? 7.815 : 5.991) || !ba_depth_positive(pb.rig, o, xc)) ? 1 : 0;
  }
  for (int kf = tid; kf < nKF; kf += BA_T)
    if (pb.kfCol[kf] >= 0) for (int k = 0; k < 7; ++k) pb.poseIO[7 * kf + k] = (float)pb.pose[7 * kf + k];
  for (int i = tid; i < nMP * 3; i += BA_T) pb.ptIO[i] = (float)pb.pt[i];
  if (tid == 0) { pb.stats[0] = its; pb.stats[1] = trials; }
}

// ---------------------------------------------------------------------------------------------------
// Grid mode: the same LM, one launch per phase over the whole chip; the accept/reject decision is taken on the
// host from three doubles read back once per trial (chi2, scale, ok).  Every reduction has a fixed order
// (block partials summed by one block; chunk partials summed per keyframe), so results are deterministic.
constexpr int GB = 256;

// LM state (grid mode, device-side control).  A phase kernel launched with gated = 1 returns at once when the solve has finished
// (launches are queued one trial ahead of the decisions) or, for the build kernels, when this trial re-solves the same system
// with a larger lambda (the previous trial was rejected).
enum { LM_ITER, LM_QMAX, LM_NBAD, LM_ITS, LM_TRIALS, LM_DONE, LM_NEEDBUILD, LM_REJECTED, LM_TICKET };
enum { LMD_CHI, LMD_LAMBDA, LMD_NI, LMD_INICHI };
__device__ __forceinline__ bool lm_skip(const BaDev& pb, int gated) { return gated && pb.lmi[LM_DONE] != 0; }
__device__ __forceinline__ bool lm_skip_build(const BaDev& pb, int gated) { return gated && (pb.lmi[LM_DONE] != 0 || pb.lmi[LM_NEEDBUILD] == 0); }
__device__ __forceinline__ bool lm_stop_requested(const BaDev& pb) {   // morb_ba_set_stop's flag and the caller's *pbStopFlag as the host loop forwards it
  return pb.stop && (__hip_atomic_load(pb.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 ||
                     __hip_atomic_load(pb.stop + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0);
}

// after the first chi2 (sum s): currentChi, and the state at the top of the first iteration.  One thread.
__device__ void lm_init(const BaDev& pb, double s) {
  const bool stop = lm_stop_requested(pb);
  pb.scal[0] = s;
  pb.lmd[LMD_CHI] = s; pb.lmd[LMD_LAMBDA] = 0; pb.lmd[LMD_NI] = 2; pb.lmd[LMD_INICHI] = s;
  pb.lmi[LM_ITER] = 0; pb.lmi[LM_QMAX] = 0; pb.lmi[LM_NBAD] = 0; pb.lmi[LM_ITS] = stop ? 0 : 1; pb.lmi[LM_TRIALS] = 0;
  pb.lmi[LM_DONE] = stop ? 1 : 0; pb.lmi[LM_NEEDBUILD] = 1; pb.lmi[LM_REJECTED] = 0;
  __hip_atomic_store(pb.lmHost + 2, stop ? 0 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(pb.lmHost + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(pb.lmHost + 1, stop ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// after a trial's chi2 (s0) and linear-model gain (s1): rho, accept / reject, the lambda schedule, and whether another trial /
// iteration follows (optimization_algorithm_levenberg.cpp:99-169 with ORB-SLAM's stop rule).  One thread.
__device__ void lm_decide(const BaDev& pb, double s0, double s1) {
  pb.scal[0] = s0; pb.scal[1] = s1;
  double currentChi = pb.lmd[LMD_CHI], lambda = pb.lmd[LMD_LAMBDA], ni = pb.lmd[LMD_NI];
  const double iniChi = pb.lmd[LMD_INICHI];
  int iter = pb.lmi[LM_ITER], qmax = pb.lmi[LM_QMAX], nBad = pb.lmi[LM_NBAD], its = pb.lmi[LM_ITS];
  double tempChi = s0;
  if (pb.scal[2] == 0.0) tempChi = 1.7976931348623157e308;   // the linear solve failed
  const double rho = (currentChi - tempChi) / (s1 + 1e-3);
  const bool accept = rho > 0 && isfinite(tempChi);
  if (accept) {
    double alpha = 1. - cube_rn(2 * rho - 1);
    alpha = fmin(alpha, 2. / 3.);
    lambda *= fmax(1. / 3., alpha);
    ni = 2;
    currentChi = tempChi;
  } else {
    lambda *= ni;
    ni *= 2;
  }
  ++qmax;
  const int trials = pb.lmi[LM_TRIALS] + 1;
  const bool stop = lm_stop_requested(pb);
  int done = 0, needBuild = 0;
  if (!(rho < 0 && qmax < 10 && !stop)) {   // the trial loop ends (:139)
    bool fin = (qmax == 10 || rho == 0);
    if (!fin) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; fin = nBad >= 3; }   // the stop rule of ORB-SLAM's g2o (:146-151)
    if (!fin) { ++iter; fin = iter >= 10 || stop; }
    if (fin) done = 1;
    else { ++its; qmax = 0; needBuild = 1; pb.lmd[LMD_INICHI] = currentChi; }
  }
  pb.lmd[LMD_CHI] = currentChi; pb.lmd[LMD_LAMBDA] = lambda; pb.lmd[LMD_NI] = ni;
  pb.lmi[LM_ITER] = iter; pb.lmi[LM_QMAX] = qmax; pb.lmi[LM_NBAD] = nBad; pb.lmi[LM_ITS] = its; pb.lmi[LM_TRIALS] = trials;
  pb.lmi[LM_DONE] = done; pb.lmi[LM_NEEDBUILD] = needBuild; pb.lmi[LM_REJECTED] = accept ? 0 : 1;
  __hip_atomic_store(pb.lmHost + 2, its, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(pb.lmHost + 3, trials, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(pb.lmHost + 1, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(pb.lmHost + 0, trials, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // decided trials: the host queues trial k + 2 when it sees k
}
// In-launch hand-off of partial sums to the workgroup that draws the last ticket (guide, guideline 16): the payload is stored write-through
// at agent scope (sc1), the storing wave waits for its stores (s_waitcnt vmcnt(0)) and only then draws its ticket with a relaxed agent-scope
// add; the last arriver reads every handed-off word at agent scope (load_l2).  No __threadfence(): on this chip an agent-scope release is a
// write-back of the XCD's L2, and one per wave (k_g_build: 324 of them, beside the landmark half's stores) was ~8 of the kernel's 18 us.
__device__ __forceinline__ void store_l2(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int draw_ticket(int* counter) { return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_l2(const double* p) {   // another workgroup of this launch wrote it: read at agent scope
  return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// mode 0: block partial sums only (host-side LM control).  mode 1 / 2 (device-side control): the last workgroup to finish adds the
// partials up in k_g_reduce's order and takes the LM decision of this trial (1) or sets up the first iteration (2).
__global__ __launch_bounds__(GB) void k_g_chi2(const BaDev* __restrict__ pbp, double* __restrict__ part, const double* __restrict__ part1, int mode) {
  __shared__ double red[4];
  __shared__ int isLast;
  const BaDev pb = *pbp;
  if (lm_skip(pb, mode == 1)) return;
  const int gid = blockIdx.x * GB + threadIdx.x;
  if (gid < pb.nKF * 7) pb.poseEval[gid] = pb.pose[gid];
  if (gid < pb.nMP * 3) pb.ptEval[gid] = pb.pt[gid];
  double s = 0;
  if (gid < pb.nE) {
    const SE3 T = load_se3(pb.pose + 7 * pb.eKF[gid]);
    double xc[3], err[3], w;
    se3_map(T, pb.pt + 3 * pb.eMP[gid], xc);
    const float* o = pb.eObs + 3 * gid;
    const bool st = !(o[2] < 0);
    const double c = ba_edge_error(pb.cam, pb.rig, st, xc, o, (double)pb.eInfo[gid], err);
    s = huber(st ? (double)(float)sqrt(7.815) : (double)(float)sqrt(5.991), c, &w);
  }
  s = block_sum_d<4>(s, red);
  if (threadIdx.x == 0) store_l2(part + blockIdx.x, s);
  if (mode == 0) return;
  if (threadIdx.x == 0) {
    wait_stores();
    isLast = draw_ticket(&pb.lmi[LM_TICKET]) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!isLast) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (every handed-off word is read with load_l2)
  const int n = gridDim.x;
  double s0 = 0, s1 = 0;
  for (int i = threadIdx.x; i < n; i += GB) { s0 += load_l2(part + i); if (mode == 1) s1 += load_l2(part1 + i); }
  s0 = block_sum_d<4>(s0, red);
  s1 = block_sum_d<4>(s1, red);
  if (threadIdx.x != 0) return;
  pb.lmi[LM_TICKET] = 0;
  if (mode == 1) lm_decide(pb, s0, s1); else lm_init(pb, s0);
}
// The landmark blocks Hll / bl of map point m (block_solver.hpp:354-480 reads them): EIGHT lanes per point, one edge each and chunk by chunk
// (a point has 5 - 8 observations here; a thread per point walked them one after the other: 20 of k_g_build's 23 us).  Each lane leaves its
// edge's twelve contributions in the wave's LDS slice and the group's first lane adds them up IN EDGE ORDER — the sums are bit for bit those
// of the serial walk.  Lanes of one wave only: no workgroup barrier.  -> max |diag Hll| of the point (first lane of the group; 0 elsewhere)
constexpr int MP_LANES = 8;
template <bool RIG>
__device__ __forceinline__ double build_mp_point(const BaDev& pb, int m, int sub, double* __restrict__ slot /* [MP_LANES][12] of the group */) {
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  double Hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
  const bool live = m < pb.nMP;
  const double* X = pb.pt + 3 * (live ? m : 0);
  const int k0 = live ? pb.mpStart[m] : 0, k1 = live ? pb.mpStart[m + 1] : 0;
  for (int kb = k0; kb < k1; kb += MP_LANES) {
    const int k = kb + sub;
    double c12[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) c12[q] = 0;
    if (k < k1) {
      const int e = pb.mpEdges[k];
      const SE3 T = load_se3(pb.pose + 7 * pb.eKF[e]);
      double xc[3], err[3], w, R[9], Jl[9];
      se3_map(T, X, xc);
      const float* o = pb.eObs + 3 * e;
      const bool st = !(o[2] < 0);
      const double info = (double)pb.eInfo[e];
      const double c = ba_edge_error(pb.cam, pb.rig, st, xc, o, info, err);
      huber(st ? deltaStereo : deltaMono, c, &w);
      q_to_R(T.q, R);
      ba_edge_jac<RIG>(pb.cam, pb.rig, st, xc, o, R, nullptr, Jl);
      const double wo = w * info;   // (mono edges: third Jacobian row and err[2] are zero, so the 3-row form is exact)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double sacc = 0;
        _Pragma("unroll") for (int i = 0; i < 3; ++i) sacc += Jl[i * 3 + r] * (-info * err[i] * w);
        c12[9 + r] = sacc;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          double h = 0;
          _Pragma("unroll") for (int i = 0; i < 3; ++i) h += Jl[i * 3 + r] * wo * Jl[i * 3 + cc];
          c12[r * 3 + cc] = h;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) slot[sub * 12 + q] = c12[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (sub == 0) {
      const int cnt = k1 - kb < MP_LANES ? k1 - kb : MP_LANES;
      for (int j = 0; j < cnt; ++j) {
#pragma unroll
        for (int q = 0; q < 9; ++q) Hl[q] += slot[j * 12 + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) bl[q] += slot[j * 12 + 9 + q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // (the slice is rewritten by the next chunk)
  }
  if (!live || sub != 0) return 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) pb.Hll[(size_t)m * 9 + k] = Hl[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) pb.b[pb.P + 3 * m + k] = bl[k];
  return fmax(fmax(fabs(Hl[0]), fabs(Hl[4])), fabs(Hl[8]));
}
template <bool RIG>
__device__ __forceinline__ void build_kf_chunk(const BaDev& pb, int c, int lane) {
  const int kf = pb.chunkKF[c];
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  const SE3 T = load_se3(pb.pose + 7 * kf);
  double R[9];
  q_to_R(T.q, R);
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0;
  const int k = pb.chunkStart[c] + lane;
  if (k < pb.chunkEnd[c]) {
    const int e = pb.kfEdges[k];
    double xc[3], err[3], w, Jp[18], Jl[9];
    se3_map(T, pb.pt + 3 * pb.eMP[e], xc);
    const float* o = pb.eObs + 3 * e;
    const bool st = !(o[2] < 0);
    const double info = (double)pb.eInfo[e];
    const double ch = ba_edge_error(pb.cam, pb.rig, st, xc, o, info, err);
    huber(st ? deltaStereo : deltaMono, ch, &w);
    ba_edge_jac<RIG>(pb.cam, pb.rig, st, xc, o, R, Jp, Jl);
    const double wo = w * info;   // (mono edges: third Jacobian row and err[2] are zero, so the 3-row form is exact)
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double sacc = 0;
      _Pragma("unroll") for (int i = 0; i < 3; ++i) sacc += Jp[i * 6 + r] * (-info * err[i] * w);
      acc[21 + r] = sacc;
#pragma unroll
      for (int cc = r; cc < 6; ++cc) {
        double h = 0;
        _Pragma("unroll") for (int i = 0; i < 3; ++i) h += Jp[i * 6 + r] * wo * Jp[i * 6 + cc];
        acc[q++] = h;
      }
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        double h = 0;
        _Pragma("unroll") for (int i = 0; i < 3; ++i) h += Jp[i * 6 + r] * wo * Jl[i * 3 + cc];
        pb.Hpl[(size_t)e * 18 + r * 3 + cc] = h;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 27; ++q) acc[q] = wave_sum_d(acc[q]);
  if (lane < 27) {
    double v = 0;
#pragma unroll
    for (int q = 0; q < 27; ++q) if (q == lane) v = acc[q];
    store_l2(pb.kfPart + (size_t)c * 27 + lane, v);
  }
}
// buildSystem in ONE launch (device-side LM control): workgroups [0, kfBlocks) take the keyframe chunks, the rest the map points;
// the last chunk of a keyframe to deliver its partial blocks adds them up in chunk order (k_g_kf_reduce's sum, whoever runs it).
// Two launches on two streams cost more in cross-stream events (~25 us per trial) than running side by side saved.
template <bool RIG>
__global__ __launch_bounds__(GB) void k_g_build(const BaDev* __restrict__ pbp, int kfBlocks) {
  const BaDev pb = *pbp;
  if (lm_skip_build(pb, 1)) return;
  // first iteration: the largest diagonal entry of the system for computeLambdaInit (:186-194) — a max is order-independent, so an
  // atomic on the bit pattern of the non-negative double keeps the result deterministic
  const bool first = pb.lmi[LM_TRIALS] == 0;
  unsigned long long* maxDiag = reinterpret_cast<unsigned long long*>(pb.scal + 3);
  if ((int)blockIdx.x >= kfBlocks) {
    __shared__ double sMp[GB * 12];
    const int g = threadIdx.x / MP_LANES, sub = threadIdx.x % MP_LANES;
    const int m = (blockIdx.x - kfBlocks) * (GB / MP_LANES) + g;
    double dm = build_mp_point<RIG>(pb, m, sub, sMp + (size_t)g * MP_LANES * 12);
    if (first) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dm = fmax(dm, __shfl_xor(dm, off, 64));
      if ((threadIdx.x & 63) == 0) atomicMax(maxDiag, __builtin_bit_cast(unsigned long long, dm));
    }
    return;
  }
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= pb.nChunks) return;
  build_kf_chunk<RIG>(pb, c, lane);
  const int kf = pb.chunkKF[c];
  int last = 0;
  wait_stores();   // (the wave's partial sums have left for memory)
  if (lane == 0) {
    const int nc = pb.kfChunkStart[kf + 1] - pb.kfChunkStart[kf];
    last = draw_ticket(&pb.kfTicket[kf]) == nc - 1;
    if (last) pb.kfTicket[kf] = 0;
  }
  if (!__builtin_amdgcn_readfirstlane(last)) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (every handed-off word is read with load_l2)
  const int col = pb.kfCol[kf];
  if (lane >= 27) return;
  double s = 0;
  for (int cc = pb.kfChunkStart[kf]; cc < pb.kfChunkStart[kf + 1]; ++cc) s += load_l2(pb.kfPart + (size_t)cc * 27 + lane);
  if (lane < 21) {
    int r = 0, q = lane;
    while (q >= 6 - r) { q -= 6 - r; ++r; }
    const int c2 = r + q;
    pb.Hpp[(size_t)col * 36 + r * 6 + c2] = s;
    pb.Hpp[(size_t)col * 36 + c2 * 6 + r] = s;
    if (first && r == c2) atomicMax(maxDiag, __builtin_bit_cast(unsigned long long, fabs(s)));
  } else {
    pb.b[6 * col + (lane - 21)] = s;
  }
}
// Hpl of (keyframe column i, landmark m) as the MFMA operands need it: a fisheye rig may observe a landmark with both cameras
// of one keyframe, i.e. through two edges — the first of them (lowest edge index) carries the sum, the others nothing.
__device__ __forceinline__ bool pair_block(const BaDev& pb, int e, int i, int m, double* __restrict__ B) {
  for (int q = 0; q < 18; ++q) B[q] = pb.Hpl[(size_t)e * 18 + q];
  if (!pb.dupPairs) return true;
  for (int k = pb.mpStart[m]; k < pb.mpStart[m + 1]; ++k) {
    const int e2 = pb.mpEdges[k];
    if (e2 == e || pb.kfCol[pb.eKF[e2]] != i) continue;
    if (e2 < e) return false;
    for (int q = 0; q < 18; ++q) B[q] += pb.Hpl[(size_t)e2 * 18 + q];
  }
  return true;
}
// Start of a trial.  gated (device-side LM control): lambda comes from the LM state; a rejected previous trial is undone here
// (pose / point backup restored instead of taken), and after an accepted one the operand W is packed too (k_g_pack_w's work).
__global__ __launch_bounds__(GB) void k_g_dinv_push(const BaDev* __restrict__ pbp, double lambda, double* __restrict__ Hs, int valuSchur, int gated) {
  const BaDev pb = *pbp;
  if (lm_skip(pb, gated)) return;
  bool restore = false, packW = false;
  const int gid = blockIdx.x * GB + threadIdx.x;
  if (gated) {
    lambda = pb.lmd[LMD_LAMBDA]; restore = pb.lmi[LM_REJECTED] != 0; packW = pb.lmi[LM_NEEDBUILD] != 0;
    if (pb.lmi[LM_TRIALS] == 0) {   // first trial: computeLambdaInit (:186-194) from the build's max diagonal; later kernels read it from the state
      lambda = pb.userLambda > 0 ? pb.userLambda : 1e-5 * pb.scal[3];
      if (gid == 0) pb.lmd[LMD_LAMBDA] = lambda;
    }
  }
  if (restore) {
    if (gid < pb.nKF * 7) pb.pose[gid] = pb.poseBk[gid];
    if (gid < pb.nMP * 3) pb.pt[gid] = pb.ptBk[gid];
  } else {
    if (gid < pb.nKF * 7) pb.poseBk[gid] = pb.pose[gid];
    if (gid < pb.nMP * 3) pb.ptBk[gid] = pb.pt[gid];
  }
  if (valuSchur && gid < pb.P * pb.P) Hs[gid] = 0;
  if (gid < pb.nMP) {
    double D[9], Di[9];
    for (int k = 0; k < 9; ++k) D[k] = pb.Hll[(size_t)gid * 9 + k];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    inv3(D, Di);
    for (int k = 0; k < 9; ++k) pb.Dinv[(size_t)gid * 9 + k] = Di[k];
  }
  if (!valuSchur && gid < pb.nE) {
    // the MFMA operand WD = Hpl (Hll + lambda I)^-1 of this observation (rows 3 m + c, columns 6 i + r); the inverse is
    // recomputed per observation so that the pack needs no second launch behind the per-point loop above
    const int i = pb.kfCol[pb.eKF[gid]];
    const int m = pb.eMP[gid];
    double B1[18];
    if (i >= 0 && pair_block(pb, gid, i, m, B1)) {
      double D[9], Di[9];
      for (int k = 0; k < 9; ++k) D[k] = pb.Hll[(size_t)m * 9 + k];
      D[0] += lambda; D[4] += lambda; D[8] += lambda;
      inv3(D, Di);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          pb.sWD[(size_t)(3 * m + c) * pb.sMp + 6 * i + r] = B1[r * 3] * Di[c] + B1[r * 3 + 1] * Di[3 + c] + B1[r * 3 + 2] * Di[6 + c];
          if (packW) pb.sW[(size_t)(3 * m + c) * pb.sMp + 6 * i + r] = B1[r * 3 + c];
        }
    }
  }
  if (packW && gid < pb.nMP * 3) pb.sW[(size_t)gid * pb.sMp + pb.P] = pb.b[pb.P + gid];
}
// reduced system from the partial products: Hs = Hpp + lambda I - C (C symmetric: the upper blocks serve both triangles),
// x[0:P] = b_p - C[:, P]
__global__ __launch_bounds__(GB) void k_g_schur_finish(const BaDev* __restrict__ pbp, double lambda, double* __restrict__ Hs, int gated) {
  const BaDev pb = *pbp;
  if (lm_skip(pb, gated)) return;
  if (gated) lambda = pb.lmd[LMD_LAMBDA];
  const int t = blockIdx.x * GB + threadIdx.x, gid = t >> 2, q = t & 3, P = pb.P;   // four lanes per element (schur_sum4)
  const bool mat = gid < P * P, rhs = !mat && gid < P * P + P;
  int r = 0, c = P;
  if (mat) { r = gid / P; c = gid - r * P; } else if (rhs) r = gid - P * P;
  const int i = r < c ? r : c, j = r < c ? c : r;
  const double cs = morbschur::schur_sum4(pb.sPart, pb.sBlkIndex, pb.sNb, pb.sNblk, pb.sNsplit, (mat || rhs) ? i : 0, (mat || rhs) ? j : 0, q);
  if (q != 0) return;
  if (mat) {
    double v = -cs;
    if (r / 6 == c / 6) { v += pb.Hpp[(size_t)(r / 6) * 36 + (r % 6) * 6 + (c % 6)]; if (r == c) v += lambda; }
    Hs[gid] = v;
  } else if (rhs) {
    pb.x[r] = pb.b[r] - cs;
  }
}
// the reduced camera system beyond ~176 unknowns (30 free keyframes and more: the reference takes every covisible keyframe,
// Optimizer.cc:1058-1070): the matrix stays in global memory, one 16-column panel at a time in LDS (dense_ldlt.h: ldlt_solve_global);
// HsG is overwritten by the factors (k_g_schur_finish rebuilds it for every trial); x = Hs^-1 x in place
__global__ __launch_bounds__(morbdense::GT) void k_g_ldlt_global(const BaDev* __restrict__ pbp, double* __restrict__ HsG, double* __restrict__ pnlG,
                                                                 int panelInLds, int gated) {
  extern __shared__ double sLd[];   // dblk | y | (the panel copies when they fit)
  __shared__ int sOk;
  const BaDev pb = *pbp;
  if (lm_skip(pb, gated)) return;
  double* pnl = panelInLds ? sLd + morbdense::global_lds_doubles(pb.P) : pnlG;
  const bool ok = morbdense::ldlt_solve_global<false>(HsG, pb.x, pb.x, pb.P, pnl, sLd, &sOk);
  if (threadIdx.x == 0) pb.scal[2] = ok ? 1.0 : 0.0;
}
// the reduced camera system with its lower triangle resident in LDS (dense_ldlt.h); x = Hs^-1 x in place
__global__ __launch_bounds__(morbdense::LT) void k_g_ldlt_lds(const BaDev* __restrict__ pbp, const double* __restrict__ HsG, int gated) {
  extern __shared__ double sLd[];
  __shared__ int sOk;
  const BaDev pb = *pbp;
  if (lm_skip(pb, gated)) return;
  const bool ok = morbdense::ldlt_solve<false>(HsG, pb.x, pb.x, pb.P, sLd, &sOk);
  if (threadIdx.x == 0) pb.scal[2] = ok ? 1.0 : 0.0;
}
// The same step with the landmark part read from the MFMA operand W (dense rows [3 nMP][Mp], column P = b_l): 16 lanes per point,
// lane q takes columns q, q + 16, ... of the point's three rows (coalesced 128-byte reads, all in flight at once) and the row sums are
// DPP reductions in a fixed order — round 2's first form walked the point's edges one dependent load chain after the other (21 us).
__global__ __launch_bounds__(GB) void k_g_backsub_update_w(const BaDev* __restrict__ pbp, double* __restrict__ part) {
  __shared__ double red[4];
  const BaDev pb = *pbp;
  if (lm_skip(pb, 1)) return;
  const double lambda = pb.lmd[LMD_LAMBDA];
  const int gid = blockIdx.x * GB + threadIdx.x, m = gid >> 4, q = gid & 15;
  const int P = pb.P;
  const bool ok = pb.scal[2] != 0.0;
  double sc = 0;
  {
    const bool live = m < pb.nMP;
    const size_t row = (size_t)3 * (live ? m : 0) * pb.sMp;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int j = q; j < P; j += 16) {
      const double xj = pb.x[j];
      a0 += pb.sW[row + j] * xj; a1 += pb.sW[row + pb.sMp + j] * xj; a2 += pb.sW[row + 2 * (size_t)pb.sMp + j] * xj;
    }
    a0 = morbwave::row_sum_f64(a0); a1 = morbwave::row_sum_f64(a1); a2 = morbwave::row_sum_f64(a2);
    if (live && q < 3) {
      double xl = 0;
      if (ok) {
        const double cl[3] = {pb.sW[row + P] - a0, pb.sW[row + pb.sMp + P] - a1, pb.sW[row + 2 * (size_t)pb.sMp + P] - a2};
        const double* Di = pb.Dinv + (size_t)m * 9;
        xl = Di[q * 3] * cl[0] + Di[q * 3 + 1] * cl[1] + Di[q * 3 + 2] * cl[2];
      }
      pb.x[P + 3 * m + q] = xl;
      pb.pt[3 * m + q] += xl;
      sc += xl * (lambda * xl + pb.b[P + 3 * m + q]);
    }
  }
  if (gid < pb.nKF) {
    const int col = pb.kfCol[gid];
    if (col >= 0) {
      double u[6];
      for (int r = 0; r < 6; ++r) { u[r] = ok ? pb.x[6 * col + r] : 0.0; sc += u[r] * (lambda * u[r] + pb.b[6 * col + r]); }
      store_se3(pb.pose + 7 * gid, se3_mul(se3_exp(u), load_se3(pb.pose + 7 * gid)));
    }
  }
  sc = block_sum_d<4>(sc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = sc;
}
__global__ __launch_bounds__(GB) void k_g_finish(const BaDev* __restrict__ pbp, int its, int trials) {
  const BaDev pb = *pbp;
  const double *pose = pb.pose, *pt = pb.pt;
  if (its < 0) {   // device-side LM control keeps the counters; a rejected last trial is undone by reading its backup
    its = pb.lmi[LM_ITS]; trials = pb.lmi[LM_TRIALS];
    if (pb.lmi[LM_REJECTED]) { pose = pb.poseBk; pt = pb.ptBk; }
  }
  const int gid = blockIdx.x * GB + threadIdx.x;
  if (gid < pb.nE) {
    const int e = gid;
    const float* o = pb.eObs + 3 * e;
    const bool st = !(o[2] < 0);
    double xc[3], err[3];
    se3_map(load_se3(pb.poseEval + 7 * pb.eKF[e]), pb.ptEval + 3 * pb.eMP[e], xc);
    const double c = ba_edge_error(pb.cam, pb.rig, st, xc, o, (double)pb.eInfo[e], err);
    se3_map(load_se3(pose + 7 * pb.eKF[e]), pt + 3 * pb.eMP[e], xc);
    pb.erase[e] = (c > (st ? 7.815 : 5.991) || !ba_depth_positive(pb.rig, o, xc)) ? 1 : 0;
  }
  if (gid < pb.nKF && pb.kfCol[gid] >= 0) for (int k = 0; k < 7; ++k) pb.poseIO[7 * gid + k] = (float)pose[7 * gid + k];
  if (gid < pb.nMP * 3) pb.ptIO[gid] = (float)pt[gid];
  if (gid == 0) { pb.stats[0] = its; pb.stats[1] = trials; }
}

// ---- device-side LM control (optimization_algorithm_levenberg.cpp:61-169 as morb_ba_solve's host loop used to run it) ----
__global__ void k_ba_reset(BaDev pb, const float* __restrict__ pose0, const float* __restrict__ pt0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < pb.nKF) {
    const SE3 s = se3_from_float(pose0 + 7 * i);
    store_se3(pb.pose + 7 * i, s);
    for (int k = 0; k < 7; ++k) pb.poseIO[7 * i + k] = pose0[7 * i + k];
  }
  if (i < pb.nMP * 3) pb.pt[i] = (double)pt0[i];
  if (i == 0) { pb.lmi[LM_TICKET] = 0; pb.scal[3] = 0; }
}

}  // namespace

// =====================================================================================================
struct morb_optimizer {
  int device = 0;
  hipStream_t stream = nullptr;
  // grid-mode LocalBA runs pairs of independent small phases side by side (per-point / per-keyframe builds, Schur
  // complement / reduced right-hand side): fork-join on a side stream
  hipStream_t side = nullptr;
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  void* work = nullptr;        // grow-only device workspace of the one-shot entry points (morb_local_inertial_ba)
  size_t workBytes = 0;
  int* lmWords = nullptr;      // 16 pinned, device-mapped ints: LM state mirror of morb_local_inertial_ba (device-side LM control)
  int* lmWordsDev = nullptr;
  void* stage = nullptr;       // grow-only pinned host buffer: the one-shot entry points gather their inputs here for a single upload
  size_t stageBytes = 0;
  bool arenaCreate = false;    // morb_ba_problem_create carves the problem from `work` / `stage` (the one-shot entry points set this around the call)
  double* scalPinned = nullptr;   // pinned scalars of an arena-mode problem
  int exactOrder = 1;             // PoseOptimization: 1 (default) = edge-order sums, the LM path of g2o decision for decision; 0 = tree sums
  int mfmaChain = 0;              // ... carried by the FP64 matrix core (this device passed k_mfma_order_selftest) instead of dependent v_add_f64
  int mfmaSelftest = -1;          // what k_mfma_order_selftest said on this device: 1 passed, 0 rejected, -1 not run (MORB_PO2_CHAIN forced the choice) or failed to run
  // outgrown workspaces / staging buffers: a kernel or copy queued earlier (on this handle's stream or a caller's) may still use them, so growth neither
  // waits for a stream nor frees (hipFree waits for the DEVICE): they are released with the handle.  Each growth asks for half as much again, so the
  // retired bytes stay below twice the final size.
  std::vector<void*> retiredDev, retiredHost;
  void* spill = nullptr;       // grow-only device buffer of the BATCH entry points (the one-shot ones own `work`): edge lists that do not fit the LDS
  size_t spillBytes = 0;
};

struct morb_ba_problem {
  morb_optimizer* opt = nullptr;
  BaDev h;                 // host copy of the device descriptor
  BaDev* d_desc = nullptr;
  std::vector<void*> allocs;
  float *d_pose0 = nullptr, *d_pt0 = nullptr;
  int* h_stop = nullptr;   // [16] pinned, device-mapped host words: [0] abort flag (morb_ba_set_stop writes it without any HIP call, kernels poll it), [1] forwarded *pbStopFlag, [4..7] LM state mirror
  const volatile unsigned char* userStop = nullptr;   // the caller's *pbStopFlag (one-shot entry points), polled by the host LM loop
  int useLds = 1;
  size_t ldsBytes = 0;
  size_t denseLds = 0;     // LDS bytes of the triangle-resident solver (0: the system is too large for it)
  int mode = 0;            // 0 = grid (one launch per LM phase, host-side accept/reject), 1 = one persistent workgroup
  hipEvent_t solved = nullptr;   // recorded behind the last morb_ba_solve on whatever stream it ran on: morb_ba_results waits for the EVENT — not for
                                 // the device, and not through the caller's stream handle, which the caller may have destroyed since
  int redBlocks = 0;
  morbschur::Plan schur;
  size_t nPairEntries = 0;   // (e1, e2) observation pairs of the sparse block-pair Schur form (flop accounting only)
  double* h_scal = nullptr;  // pinned host mirror of scal[0..3]
  double* d_ldws = nullptr;  // panel copies of the global-memory LDL^T when they do not fit LDS (dense_ldlt.h: global_panel_doubles)
  size_t globalLds = 0;      // dynamic LDS of k_g_ldlt_global
  int panelInLds = 1;
  bool arena = false;        // device memory and pinned words belong to the optimizer handle (one-shot entry points): nothing to free
};

extern "C" {

int morb_optimizer_create(morb_optimizer** out, int device) {
  MORB_REQUIRE(out, MORB_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int ndev = 0;
  MORB_HIP_CHECK(hipGetDeviceCount(&ndev));
  MORB_REQUIRE(device >= 0 && device < ndev, MORB_ERR_INVALID, "no such HIP device");
  MORB_HIP_CHECK(hipSetDevice(device));
  morb_optimizer* o = new morb_optimizer();
  o->device = device;
  if (hipStreamCreateWithFlags(&o->stream, hipStreamDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&o->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&o->evFork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&o->evJoin, hipEventDisableTiming) != hipSuccess) {
    delete o;
    set_error("cannot create stream");
    return MORB_ERR_HIP;
  }
  {  // may the matrix core carry the edge-order sums?  (one 64-thread launch per handle; MORB_PO2_CHAIN = valu | mfma overrides)
    const char* force = getenv("MORB_PO2_CHAIN");
    if (force && (!strcmp(force, "valu") || !strcmp(force, "mfma"))) o->mfmaChain = force[0] == 'm';
    else {
      int* d_bad = nullptr;
      int bad = -1;
      if (hipMalloc((void**)&d_bad, sizeof(int)) == hipSuccess && hipMemsetAsync(d_bad, 0, sizeof(int), o->stream) == hipSuccess) {
        hipLaunchKernelGGL(k_mfma_order_selftest, dim3(1), dim3(64), 0, o->stream, 1024, d_bad);
        if (hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, o->stream) != hipSuccess || hipStreamSynchronize(o->stream) != hipSuccess) bad = -1;
      }
      if (d_bad) (void)hipFree(d_bad);
      o->mfmaChain = bad == 0;
      o->mfmaSelftest = bad < 0 ? -1 : (bad == 0 ? 1 : 0);
    }
  }
  *out = o;
  return MORB_OK;
}

int morb_optimizer_device(const morb_optimizer* o) { return o ? o->device : 0; }
void* morb_optimizer_stream(const morb_optimizer* o) { return o ? (void*)o->stream : nullptr; }
int morb_optimizer_workspace(morb_optimizer* o, size_t bytes, void** out) {
  MORB_REQUIRE(o && out, MORB_ERR_INVALID, "NULL argument");
  if (bytes > o->workBytes) {
    // no hipStreamSynchronize, no hipFree on the caller's path (Optimizer.h:46-139 is all-static and entered from three threads: a tracking-thread
    // call must not wait for the LocalBundleAdjustment another thread has running on this device)
    void* fresh = nullptr;
    const size_t want = bytes + bytes / 2;
    MORB_HIP_CHECK(hipMalloc(&fresh, want));
    if (o->work) o->retiredDev.push_back(o->work);
    o->work = fresh;
    o->workBytes = want;
  }
  *out = o->work;
  return MORB_OK;
}

int morb_optimizer_spill(morb_optimizer* o, size_t bytes, void** out) {
  MORB_REQUIRE(o && out, MORB_ERR_INVALID, "NULL argument");
  if (bytes > o->spillBytes) {
    void* fresh = nullptr;
    const size_t want = bytes + bytes / 2;
    MORB_HIP_CHECK(hipMalloc(&fresh, want));
    if (o->spill) o->retiredDev.push_back(o->spill);
    o->spill = fresh;
    o->spillBytes = want;
  }
  *out = o->spill;
  return MORB_OK;
}

int morb_optimizer_lm_words(morb_optimizer* o, int** host, int** dev) {
  MORB_REQUIRE(o && host && dev, MORB_ERR_INVALID, "NULL argument");
  if (!o->lmWords) {
    MORB_HIP_CHECK(hipHostMalloc(&o->lmWords, sizeof(int) * 16, hipHostMallocMapped));
    memset(o->lmWords, 0, sizeof(int) * 16);
    MORB_HIP_CHECK(hipHostGetDevicePointer((void**)&o->lmWordsDev, o->lmWords, 0));
  }
  *host = o->lmWords; *dev = o->lmWordsDev;
  return MORB_OK;
}

int morb_optimizer_staging(morb_optimizer* o, size_t bytes, void** host) {
  MORB_REQUIRE(o && host, MORB_ERR_INVALID, "NULL argument");
  if (bytes > o->stageBytes) {
    void* fresh = nullptr;
    const size_t want = bytes + bytes / 2;
    MORB_HIP_CHECK(hipHostMalloc(&fresh, want));
    if (o->stage) o->retiredHost.push_back(o->stage);     // an upload queued from it may still be in flight
    o->stage = fresh;
    o->stageBytes = want;
  }
  *host = o->stage;
  return MORB_OK;
}

int morb_optimizer_info(const morb_optimizer* o, int* mfma_chain, int* exact_order, int* mfma_selftest) {
  MORB_REQUIRE(o, MORB_ERR_INVALID, "NULL optimizer");
  if (mfma_chain) *mfma_chain = o->mfmaChain;
  if (exact_order) *exact_order = o->exactOrder;
  if (mfma_selftest) *mfma_selftest = o->mfmaSelftest;
  return MORB_OK;
}

int morb_optimizer_set_exact_order(morb_optimizer* o, int on) {
  MORB_REQUIRE(o, MORB_ERR_INVALID, "NULL optimizer");
  o->exactOrder = on ? 1 : 0;
  return MORB_OK;
}

int morb_optimizer_sync(morb_optimizer* o) {
  MORB_REQUIRE(o, MORB_ERR_INVALID, "NULL optimizer");
  MORB_HIP_CHECK(hipSetDevice(o->device));
  MORB_HIP_CHECK(hipStreamSynchronize(o->stream));
  return MORB_OK;
}

void morb_optimizer_destroy(morb_optimizer* o) {
  if (!o) return;
  (void)hipSetDevice(o->device);
  (void)hipStreamSynchronize(o->stream);
  if (o->side) { (void)hipStreamSynchronize(o->side); (void)hipStreamDestroy(o->side); }
  if (o->evFork) (void)hipEventDestroy(o->evFork);
  if (o->evJoin) (void)hipEventDestroy(o->evJoin);
  if (o->work) (void)hipFree(o->work);
  if (o->spill) (void)hipFree(o->spill);
  for (void* w : o->retiredDev) (void)hipFree(w);
  if (o->lmWords) (void)hipHostFree(o->lmWords);
  if (o->stage) (void)hipHostFree(o->stage);
  for (void* w : o->retiredHost) (void)hipHostFree(w);
  if (o->scalPinned) (void)hipHostFree(o->scalPinned);
  (void)hipStreamDestroy(o->stream);
  delete o;
}

int morb_pose_optimization_batch(morb_optimizer* o, int nframes, int cap, const int* d_count, const uint8_t* d_hasMP,
                                 const float* d_obs, const float* d_invSigma2, const float* d_Xw, float fx, float fy,
                                 float cx, float cy, float bf, float* d_pose, uint8_t* d_outlier, int* d_nInliers,
                                 int* d_stats, void* stream) {
  MORB_REQUIRE(o && d_hasMP && d_obs && d_invSigma2 && d_Xw && d_pose && d_outlier && d_nInliers, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(o->device));
  hipStream_t st = stream ? (hipStream_t)stream : o->stream;
  Cam cam{fx, fy, cx, cy, bf};
  Rig rig;
  memset(&rig, 0, sizeof rig);
  // (tree-sum mode on small frames: k_pose_opt2's 512 threads take ONE edge each, so a frame with 513 .. 640 active edges pays a second, nearly
  // empty stage per pass — 0.447 against 0.400 ms per launch at 600 features; k_pose_opt's 256 threads with 2 - 3 edges each stay the faster
  // form there.  Frames of ~1200 features of which half hold a map point — tracking — are where the compaction of k_pose_opt2 pays.)
  const bool smallTree = !o->exactOrder && cap <= 640;
  if (pose_opt2_covers(o->exactOrder != 0, o->mfmaChain != 0, cap) && !smallTree && !getenv("MORB_PO_OLD")) {
    const int rc = launch_pose_opt2<false>(o->exactOrder != 0, o->mfmaChain != 0, nframes, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig, nullptr, d_pose,
                                           d_outlier, d_nInliers, d_stats);
    if (rc != MORB_OK) return rc;
  } else
  if (o->exactOrder) hipLaunchKernelGGL((k_pose_opt<false, true, 256>), dim3(nframes), dim3(256), 0, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig,
                     (const int*)nullptr, d_pose, d_outlier, d_nInliers, d_stats);
  else hipLaunchKernelGGL((k_pose_opt<false, false, MORB_PO_NT>), dim3(nframes), dim3(MORB_PO_NT), 0, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig,
                     (const int*)nullptr, d_pose, d_outlier, d_nInliers, d_stats);

  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_pose_optimization_fisheye_batch(morb_optimizer* o, int nframes, int cap, const int* d_count, const int* d_nLeft,
                                         const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                         const float* d_Xw, const float* camL8, const float* camR8, const float* Trl7,
                                         float* d_pose, uint8_t* d_outlier, int* d_nInliers, int* d_stats, void* stream) {
  MORB_REQUIRE(o && d_count && d_nLeft && d_hasMP && d_obs && d_invSigma2 && d_Xw && camL8 && camR8 && Trl7 && d_pose && d_outlier &&
                   d_nInliers, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(o->device));
  hipStream_t st = stream ? (hipStream_t)stream : o->stream;
  Cam cam{0, 0, 0, 0, 0};
  Rig rig;
  memcpy(rig.kbL, camL8, 32);
  memcpy(rig.kbR, camR8, 32);
  {  // g2o::SE3Quat(Trl.unit_quaternion().cast<double>(), Trl.translation().cast<double>()) incl. normalisation
    double q[4] = {Trl7[0], Trl7[1], Trl7[2], Trl7[3]};
    if (q[3] < 0) for (double& c : q) c = -c;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) rig.Trl.q[i] = q[i] / n;
    for (int i = 0; i < 3; ++i) rig.Trl.t[i] = Trl7[4 + i];
  }
  if (pose_opt2_covers(o->exactOrder != 0, o->mfmaChain != 0, cap) && !getenv("MORB_PO_OLD")) {
    const int rc = launch_pose_opt2<true>(o->exactOrder != 0, o->mfmaChain != 0, nframes, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig, d_nLeft, d_pose,
                                          d_outlier, d_nInliers, d_stats);
    if (rc != MORB_OK) return rc;
  } else
  if (o->exactOrder) hipLaunchKernelGGL((k_pose_opt<true, true, 256>), dim3(nframes), dim3(256), 0, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig,
                     d_nLeft, d_pose, d_outlier, d_nInliers, d_stats);
  else hipLaunchKernelGGL((k_pose_opt<true, false, MORB_PO_NT>), dim3(nframes), dim3(MORB_PO_NT), 0, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, cam, rig,
                     d_nLeft, d_pose, d_outlier, d_nInliers, d_stats);

  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_ba_problem_create(morb_optimizer* o, morb_ba_problem** out, int nKF, const float* kfPose, const uint8_t* kfFixed,
                           int nMP, const float* mpPos, int nE, const int* eKF, const int* eMP, const float* eObs,
                           const float* eInvSigma2, float fx, float fy, float cx, float cy, float bf,
                           int lambdaInit100) {
  MORB_REQUIRE(o && out && kfPose && kfFixed && mpPos && eKF && eMP && eObs && eInvSigma2, MORB_ERR_INVALID, "NULL argument");
  *out = nullptr;
  MORB_REQUIRE(nKF > 0 && nMP > 0 && nE > 0, MORB_ERR_INVALID, "empty problem");
  for (int e = 0; e < nE; ++e)
    MORB_REQUIRE(eKF[e] >= 0 && eKF[e] < nKF && eMP[e] >= 0 && eMP[e] < nMP, MORB_ERR_INVALID, "edge index out of range");
  MORB_HIP_CHECK(hipSetDevice(o->device));
  morb_ba_problem* p = new morb_ba_problem();
  p->opt = o;
  BaDev& h = p->h;
  memset(&h, 0, sizeof h);
  h.nKF = nKF; h.nMP = nMP; h.nE = nE;
  std::vector<int> kfCol(nKF, -1);
  // free keyframes that actually carry an edge get a column (initializeOptimization drops isolated vertices)
  std::vector<char> used(nKF, 0);
  for (int e = 0; e < nE; ++e) used[eKF[e]] = 1;
  int nFree = 0;
  for (int i = 0; i < nKF; ++i) if (!kfFixed[i] && used[i]) kfCol[i] = nFree++;
  h.nFree = nFree; h.P = 6 * nFree;
  // CSR lists in edge-id order
  std::vector<int> mpStart(nMP + 1, 0), kfStart(nKF + 1, 0), mpEdges(nE), kfEdges(nE);
  for (int e = 0; e < nE; ++e) { mpStart[eMP[e] + 1]++; kfStart[eKF[e] + 1]++; }
  for (int i = 0; i < nMP; ++i) mpStart[i + 1] += mpStart[i];
  for (int i = 0; i < nKF; ++i) kfStart[i + 1] += kfStart[i];
  {
    std::vector<int> a(mpStart.begin(), mpStart.end() - 1), b(kfStart.begin(), kfStart.end() - 1);
    for (int e = 0; e < nE; ++e) { mpEdges[a[eMP[e]]++] = e; kfEdges[b[eKF[e]]++] = e; }
  }
  h.dupPairs = 0;
  for (int m = 0; m < nMP && !h.dupPairs; ++m)
    for (int a = mpStart[m]; a < mpStart[m + 1] && !h.dupPairs; ++a)
      for (int b2 = a + 1; b2 < mpStart[m + 1]; ++b2)
        if (eKF[mpEdges[a]] == eKF[mpEdges[b2]]) { h.dupPairs = 1; break; }
  // block pairs of the reduced camera system and, per pair, the (observation, observation) entries that feed it: operands of the
  // persistent-workgroup mode.  The one-shot entry points always solve in grid
  // mode on the matrix cores, which only needs the number of entries (flop accounting): they skip the lists (0.3 ms of host work, 0.6 MB).
  std::vector<int> pairBlock, pairStart;
  std::vector<int2> pairEntries;
  const bool wantPairs = !o->arenaCreate;
  size_t nPairEntriesCount = 0;
  if (!wantPairs) {
    for (int m = 0; m < nMP; ++m) {
      size_t nf = 0;
      for (int a = mpStart[m]; a < mpStart[m + 1]; ++a) nf += kfCol[eKF[mpEdges[a]]] >= 0 ? 1 : 0;
      nPairEntriesCount += nf * (nf + 1) / 2;   // (pairs with column(e1) <= column(e2), as the lists would hold them; approximate for duplicate columns)
    }
    pairBlock.push_back(0); pairStart.assign(2, 0); pairEntries.assign(1, make_int2(0, 0));
  } else {
    const int nb = std::max(nFree, 1) * std::max(nFree, 1);
    std::vector<int> cnt(nb + 1, 0);
    auto forPairs = [&](auto&& fn) {
      for (int m = 0; m < nMP; ++m)
        for (int a = mpStart[m]; a < mpStart[m + 1]; ++a) {
          const int ea = mpEdges[a], ca = kfCol[eKF[ea]];
          if (ca < 0) continue;
          for (int b2 = mpStart[m]; b2 < mpStart[m + 1]; ++b2) {
            const int eb = mpEdges[b2], cb = kfCol[eKF[eb]];
            if (cb < 0 || cb < ca) continue;
            fn(ca * nFree + cb, ea, eb);
          }
        }
    };
    forPairs([&](int key, int, int) { cnt[key + 1]++; });
    std::vector<int> slot(nb, -1);
    int total = 0;
    for (int k = 0; k < nb; ++k) {
      if (cnt[k + 1] > 0) { slot[k] = (int)pairBlock.size(); pairBlock.push_back(k); pairStart.push_back(total); total += cnt[k + 1]; }
    }
    pairStart.push_back(total);
    pairEntries.resize(std::max(total, 1));
    std::vector<int> fill(pairStart.begin(), pairStart.end());
    forPairs([&](int key, int ea, int eb) { pairEntries[fill[slot[key]]++] = make_int2(ea, eb); });
  }
  h.nPairs = wantPairs ? (int)pairBlock.size() : 0;
  std::vector<int> chunkKF, chunkStart, chunkEnd, kfChunkStart(nKF + 1, 0);
  for (int kf = 0; kf < nKF; ++kf) {
    kfChunkStart[kf] = (int)chunkKF.size();
    if (kfCol[kf] >= 0)
      for (int k = kfStart[kf]; k < kfStart[kf + 1]; k += 64) { chunkKF.push_back(kf); chunkStart.push_back(k); chunkEnd.push_back(std::min(k + 64, kfStart[kf + 1])); }
  }
  kfChunkStart[nKF] = (int)chunkKF.size();
  h.nChunks = (int)chunkKF.size();
  bool fail = false;
  // Memory: a persistent problem (three-step API) owns one hipMalloc per array.  The one-shot entry points (arena mode) carve
  // everything from the optimizer's grow-only workspace — uploads first, mirrored in a pinned host buffer and sent in ONE copy,
  // device-only arrays behind them — because ~45 hipMalloc / hipFree pairs and ~25 synchronous copies were 3.5 ms of a 4.7 ms call.
  const bool arena = o->arenaCreate;
  p->arena = arena;
  bool dry = false;
  size_t upOff = 0, devOff = 0, upCap = 0, devCap = 0;
  char *aBase = nullptr, *stage = nullptr;
  auto up = [&](const void* src, size_t bytes) -> void* {
    if (arena) {
      const size_t sz = (std::max<size_t>(bytes, 8) + 255) & ~(size_t)255;
      size_t& off = src ? upOff : devOff;
      const size_t at = off;
      off += sz;
      if (dry) return nullptr;
      if (off > (src ? upCap : devCap)) { fail = true; return nullptr; }
      if (src) { memcpy(stage + at, src, bytes); return aBase + at; }
      return aBase + upCap + at;
    }
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(bytes, 8)) != hipSuccess) { fail = true; return nullptr; }
    p->allocs.push_back(d);
    if (src && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) fail = true;
    return d;
  };
  hipStream_t cst = o->stream;
  auto carve = [&]() {
  h.kfCol = (const int*)up(kfCol.data(), sizeof(int) * nKF);
  h.eKF = (const int*)up(eKF, sizeof(int) * nE);
  h.eMP = (const int*)up(eMP, sizeof(int) * nE);
  h.eObs = (const float*)up(eObs, sizeof(float) * 3 * nE);
  h.eInfo = (const float*)up(eInvSigma2, sizeof(float) * nE);
  h.mpStart = (const int*)up(mpStart.data(), sizeof(int) * (nMP + 1));
  h.mpEdges = (const int*)up(mpEdges.data(), sizeof(int) * nE);
  h.kfStart = (const int*)up(kfStart.data(), sizeof(int) * (nKF + 1));
  h.kfEdges = (const int*)up(kfEdges.data(), sizeof(int) * nE);
  h.pairBlock = (const int*)up(pairBlock.data(), sizeof(int) * std::max<size_t>(pairBlock.size(), 1));
  h.pairStart = (const int*)up(pairStart.data(), sizeof(int) * pairStart.size());
  h.pairEntries = (const int2*)up(pairEntries.data(), sizeof(int2) * pairEntries.size());
  p->nPairEntries = wantPairs ? pairEntries.size() : nPairEntriesCount;
  h.chunkKF = (const int*)up(chunkKF.data(), sizeof(int) * std::max<size_t>(chunkKF.size(), 1));
  h.chunkStart = (const int*)up(chunkStart.data(), sizeof(int) * std::max<size_t>(chunkStart.size(), 1));
  h.chunkEnd = (const int*)up(chunkEnd.data(), sizeof(int) * std::max<size_t>(chunkEnd.size(), 1));
  h.kfChunkStart = (const int*)up(kfChunkStart.data(), sizeof(int) * (nKF + 1));
  h.kfPart = (double*)up(nullptr, sizeof(double) * 27 * std::max<size_t>(chunkKF.size(), 1));
  p->d_ldws = (double*)up(nullptr, sizeof(double) * morbdense::global_panel_doubles(std::max(h.P, 1)));
  p->redBlocks = div_up(std::max(std::max(nE, nMP * 16), std::max(nKF * 7, 1)), GB);   // (16 lanes per point in k_g_backsub_update_w)
  h.redPart = (double*)up(nullptr, sizeof(double) * 2 * p->redBlocks);
  h.scal = (double*)up(nullptr, sizeof(double) * 8);
  {
    const morbschur::Plan sp = morbschur::make_plan(h.P + 1, 3 * nMP);
    p->schur = sp;
    std::vector<int2> blocks; std::vector<int> blkIndex((size_t)sp.nb * sp.nb, 0);
    for (int bi = 0; bi < sp.nb; ++bi) for (int bj = bi; bj < sp.nb; ++bj) { blkIndex[(size_t)bi * sp.nb + bj] = (int)blocks.size(); blocks.push_back(make_int2(bi, bj)); }
    h.sW = (double*)up(nullptr, sizeof(double) * sp.wElems());
    h.sWD = (double*)up(nullptr, sizeof(double) * sp.wElems());
    if (!dry && !fail && (arena ? (hipMemsetAsync(h.sW, 0, sizeof(double) * sp.wElems(), cst) != hipSuccess || hipMemsetAsync(h.sWD, 0, sizeof(double) * sp.wElems(), cst) != hipSuccess)
                              : (hipMemset(h.sW, 0, sizeof(double) * sp.wElems()) != hipSuccess || hipMemset(h.sWD, 0, sizeof(double) * sp.wElems()) != hipSuccess))) fail = true;
    h.sPart = (double*)up(nullptr, sizeof(double) * sp.partElems());
    h.sBlocks = (const int2*)up(blocks.data(), sizeof(int2) * blocks.size());
    h.sBlkIndex = (const int*)up(blkIndex.data(), sizeof(int) * blkIndex.size());
    h.sMp = sp.Mp; h.sNb = sp.nb; h.sNblk = sp.nblk; h.sNsplit = sp.nsplit;
  }
  const size_t nx = (size_t)h.P + 3 * (size_t)nMP;
  h.pose = (double*)up(nullptr, sizeof(double) * 7 * nKF);
  h.poseBk = (double*)up(nullptr, sizeof(double) * 7 * nKF);
  h.poseEval = (double*)up(nullptr, sizeof(double) * 7 * nKF);
  h.pt = (double*)up(nullptr, sizeof(double) * 3 * nMP);
  h.ptBk = (double*)up(nullptr, sizeof(double) * 3 * nMP);
  h.ptEval = (double*)up(nullptr, sizeof(double) * 3 * nMP);
  h.Hpp = (double*)up(nullptr, sizeof(double) * 36 * std::max(nFree, 1));
  h.Hll = (double*)up(nullptr, sizeof(double) * 9 * nMP);
  h.Dinv = (double*)up(nullptr, sizeof(double) * 9 * nMP);
  h.Hpl = (double*)up(nullptr, sizeof(double) * 18 * nE);
  h.b = (double*)up(nullptr, sizeof(double) * nx);
  h.x = (double*)up(nullptr, sizeof(double) * nx);
  h.HsG = (double*)up(nullptr, sizeof(double) * std::max<size_t>((size_t)h.P * h.P, 1));
  h.poseIO = (float*)up(nullptr, sizeof(float) * 7 * nKF);
  h.ptIO = (float*)up(nullptr, sizeof(float) * 3 * nMP);
  h.erase = (uint8_t*)up(nullptr, nE);
  h.stats = (int*)up(nullptr, sizeof(int) * 2);
  h.lmd = (double*)up(nullptr, sizeof(double) * 4);
  h.kfTicket = (int*)up(nullptr, sizeof(int) * std::max(nKF, 1));
  if (!dry && !fail && (arena ? hipMemsetAsync(h.kfTicket, 0, sizeof(int) * std::max(nKF, 1), cst) : hipMemset(h.kfTicket, 0, sizeof(int) * std::max(nKF, 1))) != hipSuccess) fail = true;
  h.lmi = (int*)up(nullptr, sizeof(int) * 16);
  // mapped host words: [0] morb_ba_set_stop, [1] the caller's *pbStopFlag as the host loop forwards it, [4..7] the LM state mirror
  if (!dry) {
    if (arena) {
      int* dv = nullptr;
      if (morb_optimizer_lm_words(o, &p->h_stop, &dv) != MORB_OK) { p->h_stop = nullptr; fail = true; }
      else { memset(p->h_stop, 0, sizeof(int) * 16); h.stop = dv; h.lmHost = dv + 4; }
    } else if (hipHostMalloc(&p->h_stop, sizeof(int) * 16, hipHostMallocMapped) != hipSuccess) { p->h_stop = nullptr; fail = true; }
    else {
      memset(p->h_stop, 0, sizeof(int) * 16);
      int* dv = nullptr;
      if (hipHostGetDevicePointer((void**)&dv, p->h_stop, 0) != hipSuccess) fail = true;
      h.stop = dv; h.lmHost = dv + 4;
    }
  }
  h.cam = Cam{fx, fy, cx, cy, bf};
  h.rig = nullptr;
  h.userLambda = lambdaInit100 ? 100.0 : 0.0;
  p->d_pose0 = (float*)up(kfPose, sizeof(float) * 7 * nKF);
  p->d_pt0 = (float*)up(mpPos, sizeof(float) * 3 * nMP);
  p->d_desc = (BaDev*)up(&h, sizeof(BaDev));
  };   // carve
  if (arena) {
    dry = true; carve(); dry = false;   // sizes
    upCap = upOff; devCap = devOff; upOff = devOff = 0;
    void *w = nullptr, *sg = nullptr;
    if (morb_optimizer_workspace(o, upCap + devCap, &w) != MORB_OK || morb_optimizer_staging(o, upCap, &sg) != MORB_OK) fail = true;
    aBase = (char*)w; stage = (char*)sg;
    if (!fail) {
      carve();
      if (!fail && hipMemcpyAsync(aBase, stage, upCap, hipMemcpyHostToDevice, cst) != hipSuccess) fail = true;   // the one upload
    }
    if (!fail) {
      if (!o->scalPinned && hipHostMalloc(&o->scalPinned, sizeof(double) * 8) != hipSuccess) fail = true;
      p->h_scal = o->scalPinned;
    }
  } else {
    carve();
    if (!fail && hipHostMalloc(&p->h_scal, sizeof(double) * 8) != hipSuccess) fail = true;
  }
  p->ldsBytes = sizeof(double) * (size_t)h.P * (h.P + 1);
  p->useLds = (p->ldsBytes <= 136 * 1024 && h.P <= 192) ? 1 : 0;
  if (!p->useLds) p->ldsBytes = 0;
  p->denseLds = sizeof(double) * morbdense::lds_doubles(h.P);
  if (p->denseLds > 156 * 1024 || h.P < 1) p->denseLds = 0;   // larger systems: the global-memory solver
  if (!fail && p->denseLds && hipFuncSetAttribute(reinterpret_cast<const void*>(k_g_ldlt_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) fail = true;
  p->globalLds = sizeof(double) * (morbdense::global_lds_doubles(std::max(h.P, 1)) + morbdense::global_panel_doubles(std::max(h.P, 1)));
  p->panelInLds = p->globalLds <= 150 * 1024 ? 1 : 0;
  if (!p->panelInLds) p->globalLds = sizeof(double) * morbdense::global_lds_doubles(std::max(h.P, 1));
  if (!fail && !p->denseLds && (p->globalLds > 150 * 1024 ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_g_ldlt_global), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)) fail = true;
  if (!fail && p->useLds && hipFuncSetAttribute(reinterpret_cast<const void*>(k_local_ba), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024) != hipSuccess)
    fail = true;
  if (fail) {
    for (void* d : p->allocs) (void)hipFree(d);
    if (p->h_stop && !p->arena) (void)hipHostFree(p->h_stop);
    delete p;
    set_error("device allocation/copy failed while creating the BA problem");
    return MORB_ERR_HIP;
  }
  *out = p;
  return MORB_OK;
}

void morb_ba_problem_destroy(morb_ba_problem* p) {
  if (!p) return;
  (void)hipSetDevice(p->opt->device);
  (void)hipStreamSynchronize(p->opt->stream);
  if (p->solved) { (void)hipEventSynchronize(p->solved); (void)hipEventDestroy(p->solved); }
  for (void* d : p->allocs) (void)hipFree(d);
  if (!p->arena) {
    if (p->h_scal) (void)hipHostFree(p->h_scal);
    if (p->h_stop) (void)hipHostFree(p->h_stop);
  }
  delete p;
}

int morb_ba_problem_create_fisheye(morb_optimizer* o, morb_ba_problem** out, int nKF, const float* kfPose, const uint8_t* kfFixed,
                                   int nMP, const float* mpPos, int nE, const int* eKF, const int* eMP, const float* eObs2,
                                   const uint8_t* eRight, const float* eInvSigma2, const float* camL8, const float* camR8,
                                   const float* Trl7, int lambdaInit100) {
  MORB_REQUIRE(out && eObs2 && eRight && camL8 && camR8 && Trl7, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nE > 0, MORB_ERR_INVALID, "empty problem");
  // the edge kind travels in the third observation slot: -2 = EdgeSE3ProjectXYZ with the left KB8 camera,
  // -3 = EdgeSE3ProjectXYZToBody (right KB8 camera behind mTrl)
  std::vector<float> obs3((size_t)nE * 3);
  for (int e = 0; e < nE; ++e) { obs3[3 * e] = eObs2[2 * e]; obs3[3 * e + 1] = eObs2[2 * e + 1]; obs3[3 * e + 2] = eRight[e] ? -3.0f : -2.0f; }
  int rc = morb_ba_problem_create(o, out, nKF, kfPose, kfFixed, nMP, mpPos, nE, eKF, eMP, obs3.data(), eInvSigma2, 0.f, 0.f, 0.f, 0.f,
                                  0.f, lambdaInit100);
  if (rc != MORB_OK) return rc;
  morb_ba_problem* p = *out;
  Rig rig;
  memcpy(rig.kbL, camL8, 32);
  memcpy(rig.kbR, camR8, 32);
  {  // g2o::SE3Quat(Trl.unit_quaternion().cast<double>(), Trl.translation().cast<double>()) incl. normalisation
    double q[4] = {Trl7[0], Trl7[1], Trl7[2], Trl7[3]};
    if (q[3] < 0) for (double& c : q) c = -c;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) rig.Trl.q[i] = q[i] / n;
    for (int i = 0; i < 3; ++i) rig.Trl.t[i] = Trl7[4 + i];
  }
  void* d_rig = nullptr;
  if (hipMalloc(&d_rig, sizeof(Rig)) != hipSuccess || hipMemcpy(d_rig, &rig, sizeof(Rig), hipMemcpyHostToDevice) != hipSuccess) {
    if (d_rig) (void)hipFree(d_rig);
    morb_ba_problem_destroy(p);
    *out = nullptr;
    set_error("cannot upload the fisheye rig");
    return MORB_ERR_HIP;
  }
  p->allocs.push_back(d_rig);
  p->h.rig = (const Rig*)d_rig;
  if (hipMemcpy(p->d_desc, &p->h, sizeof(BaDev), hipMemcpyHostToDevice) != hipSuccess) {
    morb_ba_problem_destroy(p);
    *out = nullptr;
    set_error("cannot upload the problem descriptor");
    return MORB_ERR_HIP;
  }
  return MORB_OK;
}


int morb_ba_set_mode(morb_ba_problem* p, int mode) {
  MORB_REQUIRE(p && (mode == 0 || mode == 1), MORB_ERR_INVALID, "mode must be 0 (grid) or 1 (persistent workgroup)");
  p->mode = mode;
  return MORB_OK;
}

int morb_ba_set_stop(morb_ba_problem* p, int stop) {
  MORB_REQUIRE(p, MORB_ERR_INVALID, "NULL problem");
  // No HIP call: the flag lives in pinned host memory that the device maps, so it lands while a solve is running on any
  // stream (a copy on the null stream would wait for the very kernels it is meant to stop) and from any thread.
  __atomic_store_n(p->h_stop, stop ? 1 : 0, __ATOMIC_RELEASE);
  return MORB_OK;
}

int morb_ba_solve(morb_ba_problem* p, void* stream) {
  MORB_REQUIRE(p, MORB_ERR_INVALID, "NULL problem");
  MORB_HIP_CHECK(hipSetDevice(p->opt->device));
  hipStream_t st = stream ? (hipStream_t)stream : p->opt->stream;
  if (!p->solved) MORB_HIP_CHECK(hipEventCreateWithFlags(&p->solved, hipEventDisableTiming));
  struct RecordOnExit { hipEvent_t ev; hipStream_t st; ~RecordOnExit() { (void)hipEventRecord(ev, st); } } recordOnExit{p->solved, st};
  const int n = std::max(p->h.nKF, p->h.nMP * 3);
  hipLaunchKernelGGL(k_ba_reset, dim3(div_up(n, 256)), dim3(256), 0, st, p->h, p->d_pose0, p->d_pt0);
  if (p->mode == 1) {
    hipLaunchKernelGGL(k_local_ba, dim3(1), dim3(BA_T), p->ldsBytes, st, p->d_desc, p->useLds);
    MORB_HIP_CHECK(hipGetLastError());
    return MORB_OK;
  }
  const BaDev& h = p->h;
  const BaDev* d = p->d_desc;
  const int rb = p->redBlocks;
  double* part0 = h.redPart;
  double* part1 = h.redPart + rb;
  {
    // ---- grid mode, LM control flow on the device: the host queues trial after trial, one trial ahead of the decisions, and
    // stops when the mapped `done` word says so; kernels queued behind the last decision return at once ----
    const int kfBlocks = div_up(std::max(h.nChunks, 1), 4);
    volatile int* hostw = p->h_stop;
    auto forwardStop = [&]() { if (p->userStop && *p->userStop) __atomic_store_n(p->h_stop + 1, 1, __ATOMIC_RELEASE); };
    __atomic_store_n(p->h_stop + 1, 0, __ATOMIC_RELAXED);
    for (int k = 4; k < 8; ++k) __atomic_store_n(p->h_stop + k, 0, __ATOMIC_RELAXED);
    forwardStop();
    hipLaunchKernelGGL(k_g_chi2, dim3(rb), dim3(GB), 0, st, d, part0, (const double*)part1, 2);
    constexpr int kAhead = 1;   // trials queued beyond the last decided one (the seven launches of a trial as one hipGraph: 0.89 -> 0.95 ms per solve, profiles/r04/README.md)
    for (int slot = 0; slot < 100; ++slot) {
      // buildSystem (runs only when the previous trial was accepted): keyframe chunks and map points in one launch
      if (h.rig) hipLaunchKernelGGL(k_g_build<true>, dim3(kfBlocks + div_up(h.nMP, GB / MP_LANES)), dim3(GB), 0, st, d, kfBlocks);
      else hipLaunchKernelGGL(k_g_build<false>, dim3(kfBlocks + div_up(h.nMP, GB / MP_LANES)), dim3(GB), 0, st, d, kfBlocks);
      hipLaunchKernelGGL(k_g_dinv_push, dim3(rb > div_up(h.P * h.P, GB) ? rb : div_up(h.P * h.P, GB)), dim3(GB), 0, st, d, 0.0, h.HsG, 0, 1);
      hipLaunchKernelGGL(morbschur::k_schur_mfma, dim3(p->schur.nblk, p->schur.nsplit), dim3(64), 0, st, (const double*)h.sWD,
                         (const double*)h.sW, p->schur.Mp, p->schur.ksteps, p->schur.stepsPerSplit, h.sBlocks, h.sPart, (const int*)(h.lmi + LM_DONE));
      hipLaunchKernelGGL(k_g_schur_finish, dim3(div_up(4 * (h.P * h.P + h.P), GB)), dim3(GB), 0, st, d, 0.0, h.HsG, 1);
      if (p->denseLds) hipLaunchKernelGGL(k_g_ldlt_lds, dim3(1), dim3(morbdense::LT), p->denseLds, st, d, (const double*)h.HsG, 1);
      else hipLaunchKernelGGL(k_g_ldlt_global, dim3(1), dim3(morbdense::GT), p->globalLds, st, d, h.HsG, p->d_ldws, p->panelInLds, 1);
      hipLaunchKernelGGL(k_g_backsub_update_w, dim3(rb), dim3(GB), 0, st, d, part1);
      hipLaunchKernelGGL(k_g_chi2, dim3(rb), dim3(GB), 0, st, d, part0, (const double*)part1, 1);
      MORB_HIP_CHECK(hipGetLastError());
      // wait until all but the last kAhead queued trials are decided (or the solve is done) on the mapped host words, backing off in tiers: a
      // trial takes ~110 us, so the first ~30 us are `pause` spins (the decision of a short trial is picked up at once), then the thread yields
      // its core between looks (LocalMapping's thread no longer holds a core against Tracking's for the whole solve), and a wait that outlasts
      // 2 ms — a solve stuck behind other work on the device — sleeps 50 us at a time
      unsigned spins = 0;
      const auto tWait = std::chrono::steady_clock::now();
      while (!__atomic_load_n(hostw + 5, __ATOMIC_ACQUIRE) && __atomic_load_n(hostw + 4, __ATOMIC_ACQUIRE) < slot + 1 - kAhead) {
        forwardStop();
        if ((++spins & 0x3FFu) == 0) {
          const hipError_t q = hipStreamQuery(st);
          if (q == hipSuccess) break;   // (everything queued has run: the words are final)
          if (q != hipErrorNotReady) {  // a kernel fault: the words will never change
            set_error("LocalBundleAdjustment: %s while waiting for the LM decision", hipGetErrorString(q));
            return MORB_ERR_HIP;
          }
        }
        if (spins < 2048) __builtin_ia32_pause();
        else if ((spins & 0xFF) != 0 || std::chrono::steady_clock::now() - tWait < std::chrono::milliseconds(2)) sched_yield();
        else { struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }
      }
      if (__atomic_load_n(hostw + 5, __ATOMIC_ACQUIRE)) break;
    }
    hipLaunchKernelGGL(k_g_finish, dim3(rb), dim3(GB), 0, st, d, -1, -1);
    MORB_HIP_CHECK(hipGetLastError());
    // slot limit reached without a `done`: decisions of this solve may still be on their way — let them land before the next solve resets
    // the mirror words (the usual exit has seen `done`, after which no kernel writes them)
    if (!__atomic_load_n(hostw + 5, __ATOMIC_ACQUIRE)) MORB_HIP_CHECK(hipStreamSynchronize(st));
    return MORB_OK;
  }
}

int morb_ba_schur_profile(morb_ba_problem* p, int iters, float* msPerLaunch, double* flops, double* usefulFlops) {
  MORB_REQUIRE(p && iters > 0 && msPerLaunch && flops && usefulFlops, MORB_ERR_INVALID, "bad argument");
  MORB_HIP_CHECK(hipSetDevice(p->opt->device));
  hipStream_t st = p->opt->stream;
  hipEvent_t e0, e1;
  MORB_HIP_CHECK(hipEventCreate(&e0)); MORB_HIP_CHECK(hipEventCreate(&e1));
  const morbschur::Plan& sp = p->schur;
  auto launch = [&]() { hipLaunchKernelGGL(morbschur::k_schur_mfma, dim3(sp.nblk, sp.nsplit), dim3(64), 0, st, (const double*)p->h.sWD, (const double*)p->h.sW, sp.Mp, sp.ksteps, sp.stepsPerSplit, p->h.sBlocks, p->h.sPart, (const int*)nullptr); };
  launch();
  MORB_HIP_CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) launch();
  MORB_HIP_CHECK(hipEventRecord(e1, st));
  MORB_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  MORB_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *msPerLaunch = ms / iters;
  *flops = 2.0 * sp.nblk * morbschur::SB * morbschur::SB * (double)sp.nsplit * sp.stepsPerSplit * 4;
  *usefulFlops = 2.0 * 6 * 3 * (3 + 6) * (double)p->nPairEntries;   // per (e1, e2) entry: B1 D^-1 (6x3x3) and (B1 D^-1) B2^T (6x3x6)
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return MORB_OK;
}

int morb_ba_results(morb_ba_problem* p, float* kfPose, float* mpPos, uint8_t* eraseFlag, int* stats2) {
  MORB_REQUIRE(p, MORB_ERR_INVALID, "NULL problem");
  MORB_HIP_CHECK(hipSetDevice(p->opt->device));
  // the solve may have run on a caller's stream: wait for THAT stream (not for the device: with the reference's threading a tracked
  // frame's optimisation on another handle must not wait for this solve, nor this copy for it), then copy on the handle's own stream
  // (a copy on the null stream would also wait for, and hold up, every other handle's blocking stream)
  if (p->solved) MORB_HIP_CHECK(hipEventSynchronize(p->solved));
  hipStream_t st = p->opt->stream;
  if (kfPose) MORB_HIP_CHECK(hipMemcpyAsync(kfPose, p->h.poseIO, sizeof(float) * 7 * p->h.nKF, hipMemcpyDeviceToHost, st));
  if (mpPos) MORB_HIP_CHECK(hipMemcpyAsync(mpPos, p->h.ptIO, sizeof(float) * 3 * p->h.nMP, hipMemcpyDeviceToHost, st));
  if (eraseFlag) MORB_HIP_CHECK(hipMemcpyAsync(eraseFlag, p->h.erase, p->h.nE, hipMemcpyDeviceToHost, st));
  if (stats2) MORB_HIP_CHECK(hipMemcpyAsync(stats2, p->h.stats, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
  MORB_HIP_CHECK(hipStreamSynchronize(st));
  return MORB_OK;
}

int morb_local_bundle_adjustment(morb_optimizer* o, int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos,
                                 int nE, const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2,
                                 float fx, float fy, float cx, float cy, float bf, int lambdaInit100,
                                 const unsigned char* stopFlag, uint8_t* eraseFlag, int* stats2) {
  if (stopFlag && *(const volatile unsigned char*)stopFlag) { if (stats2) stats2[0] = stats2[1] = 0; return MORB_OK; }  // :1355-1356
  morb_ba_problem* p = nullptr;
  MORB_REQUIRE(o, MORB_ERR_INVALID, "NULL optimizer");
  o->arenaCreate = true;   // the problem lives in the handle's workspace for the duration of this call
  int rc = morb_ba_problem_create(o, &p, nKF, kfPose, kfFixed, nMP, mpPos, nE, eKF, eMP, eObs, eInvSigma2, fx, fy, cx, cy, bf,
                                  lambdaInit100);
  o->arenaCreate = false;
  if (rc != MORB_OK) return rc;
  p->userStop = stopFlag;   // optimizer.setForceStopFlag(pbStopFlag) (:1142): the LM loop polls the caller's flag at every iteration and trial
  rc = morb_ba_solve(p, nullptr);
  if (rc == MORB_OK) rc = morb_ba_results(p, kfPose, mpPos, eraseFlag, stats2);
  morb_ba_problem_destroy(p);
  return rc;
}

int morb_local_bundle_adjustment_fisheye(morb_optimizer* o, int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos,
                                         int nE, const int* eKF, const int* eMP, const float* eObs2, const uint8_t* eRight,
                                         const float* eInvSigma2, const float* camL8, const float* camR8, const float* Trl7,
                                         int lambdaInit100, const unsigned char* stopFlag, uint8_t* eraseFlag, int* stats2) {
  if (stopFlag && *(const volatile unsigned char*)stopFlag) { if (stats2) stats2[0] = stats2[1] = 0; return MORB_OK; }  // :1355-1356
  morb_ba_problem* p = nullptr;
  MORB_REQUIRE(o, MORB_ERR_INVALID, "NULL optimizer");
  o->arenaCreate = true;
  int rc = morb_ba_problem_create_fisheye(o, &p, nKF, kfPose, kfFixed, nMP, mpPos, nE, eKF, eMP, eObs2, eRight, eInvSigma2, camL8,
                                          camR8, Trl7, lambdaInit100);
  o->arenaCreate = false;
  if (rc != MORB_OK) return rc;
  p->userStop = stopFlag;   // optimizer.setForceStopFlag(pbStopFlag) (:1142): the LM loop polls the caller's flag at every iteration and trial
  rc = morb_ba_solve(p, nullptr);
  if (rc == MORB_OK) rc = morb_ba_results(p, kfPose, mpPos, eraseFlag, stats2);
  morb_ba_problem_destroy(p);
  return rc;
}

}  // extern "C"
