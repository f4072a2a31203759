// Visual-inertial tracking slice of SURVEY §8(f) row N1 for MI355X (gfx950):
//   IMU::Preintegrated::IntegrateNewMeasurement        (reference src/ImuTypes.cc:191-247)      -> k_imu_preintegrate, one thread per
//                                                        measurement sequence (the recursion is sequential; frames are independent)
//   Optimizer::PoseInertialOptimizationLastKeyFrame    (reference src/Optimizer.cc:4391-4757)   -> k_pose_inertial, one workgroup per
//                                                        frame, the whole 4 x 10 Gauss-Newton schedule in ONE launch
// g2o semantics kept: Gauss-Newton (computeActiveErrors, buildSystem, dense LDL^T, update; optimization_algorithm_gauss_newton.cpp:
// 51-96), Huber weights through rho' (base_unary_edge.hpp:43-72), inlier edges keep the error of the state BEFORE the last update,
// outlier edges are re-evaluated at the final state (Optimizer.cc:4628-4640), float chi2 comparisons.
// State layout (floats at the boundary, FP64 inside): Rwb (9, row-major), twb (3), velocity (3), gyro bias (3), acc bias (3).
// The visual edges (hundreds per frame) are spread over the workgroup and reduced with wave shuffles; the 9-D inertial edge, the
// two random-walk edges and the 15 x 15 solve are wave-uniform arithmetic every thread repeats in registers (no broadcast of
// the state is needed: all threads apply the same update).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "common.h"
#include "internal_abi.h"
#include "libm_f32.h"
#include "kb8.h"
#include "dense_ldlt.h"
#include "schur_mfma.h"
#include "wave.h"

using namespace morb;

struct morb_optimizer;
extern "C" {
void* morb_optimizer_stream(const morb_optimizer*);
}

#define WAVE_SYNC_F()                                      \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

namespace {

// ---- FP32 3x3 helpers (the reference preintegrates in float) ---------------------------------------------------------------
__device__ __forceinline__ void mul33f(const float* A, const float* B, float* C) {
  float T[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
#pragma unroll
  for (int k = 0; k < 9; ++k) C[k] = T[k];
}
__device__ __forceinline__ void mul3vf(const float* A, const float* v, float* o) {
  float t[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = A[r * 3] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
__device__ __forceinline__ void hatf(const float* v, float* W) {
  W[0] = 0; W[1] = -v[2]; W[2] = v[1]; W[3] = v[2]; W[4] = 0; W[5] = -v[0]; W[6] = -v[1]; W[7] = v[0]; W[8] = 0;
}
// NormalizeRotation (ImuTypes.cc:35-39: U V^T of the SVD) = polar factor; Newton iteration X <- (X + X^-T) / 2 in FP64
__device__ void normalize_rotation_f(float* R) {
  double X[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) X[k] = R[k];
  for (int it = 0; it < 3; ++it) {   // quadratic convergence from a float-rounded rotation: 1e-7 -> 1e-14 -> below FP64 rounding
    const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
    const double inv = 1.0 / (X[0] * c00 + X[1] * c01 + X[2] * c02);
    const double C[9] = {c00, c01, c02,
                         X[2] * X[7] - X[1] * X[8], X[0] * X[8] - X[2] * X[6], X[1] * X[6] - X[0] * X[7],
                         X[1] * X[5] - X[2] * X[4], X[2] * X[3] - X[0] * X[5], X[0] * X[4] - X[1] * X[3]};
#pragma unroll
    for (int k = 0; k < 9; ++k) X[k] = 0.5 * (X[k] + C[k] * inv);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = (float)X[k];
}

// One wave per measurement sequence (the recursion over the samples is sequential, sequences are independent).  Lane 0 carries the
// 3 x 3 state (rotation, velocity, position, bias Jacobians) and writes the sample's A (9 x 9) and B (9 x 6) to LDS; the 64 lanes
// then share the two dense 9 x 9 covariance products, each entry summed in the reference's order.
__global__ __launch_bounds__(64) void k_imu_preintegrate(int nseq, const int* __restrict__ start, const float* __restrict__ acc,
                                                         const float* __restrict__ gyro, const float* __restrict__ dts,
                                                         const float* __restrict__ bias, morb_imu_preintegrated calib,
                                                         morb_imu_preintegrated* __restrict__ out) {
  __shared__ float sA[81], sB[54], sC[81], sAC[81];
  const int s = blockIdx.x, lane = threadIdx.x;
  if (s >= nseq) return;
  float dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, dV[3] = {0, 0, 0}, dP[3] = {0, 0, 0};
  float JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], avgA[3] = {0, 0, 0}, avgW[3] = {0, 0, 0}, b[6], dT = 0.f, walk[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < 9; ++k) { JRg[k] = 0; JVg[k] = 0; JVa[k] = 0; JPg[k] = 0; JPa[k] = 0; }
  for (int k = 0; k < 6; ++k) b[k] = bias[6 * s + k];
  for (int k = lane; k < 81; k += 64) sC[k] = 0.f;
  const int i0 = start[s], i1 = start[s + 1];
  for (int i = i0; i < i1; ++i) {
    const float dt = dts[i];
    if (lane == 0) {
      const float a[3] = {acc[3 * i] - b[0], acc[3 * i + 1] - b[1], acc[3 * i + 2] - b[2]};
      const float wv[3] = {gyro[3 * i] - b[3], gyro[3 * i + 1] - b[4], gyro[3 * i + 2] - b[5]};
      float Racc[3];
      mul3vf(dR, a, Racc);
      for (int k = 0; k < 3; ++k) {
        avgA[k] = (dT * avgA[k] + Racc[k] * dt) / (dT + dt);
        avgW[k] = (dT * avgW[k] + wv[k] * dt) / (dT + dt);
      }
      for (int k = 0; k < 3; ++k) {
        dP[k] = dP[k] + dV[k] * dt + 0.5f * Racc[k] * dt * dt;
        dV[k] = dV[k] + Racc[k] * dt;
      }
      // scalar factors where the C++ expressions apply them (:217-226): -dR * dt * Wacc = ((-dR) * dt) * Wacc,
      // 0.5f * dR * dt * dt * Wacc * JRg = ((((0.5f * dR) * dt) * dt) * Wacc) * JRg
      float Wacc[9], Rdt[9], Rhdt2[9], RdtW[9], Rhdt2W[9], RdtWJ[9], Rhdt2WJ[9];
      hatf(a, Wacc);
      for (int k = 0; k < 9; ++k) { Rdt[k] = dR[k] * dt; Rhdt2[k] = 0.5f * dR[k] * dt * dt; }
      mul33f(Rdt, Wacc, RdtW);
      mul33f(Rhdt2, Wacc, Rhdt2W);
      mul33f(RdtW, JRg, RdtWJ);
      mul33f(Rhdt2W, JRg, Rhdt2WJ);
      // A = [dRi^T 0 0; -dR dt Wacc, I, 0; -dR dt^2/2 Wacc, I dt, I],  B = [rightJ dt, 0; 0, dR dt; 0, dR dt^2/2]
      for (int k = 0; k < 81; ++k) sA[k] = 0.f;
      for (int k = 0; k < 54; ++k) sB[k] = 0.f;
      for (int k = 0; k < 9; ++k) sA[k * 9 + k] = 1.f;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          sA[(3 + r) * 9 + c] = -RdtW[r * 3 + c]; sA[(6 + r) * 9 + c] = -Rhdt2W[r * 3 + c];
          sB[(3 + r) * 6 + 3 + c] = Rdt[r * 3 + c]; sB[(6 + r) * 6 + 3 + c] = Rhdt2[r * 3 + c];
        }
      for (int k = 0; k < 3; ++k) sA[(6 + k) * 9 + 3 + k] = dt;
      for (int k = 0; k < 9; ++k) {
        JPa[k] = JPa[k] + JVa[k] * dt - Rhdt2[k];
        JPg[k] = JPg[k] + JVg[k] * dt - Rhdt2WJ[k];
        JVa[k] = JVa[k] - Rdt[k];
        JVg[k] = JVg[k] - RdtWJ[k];
      }
      // IntegratedRotation (ImuTypes.cc:84-107)
      float dRi[9], rJ[9];
      {
        const float x = wv[0] * dt, y = wv[1] * dt, z = wv[2] * dt;
        const float d2 = x * x + y * y + z * z, d = sqrtf(d2);
        const float v[3] = {x, y, z};
        float W[9], WW[9];
        hatf(v, W);
        mul33f(W, W, WW);
        if (d < 1e-4f) {
          for (int k = 0; k < 9; ++k) { dRi[k] = ((k & 3) == 0 ? 1.f : 0.f) + W[k]; rJ[k] = (k & 3) == 0 ? 1.f : 0.f; }
        } else {
          const float sn = morbm::sinf_glibc(d), cs = morbm::cosf_glibc(d);
          for (int k = 0; k < 9; ++k) {
            const float I = (k & 3) == 0 ? 1.f : 0.f;
            dRi[k] = I + W[k] * sn / d + WW[k] * (1.0f - cs) / d2;
            rJ[k] = I - W[k] * (1.0f - cs) / d2 + WW[k] * (d - sn) / (d2 * d);
          }
        }
      }
      mul33f(dR, dRi, dR);
      normalize_rotation_f(dR);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { sA[r * 9 + c] = dRi[c * 3 + r]; sB[r * 6 + c] = rJ[r * 3 + c] * dt; }
      float dRiT[9], t9[9];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dRiT[r * 3 + c] = dRi[c * 3 + r];
      mul33f(dRiT, JRg, t9);
      for (int k = 0; k < 9; ++k) JRg[k] = t9[k] - rJ[k] * dt;
      dT += dt;
      for (int k = 0; k < 6; ++k) walk[k] += calib.ngaWalk[k];
    }
    WAVE_SYNC_F();
    // C(0:9,0:9) = A C A^T + B Nga B^T, dense like the reference (zeros included, k ascending)
    for (int idx = lane; idx < 81; idx += 64) {
      const int r = idx / 9, c = idx - r * 9;
      float sum = 0;
      for (int k = 0; k < 9; ++k) sum += sA[r * 9 + k] * sC[k * 9 + c];
      sAC[idx] = sum;
    }
    WAVE_SYNC_F();
    float nv[2] = {0.f, 0.f};
    for (int q = 0, idx = lane; idx < 81; idx += 64, ++q) {
      const int r = idx / 9, c = idx - r * 9;
      float sum = 0;
      for (int k = 0; k < 9; ++k) sum += sAC[r * 9 + k] * sA[c * 9 + k];
      float t = 0;
      for (int k = 0; k < 6; ++k) t += sB[r * 6 + k] * calib.nga[k] * sB[c * 6 + k];
      nv[q] = sum + t;
    }
    WAVE_SYNC_F();
    for (int q = 0, idx = lane; idx < 81; idx += 64, ++q) sC[idx] = nv[q];
    WAVE_SYNC_F();
  }
  morb_imu_preintegrated* o = out + s;
  for (int k = lane; k < 225; k += 64) {
    const int r = k / 15, c = k - r * 15;
    o->C[k] = (r < 9 && c < 9) ? sC[r * 9 + c] : 0.f;
  }
  WAVE_SYNC_F();
  if (lane == 0) {
    o->dT = dT;
    for (int k = 0; k < 9; ++k) { o->dR[k] = dR[k]; o->JRg[k] = JRg[k]; o->JVg[k] = JVg[k]; o->JVa[k] = JVa[k]; o->JPg[k] = JPg[k]; o->JPa[k] = JPa[k]; }
    for (int k = 0; k < 3; ++k) { o->dV[k] = dV[k]; o->dP[k] = dP[k]; o->avgA[k] = avgA[k]; o->avgW[k] = avgW[k]; }
    for (int k = 0; k < 6; ++k) { o->b[k] = b[k]; o->nga[k] = calib.nga[k]; o->ngaWalk[k] = calib.ngaWalk[k]; o->C[(9 + k) * 15 + 9 + k] = walk[k]; }
  }
}

// ---- FP64 pieces ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mul33(const double* A, const double* B, double* C) {
  double T[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
#pragma unroll
  for (int k = 0; k < 9; ++k) C[k] = T[k];
}
__device__ __forceinline__ void mul3v(const double* A, const double* v, double* o) {
  double t[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = A[r * 3] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
__device__ __forceinline__ void transpose33(const double* A, double* T) {
  double t[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) t[r * 3 + c] = A[c * 3 + r];
#pragma unroll
  for (int k = 0; k < 9; ++k) T[k] = t[k];
}
__device__ void exp_so3(const double* w, double* R) {   // ExpSO3, G2oTypes.cc:783-796
  const double x = w[0], y = w[1], z = w[2];
  const double d2 = x * x + y * y + z * z, d = sqrt(d2);
  const double W[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  double WW[9];
  mul33(W, W, WW);
  if (d < 1e-5) {
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = ((k & 3) == 0 ? 1.0 : 0.0) + W[k] + 0.5 * WW[k];
  } else {
    const double s = sin(d), c = cos(d);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = ((k & 3) == 0 ? 1.0 : 0.0) + W[k] * s / d + WW[k] * (1.0 - c) / d2;
  }
}
__device__ void log_so3(const double* R, double* w) {   // LogSO3, G2oTypes.cc:798-811
  const double tr = R[0] + R[4] + R[8];
  w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
  const double costheta = (tr - 1.0) * 0.5f;
  if (costheta > 1 || costheta < -1) return;
  const double theta = acos(costheta), s = sin(theta);
  if (fabs(s) < 1e-5) return;
  for (int k = 0; k < 3; ++k) w[k] = theta * w[k] / s;
}
__device__ void inv_right_jacobian_so3(const double* v, double* J) {   // G2oTypes.cc:817-829
  const double x = v[0], y = v[1], z = v[2];
  const double d2 = x * x + y * y + z * z, d = sqrt(d2);
  const double W[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  if (d < 1e-5) { for (int k = 0; k < 9; ++k) J[k] = (k & 3) == 0 ? 1.0 : 0.0; return; }
  double WW[9];
  mul33(W, W, WW);
  const double k2 = 1.0 / d2 - (1.0 + cos(d)) / (2.0 * d * sin(d));
  for (int k = 0; k < 9; ++k) J[k] = ((k & 3) == 0 ? 1.0 : 0.0) + W[k] / 2 + WW[k] * k2;
}

__device__ __forceinline__ double wave_sum(double v) { return morbwave::sum_f64(v); }   // DPP, no LDS-crossbar round trips

__device__ void right_jacobian_so3(const double* v, double* J) {   // G2oTypes.cc:835-848
  const double x = v[0], y = v[1], z = v[2];
  const double d2 = x * x + y * y + z * z, d = sqrt(d2);
  const double W[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  if (d < 1e-5) { for (int k = 0; k < 9; ++k) J[k] = (k & 3) == 0 ? 1.0 : 0.0; return; }
  double WW[9];
  mul33(W, W, WW);
  for (int k = 0; k < 9; ++k) J[k] = ((k & 3) == 0 ? 1.0 : 0.0) - W[k] * (1.0 - cos(d)) / d2 + WW[k] * (d - sin(d)) / (d2 * d);
}

// ---- per-frame workspace in LDS ------------------------------------------------------------------------------------------------
// The visual edges are spread over the 256 threads; the few dense edges (inertial 9 x 24, prior 15 x 15) are written into LDS by
// one thread and multiplied out by all of them; the dense solve runs in wave 0 on the LDS matrix.
struct InertialWork {
  double H[30 * 30], b[30], x[32];
  double J[9 * 24], OJ[9 * 24], e[9], Oe[9];          // inertial edge (columns in edge order P1 V1 G1 A1 P2 V2)
  double Jp[15 * 15], OJp[15 * 15], ep[15], Oep[15];  // prior edge (last-frame variant); scratch for the one-off 9 x 9 / 15 x 15 work
  double InfoI[81], InfoG[9], InfoA[9];
  double pH[225];                                     // prior information
  double red[4][28];
  double delta[18];                                   // dR dV dP dbg of the inertial edge while state 1 is fixed
  double wPrior;
  int cnt[4][2];
  int flag;
};

// n x n inverse (Gauss-Jordan, partial pivoting); M is n x 2n scratch.  One thread.
__device__ bool invert_n(const double* A, int n, double* Ainv, double* M) {
  for (int r = 0; r < n; ++r) for (int c = 0; c < 2 * n; ++c) M[r * 2 * n + c] = c < n ? A[r * n + c] : (c - n == r ? 1.0 : 0.0);
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r) if (fabs(M[r * 2 * n + c]) > fabs(M[p * 2 * n + c])) p = r;
    if (M[p * 2 * n + c] == 0.0) return false;
    if (p != c) for (int k = 0; k < 2 * n; ++k) { const double t = M[p * 2 * n + k]; M[p * 2 * n + k] = M[c * 2 * n + k]; M[c * 2 * n + k] = t; }
    const double inv = 1.0 / M[c * 2 * n + c];
    for (int k = 0; k < 2 * n; ++k) M[c * 2 * n + k] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = M[r * 2 * n + c];
      if (f != 0.0) for (int k = 0; k < 2 * n; ++k) M[r * 2 * n + k] -= f * M[c * 2 * n + k];
    }
  }
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) Ainv[r * n + c] = M[r * 2 * n + n + c];
  return true;
}
// S (n x n, symmetric) <- V max(e, clamp) V^T with eigenvalues below `thr` set to zero (EdgeInertial's information,
// G2oTypes.cc:484-491; ConstraintPoseImu, G2oTypes.h:715-720).  The rebuild only changes S when such an eigenvalue exists:
// S - thr I positive definite (LDL^T test) <=> none does; the cyclic-Jacobi rebuild runs otherwise.  One thread; A, V: n x n scratch.
__device__ void clamp_eigenvalues(double* S, int n, double thr, double* A, double* V) {
  bool pd = true;
  for (int k = 0; k < n * n; ++k) A[k] = S[k];
  for (int k = 0; k < n; ++k) A[k * n + k] -= thr;
  for (int j = 0; j < n && pd; ++j) {
    const double d = A[j * n + j];
    if (!(d > 0)) { pd = false; break; }
    for (int r = n - 1; r > j; --r) {       // descending rows: A[c][j] for c < r is still the un-scaled column entry
      const double l = A[r * n + j] / d;
      for (int c = j + 1; c <= r; ++c) A[r * n + c] -= l * A[c * n + j];
    }
  }
  if (pd) return;
  for (int k = 0; k < n * n; ++k) { A[k] = S[k]; V[k] = 0.0; }
  for (int k = 0; k < n; ++k) V[k * n + k] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) { const double t = A[r * n + c] * A[r * n + c]; if (r == c) diag += t; else off += t; }
    if (off <= 1e-30 * diag) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (A[p * n + q] == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * A[p * n + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) { const double a = A[k * n + p], bq = A[k * n + q]; A[k * n + p] = c * a - s * bq; A[k * n + q] = s * a + c * bq; }
        for (int k = 0; k < n; ++k) { const double a = A[p * n + k], bq = A[q * n + k]; A[p * n + k] = c * a - s * bq; A[q * n + k] = s * a + c * bq; }
        for (int k = 0; k < n; ++k) { const double a = V[k * n + p], bq = V[k * n + q]; V[k * n + p] = c * a - s * bq; V[k * n + q] = s * a + c * bq; }
      }
  }
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c) {
      double s = 0;
      for (int k = 0; k < n; ++k) { const double e = A[k * n + k] < thr ? 0.0 : A[k * n + k]; s += V[r * n + k] * e * V[c * n + k]; }
      S[r * n + c] = s;
    }
}

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

// LinearSolverDense (linear_solver_dense.h:104-112): LDL^T solve, solution only when every pivot is positive.  ONE wave; lane r
// keeps row r of the matrix in registers, column entries of other rows arrive through v_readlane (the loops are fully
// unrolled, so every register index and every source lane is a compile-time constant): no LDS round trip and no barrier in
// the factorisation or the forward substitution.  The backward substitution needs L transposed: the rows go to LDS once.
template <int N>
__device__ bool wave_ldlt_solve(const double* H, int ld, const double* rhs, double* x, double* Lscr, int lane) {
  const int row = lane < N ? lane : N - 1;   // lanes >= N replicate the last row; their results are not used
  double a[N];
#pragma unroll
  for (int c = 0; c < N; ++c) a[c] = H[row * ld + c];
  double y = rhs[row], diag = 1.0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double d = morbwave::readlane_f64(a[j], j);
    ok = ok && (d > 0);
    if (row == j) diag = d;
    const double l = a[j] / d;
#pragma unroll
    for (int c = j + 1; c < N; ++c) a[c] -= l * morbwave::readlane_f64(a[j], c);   // A[r][c] -= A[r][j] A[c][j] / d
    a[j] = l;
  }
  if (!ok) return false;   // wave-uniform
#pragma unroll
  for (int j = 0; j < N - 1; ++j) {   // L y = b
    const double yj = morbwave::readlane_f64(y, j);
    if (row > j) y -= a[j] * yj;
  }
  y /= diag;
#pragma unroll
  for (int c = 0; c < N; ++c) Lscr[row * N + c] = a[c];
  WAVE_SYNC();
#pragma unroll
  for (int j = N - 1; j > 0; --j) {   // L^T x = y
    const double xj = morbwave::readlane_f64(y, j);
    if (row < j) y -= Lscr[j * N + row] * xj;
  }
  if (lane < N) x[lane] = y;
  WAVE_SYNC();
  return true;
}

// camera 0 = the pinhole camera, or on a fisheye rig (rig != 0) the left KannalaBrandt8 camera with camera 1 = the right one
// (ImuCamPose, G2oTypes.cc:74-118: Rcb[1] = Rrl Rcb[0], tcb[1] = Rrl tcb[0] + trl)
struct CamGeom {
  double Rcb[9], tcb[3], Rbc[9], tbc[3], bf;
  float fx, fy, cx, cy;
  int rig;
  float kb[2][8];
  double Rcb1[9], tcb1[3], Rbc1[9], tbc1[3];
};

struct VIState {
  double Rwb[9], twb[3], v[3], bg[3], ba[3];
  double Rcw[9], tcw[3];
  double Rcw1[9], tcw1[3];   // right camera of a rig
};
template <bool RIG>
__device__ __forceinline__ void refresh_camera_t(const CamGeom& g, VIState& S) {   // G2oTypes.cc:209-215
  double Rbw[9], tbw[3];
  transpose33(S.Rwb, Rbw);
  mul3v(Rbw, S.twb, tbw);
  for (int k = 0; k < 3; ++k) tbw[k] = -tbw[k];
  mul33(g.Rcb, Rbw, S.Rcw);
  mul3v(g.Rcb, tbw, S.tcw);
  for (int k = 0; k < 3; ++k) S.tcw[k] += g.tcb[k];
  if (RIG) {
    mul33(g.Rcb1, Rbw, S.Rcw1);
    mul3v(g.Rcb1, tbw, S.tcw1);
    for (int k = 0; k < 3; ++k) S.tcw1[k] += g.tcb1[k];
  }
}
// RIG as a template parameter keeps the right camera's pose out of the registers of the pinhole instantiations
__device__ __forceinline__ void refresh_camera(const CamGeom& g, VIState& S) { if (g.rig) refresh_camera_t<true>(g, S); else refresh_camera_t<false>(g, S); }
template <bool RIG>
__device__ __forceinline__ void load_state_t(const CamGeom& g, const float* s0, VIState& S) {
  for (int k = 0; k < 9; ++k) S.Rwb[k] = s0[k];
  for (int k = 0; k < 3; ++k) { S.twb[k] = s0[9 + k]; S.v[k] = s0[12 + k]; S.bg[k] = s0[15 + k]; S.ba[k] = s0[18 + k]; }
  refresh_camera_t<RIG>(g, S);
}
__device__ __forceinline__ void load_state(const CamGeom& g, const float* s0, VIState& S) { if (g.rig) load_state_t<true>(g, s0, S); else load_state_t<false>(g, s0, S); }
template <bool RIG>
__device__ __forceinline__ void apply_update_t(const CamGeom& g, VIState& S, const double* x) {   // ImuCamPose::Update + the additive vertices
  double t[3], dR[9];
  mul3v(S.Rwb, x + 3, t);
  for (int k = 0; k < 3; ++k) S.twb[k] += t[k];
  exp_so3(x, dR);
  mul33(S.Rwb, dR, S.Rwb);
  refresh_camera_t<RIG>(g, S);
  for (int k = 0; k < 3; ++k) { S.v[k] += x[6 + k]; S.bg[k] += x[9 + k]; S.ba[k] += x[12 + k]; }
}
__device__ __forceinline__ void apply_update(const CamGeom& g, VIState& S, const double* x) { if (g.rig) apply_update_t<true>(g, S, x); else apply_update_t<false>(g, S, x); }
// visual edge: error (obs - projection) and chi2; st = stereo
// the edge camera's KannalaBrandt8 parameters picked element by element: `g.kb[cam]` with a per-lane index made the compiler copy the whole
// CamGeom kernel argument to scratch memory to index it (736 B per thread in the rig kernels)
__device__ __forceinline__ void kb_of(const CamGeom& g, int cam, float (&p)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = cam ? g.kb[1][i] : g.kb[0][i];
}
template <bool RIG>
__device__ __forceinline__ double vis_error_t(const CamGeom& g, const VIState& S, const double* X, const float* o, bool st, double info,
                                              double* err, double* Xc, int cam) {
  if (RIG && cam) { mul3v(S.Rcw1, X, Xc); for (int k = 0; k < 3; ++k) Xc[k] += S.tcw1[k]; }
  else { mul3v(S.Rcw, X, Xc); for (int k = 0; k < 3; ++k) Xc[k] += S.tcw[k]; }
  double u, v;
  if (RIG) { double uv[2]; float kp[8]; kb_of(g, cam, kp); morbkb8::kb8_project_d(kp, Xc, uv); u = uv[0]; v = uv[1]; }   // KannalaBrandt8::project(Vector3d)
  else { u = g.fx * Xc[0] / Xc[2] + g.cx; v = g.fy * Xc[1] / Xc[2] + g.cy; }                       // Pinhole.cpp:38-44
  err[0] = (double)o[0] - u; err[1] = (double)o[1] - v; err[2] = 0;
  double c = err[0] * info * err[0] + err[1] * info * err[1];
  if (st) { const double invZ = 1 / Xc[2]; err[2] = (double)o[2] - (u - g.bf * invZ); c += err[2] * info * err[2]; }
  return c;
}
__device__ __forceinline__ double vis_error(const CamGeom& g, const VIState& S, const double* X, const float* o, bool st, double info,
                                            double* err, double* Xc, int cam = 0) {
  return g.rig ? vis_error_t<true>(g, S, X, o, st, info, err, Xc, cam) : vis_error_t<false>(g, S, X, o, st, info, err, Xc, 0);
}
// projectJac of the edge's camera (2 x 3 in pj[0..5]); pinhole: Pinhole.cpp:76-86
template <bool RIG>
__device__ __forceinline__ void cam_project_jac_t(const CamGeom& g, const double* Xc, int cam, double* pj) {
  if (RIG) { float kp[8]; kb_of(g, cam, kp); morbkb8::kb8_project_jac(kp, Xc, pj); return; }
  pj[0] = g.fx / Xc[2]; pj[1] = 0; pj[2] = -g.fx * Xc[0] / (Xc[2] * Xc[2]);
  pj[3] = 0; pj[4] = g.fy / Xc[2]; pj[5] = -g.fy * Xc[1] / (Xc[2] * Xc[2]);
}
__device__ __forceinline__ void cam_project_jac(const CamGeom& g, const double* Xc, int cam, double* pj) {
  if (g.rig) cam_project_jac_t<true>(g, Xc, cam, pj); else cam_project_jac_t<false>(g, Xc, 0, pj);
}
template <bool RIG>
__device__ __forceinline__ void vis_jacobian_t(const CamGeom& g, const double* Xc, bool st, double* J /*[3][6]*/, int cam) {   // G2oTypes.cc:361-442
  double Xb[3];
  const bool c1 = RIG && cam;
  // (element-wise selects: a pointer chosen between two members of the kernel-argument struct sends the struct to scratch memory)
  double Rbc[9], tbc[3], Rcb[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { Rbc[k] = c1 ? g.Rbc1[k] : g.Rbc[k]; Rcb[k] = c1 ? g.Rcb1[k] : g.Rcb[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) tbc[k] = c1 ? g.tbc1[k] : g.tbc[k];
  mul3v(Rbc, Xc, Xb);
  for (int k = 0; k < 3; ++k) Xb[k] += tbc[k];
  double pj[9];
  cam_project_jac_t<RIG>(g, Xc, cam, pj);
  pj[6] = pj[7] = pj[8] = 0;
  if (st) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + g.bf * (1.0 / (Xc[2] * Xc[2])); }
  const double x = Xb[0], y = Xb[1], z = Xb[2];
  // J = (proj_jac Rcb) [ -[Xb]x | I ]: SE3deriv = {0, z, -y, 1, 0, 0; -z, 0, x, 0, 1, 0; y, -x, 0, 0, 0, 1} written out — its zeros and ones cost 63 of an
  // edge's ~300 FP64 instructions when multiplied through (x * 0.0 does not fold); the same values (a + 0 = a, a * 1 = a)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double PR[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) PR[c] = pj[r * 3] * Rcb[c] + pj[r * 3 + 1] * Rcb[3 + c] + pj[r * 3 + 2] * Rcb[6 + c];
    J[r * 6 + 0] = PR[1] * -z + PR[2] * y;
    J[r * 6 + 1] = PR[0] * z + PR[2] * -x;
    J[r * 6 + 2] = PR[0] * -y + PR[1] * x;
    J[r * 6 + 3] = PR[0]; J[r * 6 + 4] = PR[1]; J[r * 6 + 5] = PR[2];
  }
}
__device__ __forceinline__ void vis_jacobian(const CamGeom& g, const double* Xc, bool st, double* J, int cam = 0) {
  if (g.rig) vis_jacobian_t<true>(g, Xc, st, J, cam); else vis_jacobian_t<false>(g, Xc, st, J, 0);
}
__device__ __forceinline__ double huber_w(double delta, double e2) {   // rho'(e2), robust_kernel_impl.cpp:65-91
  return e2 <= delta * delta ? 1.0 : delta / sqrt(e2);
}
__device__ __forceinline__ void put33(double* J, int ld, int r0, int c0, const double* B, double sgn) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) J[(r0 + r) * ld + c0 + c] = sgn * B[r * 3 + c];
}
// GetDeltaRotation / GetDeltaVelocity / GetDeltaPosition at the bias (bg1, ba1) (ImuTypes.cc:289-312): FP32 like the reference
__device__ void imu_delta(const morb_imu_preintegrated& P, const double* bg1, const double* ba1, double* dR, double* dV, double* dP,
                          double* dbg3) {
  const float b1[6] = {(float)ba1[0], (float)ba1[1], (float)ba1[2], (float)bg1[0], (float)bg1[1], (float)bg1[2]};
  const float dbg[3] = {b1[3] - P.b[3], b1[4] - P.b[4], b1[5] - P.b[5]};
  const float dba[3] = {b1[0] - P.b[0], b1[1] - P.b[1], b1[2] - P.b[2]};
  float w[3], W[9], WW[9], E[9], dRf[9];
  mul3vf(P.JRg, dbg, w);
  const float t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrtf(t2);
  hatf(w, W);
  mul33f(W, W, WW);
  if (t < 1e-5f) for (int k = 0; k < 9; ++k) E[k] = ((k & 3) == 0 ? 1.f : 0.f) + W[k] + 0.5f * WW[k];
  else { const float sn = morbm::sinf_glibc(t), cs = morbm::cosf_glibc(t); for (int k = 0; k < 9; ++k) E[k] = ((k & 3) == 0 ? 1.f : 0.f) + W[k] * sn / t + WW[k] * (1.0f - cs) / t2; }
  mul33f(P.dR, E, dRf);
  normalize_rotation_f(dRf);
  for (int k = 0; k < 9; ++k) dR[k] = dRf[k];
  float g1[3], a1[3];
  mul3vf(P.JVg, dbg, g1); mul3vf(P.JVa, dba, a1);
  for (int k = 0; k < 3; ++k) dV[k] = (double)(P.dV[k] + g1[k] + a1[k]);
  mul3vf(P.JPg, dbg, g1); mul3vf(P.JPa, dba, a1);
  for (int k = 0; k < 3; ++k) dP[k] = (double)(P.dP[k] + g1[k] + a1[k]);
  for (int k = 0; k < 3; ++k) dbg3[k] = dbg[k];
}
// EdgeInertial between state 1 (S1) and state 2 (S2): error and, when J != nullptr, the 9 x 24 Jacobian (edge column order
// P1 V1 G1 A1 P2 V2) written to LDS.  G2oTypes.cc:494-585.  One thread.
// delta != nullptr: dR dV dP dbg precomputed (state 1 fixed).  full == false: only the (P2, V2) column blocks are written.
// J must have been zeroed once: the same entries are rewritten at every call.
__device__ void inertial_edge(const morb_imu_preintegrated& P, const VIState& S1, const VIState& S2, const double* delta, bool full,
                              double* err, double* J) {
  double dR[9], dV[3], dP[3], dbg[3];
  if (delta) {
    for (int k = 0; k < 9; ++k) dR[k] = delta[k];
    for (int k = 0; k < 3; ++k) { dV[k] = delta[9 + k]; dP[k] = delta[12 + k]; dbg[k] = delta[15 + k]; }
  } else imu_delta(P, S1.bg, S1.ba, dR, dV, dP, dbg);
  const double dt = P.dT;
  double Rbw1[9], dRt[9], M[9], eR[9], er[3];
  transpose33(S1.Rwb, Rbw1);
  transpose33(dR, dRt);
  mul33(dRt, Rbw1, M);
  mul33(M, S2.Rwb, eR);
  log_so3(eR, er);
  const double g[3] = {0, 0, -(double)9.81f};
  double t[3], a1[3], a2[3];
  for (int k = 0; k < 3; ++k) t[k] = S2.v[k] - S1.v[k] - g[k] * dt;
  mul3v(Rbw1, t, a1);
  for (int k = 0; k < 3; ++k) t[k] = S2.twb[k] - S1.twb[k] - S1.v[k] * dt - g[k] * dt * dt / 2;
  mul3v(Rbw1, t, a2);
  for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = a1[k] - dV[k]; err[6 + k] = a2[k] - dP[k]; }
  if (!J) return;
  double invJr[9], T[9], T2[9], W[9];
  inv_right_jacobian_so3(er, invJr);
  // pose 2, velocity 2
  put33(J, 24, 0, 15, invJr, 1.0);
  mul33(Rbw1, S2.Rwb, T);
  put33(J, 24, 6, 18, T, 1.0);
  put33(J, 24, 3, 21, Rbw1, 1.0);
  if (!full) return;
  // pose 1
  transpose33(S2.Rwb, T2);
  mul33(invJr, T2, T); mul33(T, S1.Rwb, T2);
  put33(J, 24, 0, 0, T2, -1.0);
  W[0] = 0; W[1] = -a1[2]; W[2] = a1[1]; W[3] = a1[2]; W[4] = 0; W[5] = -a1[0]; W[6] = -a1[1]; W[7] = a1[0]; W[8] = 0;
  put33(J, 24, 3, 0, W, 1.0);
  {
    double a3[3];
    for (int k = 0; k < 3; ++k) t[k] = S2.twb[k] - S1.twb[k] - S1.v[k] * dt - 0.5 * g[k] * dt * dt;
    mul3v(Rbw1, t, a3);
    W[0] = 0; W[1] = -a3[2]; W[2] = a3[1]; W[3] = a3[2]; W[4] = 0; W[5] = -a3[0]; W[6] = -a3[1]; W[7] = a3[0]; W[8] = 0;
    put33(J, 24, 6, 0, W, 1.0);
  }
  for (int k = 0; k < 3; ++k) J[(6 + k) * 24 + 3 + k] = -1.0;
  // velocity 1
  put33(J, 24, 3, 6, Rbw1, -1.0);
  for (int k = 0; k < 9; ++k) T[k] = Rbw1[k] * dt;
  put33(J, 24, 6, 6, T, -1.0);
  // gyro bias 1
  {
    double JRg[9], w3[3], rj[9], eRt[9];
    for (int k = 0; k < 9; ++k) JRg[k] = P.JRg[k];
    mul3v(JRg, dbg, w3);
    right_jacobian_so3(w3, rj);
    transpose33(eR, eRt);
    mul33(invJr, eRt, T); mul33(T, rj, T2); mul33(T2, JRg, T);
    put33(J, 24, 0, 9, T, -1.0);
    for (int k = 0; k < 9; ++k) T[k] = P.JVg[k];
    put33(J, 24, 3, 9, T, -1.0);
    for (int k = 0; k < 9; ++k) T[k] = P.JPg[k];
    put33(J, 24, 6, 9, T, -1.0);
  }
  // acc bias 1
  for (int k = 0; k < 9; ++k) T[k] = P.JVa[k];
  put33(J, 24, 3, 12, T, -1.0);
  for (int k = 0; k < 9; ++k) T[k] = P.JPa[k];
  put33(J, 24, 6, 12, T, -1.0);
}
// EdgePriorPoseImu (G2oTypes.cc:739-766): error (15) and the 15 x 15 Jacobian in LDS.  prior = Rwb twb v bg ba (21 doubles).
__device__ void prior_edge(const double* prior, const VIState& S1, double* err, double* J) {
  double pRt[9], E[9], er[3], d[3], et[3];
  transpose33(prior, pRt);
  mul33(pRt, S1.Rwb, E);
  log_so3(E, er);
  for (int k = 0; k < 3; ++k) d[k] = S1.twb[k] - prior[9 + k];
  mul3v(pRt, d, et);
  for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = et[k]; err[6 + k] = S1.v[k] - prior[12 + k]; err[9 + k] = S1.bg[k] - prior[15 + k]; err[12 + k] = S1.ba[k] - prior[18 + k]; }
  if (!J) return;   // J zeroed once by the caller
  double invJr[9];
  inv_right_jacobian_so3(er, invJr);
  put33(J, 15, 0, 0, invJr, 1.0);
  put33(J, 15, 3, 3, E, 1.0);
  for (int k = 6; k < 15; ++k) J[k * 15 + k] = 1.0;
}

// system column of an inertial-edge column: frame = [P 0..5, V 6..8, G 9..11, A 12..14], previous frame / keyframe = 15 + the same

// LASTFRAME = false: PoseInertialOptimizationLastKeyFrame (state 1 = the keyframe, fixed: 15 unknowns)
// LASTFRAME = true : PoseInertialOptimizationLastFrame   (state 1 = the previous frame, free, with its prior: 30 unknowns)
#ifdef MORB_INERTIAL_TIMING
__device__ unsigned long long g_inertialPhase[16];
#define IMARK(k) do { __syncthreads(); if (tid == 0 && blockIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_inertialPhase[k], now_ - t0_); t0_ = now_; } } while (0)
#else
#define IMARK(k)
#endif
// GEDGE: the edge list lives in global memory (frames whose cap * 32 bytes do not fit the LDS beside the kernel's static data: ~4900 features) instead of LDS
template <bool LASTFRAME, bool RIG, bool GEDGE>
__global__ __launch_bounds__(256) void k_pose_inertial(int cap, const int* __restrict__ count, const uint8_t* __restrict__ hasMP,
                                                       const float* __restrict__ obs, const float* __restrict__ invSigma2,
                                                       const float* __restrict__ Xw, const uint8_t* __restrict__ closeFlag,
                                                       CamGeom g, const float* __restrict__ state1,
                                                       const morb_imu_preintegrated* __restrict__ pre,
                                                       const morb_imu_preintegrated* __restrict__ preKF,
                                                       const double* __restrict__ prevPrior, const int* __restrict__ nLeft,
                                                       int bRecInit, float* __restrict__ stateIO, uint8_t* __restrict__ outlier,
                                                       int* __restrict__ nInliersOut, double* __restrict__ prior, float* __restrict__ gEdge) {
  __shared__ InertialWork Wk;
  // the round's ACTIVE visual edges (map point present, not an outlier of the round before), compacted in feature order: (u, v, uR | X | info | camera)
  // as eight floats per edge.  The ten Gauss-Newton iterations of a round read them from here: each of a thread's three to five edges used to start
  // with a dependent round trip to global memory for its flags and another for its data (~2 k cycles per edge beside ~2.6 k of arithmetic).
  extern __shared__ __align__(16) float sEdgeLds[];
  float* const sEdge = GEDGE ? gEdge + (size_t)blockIdx.x * cap * 8 : sEdgeLds;   // (a compile-time choice: each instantiation knows its address space)
  __shared__ VIState sS1;   // state 1 (keyframe / previous frame): only the dense-edge threads read it, one thread updates it
  constexpr int NV = LASTFRAME ? 30 : 15;
  // threads on the visual edges; wave 3 evaluates the inertial edge and (last-frame variant) wave 2 the prior edge meanwhile
  constexpr int NVIS = LASTFRAME ? 128 : 192, NVW = NVIS / 64;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = count ? count[f] : cap;
  const size_t base = (size_t)f * cap;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  const morb_imu_preintegrated& P = pre[f];
  const double* pr = LASTFRAME ? prevPrior + (size_t)246 * f : nullptr;
  const int nL = RIG ? nLeft[f] : n;   // fisheye rig: features >= nL are right-camera observations; every edge is monocular

  int nInit = 0;
  for (int i = tid; i < n; i += 256) if (hasMP[base + i]) { ++nInit; outlier[base + i] = 0; }
  nInit = (int)wave_sum((double)nInit);
  if (lane == 0) Wk.cnt[wv][0] = nInit;
  __syncthreads();
  nInit = Wk.cnt[0][0] + Wk.cnt[1][0] + Wk.cnt[2][0] + Wk.cnt[3][0];
  __syncthreads();

  if (tid == 0) {   // informations: EdgeInertial (G2oTypes.cc:484-491), EdgeGyroRW / EdgeAccRW (Optimizer.cc:4580-4594)
    double* C9 = Wk.J;        // scratch: 81 + 162 + 162 doubles fit in J | OJ | Jp | OJp
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C9[r * 9 + c] = (double)P.C[r * 15 + c];
    if (invert_n(C9, 9, Wk.InfoI, Wk.Jp)) {
      for (int r = 0; r < 9; ++r) for (int c = r + 1; c < 9; ++c) { const double s = (Wk.InfoI[r * 9 + c] + Wk.InfoI[c * 9 + r]) / 2; Wk.InfoI[r * 9 + c] = s; Wk.InfoI[c * 9 + r] = s; }
      clamp_eigenvalues(Wk.InfoI, 9, 1e-12, Wk.Jp, Wk.OJp);
    } else for (int k = 0; k < 81; ++k) Wk.InfoI[k] = 0;
    const morb_imu_preintegrated& PK = LASTFRAME ? preKF[f] : P;
    double Cg[9], Ca[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Cg[r * 3 + c] = PK.C[(9 + r) * 15 + 9 + c]; Ca[r * 3 + c] = PK.C[(12 + r) * 15 + 12 + c]; }
    if (!invert_n(Cg, 3, Wk.InfoG, Wk.Jp)) for (int k = 0; k < 9; ++k) Wk.InfoG[k] = 0;
    if (!invert_n(Ca, 3, Wk.InfoA, Wk.Jp)) for (int k = 0; k < 9; ++k) Wk.InfoA[k] = 0;
  }
  if (LASTFRAME) for (int k = tid; k < 225; k += 256) Wk.pH[k] = pr[21 + k];
  __syncthreads();

#ifdef MORB_INERTIAL_TIMING
  unsigned long long t0_ = wall_clock64();
#endif
  IMARK(0);
  VIState S, Sprev;   // Sprev: only its camera poses are ever assigned / read
  load_state_t<RIG>(g, stateIO + 21 * f, S);
  if (tid == 0) load_state_t<RIG>(g, state1 + 21 * f, sS1);
  const VIState& S1 = sS1;
  auto keep_cameras = [&]() {
    for (int k = 0; k < 9; ++k) { Sprev.Rcw[k] = S.Rcw[k]; if (RIG) Sprev.Rcw1[k] = S.Rcw1[k]; }
    for (int k = 0; k < 3; ++k) { Sprev.tcw[k] = S.tcw[k]; if (RIG) Sprev.tcw1[k] = S.tcw1[k]; }
  };
  keep_cameras();
  __syncthreads();   // the setup above used J / Jp as scratch; sS1 is visible
  for (int k = tid; k < 216; k += 256) Wk.J[k] = 0;
  for (int k = tid; k < 225; k += 256) Wk.Jp[k] = 0;
  if (!LASTFRAME && tid == 0) imu_delta(P, S1.bg, S1.ba, Wk.delta, Wk.delta + 9, Wk.delta + 12, Wk.delta + 15);
  __syncthreads();
  const double* delta = LASTFRAME ? nullptr : Wk.delta;
  auto redsum = [&](int q) { double s = 0; for (int w = 0; w < NVW; ++w) s += Wk.red[w][q]; return s; };
  bool robust = true;
  int nBad = 0, nInl = 0;
  for (int k = tid; k < 32; k += 256) Wk.x[k] = 0;
  const float chi2MonoKF[4] = {12.f, 7.5f, 5.991f, 5.991f}, chi2MonoF[4] = {5.991f, 5.991f, 5.991f, 5.991f};
  const float chi2Stereo[4] = {15.6f, 9.8f, 7.815f, 7.815f};

  for (int it = 0; it < 4; ++it) {
    bool ok = true;
    int nAct = 0;
    for (int c0 = 0; c0 < n; c0 += 256) {
      const int i = c0 + tid;
      const bool a = i < n && hasMP[base + i] && !outlier[base + i];
      const unsigned long long m = __ballot(a);
      if (lane == 0) Wk.cnt[wv][0] = __popcll(m);
      __syncthreads();
      int off = nAct, totc = 0;
      for (int w = 0; w < 4; ++w) { const int c = Wk.cnt[w][0]; if (w < wv) off += c; totc += c; }
      if (a) {
        float* r = sEdge + 8 * (size_t)(off + __popcll(m & ((1ull << lane) - 1ull)));
        r[0] = obs[(base + i) * 3]; r[1] = obs[(base + i) * 3 + 1]; r[2] = obs[(base + i) * 3 + 2];
        r[3] = Xw[(base + i) * 3]; r[4] = Xw[(base + i) * 3 + 1]; r[5] = Xw[(base + i) * 3 + 2];
        r[6] = invSigma2[base + i]; r[7] = i >= nL ? 1.0f : 0.0f;
      }
      nAct += totc;
      __syncthreads();
    }
    for (int iter = 0; iter < 10 && ok; ++iter) {
      IMARK(1);
      keep_cameras();   // the state the active edges' errors belong to
      // ---- visual edges -> 6 x 6 block and right-hand side of the frame's pose
      if (tid < NVIS) {
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0;
        for (int e = tid; e < nAct; e += NVIS) {
          const float* o = sEdge + 8 * (size_t)e;
          const bool st = !RIG && !(o[2] < 0);
          const int cam = o[7] != 0.0f ? 1 : 0;
          const double X[3] = {(double)o[3], (double)o[4], (double)o[5]};
          const double info = (double)o[6];
          double err[3], Xc[3], J[18];
          const double c = vis_error_t<RIG>(g, S, X, o, st, info, err, Xc, cam);
          const double w = robust ? huber_w(st ? deltaStereo : deltaMono, c) : 1.0;
          vis_jacobian_t<RIG>(g, Xc, st, J, cam);   // a mono edge has a zero third row and err[2] = 0: one fully unrolled 3-row form (registers only)
          int q = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            double bb = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) bb += J[k * 6 + r] * (info * err[k]);
            acc[21 + r] -= w * bb;
#pragma unroll
            for (int cc = r; cc < 6; ++cc) {
              double h = 0;
#pragma unroll
              for (int k = 0; k < 3; ++k) h += J[k * 6 + r] * (w * info) * J[k * 6 + cc];
              acc[q++] += h;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = wave_sum(acc[k]);
        if (lane == 0) for (int k = 0; k < 27; ++k) Wk.red[wv][k] = acc[k];
      } else if (tid == 192) {
        inertial_edge(P, S1, S, delta, LASTFRAME, Wk.e, Wk.J);   // error + Jacobian to LDS
      } else if (LASTFRAME && tid == 128) {
        prior_edge(pr, S1, Wk.ep, Wk.Jp);
      }
      IMARK(2);
      __syncthreads();
      IMARK(3);
      // ---- Omega J, Omega e
      for (int k = tid; k < 9 * 24 + 9; k += 256) {
        if (k < 216) { const int r = k / 24, c = k - r * 24; double s = 0; for (int l = 0; l < 9; ++l) s += Wk.InfoI[r * 9 + l] * Wk.J[l * 24 + c]; Wk.OJ[k] = s; }
        else { const int r = k - 216; double s = 0; for (int l = 0; l < 9; ++l) s += Wk.InfoI[r * 9 + l] * Wk.e[l]; Wk.Oe[r] = s; }
      }
      if (LASTFRAME) {
        for (int k = tid; k < 225 + 15; k += 256) {
          if (k < 225) { const int r = k / 15, c = k - r * 15; double s = 0; for (int l = 0; l < 15; ++l) s += Wk.pH[r * 15 + l] * Wk.Jp[l * 15 + c]; Wk.OJp[k] = s; }
          else { const int r = k - 225; double s = 0; for (int l = 0; l < 15; ++l) s += Wk.pH[r * 15 + l] * Wk.ep[l]; Wk.Oep[r] = s; }
        }
      }
      __syncthreads();
      if (LASTFRAME && tid == 0) {   // Huber weight of the prior edge (delta 5, Optimizer.cc:4981-4984)
        double chi2 = 0;
        for (int k = 0; k < 15; ++k) chi2 += Wk.ep[k] * Wk.Oep[k];
        Wk.wPrior = huber_w(5.0, chi2);
      }
      if (LASTFRAME) __syncthreads();
      IMARK(4);
      // ---- H and b: every entry by one thread
      for (int k = tid; k < NV * NV + NV; k += 256) {
        if (k < NV * NV) {
          const int r = k / NV, c = k - r * NV;
          double h = 0;
          if (r < 6 && c < 6) {   // visual block (upper triangle was accumulated)
            const int a = r < c ? r : c, bq = r < c ? c : r;
            const int q = a * 6 - a * (a - 1) / 2 + (bq - a);
            h = redsum(q);
          }
          // inertial edge: system index -> edge column
          const int ea = r < 15 ? (r < 9 ? 15 + r : -1) : r - 15, ec = c < 15 ? (c < 9 ? 15 + c : -1) : c - 15;
          if (ea >= 0 && ec >= 0) { double s = 0; for (int l = 0; l < 9; ++l) s += Wk.J[l * 24 + ea] * Wk.OJ[l * 24 + ec]; h += s; }
          // random walks: e = bias2 - bias1 (G2oTypes.h:645-654)
          const int rb = r % 15, cb = c % 15;
          if (rb >= 9 && cb >= 9 && (rb < 12) == (cb < 12)) {
            const double* I3 = rb < 12 ? Wk.InfoG : Wk.InfoA;
            const double v = I3[(rb - (rb < 12 ? 9 : 12)) * 3 + (cb - (cb < 12 ? 9 : 12))];
            h += ((r < 15) == (c < 15)) ? v : -v;
          }
          if (LASTFRAME && r >= 15 && c >= 15) { double s = 0; for (int l = 0; l < 15; ++l) s += Wk.Jp[l * 15 + (r - 15)] * Wk.OJp[l * 15 + (c - 15)]; h += Wk.wPrior * s; }
          Wk.H[r * 30 + c] = h;
        } else {
          const int r = k - NV * NV;
          double s = 0;
          if (r < 6) s = redsum(21 + r);
          const int ea = r < 15 ? (r < 9 ? 15 + r : -1) : r - 15;
          if (ea >= 0) { double t = 0; for (int l = 0; l < 9; ++l) t += Wk.J[l * 24 + ea] * Wk.Oe[l]; s -= t; }
          const int rb = r % 15;
          if (rb >= 9) {
            const double* I3 = rb < 12 ? Wk.InfoG : Wk.InfoA;
            const double* b2 = rb < 12 ? S.bg : S.ba; const double* b1 = rb < 12 ? S1.bg : S1.ba;
            const int q = rb - (rb < 12 ? 9 : 12);
            double t = 0;
            for (int l = 0; l < 3; ++l) t += I3[q * 3 + l] * (b2[l] - b1[l]);
            s += r < 15 ? -t : t;   // J2 = I, J1 = -I
          }
          if (LASTFRAME && r >= 15) { double t = 0; for (int l = 0; l < 15; ++l) t += Wk.Jp[l * 15 + (r - 15)] * Wk.Oep[l]; s -= Wk.wPrior * t; }
          Wk.b[r] = s;
        }
      }
      __syncthreads();
      IMARK(5);
      if (wv == 0) {
        const bool good = wave_ldlt_solve<NV>(Wk.H, 30, Wk.b, Wk.x, Wk.H, lane);   // a failed solve leaves the previous x in place; L overwrites H
        if (lane == 0) Wk.flag = good ? 1 : 0;
      }
      __syncthreads();
      IMARK(6);
      const double* x = Wk.x;   // read from LDS where it is used (a register copy of the 30 unknowns pushed the kernel past 512 VGPRs: 740 B of scratch)
      ok = Wk.flag != 0;
      apply_update_t<RIG>(g, S, x);
      if (LASTFRAME && tid == 0) apply_update_t<RIG>(g, sS1, x + 15);
      __syncthreads();
      IMARK(7);
    }
    IMARK(1);
    // ---- classification (Optimizer.cc:4619-4676 / :5000-5057)
    int bad = 0, inl = 0;
    const float cm = LASTFRAME ? chi2MonoF[it] : chi2MonoKF[it];
    const float chi2close = 1.5f * cm;
    for (int i = tid; i < n; i += 256) {
      if (!hasMP[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const bool st = !RIG && !(o[2] < 0);
      const int cam = i >= nL ? 1 : 0;
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      double err[3], Xc[3];
      // an outlier's error is taken at the current state, an active edge's at the state of the last linearisation (Sprev) — picked element by
      // element: a reference chosen between the two structs at run time puts both (2 x 360 B) into scratch memory
      VIState Q;
      const bool cur = outlier[base + i] != 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) { Q.Rcw[k] = cur ? S.Rcw[k] : Sprev.Rcw[k]; if (RIG) Q.Rcw1[k] = cur ? S.Rcw1[k] : Sprev.Rcw1[k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { Q.tcw[k] = cur ? S.tcw[k] : Sprev.tcw[k]; if (RIG) Q.tcw1[k] = cur ? S.tcw1[k] : Sprev.tcw1[k]; }
      const float chi2 = (float)vis_error_t<RIG>(g, Q, X, o, st, (double)invSigma2[base + i], err, Xc, cam);
      bool isOut;
      if (st) isOut = chi2 > chi2Stereo[it];
      else {
        const bool bClose = closeFlag[base + i] != 0;
        const double r6 = cam ? S.Rcw1[6] : S.Rcw[6], r7 = cam ? S.Rcw1[7] : S.Rcw[7], r8 = cam ? S.Rcw1[8] : S.Rcw[8], tz = cam ? S.tcw1[2] : S.tcw[2];
        const bool depthPos = (r6 * X[0] + r7 * X[1] + r8 * X[2] + tz) > 0.0;
        isOut = (chi2 > cm && !bClose) || (bClose && chi2 > chi2close) || !depthPos;
      }
      outlier[base + i] = isOut ? 1 : 0;
      bad += isOut ? 1 : 0; inl += isOut ? 0 : 1;
    }
    bad = (int)wave_sum((double)bad); inl = (int)wave_sum((double)inl);
    __syncthreads();
    if (lane == 0) { Wk.cnt[wv][0] = bad; Wk.cnt[wv][1] = inl; }
    __syncthreads();
    nBad = Wk.cnt[0][0] + Wk.cnt[1][0] + Wk.cnt[2][0] + Wk.cnt[3][0];
    nInl = Wk.cnt[0][1] + Wk.cnt[1][1] + Wk.cnt[2][1] + Wk.cnt[3][1];
    __syncthreads();
    IMARK(8);
    if (it == 2) robust = false;
    if (nInit + (LASTFRAME ? 4 : 3) < 10) break;   // optimizer.edges().size() < 10
  }

  if (nInl < 30 && !bRecInit) {   // :4683-4707
    int bad = 0;
    for (int i = tid; i < n; i += 256) {
      if (!hasMP[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const bool st = !RIG && !(o[2] < 0);
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      double err[3], Xc[3];
      const float chi2 = (float)vis_error_t<RIG>(g, S, X, o, st, (double)invSigma2[base + i], err, Xc, i >= nL ? 1 : 0);
      if (chi2 < (st ? 24.f : 18.f)) outlier[base + i] = 0; else ++bad;
    }
    bad = (int)wave_sum((double)bad);
    __syncthreads();
    if (lane == 0) Wk.cnt[wv][0] = bad;
    __syncthreads();
    nBad = Wk.cnt[0][0] + Wk.cnt[1][0] + Wk.cnt[2][0] + Wk.cnt[3][0];
    __syncthreads();
  }

  if (tid == 0) {
    float* s0 = stateIO + 21 * f;
    for (int k = 0; k < 9; ++k) s0[k] = (float)S.Rwb[k];
    for (int k = 0; k < 3; ++k) { s0[9 + k] = (float)S.twb[k]; s0[12 + k] = (float)S.v[k]; s0[15 + k] = (float)S.bg[k]; s0[18 + k] = (float)S.ba[k]; }
    nInliersOut[f] = nInit - nBad;
  }
  IMARK(9);
  if (!prior) return;
  // ---- ConstraintPoseImu for the next frame (:4717-4754 / :5094-5150): un-robustified Hessians at the final state, inliers only.
  // Reference order of the 30 x 30 H: [state 1 (0..14) | frame (15..29)] = the inertial edge's own column order for its first 24.
  double acc[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) acc[k] = 0;
  for (int i = tid; i < n; i += 256) {
    if (!hasMP[base + i] || outlier[base + i]) continue;
    const float* o = obs + (base + i) * 3;
    const bool st = !RIG && !(o[2] < 0);
    const int cam = i >= nL ? 1 : 0;
    const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
    const double info = (double)invSigma2[base + i];
    double err[3], Xc[3], J[18];
    vis_error_t<RIG>(g, S, X, o, st, info, err, Xc, cam);
    vis_jacobian_t<RIG>(g, Xc, st, J, cam);
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = r; cc < 6; ++cc) {
        double h = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) h += J[k * 6 + r] * info * J[k * 6 + cc];   // mono: third row is zero
        acc[q++] += h;
      }
  }
#pragma unroll
  for (int k = 0; k < 21; ++k) acc[k] = wave_sum(acc[k]);
  __syncthreads();
  if (lane == 0) for (int k = 0; k < 21; ++k) Wk.red[wv][k] = acc[k];
  if (tid == 64) inertial_edge(P, S1, S, delta, LASTFRAME, Wk.e, Wk.J);
  if (LASTFRAME && tid == 128) prior_edge(pr, S1, Wk.ep, Wk.Jp);
  __syncthreads();
  for (int k = tid; k < 216; k += 256) { const int r = k / 24, c = k - r * 24; double s = 0; for (int l = 0; l < 9; ++l) s += Wk.InfoI[r * 9 + l] * Wk.J[l * 24 + c]; Wk.OJ[k] = s; }
  if (LASTFRAME) for (int k = tid; k < 225; k += 256) { const int r = k / 15, c = k - r * 15; double s = 0; for (int l = 0; l < 15; ++l) s += Wk.pH[r * 15 + l] * Wk.Jp[l * 15 + c]; Wk.OJp[k] = s; }
  __syncthreads();
  for (int k = tid; k < 900; k += 256) {   // H30 in the reference order
    const int r = k / 30, c = k - r * 30;
    double h = 0;
    if (r < 24 && c < 24) { double s = 0; for (int l = 0; l < 9; ++l) s += Wk.J[l * 24 + r] * Wk.OJ[l * 24 + c]; h += s; }
    const int rb = r % 15, cb = c % 15;
    if (rb >= 9 && cb >= 9 && (rb < 12) == (cb < 12)) {
      const double* I3 = rb < 12 ? Wk.InfoG : Wk.InfoA;
      const double v = I3[(rb - (rb < 12 ? 9 : 12)) * 3 + (cb - (cb < 12 ? 9 : 12))];
      if (LASTFRAME) h += ((r < 15) == (c < 15)) ? v : -v;          // GetHessian(): both vertices
      else if (r >= 15 && c >= 15) h += v;                           // GetHessian2(): the frame's bias only
    }
    if (LASTFRAME && r < 15 && c < 15) { double s = 0; for (int l = 0; l < 15; ++l) s += Wk.Jp[l * 15 + r] * Wk.OJp[l * 15 + c]; h += s; }
    if (r >= 15 && c >= 15 && r < 21 && c < 21) {
      const int a = (r < c ? r : c) - 15, bq = (r < c ? c : r) - 15;
      const int q = a * 6 - a * (a - 1) / 2 + (bq - a);
      h += Wk.red[0][q] + Wk.red[1][q] + Wk.red[2][q] + Wk.red[3][q];
    }
    Wk.H[k] = h;
  }
  __syncthreads();
  double* Hout = prior + (size_t)246 * f + 21;
  if (!LASTFRAME) {
    // GetHessian2 of the inertial edge = its (P2, V2) columns: rows / cols 15..23 of H30; the result is the trailing 15 x 15
    for (int k = tid; k < 225; k += 256) { const int r = k / 15, c = k - r * 15; Wk.Jp[k] = Wk.H[(15 + r) * 30 + 15 + c]; }
    __syncthreads();
  } else {
    // Optimizer::Marginalize(H, 0, 14) (Optimizer.cc:2898-2977): Hcc - Hcp pinv(Hpp) Hpc, singular values <= 1e-6 dropped.
    // All singular values above the threshold (Hpp - 1e-6 I positive definite) <=> the pseudo-inverse is the inverse: LDL^T
    // solves; otherwise the eigen-decomposition path below.
    // Hpp -> Jp (factor), Hpc -> OJp (15 right-hand sides, solved in place column by column)
    for (int k = tid; k < 225; k += 256) { const int r = k / 15, c = k - r * 15; Wk.Jp[k] = 0.5 * (Wk.H[r * 30 + c] + Wk.H[c * 30 + r]); Wk.OJp[k] = Wk.H[r * 30 + 15 + c]; }
    __syncthreads();
    if (tid == 0) {
      // test on a copy (J | OJ hold 432 doubles)
      double* T = Wk.J;
      for (int k = 0; k < 225; ++k) T[k] = Wk.Jp[k];
      for (int k = 0; k < 15; ++k) T[k * 15 + k] -= 1e-6;
      bool pd = true;
      for (int j = 0; j < 15 && pd; ++j) {
        const double d = T[j * 15 + j];
        if (!(d > 0)) { pd = false; break; }
        for (int r = 14; r > j; --r) { const double l = T[r * 15 + j] / d; for (int c = j + 1; c <= r; ++c) T[r * 15 + c] -= l * T[c * 15 + j]; }
      }
      Wk.flag = pd ? 1 : 0;
    }
    __syncthreads();
    if (Wk.flag) {
      // X = Hpp^-1 Hpc: factor once (wave 0), then 15 substitutions
      if (wv == 0) {
        for (int j = 0; j < 15; ++j) {
          WAVE_SYNC();
          const double d = Wk.Jp[j * 15 + j];
          const int m = 14 - j, pairs = m * (m + 1) / 2;
          for (int p = lane; p < pairs; p += 64) {
            int rr = (int)((sqrtf(8.f * (float)p + 1.f) - 1.f) * 0.5f);
            while ((rr + 1) * (rr + 2) / 2 <= p) ++rr;
            while (rr * (rr + 1) / 2 > p) --rr;
            const int cc = p - rr * (rr + 1) / 2;
            const int r = j + 1 + rr, c = j + 1 + cc;
            Wk.Jp[r * 15 + c] -= Wk.Jp[r * 15 + j] * Wk.Jp[c * 15 + j] / d;
          }
          WAVE_SYNC();
          for (int r = j + 1 + lane; r < 15; r += 64) Wk.Jp[r * 15 + j] /= d;
        }
        WAVE_SYNC();
      }
      __syncthreads();
      if (tid < 15) {   // one right-hand side (column tid of Hpc) per thread
        double y[15];
        for (int r = 0; r < 15; ++r) { double s = Wk.OJp[r * 15 + tid]; for (int c = 0; c < r; ++c) s -= Wk.Jp[r * 15 + c] * y[c]; y[r] = s; }
        for (int r = 0; r < 15; ++r) y[r] /= Wk.Jp[r * 15 + r];
        for (int r = 14; r >= 0; --r) { double s = y[r]; for (int c = r + 1; c < 15; ++c) s -= Wk.Jp[c * 15 + r] * y[c]; y[r] = s; }
        for (int r = 0; r < 15; ++r) Wk.OJp[r * 15 + tid] = y[r];
      }
      __syncthreads();
    } else {
      if (tid == 0) {   // pinv by the eigen-decomposition (rare): V diag(1/e, |e| > 1e-6) V^T
        double* A = Wk.J;          // 225 (J | OJ = 432 doubles)
        double* V = Wk.J + 225;    // needs 225: spills into OJ's tail + e/Oe ... keep within J|OJ: 432 < 450 -> use b/x region too
        // 450 doubles needed: J (216) + OJ (216) + e (9) + Oe (9) are contiguous in InertialWork
        for (int k = 0; k < 225; ++k) { A[k] = Wk.Jp[k]; V[k] = 0.0; }
        for (int k = 0; k < 15; ++k) V[k * 15 + k] = 1.0;
        for (int sweep = 0; sweep < 60; ++sweep) {
          double off = 0, diag = 0;
          for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) { const double t = A[r * 15 + c] * A[r * 15 + c]; if (r == c) diag += t; else off += t; }
          if (off <= 1e-30 * diag) break;
          for (int p = 0; p < 15; ++p)
            for (int q = p + 1; q < 15; ++q) {
              if (A[p * 15 + q] == 0.0) continue;
              const double theta = (A[q * 15 + q] - A[p * 15 + p]) / (2.0 * A[p * 15 + q]);
              const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
              const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
              for (int k = 0; k < 15; ++k) { const double a = A[k * 15 + p], bq = A[k * 15 + q]; A[k * 15 + p] = c * a - s * bq; A[k * 15 + q] = s * a + c * bq; }
              for (int k = 0; k < 15; ++k) { const double a = A[p * 15 + k], bq = A[q * 15 + k]; A[p * 15 + k] = c * a - s * bq; A[q * 15 + k] = s * a + c * bq; }
              for (int k = 0; k < 15; ++k) { const double a = V[k * 15 + p], bq = V[k * 15 + q]; V[k * 15 + p] = c * a - s * bq; V[k * 15 + q] = s * a + c * bq; }
            }
        }
        // Jp <- pinv
        for (int r = 0; r < 15; ++r)
          for (int c = 0; c < 15; ++c) {
            double s = 0;
            for (int k = 0; k < 15; ++k) { const double e = A[k * 15 + k]; if (fabs(e) > 1e-6) s += V[r * 15 + k] * (1.0 / e) * V[c * 15 + k]; }
            Wk.Jp[r * 15 + c] = s;
          }
        // OJp <- pinv * Hpc
        for (int c = 0; c < 15; ++c) {
          double y[15];
          for (int r = 0; r < 15; ++r) { double s = 0; for (int k = 0; k < 15; ++k) s += Wk.Jp[r * 15 + k] * Wk.OJp[k * 15 + c]; y[r] = s; }
          for (int r = 0; r < 15; ++r) Wk.OJp[r * 15 + c] = y[r];
        }
      }
      __syncthreads();
    }
    // Hcc - Hcp X  -> Jp
    double hv = 0;
    if (tid < 225) {
      const int r = tid / 15, c = tid - r * 15;
      double s = 0;
      for (int k = 0; k < 15; ++k) s += Wk.H[(15 + r) * 30 + k] * Wk.OJp[k * 15 + c];
      hv = Wk.H[(15 + r) * 30 + 15 + c] - s;
    }
    __syncthreads();
    if (tid < 225) Wk.Jp[tid] = hv;
    __syncthreads();
  }
  if (tid == 0) clamp_eigenvalues(Wk.Jp, 15, 1e-12, Wk.J, Wk.OJp);   // ConstraintPoseImu's constructor; J | OJ | e | Oe = 450 doubles
  __syncthreads();
  IMARK(10);
  for (int k = tid; k < 225; k += 256) Hout[k] = Wk.Jp[k];
  if (tid == 0) {
    double* out = prior + (size_t)246 * f;
    for (int k = 0; k < 9; ++k) out[k] = S.Rwb[k];
    for (int k = 0; k < 3; ++k) { out[9 + k] = S.twb[k]; out[12 + k] = S.v[k]; out[15 + k] = S.bg[k]; out[18 + k] = S.ba[k]; }
  }
}


// =====================================================================================================================================
// LocalInertialBA (reference src/Optimizer.cc:2324-2897): the temporal window of keyframes with 15-dof states, the points they see
// behind a Schur complement, the chain of inertial / random-walk edges.  g2o Levenberg-Marquardt semantics
// (optimization_algorithm_levenberg.cpp:61-194, block_solver.hpp) with the accept / reject decision on the host from one small
// read-back per trial, like the grid mode of LocalBundleAdjustment.  One launch per phase over the whole chip:
//   k_iba_errors   computeActiveErrors + activeRobustChi2 (one thread per visual edge / per inertial link)
//   k_iba_points   per point: Hll, bl                      k_iba_kf   per 64-edge chunk of a keyframe: Hpp, bp (wave sums), Hpl
//   k_iba_links    inertial + random-walk edges into the dense 15 N x 15 N system
//   k_iba_pack_w / k_iba_pack_wd + k_schur_mfma + k_iba_schur_finish   Schur complement of the points on the FP64 matrix cores
//   k_iba_solve    one workgroup: dense LDL^T          k_iba_update   back-substitution of the points, oplus of all vertices
// =====================================================================================================================================
struct IbaDev {
  int nKF, nMP, nE, nI, P, nChunks;
  const int *eKF, *eMP;
  const float *eObs, *eInfo;
  const int *ptStart, *ptEdges;                 // CSR by point
  const int *kfEdges, *chunkKF, *chunkStart, *chunkEnd;   // edges of free keyframes in chunks of <= 64
  double* kfPart; int* kfTicket;                          // per-chunk partial sums of k_iba_kf (27 each), arrival counter per keyframe
  const int* linkOrder;                                   // inertial links sorted by colour: links of one colour share no keyframe
  const int* col;                               // [nKF] index of an optimizable keyframe or -1
  const int *iKF1, *iKF2;
  const morb_imu_preintegrated* iPre;
  const uint8_t* iRobust;
  const uint8_t* mpClose;
  const uint8_t* eRight;                        // fisheye rig: edge on the right camera (NULL: pinhole)
  double* S;                                    // [nKF][33]: Rwb twb v bg ba | Rcw tcw
  double* pts;                                  // [nMP][3]
  double *vErr, *iErr, *gErr, *aErr;            // errors of the last computeActiveErrors
  double *InfoI, *InfoG, *InfoA;
  double *H, *b, *Hll, *Hpl, *Hs, *bs, *x;      // b, x: P + 3 nMP
  double* scal;                                 // [0] robust chi2 of the last k_iba_errors, [2] solve ok
  double *partChi, *partScale; int nbUpdate;    // per-block partial sums of k_iba_errors / k_iba_update (added up in block order)
  // Schur complement on the FP64 matrix cores (schur_mfma.h): dense K-major operands (columns 6 c + r of the pose parts, plus
  // the right-hand-side column 6 nOpt), partial products, block directory
  double *sW, *sWD, *sPart;
  const int2* sBlocks;
  const int* sBlkIndex;
  int sMp, sNb, sNblk, sNsplit;
  // device-side LM control (see optimizer.hip, grid-mode LocalBA): the loop state, its mirror in mapped host memory, the backups
  double* lmd;                                  // IBA_LMD_*: currentChi, lambda, ni, iniChi, chi2 of the last evaluated trial
  int* lmi;                                     // IBA_LM_*
  int* lmHost;                                  // [0] decided trials, [1] done
  double *Sbk, *ptsBk;                          // state / points before the trial
  double* pnlG;                                 // panel copies of the global-memory LDL^T when they do not fit LDS
  int nS, nPts, optIt;
  CamGeom g;
};
enum { IBA_LM_ITER, IBA_LM_QMAX, IBA_LM_NBAD, IBA_LM_ITS, IBA_LM_TRIALS, IBA_LM_DONE, IBA_LM_NEEDBUILD, IBA_LM_REJECTED, IBA_LM_TICKET };
enum { IBA_LMD_CHI, IBA_LMD_LAMBDA, IBA_LMD_NI, IBA_LMD_INICHI, IBA_LMD_LASTCHI };
__device__ __forceinline__ bool iba_done(const IbaDev& D) { return D.lmi[IBA_LM_DONE] != 0; }
__device__ __forceinline__ bool iba_no_build(const IbaDev& D) { return D.lmi[IBA_LM_DONE] != 0 || D.lmi[IBA_LM_NEEDBUILD] == 0; }
__device__ __forceinline__ void iba_load(const CamGeom& g, const double* s, VIState& V) {
  for (int k = 0; k < 9; ++k) { V.Rwb[k] = s[k]; V.Rcw[k] = s[21 + k]; }
  for (int k = 0; k < 3; ++k) { V.twb[k] = s[9 + k]; V.v[k] = s[12 + k]; V.bg[k] = s[15 + k]; V.ba[k] = s[18 + k]; V.tcw[k] = s[30 + k]; }
  if (g.rig) refresh_camera(g, V);   // the right camera's pose is derived, not stored
}
__device__ __forceinline__ void iba_store(const VIState& V, double* s) {
  for (int k = 0; k < 9; ++k) { s[k] = V.Rwb[k]; s[21 + k] = V.Rcw[k]; }
  for (int k = 0; k < 3; ++k) { s[9 + k] = V.twb[k]; s[12 + k] = V.v[k]; s[15 + k] = V.bg[k]; s[18 + k] = V.ba[k]; s[30 + k] = V.tcw[k]; }
}
__device__ __forceinline__ double huber_rho(double delta, double e2) {   // rho(e2)
  return e2 <= delta * delta ? e2 : 2 * sqrt(e2) * delta - delta * delta;
}
// Sum over the block, in a fixed order (DPP tree per wave, then the waves in turn): thread 0 holds it.  The blocks' sums are stored one per
// block and added up in block order by whoever consumes them — a floating-point atomicAdd per block would make the chi2, through rho and
// lambda every later iterate, depend on the order in which the blocks happen to finish (seen as last-bit differences between two runs).
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double red[16];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0) for (int w = 0; w < nw; ++w) s += red[w];
  __syncthreads();
  return s;
}
// The partial sums of n workgroups added up by one full wave, always in the same order (lane k takes partials k, k + 64, ...; then the DPP
// tree): every lane returns the total.  The partials were written by other workgroups: agent-scope loads.
__device__ __forceinline__ double ordered_sum_wave(const double* part, int n) {
  double s = 0;
  for (int k = threadIdx.x & 63; k < n; k += 64)
    s += __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return wave_sum(s);
}
__device__ __forceinline__ double iba_vis_chi2(const IbaDev& D, int e) {
  const double info = (double)D.eInfo[e];
  const double* er = D.vErr + 3 * (size_t)e;
  return er[0] * info * er[0] + er[1] * info * er[1] + er[2] * info * er[2];   // mono: er[2] = 0
}
__device__ __forceinline__ double iba_quad(const double* e, const double* Om, int n) {
  double s = 0;
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) s += e[r] * Om[r * n + c] * e[c];
  return s;
}

__global__ void k_iba_setup_kf(IbaDev D, const float* __restrict__ kfState) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D.nKF) return;
  VIState V;
  load_state(D.g, kfState + 21 * k, V);
  iba_store(V, D.S + 33 * (size_t)k);
}
__global__ void k_iba_setup_pts(IbaDev D, const float* __restrict__ mpPos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < D.nPts) D.pts[k] = (double)mpPos[k];
}
// informations of the inertial links: one workgroup per link, the 9 x 9 work arrays in LDS (one thread computes)
__global__ __launch_bounds__(64) void k_iba_setup_links(IbaDev D, const float* __restrict__ infoScale) {
  __shared__ double C9[81], M[162], V[162], Inf[81];
  const int i = blockIdx.x;
  if (threadIdx.x == 0) {
    const morb_imu_preintegrated& P = D.iPre[i];
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C9[r * 9 + c] = (double)P.C[r * 15 + c];
    if (invert_n(C9, 9, Inf, M)) {
      for (int r = 0; r < 9; ++r) for (int c = r + 1; c < 9; ++c) { const double s = (Inf[r * 9 + c] + Inf[c * 9 + r]) / 2; Inf[r * 9 + c] = s; Inf[c * 9 + r] = s; }
      clamp_eigenvalues(Inf, 9, 1e-12, M, V);
    } else for (int k = 0; k < 81; ++k) Inf[k] = 0;
    double Cg[9], Ca[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Cg[r * 3 + c] = P.C[(9 + r) * 15 + 9 + c]; Ca[r * 3 + c] = P.C[(12 + r) * 15 + 12 + c]; }
    if (!invert_n(Cg, 3, D.InfoG + (size_t)i * 9, M)) for (int k = 0; k < 9; ++k) D.InfoG[(size_t)i * 9 + k] = 0;
    if (!invert_n(Ca, 3, D.InfoA + (size_t)i * 9, M)) for (int k = 0; k < 9; ++k) D.InfoA[(size_t)i * 9 + k] = 0;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 81; k += 64) D.InfoI[(size_t)i * 81 + k] = Inf[k] * (double)infoScale[i];
}

// computeActiveErrors + activeRobustChi2 -> scal[0] (the last workgroup to finish adds the blocks' partial sums up in block order)
__device__ void iba_lm_decide(const IbaDev& D, double trialChi2, double gain);
// mode 0: computeActiveErrors + activeRobustChi2.  mode 1 (a trial under device-side LM control):
// returns at once when the solve has finished; the last workgroup to finish takes the trial's accept / reject decision.
__global__ __launch_bounds__(256) void k_iba_errors(IbaDev D, int mode) {
  if (mode == 1 && iba_done(D)) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815), deltaI = sqrt(16.92);
  double c = 0;
  if (t < D.nE) {
    VIState V;
    iba_load(D.g, D.S + 33 * (size_t)D.eKF[t], V);
    const float* o = D.eObs + 3 * (size_t)t;
    const bool st = !D.g.rig && !(o[2] < 0);
    const double* Xp = D.pts + 3 * (size_t)D.eMP[t];
    const double X[3] = {Xp[0], Xp[1], Xp[2]};
    double err[3], Xc[3];
    const double chi = vis_error(D.g, V, X, o, st, (double)D.eInfo[t], err, Xc, D.eRight && D.eRight[t] ? 1 : 0);
    for (int k = 0; k < 3; ++k) D.vErr[3 * (size_t)t + k] = err[k];
    c = huber_rho(st ? deltaStereo : deltaMono, chi);
  } else if (t - D.nE < D.nI) {
    const int i = t - D.nE;
    VIState V1, V2;
    iba_load(D.g, D.S + 33 * (size_t)D.iKF1[i], V1);
    iba_load(D.g, D.S + 33 * (size_t)D.iKF2[i], V2);
    double err[9];
    inertial_edge(D.iPre[i], V1, V2, nullptr, true, err, nullptr);
    double ge[3], ae[3];
    for (int k = 0; k < 9; ++k) D.iErr[9 * (size_t)i + k] = err[k];
    for (int k = 0; k < 3; ++k) { ge[k] = V2.bg[k] - V1.bg[k]; ae[k] = V2.ba[k] - V1.ba[k]; D.gErr[3 * i + k] = ge[k]; D.aErr[3 * i + k] = ae[k]; }
    const double ci = iba_quad(err, D.InfoI + (size_t)i * 81, 9);
    c = (D.iRobust[i] ? huber_rho(deltaI, ci) : ci) + iba_quad(ge, D.InfoG + (size_t)i * 9, 3) + iba_quad(ae, D.InfoA + (size_t)i * 9, 3);
  }
  const double bs = block_sum(c);
  __shared__ int isLast;
  if (threadIdx.x == 0) {   // hand-off without __threadfence() (an L2 write-back per workgroup on this chip): write-through store, wait, relaxed ticket
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(D.partChi + blockIdx.x), __builtin_bit_cast(unsigned long long, bs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    isLast = __hip_atomic_fetch_add(&D.lmi[IBA_LM_TICKET], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!isLast || threadIdx.x >= 64) return;   // the last workgroup's first wave adds the partials up
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (ordered_sum_wave reads every handed-off word at agent scope)
  const double chi2 = ordered_sum_wave(D.partChi, (int)gridDim.x);
  const double gain = mode == 1 ? ordered_sum_wave(D.partScale, D.nbUpdate) : 0.0;   // k_iba_update's blocks
  if (threadIdx.x != 0) return;
  D.lmi[IBA_LM_TICKET] = 0;
  D.scal[0] = chi2;
  if (mode == 1) iba_lm_decide(D, chi2, gain);
}
// optimization_algorithm_levenberg.cpp:99-169 with ORB-SLAM's stop rule, as local_inertial_ba_impl's host loop ran it: rho from
// the trial's chi2, the linear-model gain (the sum of k_iba_update's per-block partials) and the solver's flag (scal[2]).  One thread.
__device__ void iba_lm_decide(const IbaDev& D, double trialChi2, double gain) {
  const double s0 = trialChi2;
  const double s1 = gain;
  const bool ok2 = D.scal[2] != 0.0;
  double currentChi = D.lmd[IBA_LMD_CHI], lambda = D.lmd[IBA_LMD_LAMBDA], ni = D.lmd[IBA_LMD_NI];
  const double iniChi = D.lmd[IBA_LMD_INICHI];
  int iter = D.lmi[IBA_LM_ITER], qmax = D.lmi[IBA_LM_QMAX], nBad = D.lmi[IBA_LM_NBAD], its = D.lmi[IBA_LM_ITS];
  double tempChi = s0;
  if (!ok2) tempChi = 1.7976931348623157e308;
  const double rho = (currentChi - tempChi) / (s1 + 1e-3);
  const bool accept = rho > 0 && isfinite(tempChi);
  if (accept) {
    double alpha = 1. - pow((2 * rho - 1), 3);
    alpha = fmin(alpha, 2. / 3.);
    lambda *= fmax(1. / 3., alpha);
    ni = 2;
    currentChi = tempChi;
  } else {
    lambda *= ni;
    ni *= 2;
  }
  ++qmax;
  const int trials = D.lmi[IBA_LM_TRIALS] + 1;
  int done = 0, needBuild = 0;
  if (!(rho < 0 && qmax < 10)) {   // the trial loop ends
    // (a NaN rho — NaN errors — leaves the loop with the trial rejected: the host loop went on from the restored state; here the
    // solve ends, the stored errors belong to the rejected state)
    bool fin = (qmax == 10 || rho == 0 || !(rho == rho));
    if (!fin) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; fin = nBad >= 3; }
    if (!fin) { ++iter; fin = iter >= D.optIt; }
    if (fin) done = 1;
    else { ++its; qmax = 0; needBuild = 1; D.lmd[IBA_LMD_INICHI] = currentChi; }
  }
  D.lmd[IBA_LMD_CHI] = currentChi; D.lmd[IBA_LMD_LAMBDA] = lambda; D.lmd[IBA_LMD_NI] = ni; D.lmd[IBA_LMD_LASTCHI] = s0;
  D.lmi[IBA_LM_ITER] = iter; D.lmi[IBA_LM_QMAX] = qmax; D.lmi[IBA_LM_NBAD] = nBad; D.lmi[IBA_LM_ITS] = its; D.lmi[IBA_LM_TRIALS] = trials;
  D.lmi[IBA_LM_DONE] = done; D.lmi[IBA_LM_NEEDBUILD] = needBuild; D.lmi[IBA_LM_REJECTED] = accept ? 0 : 1;
  __hip_atomic_store(D.lmHost + 1, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(D.lmHost + 0, trials, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// after the last slot: a rejected last trial is undone
__global__ __launch_bounds__(256) void k_iba_end(IbaDev D) {
  if (!D.lmi[IBA_LM_REJECTED]) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < D.nS) D.S[t] = D.Sbk[t];
  if (t < D.nPts) D.pts[t] = D.ptsBk[t];
}
__global__ void k_iba_lm_init(IbaDev D, double chi, double lambda0) {
  D.lmd[IBA_LMD_CHI] = chi; D.lmd[IBA_LMD_LAMBDA] = lambda0; D.lmd[IBA_LMD_NI] = 2; D.lmd[IBA_LMD_INICHI] = chi; D.lmd[IBA_LMD_LASTCHI] = chi;
  for (int k = 0; k < 16; ++k) D.lmi[k] = 0;
  D.lmi[IBA_LM_ITS] = 1; D.lmi[IBA_LM_NEEDBUILD] = 1;
  D.scal[0] = 0.0; D.scal[1] = 0.0;
}

// point side of buildSystem: Hll, bl (one thread per point over its edges, no atomics)
// (the launch starts with k_iba_begin's work — backup / restore, clearing the accumulators — on as many threads as that needs)
__device__ __forceinline__ void iba_begin_part(const IbaDev& D, int t) {
  const bool restore = D.lmi[IBA_LM_REJECTED] != 0;
  if (restore) {
    if (t < D.nS) D.S[t] = D.Sbk[t];
    if (t < D.nPts) D.pts[t] = D.ptsBk[t];
  } else {
    if (t < D.nS) D.Sbk[t] = D.S[t];
    if (t < D.nPts) D.ptsBk[t] = D.pts[t];
  }
  if (D.lmi[IBA_LM_NEEDBUILD]) {
    if (t < D.P * D.P) D.H[t] = 0.0;
    if (t < D.P) D.b[t] = 0.0;
    if (t < 18 * D.nE) D.Hpl[t] = 0.0;
  }
}
constexpr int IBA_PT_LANES = 8;   // lanes per point in k_iba_points
__global__ __launch_bounds__(256) void k_iba_points(IbaDev D) {
  __shared__ double sPt[256 * 12];
  if (iba_done(D)) return;
  iba_begin_part(D, blockIdx.x * 256 + threadIdx.x);
  if (D.lmi[IBA_LM_NEEDBUILD] == 0) return;   // (a rebuild follows an accepted trial: nothing was restored, the points read below are current)
  // EIGHT lanes per point, one edge each and chunk by chunk; the twelve contributions of an edge go to the wave's LDS slice and the group's first lane
  // adds them up in edge order: the sums of the one-thread-per-point walk this replaces, bit for bit (as in LocalBundleAdjustment's k_g_build)
  const int g = threadIdx.x / IBA_PT_LANES, sub = threadIdx.x % IBA_PT_LANES;
  const int l = blockIdx.x * (256 / IBA_PT_LANES) + g;
  const bool live = l < D.nMP;
  double* slot = sPt + (size_t)g * IBA_PT_LANES * 12;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  double Hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
  const size_t lp = live ? (size_t)l : 0;
  const double X[3] = {D.pts[3 * lp], D.pts[3 * lp + 1], D.pts[3 * lp + 2]};
  const int k0 = live ? D.ptStart[l] : 0, k1 = live ? D.ptStart[l + 1] : 0;
  for (int kb = k0; kb < k1; kb += IBA_PT_LANES) {
    const int k = kb + sub;
    double c12[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) c12[q] = 0;
    if (k < k1) {
      const int e = D.ptEdges[k];
      VIState V;
      iba_load(D.g, D.S + 33 * (size_t)D.eKF[e], V);
      const int cam = D.eRight && D.eRight[e] ? 1 : 0;
      const double* Rc = cam ? V.Rcw1 : V.Rcw;
      const double* tc = cam ? V.tcw1 : V.tcw;
      const float* o = D.eObs + 3 * (size_t)e;
      const bool st = !D.g.rig && !(o[2] < 0);
      const double info = (double)D.eInfo[e];
      const double w = huber_w(st ? deltaStereo : deltaMono, iba_vis_chi2(D, e));
      double Xc[3];
      for (int r = 0; r < 3; ++r) Xc[r] = Rc[r * 3] * X[0] + Rc[r * 3 + 1] * X[1] + Rc[r * 3 + 2] * X[2] + tc[r];
      // -proj_jac * Rcw (G2oTypes.cc:334-415)
      double pj[9];
      cam_project_jac(D.g, Xc, cam, pj);
      pj[6] = pj[7] = pj[8] = 0;
      if (st) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + D.g.bf * (1.0 / (Xc[2] * Xc[2])); }
      double Jl[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Jl[r * 3 + c] = -(pj[r * 3] * Rc[c] + pj[r * 3 + 1] * Rc[3 + c] + pj[r * 3 + 2] * Rc[6 + c]);
      const double* er = D.vErr + 3 * (size_t)e;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        double sm = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) sm += Jl[i * 3 + r] * info * er[i];
        c12[9 + r] = w * sm;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double h = 0;
#pragma unroll
          for (int i = 0; i < 3; ++i) h += Jl[i * 3 + r] * (w * info) * Jl[i * 3 + c];
          c12[r * 3 + c] = h;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) slot[sub * 12 + q] = c12[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (sub == 0) {
      const int cnt = k1 - kb < IBA_PT_LANES ? k1 - kb : IBA_PT_LANES;
      for (int j = 0; j < cnt; ++j) {
#pragma unroll
        for (int q = 0; q < 9; ++q) Hl[q] += slot[j * 12 + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) bl[q] -= slot[j * 12 + 9 + q];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // (the slice is rewritten by the next chunk)
  }
  if (!live || sub != 0) return;
  for (int k = 0; k < 9; ++k) D.Hll[(size_t)l * 9 + k] = Hl[k];
  for (int k = 0; k < 3; ++k) D.b[D.P + 3 * (size_t)l + k] = bl[k];
}
// keyframe side: one wave per chunk of <= 64 edges of one optimizable keyframe
__global__ __launch_bounds__(256) void k_iba_kf(IbaDev D) {
  if (iba_no_build(D)) return;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= D.nChunks) return;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  const int kf = D.chunkKF[c];
  VIState V;
  iba_load(D.g, D.S + 33 * (size_t)kf, V);
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0;
  const int k = D.chunkStart[c] + lane;
  if (k < D.chunkEnd[c]) {
    const int e = D.kfEdges[k];
    const float* o = D.eObs + 3 * (size_t)e;
    const bool st = !D.g.rig && !(o[2] < 0);
    const int cam = D.eRight && D.eRight[e] ? 1 : 0;
    double Rc[9], tc[3];   // (element-wise selects: a pointer chosen between two members of V sends V to scratch memory)
#pragma unroll
    for (int r = 0; r < 9; ++r) Rc[r] = cam ? V.Rcw1[r] : V.Rcw[r];
#pragma unroll
    for (int r = 0; r < 3; ++r) tc[r] = cam ? V.tcw1[r] : V.tcw[r];
    const double info = (double)D.eInfo[e];
    const double w = huber_w(st ? deltaStereo : deltaMono, iba_vis_chi2(D, e));
    const double* Xp = D.pts + 3 * (size_t)D.eMP[e];
    const double X[3] = {Xp[0], Xp[1], Xp[2]};
    double Xc[3], Jp[18], Jl[9];
    mul3v(Rc, X, Xc);
    for (int r = 0; r < 3; ++r) Xc[r] += tc[r];
    vis_jacobian(D.g, Xc, st, Jp, cam);
    double pj[9];
    cam_project_jac(D.g, Xc, cam, pj);
    pj[6] = pj[7] = pj[8] = 0;
    if (st) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + D.g.bf * (1.0 / (Xc[2] * Xc[2])); }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) Jl[r * 3 + cc] = -(pj[r * 3] * Rc[cc] + pj[r * 3 + 1] * Rc[3 + cc] + pj[r * 3 + 2] * Rc[6 + cc]);
    const double* er = D.vErr + 3 * (size_t)e;
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double sm = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) sm += Jp[i * 6 + r] * info * er[i];
      acc[21 + r] = -w * sm;
#pragma unroll
      for (int cc = r; cc < 6; ++cc) {
        double h = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) h += Jp[i * 6 + r] * (w * info) * Jp[i * 6 + cc];
        acc[q++] = h;
      }
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        double h = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) h += Jp[i * 6 + r] * (w * info) * Jl[i * 3 + cc];
        D.Hpl[(size_t)e * 18 + r * 3 + cc] = h;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 27; ++q) acc[q] = wave_sum(acc[q]);
  // A keyframe's chunks are added up in chunk order by whichever of them finishes last (floating-point atomics would add them in
  // arrival order: the Hessian, and with it every iterate, would differ in the last bits from run to run).  Nothing else has touched the
  // keyframe's pose block yet — the links are launched afterwards — so the sum is stored with plain read-modify-writes.
  int first = c, end = c + 1, last = 0;
  if (lane == 0) {   // (hand-off as in k_iba_errors: write-through stores, wait, relaxed ticket; the last arriver reads at agent scope)
#pragma unroll
    for (int q = 0; q < 27; ++q)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(D.kfPart + (size_t)c * 27 + q), __builtin_bit_cast(unsigned long long, acc[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (first > 0 && D.chunkKF[first - 1] == kf) --first;
    while (end < D.nChunks && D.chunkKF[end] == kf) ++end;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(&D.kfTicket[kf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == end - first - 1;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  first = __builtin_amdgcn_readfirstlane(first); end = __builtin_amdgcn_readfirstlane(end);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0) D.kfTicket[kf] = 0;
  if (lane < 27) {
    double sum = 0;
    const double* part = D.kfPart + lane;
    for (int j = first; j < end; ++j)
      sum += __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part + (size_t)j * 27), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int o = 15 * D.col[kf];
    if (lane >= 21) D.b[o + lane - 21] += sum;
    else {
      int r = 0, q0 = 0;
      while (lane >= q0 + 6 - r) { q0 += 6 - r; ++r; }
      const int cc = r + lane - q0;
      D.H[(size_t)(o + r) * D.P + o + cc] += sum;
      if (cc != r) D.H[(size_t)(o + cc) * D.P + o + r] += sum;
    }
  }
}
// inertial + random-walk edges: one 64-thread workgroup per link; thread 0 evaluates the edge (error, 9 x 24 Jacobian) into LDS, all
// threads multiply out J^T Omega J
// One launch per colour: the links of a launch share no keyframe (a chain of consecutive keyframes has two colours), so every entry of H
// receives its contributions — the keyframe's visual block, then its links colour by colour — in the same order in every run.
__global__ __launch_bounds__(64) void k_iba_links(IbaDev D, int base) {
  __shared__ double J[216], OJ[216], Oe[9];
  __shared__ double sw;
  if (iba_no_build(D)) return;
  const int i = D.linkOrder[base + blockIdx.x], tid = threadIdx.x;
  const int k1 = D.iKF1[i], k2 = D.iKF2[i];
  for (int k = tid; k < 216; k += 64) J[k] = 0;
  __syncthreads();
  const double* er = D.iErr + 9 * (size_t)i;   // errors of computeActiveErrors (same state)
  const double* Om = D.InfoI + (size_t)i * 81;
  if (tid == 0) {
    VIState V1, V2;
    iba_load(D.g, D.S + 33 * (size_t)k1, V1);
    iba_load(D.g, D.S + 33 * (size_t)k2, V2);
    double errNow[9];
    inertial_edge(D.iPre[i], V1, V2, nullptr, true, errNow, J);
    sw = D.iRobust[i] ? huber_w(sqrt(16.92), iba_quad(er, Om, 9)) : 1.0;
  }
  __syncthreads();
  for (int k = tid; k < 216 + 9; k += 64) {
    if (k < 216) { const int r = k / 24, c = k - r * 24; double sm = 0; for (int l = 0; l < 9; ++l) sm += Om[r * 9 + l] * J[l * 24 + c]; OJ[k] = sm; }
    else { const int r = k - 216; double sm = 0; for (int l = 0; l < 9; ++l) sm += Om[r * 9 + l] * er[l]; Oe[r] = sm; }
  }
  __syncthreads();
  const double w = sw;
  const int c1 = D.col[k1], c2 = D.col[k2];
  for (int idx = tid; idx < 24 * 24 + 24; idx += 64) {
    const int a = idx < 576 ? idx / 24 : idx - 576;
    const int ca = a < 15 ? (c1 >= 0 ? 15 * c1 + a : -1) : (c2 >= 0 ? 15 * c2 + (a - 15) : -1);
    if (ca < 0) continue;
    if (idx < 576) {
      const int b2 = idx - a * 24;
      const int cb = b2 < 15 ? (c1 >= 0 ? 15 * c1 + b2 : -1) : (c2 >= 0 ? 15 * c2 + (b2 - 15) : -1);
      if (cb < 0) continue;
      double h = 0;
      for (int k = 0; k < 9; ++k) h += J[k * 24 + a] * OJ[k * 24 + b2];
      if (h != 0.0) unsafeAtomicAdd(&D.H[(size_t)ca * D.P + cb], w * h);
    } else {
      double sm = 0;
      for (int k = 0; k < 9; ++k) sm += J[k * 24 + a] * Oe[k];
      unsafeAtomicAdd(&D.b[ca], -w * sm);
    }
  }
  __syncthreads();   // (the random-walk terms land on entries of the bias blocks the loop above has added to: after it, not in a race with it)
  if (tid < 18) {   // EdgeGyroRW / EdgeAccRW: e = bias2 - bias1; thread = (type, row r, column c)
    const int t = tid / 9, r = (tid % 9) / 3, c = tid % 3;
    const double* I3 = (t == 0 ? D.InfoG : D.InfoA) + (size_t)i * 9;
    const double* e3 = (t == 0 ? D.gErr : D.aErr) + 3 * (size_t)i;
    const int o1 = c1 >= 0 ? 15 * c1 + 9 + 3 * t : -1, o2 = c2 >= 0 ? 15 * c2 + 9 + 3 * t : -1;
    const double v = I3[r * 3 + c];
    if (o2 >= 0) unsafeAtomicAdd(&D.H[(size_t)(o2 + r) * D.P + o2 + c], v);
    if (o1 >= 0) unsafeAtomicAdd(&D.H[(size_t)(o1 + r) * D.P + o1 + c], v);
    if (o1 >= 0 && o2 >= 0) { unsafeAtomicAdd(&D.H[(size_t)(o1 + r) * D.P + o2 + c], -v); unsafeAtomicAdd(&D.H[(size_t)(o2 + r) * D.P + o1 + c], -v); }
    if (c == 0) {
      double sm = 0;
      for (int k = 0; k < 3; ++k) sm += I3[r * 3 + k] * e3[k];
      if (o2 >= 0) unsafeAtomicAdd(&D.b[o2 + r], -sm);
      if (o1 >= 0) unsafeAtomicAdd(&D.b[o1 + r], sm);
    }
  }
}
__device__ __forceinline__ bool inv3(const double* D3, double* I) {
  const double a = D3[0], b = D3[1], c = D3[2], d = D3[3], e = D3[4], f = D3[5], g = D3[6], h = D3[7], i = D3[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  if (det == 0.0) { for (int k = 0; k < 9; ++k) I[k] = 0; return false; }
  const double inv = 1.0 / det;
  I[0] = A * inv; I[1] = -(b * i - c * h) * inv; I[2] = (b * f - c * e) * inv;
  I[3] = B * inv; I[4] = (a * i - c * g) * inv; I[5] = -(a * f - c * d) * inv;
  I[6] = C * inv; I[7] = -(a * h - b * g) * inv; I[8] = (a * e - b * d) * inv;
  return true;
}
// MFMA operands of the Schur complement (schur_mfma.h).  W: Hpl of every observation of an optimizable keyframe and b_l in the extra
// column (once per outer iteration, after buildSystem); WD = Hpl (Hll + lambda I)^-1 (every trial).
// Hpl of (keyframe column c1, point m) as the MFMA operands need it: on a fisheye rig a keyframe may observe a point with both
// cameras, i.e. through two edges — the first of them (lowest edge index) carries the sum, the others nothing.
__device__ __forceinline__ bool iba_pair_block(const IbaDev& D, int e, int c1, int m, double* __restrict__ B) {
  for (int q = 0; q < 18; ++q) B[q] = D.Hpl[(size_t)e * 18 + q];
  for (int k = D.ptStart[m]; k < D.ptStart[m + 1]; ++k) {
    const int e2 = D.ptEdges[k];
    if (e2 == e || D.col[D.eKF[e2]] != c1) continue;
    if (e2 < e) return false;
    for (int q = 0; q < 18; ++q) B[q] += D.Hpl[(size_t)e2 * 18 + q];
  }
  return true;
}
__global__ __launch_bounds__(256) void k_iba_pack_w(IbaDev D) {
  if (iba_no_build(D)) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int M = 6 * (D.P / 15);
  if (t < D.nE) {
    const int c1 = D.col[D.eKF[t]];
    const int m = D.eMP[t];
    double B1[18];
    if (c1 >= 0 && iba_pair_block(D, t, c1, m, B1)) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) D.sW[(size_t)(3 * m + c) * D.sMp + 6 * c1 + r] = B1[r * 3 + c];
    }
  }
  if (t < 3 * D.nMP) D.sW[(size_t)t * D.sMp + M] = D.b[D.P + t];
}
// (the launch also sets Hs = H + lambda I, bs = b on its first P * P + P threads: one launch fewer per trial)
__global__ __launch_bounds__(256) void k_iba_pack_wd(IbaDev D) {
  if (iba_done(D)) return;
  const double lambda = D.lmd[IBA_LMD_LAMBDA];
  const int t = blockIdx.x * 256 + threadIdx.x;
  {
    const int n = D.P * D.P;
    if (t < n) { const int r = t / D.P, c = t - r * D.P; D.Hs[t] = D.H[t] + (r == c ? lambda : 0.0); }
    else if (t - n < D.P) D.bs[t - n] = D.b[t - n];
  }
  if (t >= D.nE) return;
  const int c1 = D.col[D.eKF[t]];
  if (c1 < 0) return;
  const int m = D.eMP[t];
  double B1[18];
  if (!iba_pair_block(D, t, c1, m, B1)) return;
  double Dm[9], Di[9];
  for (int k = 0; k < 9; ++k) Dm[k] = D.Hll[(size_t)m * 9 + k];
  Dm[0] += lambda; Dm[4] += lambda; Dm[8] += lambda;
  inv3(Dm, Di);
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      D.sWD[(size_t)(3 * m + c) * D.sMp + 6 * c1 + r] = B1[r * 3] * Di[c] + B1[r * 3 + 1] * Di[3 + c] + B1[r * 3 + 2] * Di[6 + c];
}
// Hs(pose rows / columns) -= C, bs(pose rows) -= C[:, M], from the partial products (four lanes per element, fixed order)
__global__ __launch_bounds__(256) void k_iba_schur_finish(IbaDev D) {
  if (iba_done(D)) return;
  const int t = blockIdx.x * 256 + threadIdx.x, gid = t >> 2, q = t & 3;
  const int M = 6 * (D.P / 15);
  const bool mat = gid < M * M, rhs = !mat && gid < M * M + M;
  int r = 0, c = M;
  if (mat) { r = gid / M; c = gid - r * M; } else if (rhs) r = gid - M * M;
  const int i = r < c ? r : c, j = r < c ? c : r;
  const double cs = morbschur::schur_sum4(D.sPart, D.sBlkIndex, D.sNb, D.sNblk, D.sNsplit, (mat || rhs) ? i : 0, (mat || rhs) ? j : 0, q);
  if (q != 0) return;
  if (mat) D.Hs[(size_t)(15 * (r / 6) + r % 6) * D.P + 15 * (c / 6) + c % 6] -= cs;
  else if (rhs) D.bs[15 * (r / 6) + r % 6] -= cs;
}
// dense LDL^T + solve with the lower triangle resident in LDS (dense_ldlt.h): windows up to 15 N = 165
__global__ __launch_bounds__(morbdense::LT) void k_iba_solve_lds(IbaDev D) {
  extern __shared__ double sLd[];
  __shared__ int sOk;
  if (iba_done(D)) return;
  const bool ok = morbdense::ldlt_solve<true>(D.Hs, D.bs, D.x, D.P, sLd, &sOk);
  if (threadIdx.x == 0) D.scal[2] = ok ? 1.0 : 0.0;
}
// larger windows (bLarge, or more than 11 optimizable keyframes): the matrix stays in global memory, one 16-column panel at a time in LDS
// (dense_ldlt.h: ldlt_solve_global; the panel copies move to a global scratch when even they do not fit LDS, P > ~530)
__global__ __launch_bounds__(morbdense::GT) void k_iba_solve_blocked(IbaDev D, int panelInLds) {
  extern __shared__ double sm[];   // dblk[NB * NBP] | y[n] | (pnlL | pnlU when they fit)
  __shared__ int sOk;
  if (iba_done(D)) return;
  double* pnl = panelInLds ? sm + morbdense::global_lds_doubles(D.P) : D.pnlG;
  const bool ok = morbdense::ldlt_solve_global<true>(D.Hs, D.bs, D.x, D.P, pnl, sm, &sOk);
  if (threadIdx.x == 0) D.scal[2] = ok ? 1.0 : 0.0;
}

// back-substitution of the points + oplus of every vertex + the LM scale  sum x (lambda x + b) -> partScale[block]
__global__ __launch_bounds__(256) void k_iba_update(IbaDev D) {
  if (iba_done(D)) return;
  const double lambda = D.lmd[IBA_LMD_LAMBDA];
  const int t = blockIdx.x * 256 + threadIdx.x;
  double sc = 0;
  if (t < D.nMP) {
    const int l = t;
    double Dm[9], Di[9];
    for (int k = 0; k < 9; ++k) Dm[k] = D.Hll[(size_t)l * 9 + k];
    Dm[0] += lambda; Dm[4] += lambda; Dm[8] += lambda;
    inv3(Dm, Di);
    const double* bl = D.b + D.P + 3 * (size_t)l;
    double cl[3] = {bl[0], bl[1], bl[2]};
    for (int k = D.ptStart[l]; k < D.ptStart[l + 1]; ++k) {
      const int e = D.ptEdges[k];
      const int c1 = D.col[D.eKF[e]];
      if (c1 < 0) continue;
      const double* B = D.Hpl + (size_t)e * 18;
      const double* xp = D.x + 15 * c1;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 6; ++r) cl[c] -= B[r * 3 + c] * xp[r];
    }
    for (int r = 0; r < 3; ++r) {
      const double xl = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
      D.x[D.P + 3 * (size_t)l + r] = xl;
      D.pts[3 * (size_t)l + r] += xl;
      sc += xl * (lambda * xl + bl[r]);
    }
  } else if (t - D.nMP < D.nKF) {
    const int k = t - D.nMP;
    const int c = D.col[k];
    if (c >= 0) {
      VIState V;
      iba_load(D.g, D.S + 33 * (size_t)k, V);
      double dx[15];
      for (int q = 0; q < 15; ++q) { dx[q] = D.x[15 * c + q]; sc += dx[q] * (lambda * dx[q] + D.b[15 * c + q]); }
      apply_update(D.g, V, dx);
      iba_store(V, D.S + 33 * (size_t)k);
    }
  }
  const double bs = block_sum(sc);
  if (threadIdx.x == 0) D.partScale[blockIdx.x] = bs;
}
// erase flags (:2773-2801) from the errors of the last computeActiveErrors, and the float outputs
__global__ __launch_bounds__(256) void k_iba_finish(IbaDev D, uint8_t* __restrict__ erase, float* __restrict__ kfOut, float* __restrict__ mpOut) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < D.nE) {
    const float* o = D.eObs + 3 * (size_t)t;
    const bool st = !D.g.rig && !(o[2] < 0);
    const double c = iba_vis_chi2(D, t);
    bool er;
    if (st) er = c > (double)7.815f;
    else {
      const bool bClose = D.mpClose[D.eMP[t]] != 0;
      VIState V;
      iba_load(D.g, D.S + 33 * (size_t)D.eKF[t], V);
      const int cam = D.eRight && D.eRight[t] ? 1 : 0;
      const double r6 = cam ? V.Rcw1[6] : V.Rcw[6], r7 = cam ? V.Rcw1[7] : V.Rcw[7], r8 = cam ? V.Rcw1[8] : V.Rcw[8], tz = cam ? V.tcw1[2] : V.tcw[2];
      const double* X = D.pts + 3 * (size_t)D.eMP[t];
      const bool depthPos = (r6 * X[0] + r7 * X[1] + r8 * X[2] + tz) > 0.0;
      er = (c > (double)5.991f && !bClose) || (c > (double)(1.5f * 5.991f) && bClose) || !depthPos;
    }
    erase[t] = er ? 1 : 0;
  }
  if (t < D.nKF && D.col[t] >= 0) for (int k = 0; k < 21; ++k) kfOut[21 * (size_t)t + k] = (float)D.S[33 * (size_t)t + k];
  if (t < 3 * D.nMP) mpOut[t] = (float)D.pts[t];
}

}  // namespace

extern "C" {

int morb_imu_preintegrate_batch(morb_optimizer* o, int nseq, const int* d_start, const float* d_acc, const float* d_gyro,
                                const float* d_dt, const float* d_bias, const float* ngaDiag6, const float* walkDiag6,
                                morb_imu_preintegrated* d_out, void* stream) {
  MORB_REQUIRE(o && d_start && d_acc && d_gyro && d_dt && d_bias && ngaDiag6 && walkDiag6 && d_out, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nseq > 0, MORB_ERR_INVALID, "nseq must be positive");
  MORB_HIP_CHECK(hipSetDevice(morb_optimizer_device(o)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_optimizer_stream(o);
  morb_imu_preintegrated calib;
  memset(&calib, 0, sizeof calib);
  memcpy(calib.nga, ngaDiag6, sizeof(float) * 6);
  memcpy(calib.ngaWalk, walkDiag6, sizeof(float) * 6);
  hipLaunchKernelGGL(k_imu_preintegrate, dim3(nseq), dim3(64), 0, st, nseq, d_start, d_acc, d_gyro, d_dt, d_bias, calib, d_out);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

// ImuCamPose's constant part: Tcb = Tbc^-1 and, on a fisheye rig, the right camera behind Trl (G2oTypes.cc:96-113).
// rig28 = left KB8 (8), right KB8 (8), Trl rotation (9, row-major) + translation (3), or NULL for one pinhole camera.
static void make_geom(const float* Tbc12, float fx, float fy, float cx, float cy, float bf, const float* rig28, CamGeom& g) {
  memset(&g, 0, sizeof g);
  for (int k = 0; k < 9; ++k) g.Rbc[k] = Tbc12[k];
  for (int k = 0; k < 3; ++k) g.tbc[k] = Tbc12[9 + k];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g.Rcb[r * 3 + c] = g.Rbc[c * 3 + r];   // mTcb = mTbc.inverse()
  for (int r = 0; r < 3; ++r) g.tcb[r] = -(g.Rcb[r * 3] * g.tbc[0] + g.Rcb[r * 3 + 1] * g.tbc[1] + g.Rcb[r * 3 + 2] * g.tbc[2]);
  g.bf = bf; g.fx = fx; g.fy = fy; g.cx = cx; g.cy = cy;
  if (!rig28) return;
  g.rig = 1;
  memcpy(g.kb[0], rig28, 32); memcpy(g.kb[1], rig28 + 8, 32);
  double Rrl[9], trl[3];
  for (int k = 0; k < 9; ++k) Rrl[k] = rig28[16 + k];
  for (int k = 0; k < 3; ++k) trl[k] = rig28[25 + k];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) g.Rcb1[r * 3 + c] = Rrl[r * 3] * g.Rcb[c] + Rrl[r * 3 + 1] * g.Rcb[3 + c] + Rrl[r * 3 + 2] * g.Rcb[6 + c];
    g.tcb1[r] = Rrl[r * 3] * g.tcb[0] + Rrl[r * 3 + 1] * g.tcb[1] + Rrl[r * 3 + 2] * g.tcb[2] + trl[r];
  }
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g.Rbc1[r * 3 + c] = g.Rcb1[c * 3 + r];
  for (int r = 0; r < 3; ++r) g.tbc1[r] = -(g.Rbc1[r * 3] * g.tcb1[0] + g.Rbc1[r * 3 + 1] * g.tcb1[1] + g.Rbc1[r * 3 + 2] * g.tcb1[2]);
}

static int launch_pose_inertial(bool lastFrame, morb_optimizer* o, int nframes, int cap, const int* d_count, const uint8_t* d_hasMP,
                                const float* d_obs, const float* d_invSigma2, const float* d_Xw, const uint8_t* d_close, float fx,
                                float fy, float cx, float cy, float bf, const float* Tbc12, const float* d_state1,
                                const morb_imu_preintegrated* d_pre, const morb_imu_preintegrated* d_preKF,
                                const double* d_prevPrior, int bRecInit, float* d_state, uint8_t* d_outlier, int* d_nInliers,
                                double* d_prior, void* stream, const int* d_nLeft = nullptr, const float* rig28 = nullptr) {
  MORB_REQUIRE(o && d_hasMP && d_obs && d_invSigma2 && d_Xw && d_close && Tbc12 && d_state1 && d_pre && d_state && d_outlier && d_nInliers,
               MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(!lastFrame || (d_preKF && d_prevPrior), MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(!rig28 || d_nLeft, MORB_ERR_INVALID, "a fisheye rig needs d_nLeft");
  MORB_REQUIRE(nframes > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_optimizer_device(o)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_optimizer_stream(o);
  CamGeom g;
  make_geom(Tbc12, fx, fy, cx, cy, bf, rig28, g);
  const size_t edgeLds = (size_t)cap * 32;   // the active visual edges of a frame, eight floats each
  // (round 5 refused frames whose list does not fit the LDS; now they take the same kernel with the list in a global spill buffer of the handle)
  const bool spill = edgeLds + sizeof(InertialWork) + 1024 > 160 * 1024;
  void* gEdge = nullptr;
  if (spill) { const int rc = morb_optimizer_spill(o, (size_t)nframes * edgeLds, &gEdge); if (rc != MORB_OK) return rc; }
#define MORB_LAUNCH_PI2(LF, RG, GE)                                                                                                  \
  do {                                                                                                                               \
    const size_t lds_ = GE ? 0 : edgeLds;                                                                                            \
    MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pose_inertial<LF, RG, GE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
    hipLaunchKernelGGL((k_pose_inertial<LF, RG, GE>), dim3(nframes), dim3(256), lds_, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, d_close, \
                       g, d_state1, d_pre, d_preKF, d_prevPrior, d_nLeft, bRecInit, d_state, d_outlier, d_nInliers, d_prior, (float*)gEdge);  \
  } while (0)
#define MORB_LAUNCH_PI(LF, RG) do { if (spill) MORB_LAUNCH_PI2(LF, RG, true); else MORB_LAUNCH_PI2(LF, RG, false); } while (0)
  if (lastFrame) { if (rig28) MORB_LAUNCH_PI(true, true); else MORB_LAUNCH_PI(true, false); }
  else { if (rig28) MORB_LAUNCH_PI(false, true); else MORB_LAUNCH_PI(false, false); }
#undef MORB_LAUNCH_PI2
#undef MORB_LAUNCH_PI
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_pose_inertial_optimization_last_keyframe_batch(morb_optimizer* o, int nframes, int cap, const int* d_count,
                                                        const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                        const float* d_Xw, const uint8_t* d_close, float fx, float fy, float cx,
                                                        float cy, float bf, const float* Tbc12, const float* d_kfState,
                                                        const morb_imu_preintegrated* d_pre, int bRecInit, float* d_state,
                                                        uint8_t* d_outlier, int* d_nInliers, double* d_prior, void* stream) {
  return launch_pose_inertial(false, o, nframes, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, d_close, fx, fy, cx, cy, bf, Tbc12,
                              d_kfState, d_pre, nullptr, nullptr, bRecInit, d_state, d_outlier, d_nInliers, d_prior, stream);
}

int morb_pose_inertial_optimization_last_frame_batch(morb_optimizer* o, int nframes, int cap, const int* d_count,
                                                     const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                     const float* d_Xw, const uint8_t* d_close, float fx, float fy, float cx, float cy,
                                                     float bf, const float* Tbc12, const float* d_prevState,
                                                     const morb_imu_preintegrated* d_preFrame, const morb_imu_preintegrated* d_preKF,
                                                     const double* d_prevPrior, int bRecInit, float* d_state, uint8_t* d_outlier,
                                                     int* d_nInliers, double* d_prior, void* stream) {
  return launch_pose_inertial(true, o, nframes, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, d_close, fx, fy, cx, cy, bf, Tbc12,
                              d_prevState, d_preFrame, d_preKF, d_prevPrior, bRecInit, d_state, d_outlier, d_nInliers, d_prior, stream);
}

int morb_pose_inertial_optimization_last_keyframe_fisheye_batch(morb_optimizer* o, int nframes, int cap, const int* d_count, const int* d_nLeft,
                                                                const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                                const float* d_Xw, const uint8_t* d_close, const float* rig28,
                                                                const float* Tbc12, const float* d_kfState,
                                                                const morb_imu_preintegrated* d_pre, int bRecInit, float* d_state,
                                                                uint8_t* d_outlier, int* d_nInliers, double* d_prior, void* stream) {
  MORB_REQUIRE(rig28 && d_nLeft, MORB_ERR_INVALID, "NULL argument");
  return launch_pose_inertial(false, o, nframes, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, d_close, 0, 0, 0, 0, 0, Tbc12, d_kfState, d_pre,
                              nullptr, nullptr, bRecInit, d_state, d_outlier, d_nInliers, d_prior, stream, d_nLeft, rig28);
}

int morb_pose_inertial_optimization_last_frame_fisheye_batch(morb_optimizer* o, int nframes, int cap, const int* d_count, const int* d_nLeft,
                                                             const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                             const float* d_Xw, const uint8_t* d_close, const float* rig28, const float* Tbc12,
                                                             const float* d_prevState, const morb_imu_preintegrated* d_preFrame,
                                                             const morb_imu_preintegrated* d_preKF, const double* d_prevPrior, int bRecInit,
                                                             float* d_state, uint8_t* d_outlier, int* d_nInliers, double* d_prior,
                                                             void* stream) {
  MORB_REQUIRE(rig28 && d_nLeft, MORB_ERR_INVALID, "NULL argument");
  return launch_pose_inertial(true, o, nframes, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, d_close, 0, 0, 0, 0, 0, Tbc12, d_prevState,
                              d_preFrame, d_preKF, d_prevPrior, bRecInit, d_state, d_outlier, d_nInliers, d_prior, stream, d_nLeft, rig28);
}

// static void Optimizer::LocalInertialBA(KeyFrame*, bool* pbStopFlag, Map*, int&, int&, int&, int&, bool bLarge, bool bRecInit)
// on the flattened graph (see include/morb_hip.h).  HOST pointers.  eRight / rig28 != NULL: fisheye rig.
static int local_inertial_ba_impl(morb_optimizer* o, int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos,
                           const uint8_t* mpClose, int nE, const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2,
                           int nI, const int* iKF1, const int* iKF2, const morb_imu_preintegrated* iPre, const uint8_t* iRobust,
                           const float* iInfoScale, float fx, float fy, float cx, float cy, float bf, const float* Tbc12, int bLarge,
                           uint8_t* eraseFlag, int* stats3, const uint8_t* eRight, const float* rig28) {
  MORB_REQUIRE((eRight == nullptr) == (rig28 == nullptr), MORB_ERR_INVALID, "eRight and rig28 go together");
  MORB_REQUIRE(o && kfState21 && kfKind && mpPos && mpClose && eKF && eMP && eObs && eInvSigma2 && iKF1 && iKF2 && iPre && iRobust &&
                   iInfoScale && Tbc12 && eraseFlag, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nKF > 0 && nMP > 0 && nE > 0 && nI >= 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_optimizer_device(o)));
  hipStream_t st = (hipStream_t)morb_optimizer_stream(o);
  // ---- host-side graph layout
  std::vector<int> col(nKF, -1);
  int nOpt = 0;
  for (int k = 0; k < nKF; ++k) if (kfKind[k] == 0) col[k] = nOpt++;
  MORB_REQUIRE(nOpt > 0, MORB_ERR_INVALID, "no optimizable keyframe");
  const int P = 15 * nOpt;
  for (int e = 0; e < nE; ++e) MORB_REQUIRE(eKF[e] >= 0 && eKF[e] < nKF && eMP[e] >= 0 && eMP[e] < nMP, MORB_ERR_INVALID, "edge index out of range");
  for (int i = 0; i < nI; ++i) MORB_REQUIRE(iKF1[i] >= 0 && iKF1[i] < nKF && iKF2[i] >= 0 && iKF2[i] < nKF, MORB_ERR_INVALID, "link index out of range");
  std::vector<int> ptStart(nMP + 1, 0), ptEdges(nE);
  for (int e = 0; e < nE; ++e) ++ptStart[eMP[e] + 1];
  for (int l = 0; l < nMP; ++l) ptStart[l + 1] += ptStart[l];
  { std::vector<int> fill(ptStart.begin(), ptStart.end() - 1); for (int e = 0; e < nE; ++e) ptEdges[fill[eMP[e]]++] = e; }
  std::vector<std::vector<int>> byKF(nKF);
  for (int e = 0; e < nE; ++e) if (col[eKF[e]] >= 0) byKF[eKF[e]].push_back(e);
  std::vector<int> kfEdges, chunkKF, chunkStart, chunkEnd;
  for (int k = 0; k < nKF; ++k)
    for (size_t s0 = 0; s0 < byKF[k].size(); s0 += 64) {
      chunkKF.push_back(k); chunkStart.push_back((int)kfEdges.size() + 0);
      const size_t s1 = std::min(byKF[k].size(), s0 + 64);
      for (size_t q = s0; q < s1; ++q) kfEdges.push_back(byKF[k][q]);
      chunkEnd.push_back((int)kfEdges.size());
    }
  const int nChunks = (int)chunkKF.size();

  // ---- device memory: one arena carved from the handle's grow-only workspace
  const size_t nS = (size_t)33 * nKF, nPts = (size_t)3 * nMP, nX = (size_t)P + 3 * nMP;
  const int nI1 = std::max(nI, 1);
  const morbschur::Plan splan = morbschur::make_plan(6 * nOpt + 1, 3 * nMP);
  size_t arenaBytes = 0;
  auto reserve = [&](size_t bytes) { arenaBytes += (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255; };
  for (size_t b : {sizeof(int) * (size_t)nE, sizeof(int) * (size_t)nE, sizeof(float) * 3 * (size_t)nE, sizeof(float) * (size_t)nE,
                   sizeof(int) * (size_t)(nMP + 1), sizeof(int) * (size_t)nE, sizeof(int) * kfEdges.size(), sizeof(int) * (size_t)nChunks,
                   sizeof(int) * (size_t)nChunks, sizeof(int) * (size_t)nChunks, sizeof(int) * (size_t)nKF, sizeof(int) * (size_t)nI,
                   sizeof(int) * (size_t)nI, sizeof(morb_imu_preintegrated) * (size_t)nI, (size_t)nI, (size_t)nMP, sizeof(float) * (size_t)nI,
                   sizeof(float) * 21 * (size_t)nKF, sizeof(float) * 3 * (size_t)nMP,
                   sizeof(double) * nS, sizeof(double) * nS, sizeof(double) * nPts, sizeof(double) * nPts, sizeof(double) * 3 * (size_t)nE,
                   sizeof(double) * 9 * (size_t)nI1, sizeof(double) * 3 * (size_t)nI1, sizeof(double) * 3 * (size_t)nI1,
                   sizeof(double) * 81 * (size_t)nI1, sizeof(double) * 9 * (size_t)nI1, sizeof(double) * 9 * (size_t)nI1,
                   sizeof(double) * (size_t)P * P, sizeof(double) * (size_t)P * P, sizeof(double) * nX, sizeof(double) * (size_t)P,
                   sizeof(double) * nX, sizeof(double) * 9 * (size_t)nMP, sizeof(double) * 18 * (size_t)nE, sizeof(double) * 4,
                   (size_t)nE, (size_t)nE, sizeof(double) * splan.wElems(), sizeof(double) * splan.wElems(), sizeof(double) * splan.partElems(), sizeof(double) * morbdense::global_panel_doubles(P),
                   sizeof(int2) * (size_t)splan.nblk, sizeof(int) * (size_t)splan.nb * splan.nb, sizeof(double) * 8, sizeof(int) * 16,
                   sizeof(int) * (size_t)nI, sizeof(double) * 27 * (size_t)nChunks, sizeof(int) * (size_t)nKF,
                   sizeof(double) * (size_t)div_up(nE + nI, 256), sizeof(double) * (size_t)div_up(nMP + nKF, 256)})
    reserve(b);
  void* arena = nullptr;
  { const int rc = morb_optimizer_workspace(o, arenaBytes, &arena); if (rc != MORB_OK) return rc; }
  size_t arenaOff = 0;
  auto dalloc = [&](size_t bytes) -> void* {
    void* p = (char*)arena + arenaOff;
    arenaOff += (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    return arenaOff <= arenaBytes ? p : nullptr;
  };
  auto cleanup = [&]() {};
  // host -> device: every array goes into a pinned mirror of the arena's upload prefix first and crosses PCIe in ONE copy (twenty
  // pageable hipMemcpyAsync calls, each staged and synchronised by the runtime, were ~0.25 ms of a 1.8 ms call)
  void* stage = nullptr;
  size_t stageCap = 0;
  {
    size_t upBytes = 0;
    for (size_t b : {sizeof(int) * (size_t)nE, sizeof(int) * (size_t)nE, sizeof(float) * 3 * (size_t)nE, sizeof(float) * (size_t)nE,
                     sizeof(int) * (size_t)(nMP + 1), sizeof(int) * (size_t)nE, sizeof(int) * kfEdges.size(), sizeof(int) * (size_t)nChunks,
                     sizeof(int) * (size_t)nChunks, sizeof(int) * (size_t)nChunks, sizeof(int) * (size_t)nKF, sizeof(int) * (size_t)nI,
                     sizeof(int) * (size_t)nI, sizeof(morb_imu_preintegrated) * (size_t)nI, (size_t)nI, (size_t)nMP, sizeof(float) * (size_t)nI,
                     sizeof(float) * 21 * (size_t)nKF, sizeof(float) * 3 * (size_t)nMP, (size_t)nE, sizeof(int2) * (size_t)splan.nblk,
                     sizeof(int) * (size_t)splan.nb * splan.nb, sizeof(int) * (size_t)nI})
      upBytes += (std::max<size_t>(b, 16) + 255) & ~(size_t)255;
    const int rc = morb_optimizer_staging(o, upBytes, &stage);
    if (rc != MORB_OK) return rc;
    stageCap = upBytes;
  }
  size_t upHi = 0;
  auto up = [&](const void* h, size_t bytes) -> void* {
    const size_t off = arenaOff;
    void* d = dalloc(bytes);
    if (d && bytes) {
      if (off + bytes > stageCap) return nullptr;   // (cannot happen: the uploads are the arena's first allocations, sized above)
      memcpy((char*)stage + off, h, bytes); upHi = off + bytes;
    }
    return d;
  };
  IbaDev D;
  memset(&D, 0, sizeof D);
  D.nKF = nKF; D.nMP = nMP; D.nE = nE; D.nI = nI; D.P = P; D.nChunks = nChunks;
  D.eKF = (const int*)up(eKF, sizeof(int) * nE); D.eMP = (const int*)up(eMP, sizeof(int) * nE);
  D.eObs = (const float*)up(eObs, sizeof(float) * 3 * nE); D.eInfo = (const float*)up(eInvSigma2, sizeof(float) * nE);
  D.ptStart = (const int*)up(ptStart.data(), sizeof(int) * (nMP + 1)); D.ptEdges = (const int*)up(ptEdges.data(), sizeof(int) * nE);
  D.kfEdges = (const int*)up(kfEdges.data(), sizeof(int) * kfEdges.size());
  D.chunkKF = (const int*)up(chunkKF.data(), sizeof(int) * nChunks); D.chunkStart = (const int*)up(chunkStart.data(), sizeof(int) * nChunks);
  D.chunkEnd = (const int*)up(chunkEnd.data(), sizeof(int) * nChunks);
  D.col = (const int*)up(col.data(), sizeof(int) * nKF);
  D.iKF1 = (const int*)up(iKF1, sizeof(int) * nI); D.iKF2 = (const int*)up(iKF2, sizeof(int) * nI);
  D.iPre = (const morb_imu_preintegrated*)up(iPre, sizeof(morb_imu_preintegrated) * nI);
  D.iRobust = (const uint8_t*)up(iRobust, nI); D.mpClose = (const uint8_t*)up(mpClose, nMP);
  float* d_scale = (float*)up(iInfoScale, sizeof(float) * nI);
  float* d_kfIn = (float*)up(kfState21, sizeof(float) * 21 * nKF);
  float* d_mpIn = (float*)up(mpPos, sizeof(float) * 3 * nMP);
  D.eRight = eRight ? (const uint8_t*)up(eRight, nE) : nullptr;
  {
    std::vector<int2> blocks; std::vector<int> blkIndex((size_t)splan.nb * splan.nb, 0);
    for (int bi = 0; bi < splan.nb; ++bi) for (int bj = bi; bj < splan.nb; ++bj) { blkIndex[(size_t)bi * splan.nb + bj] = (int)blocks.size(); blocks.push_back(make_int2(bi, bj)); }
    D.sBlocks = (const int2*)up(blocks.data(), sizeof(int2) * blocks.size()); D.sBlkIndex = (const int*)up(blkIndex.data(), sizeof(int) * blkIndex.size());
  }
  // colour the inertial links so that links of one colour share no keyframe (greedy; a chain of consecutive keyframes alternates 0 / 1)
  std::vector<int> linkOrder, colourStart;
  {
    std::vector<int> colour(nI, 0);
    int nColours = 0;
    for (int i = 0; i < nI; ++i) {
      int c = 0;
      for (bool clash = true; clash; ) {
        clash = false;
        for (int j = 0; j < i && !clash; ++j)
          clash = colour[j] == c && (iKF1[j] == iKF1[i] || iKF1[j] == iKF2[i] || iKF2[j] == iKF1[i] || iKF2[j] == iKF2[i]);
        if (clash) ++c;
      }
      colour[i] = c; nColours = std::max(nColours, c + 1);
    }
    for (int c = 0; c < nColours; ++c) { colourStart.push_back((int)linkOrder.size()); for (int i = 0; i < nI; ++i) if (colour[i] == c) linkOrder.push_back(i); }
    colourStart.push_back((int)linkOrder.size());
  }
  D.linkOrder = (const int*)up(linkOrder.data(), sizeof(int) * nI);
  MORB_REQUIRE(D.sBlkIndex != nullptr && D.linkOrder != nullptr && arenaOff <= arenaBytes, MORB_ERR_HIP, "workspace carve-up overflow in morb_local_inertial_ba");
  if (hipMemcpyAsync(arena, stage, upHi, hipMemcpyHostToDevice, st) != hipSuccess) return MORB_ERR_HIP;   // the one upload
  D.kfPart = (double*)dalloc(sizeof(double) * 27 * (size_t)nChunks); D.kfTicket = (int*)dalloc(sizeof(int) * (size_t)nKF);
  MORB_REQUIRE(D.kfTicket != nullptr, MORB_ERR_HIP, "workspace carve-up overflow in morb_local_inertial_ba");
  (void)hipMemsetAsync(D.kfTicket, 0, sizeof(int) * (size_t)nKF, st);   // (the last chunk of a keyframe to arrive resets its counter)
  D.S = (double*)dalloc(sizeof(double) * nS); D.Sbk = (double*)dalloc(sizeof(double) * nS);
  D.pts = (double*)dalloc(sizeof(double) * nPts); D.ptsBk = (double*)dalloc(sizeof(double) * nPts);
  D.nS = (int)nS; D.nPts = (int)nPts;
  D.lmd = (double*)dalloc(sizeof(double) * 8); D.lmi = (int*)dalloc(sizeof(int) * 16);
  if (D.lmi) (void)hipMemsetAsync(D.lmi, 0, sizeof(int) * 16, st);   // (k_iba_errors' arrival counter is live before k_iba_lm_init)
  D.vErr = (double*)dalloc(sizeof(double) * 3 * nE); D.iErr = (double*)dalloc(sizeof(double) * 9 * std::max(nI, 1));
  D.gErr = (double*)dalloc(sizeof(double) * 3 * std::max(nI, 1)); D.aErr = (double*)dalloc(sizeof(double) * 3 * std::max(nI, 1));
  D.InfoI = (double*)dalloc(sizeof(double) * 81 * std::max(nI, 1)); D.InfoG = (double*)dalloc(sizeof(double) * 9 * std::max(nI, 1));
  D.InfoA = (double*)dalloc(sizeof(double) * 9 * std::max(nI, 1));
  D.H = (double*)dalloc(sizeof(double) * (size_t)P * P); D.Hs = (double*)dalloc(sizeof(double) * (size_t)P * P);
  D.b = (double*)dalloc(sizeof(double) * nX); D.bs = (double*)dalloc(sizeof(double) * P); D.x = (double*)dalloc(sizeof(double) * nX);
  D.Hll = (double*)dalloc(sizeof(double) * 9 * nMP); D.Hpl = (double*)dalloc(sizeof(double) * 18 * nE);
  D.scal = (double*)dalloc(sizeof(double) * 4);
  D.partChi = (double*)dalloc(sizeof(double) * div_up(nE + nI, 256)); D.partScale = (double*)dalloc(sizeof(double) * div_up(nMP + nKF, 256)); D.nbUpdate = div_up(nMP + nKF, 256);
  D.pnlG = (double*)dalloc(sizeof(double) * morbdense::global_panel_doubles(P));
  uint8_t* d_erase = (uint8_t*)dalloc(nE);
  MORB_REQUIRE(d_erase != nullptr && arenaOff <= arenaBytes, MORB_ERR_HIP, "workspace carve-up overflow in morb_local_inertial_ba");
  (void)hipMemsetAsync(D.x, 0, sizeof(double) * nX, st);   // the solver's x before the first solve
  make_geom(Tbc12, fx, fy, cx, cy, bf, rig28, D.g);
  {
    D.sW = (double*)dalloc(sizeof(double) * splan.wElems()); D.sWD = (double*)dalloc(sizeof(double) * splan.wElems());
    D.sPart = (double*)dalloc(sizeof(double) * splan.partElems());
    MORB_REQUIRE(D.sPart != nullptr && arenaOff <= arenaBytes, MORB_ERR_HIP, "workspace carve-up overflow in morb_local_inertial_ba");
    D.sMp = splan.Mp; D.sNb = splan.nb; D.sNblk = splan.nblk; D.sNsplit = splan.nsplit;
    // the operands' zero pattern is this graph's: the workspace is reused from call to call
    (void)hipMemsetAsync(D.sW, 0, sizeof(double) * splan.wElems(), st);
    (void)hipMemsetAsync(D.sWD, 0, sizeof(double) * splan.wElems(), st);
  }

  auto fail = [&](const char* what) { cleanup(); set_error("%s", what); return MORB_ERR_HIP; };
  double h[4];
  auto errors = [&](double* chi) -> bool {   // computeActiveErrors + activeRobustChi2
    if (hipMemsetAsync(D.scal, 0, sizeof(double) * 4, st) != hipSuccess) return false;
    hipLaunchKernelGGL(k_iba_errors, dim3(div_up(nE + nI, 256)), dim3(256), 0, st, D, 0);
    if (hipMemcpyAsync(h, D.scal, sizeof(double) * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    *chi = h[0];
    return true;
  };
  hipLaunchKernelGGL(k_iba_setup_kf, dim3(div_up(nKF, 64)), dim3(64), 0, st, D, d_kfIn);
  if (nI) hipLaunchKernelGGL(k_iba_setup_links, dim3(nI), dim3(64), 0, st, D, d_scale);
  hipLaunchKernelGGL(k_iba_setup_pts, dim3(div_up((int)nPts, 256)), dim3(256), 0, st, D, (const float*)d_mpIn);   // points to FP64
  const int Mpose = 6 * nOpt;
  size_t blockedLds = sizeof(double) * (morbdense::global_lds_doubles(P) + morbdense::global_panel_doubles(P));
  const int panelInLds = blockedLds <= 150 * 1024 ? 1 : 0;
  if (!panelInLds) blockedLds = sizeof(double) * morbdense::global_lds_doubles(P);
  MORB_REQUIRE(blockedLds <= 150 * 1024, MORB_ERR_CAPACITY, "window too large for the dense solver");
  if (blockedLds > 48 * 1024)
    MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_iba_solve_blocked), hipFuncAttributeMaxDynamicSharedMemorySize, (int)blockedLds));
  size_t denseLds = sizeof(double) * morbdense::lds_doubles(P);
  if (denseLds > 156 * 1024) denseLds = 0;   // larger windows (bLarge) use the global-memory solver
  if (denseLds) MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_iba_solve_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  double chi = 0;
  if (!errors(&chi)) return fail("k_iba_errors failed");
  const float err0 = (float)chi;
  // ---- Levenberg-Marquardt with the control flow on the device (as grid-mode LocalBA, optimizer.hip): one slot = one trial; the host queues
  // slots one ahead of the decisions and stops when the mapped `done` word appears; kernels queued behind the last decision return at once
  int trials = 0, outer = 0;
  {
    int* hostw = nullptr;
    int* hostwDev = nullptr;
    MORB_REQUIRE(morb_optimizer_lm_words(o, &hostw, &hostwDev) == MORB_OK, MORB_ERR_HIP, "cannot map the LM state words");
    D.lmHost = hostwDev;
    D.optIt = bLarge ? 4 : 10;
    __atomic_store_n(hostw + 0, 0, __ATOMIC_RELAXED); __atomic_store_n(hostw + 1, 0, __ATOMIC_RELEASE);
    hipLaunchKernelGGL(k_iba_lm_init, dim3(1), dim3(1), 0, st, D, chi, bLarge ? 1e-2 : 1e0);
    const int beginGrid = div_up((int)std::max<size_t>(std::max<size_t>(std::max<size_t>(nS, nPts), (size_t)nMP * 8 /* k_iba_points: eight lanes per point */), std::max<size_t>((size_t)P * P, (size_t)18 * nE)), 256);
    for (int slot = 0; slot < 120; ++slot) {
      // backup / restore, then buildSystem (which runs only when the previous trial was accepted)
      hipLaunchKernelGGL(k_iba_points, dim3(beginGrid), dim3(256), 0, st, D);
      if (nChunks) hipLaunchKernelGGL(k_iba_kf, dim3(div_up(nChunks, 4)), dim3(256), 0, st, D);
      for (size_t c = 0; c + 1 < colourStart.size(); ++c)
        hipLaunchKernelGGL(k_iba_links, dim3(colourStart[c + 1] - colourStart[c]), dim3(64), 0, st, D, colourStart[c]);
      hipLaunchKernelGGL(k_iba_pack_w, dim3(div_up(std::max(nE, 3 * nMP), 256)), dim3(256), 0, st, D);
      // the trial: Schur complement of the points on the FP64 matrix cores, one dense product for matrix and right-hand side
      hipLaunchKernelGGL(k_iba_pack_wd, dim3(div_up(std::max(nE, P * P + P), 256)), dim3(256), 0, st, D);
      hipLaunchKernelGGL(morbschur::k_schur_mfma, dim3(splan.nblk, splan.nsplit), dim3(64), 0, st, (const double*)D.sWD, (const double*)D.sW,
                         splan.Mp, splan.ksteps, splan.stepsPerSplit, D.sBlocks, D.sPart, (const int*)(D.lmi + IBA_LM_DONE));
      hipLaunchKernelGGL(k_iba_schur_finish, dim3(div_up(4 * (Mpose * Mpose + Mpose), 256)), dim3(256), 0, st, D);
      if (denseLds) hipLaunchKernelGGL(k_iba_solve_lds, dim3(1), dim3(morbdense::LT), denseLds, st, D);
      else hipLaunchKernelGGL(k_iba_solve_blocked, dim3(1), dim3(morbdense::GT), blockedLds, st, D, panelInLds);
      // a failed solve leaves x as it was (zero at the first trial): g2o still applies the update
      hipLaunchKernelGGL(k_iba_update, dim3(div_up(nMP + nKF, 256)), dim3(256), 0, st, D);
      hipLaunchKernelGGL(k_iba_errors, dim3(div_up(nE + nI, 256)), dim3(256), 0, st, D, 1);
      if (hipGetLastError() != hipSuccess) return fail("LocalInertialBA trial failed");
      unsigned spins = 0;
      while (!__atomic_load_n(hostw + 1, __ATOMIC_ACQUIRE) && __atomic_load_n(hostw + 0, __ATOMIC_ACQUIRE) < slot) {   // one slot ahead
        if ((++spins & 0x3FFu) == 0) {
          const hipError_t q = hipStreamQuery(st);
          if (q == hipSuccess) break;   // (everything queued has run: the words are final)
          if (q != hipErrorNotReady) return fail("LocalInertialBA: device error while waiting for the LM decision");
        }
        __builtin_ia32_pause();
      }
      if (__atomic_load_n(hostw + 1, __ATOMIC_ACQUIRE)) break;
    }
    hipLaunchKernelGGL(k_iba_end, dim3(div_up((int)std::max<size_t>(nS, nPts), 256)), dim3(256), 0, st, D);
    // the loop's counters and the chi2 of the last evaluated trial (h[0] below)
    int hi[16];
    double hd[8];
    if (hipMemcpyAsync(hi, D.lmi, sizeof hi, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hd, D.lmd, sizeof hd, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
      return fail("LocalInertialBA failed");
    outer = hi[IBA_LM_ITS]; trials = hi[IBA_LM_TRIALS];
    h[0] = hd[IBA_LMD_LASTCHI];
  }
  // activeRobustChi2 of the last computed errors = h[0] of the last trial (or the initial one)
  const float errEnd = (float)(trials ? h[0] : chi);
  int okFlag = 1;
  if ((2 * err0 < errEnd || std::isnan(err0) || std::isnan(errEnd)) && !bLarge) okFlag = 0;   // "FAIL LOCAL-INERTIAL BA" (:2808-2813)
  memset(eraseFlag, 0, nE);
  if (okFlag) {
    hipLaunchKernelGGL(k_iba_finish, dim3(div_up(std::max(nE, std::max(nKF, 3 * nMP)), 256)), dim3(256), 0, st, D, d_erase, d_kfIn, d_mpIn);
    (void)hipMemcpyAsync(eraseFlag, d_erase, nE, hipMemcpyDeviceToHost, st);
    (void)hipMemcpyAsync(kfState21, d_kfIn, sizeof(float) * 21 * nKF, hipMemcpyDeviceToHost, st);
    (void)hipMemcpyAsync(mpPos, d_mpIn, sizeof(float) * 3 * nMP, hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) return fail("LocalInertialBA read-back failed");
  }
  if (stats3) { stats3[0] = outer; stats3[1] = trials; stats3[2] = okFlag; }
  cleanup();
  return MORB_OK;
}

int morb_local_inertial_ba(morb_optimizer* o, int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos,
                           const uint8_t* mpClose, int nE, const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2,
                           int nI, const int* iKF1, const int* iKF2, const morb_imu_preintegrated* iPre, const uint8_t* iRobust,
                           const float* iInfoScale, float fx, float fy, float cx, float cy, float bf, const float* Tbc12, int bLarge,
                           uint8_t* eraseFlag, int* stats3) {
  return local_inertial_ba_impl(o, nKF, kfState21, kfKind, nMP, mpPos, mpClose, nE, eKF, eMP, eObs, eInvSigma2, nI, iKF1, iKF2, iPre, iRobust,
                                iInfoScale, fx, fy, cx, cy, bf, Tbc12, bLarge, eraseFlag, stats3, nullptr, nullptr);
}
int morb_local_inertial_ba_fisheye(morb_optimizer* o, int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos,
                                   const uint8_t* mpClose, int nE, const int* eKF, const int* eMP, const float* eObs, const uint8_t* eRight,
                                   const float* eInvSigma2, int nI, const int* iKF1, const int* iKF2, const morb_imu_preintegrated* iPre,
                                   const uint8_t* iRobust, const float* iInfoScale, const float* rig28, const float* Tbc12, int bLarge,
                                   uint8_t* eraseFlag, int* stats3) {
  MORB_REQUIRE(eRight && rig28, MORB_ERR_INVALID, "NULL argument");
  return local_inertial_ba_impl(o, nKF, kfState21, kfKind, nMP, mpPos, mpClose, nE, eKF, eMP, eObs, eInvSigma2, nI, iKF1, iKF2, iPre, iRobust,
                                iInfoScale, 0, 0, 0, 0, 0, Tbc12, bLarge, eraseFlag, stats3, eRight, rig28);
}

#ifdef MORB_INERTIAL_TIMING
int morb_inertial_timing(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; MORB_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_inertialPhase), z, sizeof(z))); return 0; }
  MORB_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_inertialPhase), 16 * sizeof(unsigned long long)));
  return 0;
}
#endif

}  // extern "C"
