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

#include <cstring>

#include "common.h"

using namespace morb;

struct morb_optimizer;
extern "C" {
int morb_optimizer_device(const morb_optimizer*);
void* morb_optimizer_stream(const morb_optimizer*);
}

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
  for (int it = 0; it < 6; ++it) {
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

__global__ __launch_bounds__(64) void k_imu_preintegrate(int nseq, const int* __restrict__ start, const float* __restrict__ acc,
                                                         const float* __restrict__ gyro, const float* __restrict__ dts,
                                                         const float* __restrict__ bias, morb_imu_preintegrated calib,
                                                         morb_imu_preintegrated* __restrict__ out) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= nseq) return;
  morb_imu_preintegrated P;
  memset(&P, 0, sizeof P);
  P.dR[0] = P.dR[4] = P.dR[8] = 1.f;
  for (int k = 0; k < 6; ++k) { P.b[k] = bias[6 * s + k]; P.nga[k] = calib.nga[k]; P.ngaWalk[k] = calib.ngaWalk[k]; }
  float C9[81];   // the 9 x 9 block that the recursion touches; the bias-walk block is diagonal
  for (int k = 0; k < 81; ++k) C9[k] = 0.f;
  float walk[6] = {0, 0, 0, 0, 0, 0};
  for (int i = start[s]; i < start[s + 1]; ++i) {
    const float dt = dts[i];
    const float a[3] = {acc[3 * i] - P.b[0], acc[3 * i + 1] - P.b[1], acc[3 * i + 2] - P.b[2]};
    const float wv[3] = {gyro[3 * i] - P.b[3], gyro[3 * i + 1] - P.b[4], gyro[3 * i + 2] - P.b[5]};
    float Racc[3];
    mul3vf(P.dR, a, Racc);
    for (int k = 0; k < 3; ++k) {
      P.avgA[k] = (P.dT * P.avgA[k] + Racc[k] * dt) / (P.dT + dt);
      P.avgW[k] = (P.dT * P.avgW[k] + wv[k] * dt) / (P.dT + dt);
    }
    for (int k = 0; k < 3; ++k) {
      P.dP[k] = P.dP[k] + P.dV[k] * dt + 0.5f * Racc[k] * dt * dt;
      P.dV[k] = P.dV[k] + Racc[k] * dt;
    }
    float Wacc[9], RW[9], RWJ[9];
    hatf(a, Wacc);
    mul33f(P.dR, Wacc, RW);
    mul33f(RW, P.JRg, RWJ);
    // A = [dRi^T 0 0; -dR dt Wacc, I, 0; -dR dt^2/2 Wacc, I dt, I],  B = [rightJ dt, 0; 0, dR dt; 0, dR dt^2/2]
    float A10[9], A20[9], B11[9], B21[9];
    for (int k = 0; k < 9; ++k) {
      A10[k] = -RW[k] * dt; A20[k] = -0.5f * RW[k] * dt * dt;
      B11[k] = P.dR[k] * dt; B21[k] = 0.5f * P.dR[k] * dt * dt;
    }
    for (int k = 0; k < 9; ++k) {
      P.JPa[k] = P.JPa[k] + P.JVa[k] * dt - 0.5f * P.dR[k] * dt * dt;
      P.JPg[k] = P.JPg[k] + P.JVg[k] * dt - 0.5f * RWJ[k] * dt * dt;
      P.JVa[k] = P.JVa[k] - P.dR[k] * dt;
      P.JVg[k] = P.JVg[k] - RWJ[k] * dt;
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
        const float sn = sinf(d), cs = cosf(d);
        for (int k = 0; k < 9; ++k) {
          const float I = (k & 3) == 0 ? 1.f : 0.f;
          dRi[k] = I + W[k] * sn / d + WW[k] * (1.0f - cs) / d2;
          rJ[k] = I - W[k] * (1.0f - cs) / d2 + WW[k] * (d - sn) / (d2 * d);
        }
      }
    }
    mul33f(P.dR, dRi, P.dR);
    normalize_rotation_f(P.dR);
    // dense 9 x 9 products like the reference (A and B as full matrices, zeros included, same summation order)
    float A[81], B[54];
    for (int k = 0; k < 81; ++k) A[k] = 0.f;
    for (int k = 0; k < 54; ++k) B[k] = 0.f;
    for (int k = 0; k < 9; ++k) A[k * 9 + k] = 1.f;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        A[r * 9 + c] = dRi[c * 3 + r];
        A[(3 + r) * 9 + c] = A10[r * 3 + c]; A[(6 + r) * 9 + c] = A20[r * 3 + c];
        B[r * 6 + c] = rJ[r * 3 + c] * dt;
        B[(3 + r) * 6 + 3 + c] = B11[r * 3 + c]; B[(6 + r) * 6 + 3 + c] = B21[r * 3 + c];
      }
    for (int k = 0; k < 3; ++k) A[(6 + k) * 9 + 3 + k] = dt;
    float AC[81];
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) {
        float sum = 0;
        for (int k = 0; k < 9; ++k) sum += A[r * 9 + k] * C9[k * 9 + c];
        AC[r * 9 + c] = sum;
      }
    float N[81];
    for (int r = 0; r < 9; ++r)
      for (int c = 0; c < 9; ++c) {
        float sum = 0;
        for (int k = 0; k < 9; ++k) sum += AC[r * 9 + k] * A[c * 9 + k];
        float t = 0;
        for (int k = 0; k < 6; ++k) t += B[r * 6 + k] * P.nga[k] * B[c * 6 + k];
        N[r * 9 + c] = sum + t;
      }
    for (int k = 0; k < 81; ++k) C9[k] = N[k];
    for (int k = 0; k < 6; ++k) walk[k] += P.ngaWalk[k];
    float dRiT[9], t9[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dRiT[r * 3 + c] = dRi[c * 3 + r];
    mul33f(dRiT, P.JRg, t9);
    for (int k = 0; k < 9; ++k) P.JRg[k] = t9[k] - rJ[k] * dt;
    P.dT += dt;
  }
  for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) P.C[r * 15 + c] = C9[r * 9 + c];
  for (int k = 0; k < 6; ++k) P.C[(9 + k) * 15 + 9 + k] = walk[k];
  out[s] = P;
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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Shared per-frame constants prepared by thread 0
struct InertialConst {
  double dR[9], dV[3], dP[3], dt;
  double Rbw1[9], twb1[3], v1[3], bg1[3], ba1[3];
  double InfoI[81], InfoG[9], InfoA[9];
};

// n x n inverse in place (Gauss-Jordan, partial pivoting); M is n x 2n scratch
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
// EdgeInertial's information (G2oTypes.cc:484-491): inverse, symmetrise, clamp eigenvalues below 1e-12 to zero.  The eigen
// rebuild only changes the matrix when such an eigenvalue exists: S - 1e-12 I positive definite (LDL^T test) <=> none does,
// and the cyclic-Jacobi rebuild runs otherwise.  S (9 x 9) in/out; scratch >= 171 doubles.
__device__ void clamp_information(double* S, double* scratch) {
  double* A = scratch;          // 81
  double* V = scratch + 81;     // 81
  bool pd = true;
  for (int k = 0; k < 81; ++k) A[k] = S[k];
  for (int k = 0; k < 9; ++k) A[k * 9 + k] -= 1e-12;
  for (int j = 0; j < 9 && pd; ++j) {
    const double d = A[j * 9 + j];
    if (!(d > 0)) { pd = false; break; }
    for (int r = j + 1; r < 9; ++r) {
      const double l = A[r * 9 + j] / d;
      for (int c = j + 1; c <= r; ++c) A[r * 9 + c] -= l * A[c * 9 + j];
    }
  }
  if (pd) return;
  for (int k = 0; k < 81; ++k) { A[k] = S[k]; V[k] = (k % 10 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) { const double t = A[r * 9 + c] * A[r * 9 + c]; if (r == c) diag += t; else off += t; }
    if (off <= 1e-30 * diag) break;
    for (int p = 0; p < 9; ++p)
      for (int q = p + 1; q < 9; ++q) {
        if (A[p * 9 + q] == 0.0) continue;
        const double theta = (A[q * 9 + q] - A[p * 9 + p]) / (2.0 * A[p * 9 + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 9; ++k) { const double a = A[k * 9 + p], b = A[k * 9 + q]; A[k * 9 + p] = c * a - s * b; A[k * 9 + q] = s * a + c * b; }
        for (int k = 0; k < 9; ++k) { const double a = A[p * 9 + k], b = A[q * 9 + k]; A[p * 9 + k] = c * a - s * b; A[q * 9 + k] = s * a + c * b; }
        for (int k = 0; k < 9; ++k) { const double a = V[k * 9 + p], b = V[k * 9 + q]; V[k * 9 + p] = c * a - s * b; V[k * 9 + q] = s * a + c * b; }
      }
  }
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      double s = 0;
      for (int k = 0; k < 9; ++k) { const double e = A[k * 9 + k] < 1e-12 ? 0.0 : A[k * 9 + k]; s += V[r * 9 + k] * e * V[c * 9 + k]; }
      S[r * 9 + c] = s;
    }
}

struct CamGeom { double Rcb[9], tcb[3], Rbc[9], tbc[3], bf; float fx, fy, cx, cy; };

struct VIState {
  double Rwb[9], twb[3], v[3], bg[3], ba[3];
  double Rcw[9], tcw[3];
};
__device__ void refresh_camera(const CamGeom& g, VIState& S) {   // G2oTypes.cc:209-215
  double Rbw[9], tbw[3];
  transpose33(S.Rwb, Rbw);
  mul3v(Rbw, S.twb, tbw);
  for (int k = 0; k < 3; ++k) tbw[k] = -tbw[k];
  mul33(g.Rcb, Rbw, S.Rcw);
  mul3v(g.Rcb, tbw, S.tcw);
  for (int k = 0; k < 3; ++k) S.tcw[k] += g.tcb[k];
}
__device__ void apply_update(const CamGeom& g, VIState& S, const double* x) {   // ImuCamPose::Update + the additive vertices
  double t[3], dR[9];
  mul3v(S.Rwb, x + 3, t);
  for (int k = 0; k < 3; ++k) S.twb[k] += t[k];
  exp_so3(x, dR);
  mul33(S.Rwb, dR, S.Rwb);
  refresh_camera(g, S);
  for (int k = 0; k < 3; ++k) { S.v[k] += x[6 + k]; S.bg[k] += x[9 + k]; S.ba[k] += x[12 + k]; }
}
// visual edge: error (obs - projection) and chi2; st = stereo
__device__ __forceinline__ double vis_error(const CamGeom& g, const VIState& S, const double* X, const float* o, bool st, double info,
                                            double* err, double* Xc) {
  mul3v(S.Rcw, X, Xc);
  for (int k = 0; k < 3; ++k) Xc[k] += S.tcw[k];
  const double u = g.fx * Xc[0] / Xc[2] + g.cx, v = g.fy * Xc[1] / Xc[2] + g.cy;   // Pinhole.cpp:38-44
  err[0] = (double)o[0] - u; err[1] = (double)o[1] - v; err[2] = 0;
  double c = err[0] * info * err[0] + err[1] * info * err[1];
  if (st) { const double invZ = 1 / Xc[2]; err[2] = (double)o[2] - (u - g.bf * invZ); c += err[2] * info * err[2]; }
  return c;
}
__device__ __forceinline__ void vis_jacobian(const CamGeom& g, const double* Xc, bool st, double* J /*[3][6]*/) {   // G2oTypes.cc:361-442
  double Xb[3];
  mul3v(g.Rbc, Xc, Xb);
  for (int k = 0; k < 3; ++k) Xb[k] += g.tbc[k];
  double pj[9];
  pj[0] = g.fx / Xc[2]; pj[1] = 0; pj[2] = -g.fx * Xc[0] / (Xc[2] * Xc[2]);
  pj[3] = 0; pj[4] = g.fy / Xc[2]; pj[5] = -g.fy * Xc[1] / (Xc[2] * Xc[2]);
  pj[6] = pj[7] = pj[8] = 0;
  if (st) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + g.bf * (1.0 / (Xc[2] * Xc[2])); }
  const double x = Xb[0], y = Xb[1], z = Xb[2];
  const double Sd[18] = {0.0, z, -y, 1.0, 0.0, 0.0, -z, 0.0, x, 0.0, 1.0, 0.0, y, -x, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double PR[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) PR[c] = pj[r * 3] * g.Rcb[c] + pj[r * 3 + 1] * g.Rcb[3 + c] + pj[r * 3 + 2] * g.Rcb[6 + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) J[r * 6 + c] = PR[0] * Sd[c] + PR[1] * Sd[6 + c] + PR[2] * Sd[12 + c];
  }
}
__device__ __forceinline__ double huber_w(double delta, double e2) {   // rho'(e2), robust_kernel_impl.cpp:65-91
  return e2 <= delta * delta ? 1.0 : delta / sqrt(e2);
}
// inertial edge at state S: error and the Jacobian blocks w.r.t. (pose 2, velocity 2) as one 9 x 9 matrix J   G2oTypes.cc:494-585
__device__ void inertial_edge(const InertialConst& K, const VIState& S, double* err, double* J) {
  double dRt[9], M[9], eR[9], er[3];
  transpose33(K.dR, dRt);
  mul33(dRt, K.Rbw1, M);
  mul33(M, S.Rwb, eR);
  log_so3(eR, er);
  const double g[3] = {0, 0, -(double)9.81f};
  double t[3], ev[3], ep[3];
  for (int k = 0; k < 3; ++k) t[k] = S.v[k] - K.v1[k] - g[k] * K.dt;
  mul3v(K.Rbw1, t, ev);
  for (int k = 0; k < 3; ++k) t[k] = S.twb[k] - K.twb1[k] - K.v1[k] * K.dt - g[k] * K.dt * K.dt / 2;
  mul3v(K.Rbw1, t, ep);
  for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = ev[k] - K.dV[k]; err[6 + k] = ep[k] - K.dP[k]; }
  if (J) {
    double invJr[9], RR[9];
    inv_right_jacobian_so3(er, invJr);
    mul33(K.Rbw1, S.Rwb, RR);
    for (int k = 0; k < 81; ++k) J[k] = 0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) { J[r * 9 + c] = invJr[r * 3 + c]; J[(6 + r) * 9 + 3 + c] = RR[r * 3 + c]; J[(3 + r) * 9 + 6 + c] = K.Rbw1[r * 3 + c]; }
  }
}
// H += J^T Info J (9 x 9 into the 15 x 15), b -= J^T Info e
__device__ void add_inertial(const InertialConst& K, const double* err, const double* J, double* H, double* b) {
  for (int r = 0; r < 9; ++r) {
    double JtO[9];   // row r of J^T Info
    for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += J[k * 9 + r] * K.InfoI[k * 9 + c]; JtO[c] = s; }
    if (b) { double s = 0; for (int k = 0; k < 9; ++k) s += JtO[k] * err[k]; b[r] -= s; }
    for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += JtO[k] * J[k * 9 + c]; H[r * 15 + c] += s; }
  }
}
// LinearSolverDense: LDL^T, solution only when positive (linear_solver_dense.h:104-112)
__device__ bool ldlt15(double* A /* 15 x 15, destroyed */, const double* rhs, double* x) {
  for (int j = 0; j < 15; ++j) {
    const double d = A[j * 15 + j];
    if (!(d > 0)) return false;
    for (int r = j + 1; r < 15; ++r) {
      const double l = A[r * 15 + j] / d;
      for (int c = j + 1; c <= r; ++c) A[r * 15 + c] -= l * A[c * 15 + j];
      A[r * 15 + j] = l;
    }
  }
  double y[15];
  for (int r = 0; r < 15; ++r) { double s = rhs[r]; for (int c = 0; c < r; ++c) s -= A[r * 15 + c] * y[c]; y[r] = s; }
  for (int k = 0; k < 15; ++k) y[k] /= A[k * 15 + k];
  for (int r = 14; r >= 0; --r) { double s = y[r]; for (int c = r + 1; c < 15; ++c) s -= A[c * 15 + r] * y[c]; y[r] = s; }
  for (int k = 0; k < 15; ++k) x[k] = y[k];
  return true;
}

__global__ __launch_bounds__(256) void k_pose_inertial(int cap, const int* __restrict__ count, const uint8_t* __restrict__ hasMP,
                                                       const float* __restrict__ obs, const float* __restrict__ invSigma2,
                                                       const float* __restrict__ Xw, const uint8_t* __restrict__ closeFlag,
                                                       CamGeom g, const float* __restrict__ kfState,
                                                       const morb_imu_preintegrated* __restrict__ pre, int bRecInit,
                                                       float* __restrict__ stateIO, uint8_t* __restrict__ outlier,
                                                       int* __restrict__ nInliersOut, double* __restrict__ prior) {
  __shared__ InertialConst K;
  __shared__ double sScratch[9 * 18 + 81];
  __shared__ double sRed[4][28];
  __shared__ double sX[16];
  __shared__ int sCnt[4][2];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = count ? count[f] : cap;
  const size_t base = (size_t)f * cap;
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);

  int nInit = 0;
  for (int i = tid; i < n; i += 256) if (hasMP[base + i]) { ++nInit; outlier[base + i] = 0; }
  {
    nInit = (int)wave_sum((double)nInit);
    if (lane == 0) sCnt[wv][0] = nInit;
    __syncthreads();
    nInit = sCnt[0][0] + sCnt[1][0] + sCnt[2][0] + sCnt[3][0];
    __syncthreads();
  }

  if (tid == 0) {   // the inertial edge's constants (GetDelta*, ImuTypes.cc:289-312; information, G2oTypes.cc:484-491)
    const morb_imu_preintegrated& P = pre[f];
    const float* ks = kfState + 21 * f;
    double Rwb1[9];
    for (int k = 0; k < 9; ++k) Rwb1[k] = ks[k];
    transpose33(Rwb1, K.Rbw1);
    for (int k = 0; k < 3; ++k) { K.twb1[k] = ks[9 + k]; K.v1[k] = ks[12 + k]; K.bg1[k] = ks[15 + k]; K.ba1[k] = ks[18 + k]; }
    const float dbg[3] = {ks[15] - P.b[3], ks[16] - P.b[4], ks[17] - P.b[5]};
    const float dba[3] = {ks[18] - P.b[0], ks[19] - P.b[1], ks[20] - P.b[2]};
    float w[3], W[9], WW[9], E[9], dRf[9];
    mul3vf(P.JRg, dbg, w);
    const float t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrtf(t2);
    hatf(w, W);
    mul33f(W, W, WW);
    if (t < 1e-5f) for (int k = 0; k < 9; ++k) E[k] = ((k & 3) == 0 ? 1.f : 0.f) + W[k] + 0.5f * WW[k];
    else { const float sn = sinf(t), cs = cosf(t); for (int k = 0; k < 9; ++k) E[k] = ((k & 3) == 0 ? 1.f : 0.f) + W[k] * sn / t + WW[k] * (1.0f - cs) / t2; }
    mul33f(P.dR, E, dRf);
    normalize_rotation_f(dRf);
    for (int k = 0; k < 9; ++k) K.dR[k] = dRf[k];
    float g1[3], a1[3];
    mul3vf(P.JVg, dbg, g1); mul3vf(P.JVa, dba, a1);
    for (int k = 0; k < 3; ++k) K.dV[k] = (double)(P.dV[k] + g1[k] + a1[k]);
    mul3vf(P.JPg, dbg, g1); mul3vf(P.JPa, dba, a1);
    for (int k = 0; k < 3; ++k) K.dP[k] = (double)(P.dP[k] + g1[k] + a1[k]);
    K.dt = P.dT;
    double C9[81];
    for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C9[r * 9 + c] = (double)P.C[r * 15 + c];
    if (invert_n(C9, 9, K.InfoI, sScratch)) {
      for (int r = 0; r < 9; ++r) for (int c = r + 1; c < 9; ++c) { const double s = (K.InfoI[r * 9 + c] + K.InfoI[c * 9 + r]) / 2; K.InfoI[r * 9 + c] = s; K.InfoI[c * 9 + r] = s; }
      clamp_information(K.InfoI, sScratch);
    } else for (int k = 0; k < 81; ++k) K.InfoI[k] = 0;
    double Cg[9], Ca[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Cg[r * 3 + c] = P.C[(9 + r) * 15 + 9 + c]; Ca[r * 3 + c] = P.C[(12 + r) * 15 + 12 + c]; }
    if (!invert_n(Cg, 3, K.InfoG, sScratch)) for (int k = 0; k < 9; ++k) K.InfoG[k] = 0;
    if (!invert_n(Ca, 3, K.InfoA, sScratch)) for (int k = 0; k < 9; ++k) K.InfoA[k] = 0;
  }
  __syncthreads();

  VIState S, Sprev;
  {
    const float* s0 = stateIO + 21 * f;
    for (int k = 0; k < 9; ++k) S.Rwb[k] = s0[k];
    for (int k = 0; k < 3; ++k) { S.twb[k] = s0[9 + k]; S.v[k] = s0[12 + k]; S.bg[k] = s0[15 + k]; S.ba[k] = s0[18 + k]; }
    refresh_camera(g, S);
    Sprev = S;
  }
  bool robust = true;
  int nBad = 0, nInl = 0;
  double xPrev[15];
  for (int k = 0; k < 15; ++k) xPrev[k] = 0;
  const float chi2Mono[4] = {12.f, 7.5f, 5.991f, 5.991f}, chi2Stereo[4] = {15.6f, 9.8f, 7.815f, 7.815f};

  for (int it = 0; it < 4; ++it) {
    bool ok = true;
    for (int iter = 0; iter < 10 && ok; ++iter) {
      Sprev = S;   // the state the active edges' errors belong to
      double acc[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[k] = 0;
      for (int i = tid; i < n; i += 256) {
        if (!hasMP[base + i] || outlier[base + i]) continue;
        const float* o = obs + (base + i) * 3;
        const bool st = !(o[2] < 0);
        const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
        const double info = (double)invSigma2[base + i];
        double err[3], Xc[3], J[18];
        const double c = vis_error(g, S, X, o, st, info, err, Xc);
        const double w = robust ? huber_w(st ? deltaStereo : deltaMono, c) : 1.0;
        vis_jacobian(g, Xc, st, J);
        const int d = st ? 3 : 2;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double bb = 0;
          for (int k = 0; k < d; ++k) bb += J[k * 6 + r] * (info * err[k]);
          acc[21 + r] -= w * bb;
#pragma unroll
          for (int cc = r; cc < 6; ++cc) {
            double h = 0;
            for (int k = 0; k < d; ++k) h += J[k * 6 + r] * (w * info) * J[k * 6 + cc];
            acc[q++] += h;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[k] = wave_sum(acc[k]);
      __syncthreads();
      if (lane == 0) for (int k = 0; k < 27; ++k) sRed[wv][k] = acc[k];
      __syncthreads();
      if (tid == 0) {
        double H[225], b[15];
        for (int k = 0; k < 225; ++k) H[k] = 0;
        for (int k = 0; k < 15; ++k) b[k] = 0;
        int q = 0;
        for (int r = 0; r < 6; ++r) for (int cc = r; cc < 6; ++cc) { const double t = sRed[0][q] + sRed[1][q] + sRed[2][q] + sRed[3][q]; H[r * 15 + cc] = t; H[cc * 15 + r] = t; ++q; }
        for (int r = 0; r < 6; ++r) b[r] = sRed[0][21 + r] + sRed[1][21 + r] + sRed[2][21 + r] + sRed[3][21 + r];
        double err[9];
        double* J = sScratch;   // 81
        inertial_edge(K, S, err, J);
        add_inertial(K, err, J, H, b);
        for (int r = 0; r < 3; ++r) {   // EdgeGyroRW / EdgeAccRW (G2oTypes.h:645-654)
          double sg = 0, sa = 0;
          for (int k = 0; k < 3; ++k) { sg += K.InfoG[r * 3 + k] * (S.bg[k] - K.bg1[k]); sa += K.InfoA[r * 3 + k] * (S.ba[k] - K.ba1[k]); }
          b[9 + r] -= sg; b[12 + r] -= sa;
          for (int c = 0; c < 3; ++c) { H[(9 + r) * 15 + 9 + c] += K.InfoG[r * 3 + c]; H[(12 + r) * 15 + 12 + c] += K.InfoA[r * 3 + c]; }
        }
        double x[15];
        for (int k = 0; k < 15; ++k) x[k] = xPrev[k];   // a failed solve leaves the solver's previous x in place
        const bool good = ldlt15(H, b, x);
        for (int k = 0; k < 15; ++k) sX[k] = x[k];
        sX[15] = good ? 1.0 : 0.0;
      }
      __syncthreads();
      double x[15];
      for (int k = 0; k < 15; ++k) { x[k] = sX[k]; xPrev[k] = x[k]; }
      ok = sX[15] != 0.0;
      apply_update(g, S, x);
    }
    // ---- classification (Optimizer.cc:4619-4676)
    int bad = 0, inl = 0;
    const float chi2close = 1.5f * chi2Mono[it];
    for (int i = tid; i < n; i += 256) {
      if (!hasMP[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const bool st = !(o[2] < 0);
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      double err[3], Xc[3];
      const float chi2 = (float)vis_error(g, outlier[base + i] ? S : Sprev, X, o, st, (double)invSigma2[base + i], err, Xc);
      bool isOut;
      if (st) isOut = chi2 > chi2Stereo[it];
      else {
        const bool bClose = closeFlag[base + i] != 0;
        const bool depthPos = (S.Rcw[6] * X[0] + S.Rcw[7] * X[1] + S.Rcw[8] * X[2] + S.tcw[2]) > 0.0;
        isOut = (chi2 > chi2Mono[it] && !bClose) || (bClose && chi2 > chi2close) || !depthPos;
      }
      outlier[base + i] = isOut ? 1 : 0;
      bad += isOut ? 1 : 0; inl += isOut ? 0 : 1;
    }
    bad = (int)wave_sum((double)bad); inl = (int)wave_sum((double)inl);
    __syncthreads();
    if (lane == 0) { sCnt[wv][0] = bad; sCnt[wv][1] = inl; }
    __syncthreads();
    nBad = sCnt[0][0] + sCnt[1][0] + sCnt[2][0] + sCnt[3][0];
    nInl = sCnt[0][1] + sCnt[1][1] + sCnt[2][1] + sCnt[3][1];
    __syncthreads();
    if (it == 2) robust = false;
    if (nInit + 3 < 10) break;   // optimizer.edges().size() < 10
  }

  if (nInl < 30 && !bRecInit) {   // :4683-4707
    int bad = 0;
    for (int i = tid; i < n; i += 256) {
      if (!hasMP[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const bool st = !(o[2] < 0);
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      double err[3], Xc[3];
      const float chi2 = (float)vis_error(g, S, X, o, st, (double)invSigma2[base + i], err, Xc);
      if (chi2 < (st ? 24.f : 18.f)) outlier[base + i] = 0; else ++bad;
    }
    bad = (int)wave_sum((double)bad);
    __syncthreads();
    if (lane == 0) sCnt[wv][0] = bad;
    __syncthreads();
    nBad = sCnt[0][0] + sCnt[1][0] + sCnt[2][0] + sCnt[3][0];
    __syncthreads();
  }

  if (tid == 0) {
    float* s0 = stateIO + 21 * f;
    for (int k = 0; k < 9; ++k) s0[k] = (float)S.Rwb[k];
    for (int k = 0; k < 3; ++k) { s0[9 + k] = (float)S.twb[k]; s0[12 + k] = (float)S.v[k]; s0[15 + k] = (float)S.bg[k]; s0[18 + k] = (float)S.ba[k]; }
    nInliersOut[f] = nInit - nBad;
  }
  if (prior) {   // ConstraintPoseImu (:4717-4754): H from the un-robustified Jacobians at the final state, inliers only
    double acc[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) acc[k] = 0;
    for (int i = tid; i < n; i += 256) {
      if (!hasMP[base + i] || outlier[base + i]) continue;
      const float* o = obs + (base + i) * 3;
      const bool st = !(o[2] < 0);
      const double X[3] = {(double)Xw[(base + i) * 3], (double)Xw[(base + i) * 3 + 1], (double)Xw[(base + i) * 3 + 2]};
      const double info = (double)invSigma2[base + i];
      double err[3], Xc[3], J[18];
      vis_error(g, S, X, o, st, info, err, Xc);
      vis_jacobian(g, Xc, st, J);
      const int d = st ? 3 : 2;
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = r; cc < 6; ++cc) {
          double h = 0;
          for (int k = 0; k < d; ++k) h += J[k * 6 + r] * info * J[k * 6 + cc];
          acc[q++] += h;
        }
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) acc[k] = wave_sum(acc[k]);
    __syncthreads();
    if (lane == 0) for (int k = 0; k < 21; ++k) sRed[wv][k] = acc[k];
    __syncthreads();
    if (tid == 0) {
      double* out = prior + (size_t)246 * f;
      for (int k = 0; k < 9; ++k) out[k] = S.Rwb[k];
      for (int k = 0; k < 3; ++k) { out[9 + k] = S.twb[k]; out[12 + k] = S.v[k]; out[15 + k] = S.bg[k]; out[18 + k] = S.ba[k]; }
      double* H = out + 21;
      for (int k = 0; k < 225; ++k) H[k] = 0;
      double err[9];
      double* J = sScratch;
      inertial_edge(K, S, err, J);
      add_inertial(K, err, J, H, nullptr);
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { H[(9 + r) * 15 + 9 + c] += K.InfoG[r * 3 + c]; H[(12 + r) * 15 + 12 + c] += K.InfoA[r * 3 + c]; }
      int q = 0;
      for (int r = 0; r < 6; ++r)
        for (int cc = r; cc < 6; ++cc) {
          const double t = sRed[0][q] + sRed[1][q] + sRed[2][q] + sRed[3][q];
          H[r * 15 + cc] += t;
          if (cc != r) H[cc * 15 + r] += t;
          ++q;
        }
    }
  }
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
  hipLaunchKernelGGL(k_imu_preintegrate, dim3(div_up(nseq, 64)), dim3(64), 0, st, nseq, d_start, d_acc, d_gyro, d_dt, d_bias, calib, d_out);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_pose_inertial_optimization_last_keyframe_batch(morb_optimizer* o, int nframes, int cap, const int* d_count,
                                                        const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                        const float* d_Xw, const uint8_t* d_close, float fx, float fy, float cx,
                                                        float cy, float bf, const float* Tbc12, const float* d_kfState,
                                                        const morb_imu_preintegrated* d_pre, int bRecInit, float* d_state,
                                                        uint8_t* d_outlier, int* d_nInliers, double* d_prior, void* stream) {
  MORB_REQUIRE(o && d_hasMP && d_obs && d_invSigma2 && d_Xw && d_close && Tbc12 && d_kfState && d_pre && d_state && d_outlier && d_nInliers,
               MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_optimizer_device(o)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_optimizer_stream(o);
  CamGeom g;
  for (int k = 0; k < 9; ++k) g.Rbc[k] = Tbc12[k];
  for (int k = 0; k < 3; ++k) g.tbc[k] = Tbc12[9 + k];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g.Rcb[r * 3 + c] = g.Rbc[c * 3 + r];   // mTcb = mTbc.inverse()
  for (int r = 0; r < 3; ++r) g.tcb[r] = -(g.Rcb[r * 3] * g.tbc[0] + g.Rcb[r * 3 + 1] * g.tbc[1] + g.Rcb[r * 3 + 2] * g.tbc[2]);
  g.bf = bf; g.fx = fx; g.fy = fy; g.cx = cx; g.cy = cy;
  hipLaunchKernelGGL(k_pose_inertial, dim3(nframes), dim3(256), 0, st, cap, d_count, d_hasMP, d_obs, d_invSigma2, d_Xw, d_close, g,
                     d_kfState, d_pre, bRecInit, d_state, d_outlier, d_nInliers, d_prior);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

}  // extern "C"
