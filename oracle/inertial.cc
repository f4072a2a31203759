// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
//
// CPU restatement of the first slice of SURVEY §8(f) row N1 (visual-inertial tracking):
//   IMU::Preintegrated::IntegrateNewMeasurement / GetDelta{Rotation,Velocity,Position}   src/ImuTypes.cc:84-107, :191-247, :289-312
//   Optimizer::PoseInertialOptimizationLastKeyFrame                                      src/Optimizer.cc:4391-4757
//   ImuCamPose (Project / ProjectStereo / isDepthPositive / Update)                      src/G2oTypes.cc:74-216
//   EdgeMonoOnlyPose / EdgeStereoOnlyPose                                                include/G2oTypes.h:390-423, :466-493, src/G2oTypes.cc:361-442
//   EdgeInertial (information, error, Jacobians, GetHessian2)                            src/G2oTypes.cc:472-585, include/G2oTypes.h:531-537
//   EdgeGyroRW / EdgeAccRW                                                               include/G2oTypes.h:635-704
//   ExpSO3 / LogSO3 / InverseRightJacobianSO3                                            src/G2oTypes.cc:779-829
//   g2o Gauss-Newton (computeActiveErrors, buildSystem, dense LDLT solve, update)        Thirdparty/g2o/g2o/core/optimization_algorithm_gauss_newton.cpp:51-96,
//                                                                                        core/sparse_optimizer.cpp:354-420, solvers/linear_solver_dense.h:65-113
//   Pinhole::project / projectJac (double overloads, float parameters)                   src/CameraModels/Pinhole.cpp:38-44, :76-86
// PARITY UNPINNED: the reference's tests hold no vectors for these functions and the reference does not build here (Eigen,
// Sophus, g2o's Eigen dependency are absent).  Eigen / Sophus pieces are restated from their published formulas:
//   * NormalizeRotation = U V^T of a JacobiSVD: restated as the polar factor (Newton iteration in FP64).  In ExpSO3 (FP64) the
//     argument is already orthonormal to rounding, so the projection is omitted there; ImuCamPose::Update calls
//     NormalizeRotation(Rwb) and DISCARDS the result (src/G2oTypes.cc:204-207), i.e. it does not normalise: restated as such.
//   * Sophus::SO3f::exp(v).matrix() is restated with Rodrigues' formula in float.
//   * Eigen::LDLT (pivoted) of the 15 x 15 system: restated as an LDL^T with diagonal pivoting and the same isPositive() test.
//   * EdgeInertial's information: inverse of C(0:9,0:9) in FP64, symmetrised, eigenvalues below 1e-12 clamped to zero
//     (cyclic Jacobi eigen-decomposition instead of Eigen::SelfAdjointEigenSolver).
// Pinhole cameras only (mono and rectified-stereo observations of camera 0); the KannalaBrandt8 second camera is not restated.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "matcher.h"
#include "orb_oracle.h"

namespace orc {
namespace imu {

// ---- float 3x3 helpers (row-major) ------------------------------------------------------------------------------------
static void mul33f(const float* A, const float* B, float* C) {
  float T[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
  memcpy(C, T, sizeof T);
}
static void mul3vf(const float* A, const float* v, float* o) {
  float t[3];
  for (int r = 0; r < 3; ++r) t[r] = A[r * 3] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2];
  memcpy(o, t, sizeof t);
}
static void hatf(const float* v, float* W) {
  W[0] = 0; W[1] = -v[2]; W[2] = v[1]; W[3] = v[2]; W[4] = 0; W[5] = -v[0]; W[6] = -v[1]; W[7] = v[0]; W[8] = 0;
}
static void transpose33f(const float* A, float* T) {
  float t[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) t[r * 3 + c] = A[c * 3 + r];
  memcpy(T, t, sizeof t);
}
// polar factor of a near-rotation (= U V^T of its SVD), FP64 Newton iteration X <- (X + X^-T) / 2
static void polar33(const double* A, double* Q) {
  double X[9];
  memcpy(X, A, sizeof X);
  for (int it = 0; it < 6; ++it) {
    const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
    const double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
    const double inv = 1.0 / det;
    // inverse transpose = cofactor matrix / det
    const double C[9] = {c00, c01, c02,
                         X[2] * X[7] - X[1] * X[8], X[0] * X[8] - X[2] * X[6], X[1] * X[6] - X[0] * X[7],
                         X[1] * X[5] - X[2] * X[4], X[2] * X[3] - X[0] * X[5], X[0] * X[4] - X[1] * X[3]};
    for (int k = 0; k < 9; ++k) X[k] = 0.5 * (X[k] + C[k] * inv);
  }
  memcpy(Q, X, sizeof X);
}
static void normalizeRotationF(float* R) {  // ImuTypes.cc:35-39
  double A[9], Q[9];
  for (int k = 0; k < 9; ++k) A[k] = R[k];
  polar33(A, Q);
  for (int k = 0; k < 9; ++k) R[k] = (float)Q[k];
}

// IntegratedRotation (ImuTypes.cc:84-107)
static void integratedRotation(const float* w, const float* b, float dt, float* deltaR, float* rightJ) {
  const float eps = 1e-4f;  // ImuTypes.cc:33
  const float x = (w[0] - b[3]) * dt, y = (w[1] - b[4]) * dt, z = (w[2] - b[5]) * dt;
  const float d2 = x * x + y * y + z * z, d = std::sqrt(d2);
  const float v[3] = {x, y, z};
  float W[9], WW[9];
  hatf(v, W);
  mul33f(W, W, WW);
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d < eps) {
    for (int k = 0; k < 9; ++k) { deltaR[k] = I[k] + W[k]; rightJ[k] = I[k]; }
  } else {
    const float s = std::sin(d), c = std::cos(d);
    for (int k = 0; k < 9; ++k) {
      deltaR[k] = I[k] + W[k] * s / d + WW[k] * (1.0f - c) / d2;
      rightJ[k] = I[k] - W[k] * (1.0f - c) / d2 + WW[k] * (d - s) / (d2 * d);
    }
  }
}

static void initialize(orc_imu_preintegrated* P, const float* bias, const float* ngaDiag, const float* walkDiag) {  // :152-170
  memset(P, 0, sizeof *P);
  P->dR[0] = P->dR[4] = P->dR[8] = 1.f;
  memcpy(P->b, bias, sizeof(float) * 6);
  memcpy(P->nga, ngaDiag, sizeof(float) * 6);
  memcpy(P->ngaWalk, walkDiag, sizeof(float) * 6);
}

// Preintegrated::IntegrateNewMeasurement (ImuTypes.cc:191-247); bias layout b = (bax, bay, baz, bwx, bwy, bwz)
static void integrate(orc_imu_preintegrated* P, const float* a, const float* w, float dt) {
  float A[81], B[54];
  memset(A, 0, sizeof A); memset(B, 0, sizeof B);
  for (int k = 0; k < 9; ++k) A[k * 9 + k] = 1.f;
  const float acc[3] = {a[0] - P->b[0], a[1] - P->b[1], a[2] - P->b[2]};
  const float accW[3] = {w[0] - P->b[3], w[1] - P->b[4], w[2] - P->b[5]};
  float Racc[3];
  mul3vf(P->dR, acc, Racc);
  for (int k = 0; k < 3; ++k) {
    P->avgA[k] = (P->dT * P->avgA[k] + Racc[k] * dt) / (P->dT + dt);
    P->avgW[k] = (P->dT * P->avgW[k] + accW[k] * dt) / (P->dT + dt);
  }
  for (int k = 0; k < 3; ++k) {
    P->dP[k] = P->dP[k] + P->dV[k] * dt + 0.5f * Racc[k] * dt * dt;
    P->dV[k] = P->dV[k] + Racc[k] * dt;
  }
  // Scalar factors are applied where the C++ expressions apply them (operator precedence, left to right):
  // -dR * dt * Wacc = ((-dR) * dt) * Wacc, 0.5f * dR * dt * dt * Wacc * JRg = ((((0.5f * dR) * dt) * dt) * Wacc) * JRg  (:217-226)
  float Wacc[9], Rdt[9], Rhdt2[9], RdtW[9], Rhdt2W[9];
  hatf(acc, Wacc);
  for (int k = 0; k < 9; ++k) { Rdt[k] = P->dR[k] * dt; Rhdt2[k] = 0.5f * P->dR[k] * dt * dt; }
  mul33f(Rdt, Wacc, RdtW);       // dR * dt * Wacc
  mul33f(Rhdt2, Wacc, Rhdt2W);   // 0.5f * dR * dt * dt * Wacc
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      A[(3 + r) * 9 + c] = -RdtW[r * 3 + c];      // (negation commutes with every rounding)
      A[(6 + r) * 9 + c] = -Rhdt2W[r * 3 + c];
      B[(3 + r) * 6 + 3 + c] = Rdt[r * 3 + c];
      B[(6 + r) * 6 + 3 + c] = Rhdt2[r * 3 + c];
    }
  for (int k = 0; k < 3; ++k) A[(6 + k) * 9 + 3 + k] = dt;
  // bias-correction Jacobians of position and velocity
  float RdtWJ[9], Rhdt2WJ[9];
  mul33f(RdtW, P->JRg, RdtWJ);
  mul33f(Rhdt2W, P->JRg, Rhdt2WJ);
  for (int k = 0; k < 9; ++k) {
    P->JPa[k] = P->JPa[k] + P->JVa[k] * dt - Rhdt2[k];
    P->JPg[k] = P->JPg[k] + P->JVg[k] * dt - Rhdt2WJ[k];
    P->JVa[k] = P->JVa[k] - Rdt[k];
    P->JVg[k] = P->JVg[k] - RdtWJ[k];
  }
  float dRi[9], rJ[9], dRiT[9];
  integratedRotation(w, P->b, dt, dRi, rJ);
  mul33f(P->dR, dRi, P->dR);
  normalizeRotationF(P->dR);
  transpose33f(dRi, dRiT);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { A[r * 9 + c] = dRiT[r * 3 + c]; B[r * 6 + c] = rJ[r * 3 + c] * dt; }
  // C(0:9,0:9) = A C A^T + B Nga B^T ;  C(9:15,9:15) += NgaWalk   (C is 15 x 15 row-major)
  float AC[81], N[81];
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      float s = 0;
      for (int k = 0; k < 9; ++k) s += A[r * 9 + k] * P->C[k * 15 + c];
      AC[r * 9 + c] = s;
    }
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      float s = 0;
      for (int k = 0; k < 9; ++k) s += AC[r * 9 + k] * A[c * 9 + k];
      float t = 0;
      for (int k = 0; k < 6; ++k) t += B[r * 6 + k] * P->nga[k] * B[c * 6 + k];
      N[r * 9 + c] = s + t;
    }
  for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) P->C[r * 15 + c] = N[r * 9 + c];
  for (int k = 0; k < 6; ++k) P->C[(9 + k) * 15 + 9 + k] += P->ngaWalk[k];
  // JRg = dRi^T JRg - rightJ dt
  float t9[9];
  mul33f(dRiT, P->JRg, t9);
  for (int k = 0; k < 9; ++k) P->JRg[k] = t9[k] - rJ[k] * dt;
  P->dT += dt;
}

// ---- FP64 pieces ----------------------------------------------------------------------------------------------------------
static void mul33(const double* A, const double* B, double* C) {
  double T[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
  memcpy(C, T, sizeof T);
}
static void mul3v(const double* A, const double* v, double* o) {
  double t[3];
  for (int r = 0; r < 3; ++r) t[r] = A[r * 3] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2];
  memcpy(o, t, sizeof t);
}
static void mulT3v(const double* A, const double* v, double* o) {   // A^T v
  double t[3];
  for (int r = 0; r < 3; ++r) t[r] = A[r] * v[0] + A[3 + r] * v[1] + A[6 + r] * v[2];
  memcpy(o, t, sizeof t);
}
static void transpose33(const double* A, double* T) {
  double t[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) t[r * 3 + c] = A[c * 3 + r];
  memcpy(T, t, sizeof t);
}
static void expSO3(const double* w, double* R) {  // G2oTypes.cc:783-796 (projection omitted, see header)
  const double x = w[0], y = w[1], z = w[2];
  const double d2 = x * x + y * y + z * z, d = std::sqrt(d2);
  const double W[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  double WW[9];
  mul33(W, W, WW);
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d < 1e-5) for (int k = 0; k < 9; ++k) R[k] = I[k] + W[k] + 0.5 * WW[k];
  else {
    const double s = std::sin(d), c = std::cos(d);
    for (int k = 0; k < 9; ++k) R[k] = I[k] + W[k] * s / d + WW[k] * (1.0 - c) / d2;
  }
}
static void logSO3(const double* R, double* w) {  // G2oTypes.cc:798-811
  const double tr = R[0] + R[4] + R[8];
  w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
  const double costheta = (tr - 1.0) * 0.5f;
  if (costheta > 1 || costheta < -1) return;
  const double theta = std::acos(costheta), s = std::sin(theta);
  if (std::fabs(s) < 1e-5) return;
  for (int k = 0; k < 3; ++k) w[k] = theta * w[k] / s;
}
static void invRightJacobianSO3(const double* v, double* J) {  // G2oTypes.cc:817-829
  const double x = v[0], y = v[1], z = v[2];
  const double d2 = x * x + y * y + z * z, d = std::sqrt(d2);
  const double W[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d < 1e-5) { memcpy(J, I, sizeof I); return; }
  double WW[9];
  mul33(W, W, WW);
  const double k2 = 1.0 / d2 - (1.0 + std::cos(d)) / (2.0 * d * std::sin(d));
  for (int k = 0; k < 9; ++k) J[k] = I[k] + W[k] / 2 + WW[k] * k2;
}

static void rightJacobianSO3(const double* v, double* J) {  // G2oTypes.cc:835-848
  const double x = v[0], y = v[1], z = v[2];
  const double d2 = x * x + y * y + z * z, d = std::sqrt(d2);
  const double W[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d < 1e-5) { memcpy(J, I, sizeof I); return; }
  double WW[9];
  mul33(W, W, WW);
  for (int k = 0; k < 9; ++k) J[k] = I[k] - W[k] * (1.0 - std::cos(d)) / d2 + WW[k] * (d - std::sin(d)) / (d2 * d);
}
static void hat(const double* v, double* W) {
  W[0] = 0; W[1] = -v[2]; W[2] = v[1]; W[3] = v[2]; W[4] = 0; W[5] = -v[0]; W[6] = -v[1]; W[7] = v[0]; W[8] = 0;
}

// general n x n inverse, Gauss-Jordan with partial pivoting (Eigen: PartialPivLU based inverse)
static bool invertN(const double* A, int n, double* Ainv) {
  std::vector<double> M((size_t)n * 2 * n, 0.0);
  for (int r = 0; r < n; ++r) { for (int c = 0; c < n; ++c) M[(size_t)r * 2 * n + c] = A[r * n + c]; M[(size_t)r * 2 * n + n + r] = 1.0; }
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r) if (std::fabs(M[(size_t)r * 2 * n + c]) > std::fabs(M[(size_t)p * 2 * n + c])) p = r;
    if (M[(size_t)p * 2 * n + c] == 0.0) return false;
    if (p != c) for (int k = 0; k < 2 * n; ++k) std::swap(M[(size_t)p * 2 * n + k], M[(size_t)c * 2 * n + k]);
    const double inv = 1.0 / M[(size_t)c * 2 * n + c];
    for (int k = 0; k < 2 * n; ++k) M[(size_t)c * 2 * n + k] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = M[(size_t)r * 2 * n + c];
      if (f != 0.0) for (int k = 0; k < 2 * n; ++k) M[(size_t)r * 2 * n + k] -= f * M[(size_t)c * 2 * n + k];
    }
  }
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) Ainv[r * n + c] = M[(size_t)r * 2 * n + n + c];
  return true;
}
// symmetric eigen-decomposition, cyclic Jacobi: A = V diag(e) V^T
static void jacobiEig(const double* A, int n, double* e, double* V) {
  std::vector<double> a(A, A + (size_t)n * n);
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) V[r * n + c] = r == c ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) (r == c ? diag : off) += a[r * n + c] * a[r * n + c];
    if (off <= 1e-30 * diag) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (a[p * n + q] == 0.0) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * a[p * n + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int k = 0; k < n; ++k) e[k] = a[k * n + k];
}
// EdgeInertial's information (G2oTypes.cc:484-491)
static void inertialInformation(const float* C15, double* Info) {
  double C9[81], Inv[81];
  for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) C9[r * 9 + c] = (double)C15[r * 15 + c];
  if (!invertN(C9, 9, Inv)) { memset(Info, 0, sizeof(double) * 81); return; }
  double S[81];
  for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) S[r * 9 + c] = (Inv[r * 9 + c] + Inv[c * 9 + r]) / 2;
  double e[9], V[81];
  jacobiEig(S, 9, e, V);
  for (int k = 0; k < 9; ++k) if (e[k] < 1e-12) e[k] = 0;
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      double s = 0;
      for (int k = 0; k < 9; ++k) s += V[r * 9 + k] * e[k] * V[c * 9 + k];
      Info[r * 9 + c] = s;
    }
}

// ConstraintPoseImu's constructor (G2oTypes.h:711-721): eigenvalues of H below 1e-12 are set to zero
static void clampConstraintH(double* H15) {
  double e[15], V[225], S[225];
  jacobiEig(H15, 15, e, V);
  for (int k = 0; k < 15; ++k) if (e[k] < 1e-12) e[k] = 0;
  for (int r = 0; r < 15; ++r)
    for (int c = 0; c < 15; ++c) { double t = 0; for (int k = 0; k < 15; ++k) t += V[r * 15 + k] * e[k] * V[c * 15 + k]; S[r * 15 + c] = t; }
  memcpy(H15, S, sizeof S);
}
// Optimizer::Marginalize(H, 0, 14) of a 30 x 30 H (Optimizer.cc:2898-2977), returning the trailing 15 x 15 block:
// Hcc - Hcp pinv(Hpp) Hpc with the pseudo-inverse from the SVD (singular values <= 1e-6 dropped).  Hpp is symmetric, so its
// SVD is its eigen-decomposition with singular values |e| (restated with the cyclic Jacobi solver).
static void marginalizeFirst15(const double* H30, double* out15) {
  double Hpp[225], e[15], V[225], P[225];
  for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) Hpp[r * 15 + c] = H30[r * 30 + c];
  for (int r = 0; r < 15; ++r) for (int c = r + 1; c < 15; ++c) { const double m = 0.5 * (Hpp[r * 15 + c] + Hpp[c * 15 + r]); Hpp[r * 15 + c] = m; Hpp[c * 15 + r] = m; }
  jacobiEig(Hpp, 15, e, V);
  for (int r = 0; r < 15; ++r)
    for (int c = 0; c < 15; ++c) {
      double t = 0;
      for (int k = 0; k < 15; ++k) if (std::fabs(e[k]) > 1e-6) t += V[r * 15 + k] * (1.0 / e[k]) * V[c * 15 + k];
      P[r * 15 + c] = t;
    }
  for (int r = 0; r < 15; ++r)
    for (int c = 0; c < 15; ++c) {
      double t = 0;
      for (int k = 0; k < 15; ++k) for (int l = 0; l < 15; ++l) t += H30[(15 + r) * 30 + k] * P[k * 15 + l] * H30[l * 30 + 15 + c];
      out15[r * 15 + c] = H30[(15 + r) * 30 + 15 + c] - t;
    }
}

// Eigen::LDLT with diagonal pivoting; solves H x = b when the factorisation is positive (linear_solver_dense.h:104-112)
static bool ldltSolve(const double* Hin, const double* b, int n, double* x) {
  std::vector<double> A(Hin, Hin + (size_t)n * n);
  std::vector<int> perm(n);
  for (int k = 0; k < n; ++k) perm[k] = k;
  bool positive = true;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int r = k + 1; r < n; ++r) if (std::fabs(A[r * n + r]) > std::fabs(A[p * n + p])) p = r;
    if (p != k) {
      for (int c = 0; c < n; ++c) std::swap(A[k * n + c], A[p * n + c]);
      for (int r = 0; r < n; ++r) std::swap(A[r * n + k], A[r * n + p]);
      std::swap(perm[k], perm[p]);
    }
    const double d = A[k * n + k];
    if (!(d > 0)) positive = false;
    if (d == 0) continue;
    std::vector<double> col(n);
    for (int r = k + 1; r < n; ++r) col[r] = A[r * n + k];
    for (int r = k + 1; r < n; ++r) {
      const double l = col[r] / d;
      for (int c = k + 1; c <= r; ++c) { A[r * n + c] -= l * col[c]; A[c * n + r] = A[r * n + c]; }
      A[r * n + k] = l;
    }
  }
  if (!positive) return false;
  std::vector<double> y(n);
  for (int k = 0; k < n; ++k) y[k] = b[perm[k]];
  for (int r = 0; r < n; ++r) for (int c = 0; c < r; ++c) y[r] -= A[r * n + c] * y[c];
  for (int k = 0; k < n; ++k) y[k] /= A[k * n + k];
  for (int r = n - 1; r >= 0; --r) for (int c = r + 1; c < n; ++c) y[r] -= A[c * n + r] * y[c];
  for (int k = 0; k < n; ++k) x[perm[k]] = y[k];
  return true;
}

struct CamPose {  // ImuCamPose: one pinhole camera, or the two KannalaBrandt8 cameras of a fisheye rig (G2oTypes.cc:74-118)
  double Rwb[9], twb[3];
  double Rcw[9], tcw[3];
  double Rcb[9], tcb[3], Rbc[9], tbc[3];
  double bf;
  float fx, fy, cx, cy;
  bool rig = false;           // pCamera2 != NULL: camera 0 / 1 = left / right KB8
  orc::kb8::Cam kb[2];
  double Rcw1[9], tcw1[3], Rcb1[9], tcb1[3], Rbc1[9], tbc1[3];
  // rig28 = left KB8 (8), right KB8 (8), Trl rotation (9, row-major) + translation (3)   (G2oTypes.cc:104-113)
  void setRig(const float* rig28) {
    rig = true;
    memcpy(kb[0].p, rig28, 32); memcpy(kb[1].p, rig28 + 8, 32);
    double Rrl[9], trl[3];
    for (int k = 0; k < 9; ++k) Rrl[k] = rig28[16 + k];
    for (int k = 0; k < 3; ++k) trl[k] = rig28[25 + k];
    mul33(Rrl, Rcb, Rcb1);
    mul3v(Rrl, tcb, tcb1);
    for (int k = 0; k < 3; ++k) tcb1[k] += trl[k];
    transpose33(Rcb1, Rbc1);
    mul3v(Rbc1, tcb1, tbc1);
    for (double& c : tbc1) c = -c;
  }
  const double* R_cw(int cam) const { return cam ? Rcw1 : Rcw; }
  const double* t_cw(int cam) const { return cam ? tcw1 : tcw; }
  const double* R_cb(int cam) const { return cam ? Rcb1 : Rcb; }
  const double* R_bc(int cam) const { return cam ? Rbc1 : Rbc; }
  const double* t_bc(int cam) const { return cam ? tbc1 : tbc; }
  void refreshCamera() {   // G2oTypes.cc:209-215
    double Rbw[9], tbw[3];
    transpose33(Rwb, Rbw);
    mul3v(Rbw, twb, tbw);
    for (double& c : tbw) c = -c;
    mul33(Rcb, Rbw, Rcw);
    mul3v(Rcb, tbw, tcw);
    for (int k = 0; k < 3; ++k) tcw[k] += tcb[k];
    if (rig) {
      mul33(Rcb1, Rbw, Rcw1);
      mul3v(Rcb1, tbw, tcw1);
      for (int k = 0; k < 3; ++k) tcw1[k] += tcb1[k];
    }
  }
  void update(const double* pu) {   // ImuCamPose::Update (G2oTypes.cc:192-216)
    double t[3], dR[9];
    mul3v(Rwb, pu + 3, t);
    for (int k = 0; k < 3; ++k) twb[k] += t[k];
    expSO3(pu, dR);
    mul33(Rwb, dR, Rwb);
    refreshCamera();
  }
  void camPoint(const double* Xw, double* Xc, int cam = 0) const { mul3v(R_cw(cam), Xw, Xc); for (int k = 0; k < 3; ++k) Xc[k] += t_cw(cam)[k]; }
  void project(const double* Xc, double* uv, int cam = 0) const {
    if (rig) { orc::kb8::projectD(kb[cam], Xc, uv); return; }   // KannalaBrandt8::project(Vector3d)
    uv[0] = fx * Xc[0] / Xc[2] + cx;                            // Pinhole.cpp:38-44
    uv[1] = fy * Xc[1] / Xc[2] + cy;
  }
  void projectJac(const double* Xc, double* J, int cam = 0) const {
    if (rig) { orc::kb8::projectJac(kb[cam], Xc, J); return; }  // KannalaBrandt8::projectJac
    J[0] = fx / Xc[2]; J[1] = 0; J[2] = -fx * Xc[0] / (Xc[2] * Xc[2]);   // Pinhole.cpp:76-86
    J[3] = 0; J[4] = fy / Xc[2]; J[5] = -fy * Xc[1] / (Xc[2] * Xc[2]);
  }
};

struct VisEdge {
  int idx; bool stereo; double obs[3]; double Xw[3]; double info; bool close;
  int cam = 0;   // EdgeMonoOnlyPose(Xw, cam_idx) / EdgeMono(cam_idx)
  int level = 0; bool robust = true;
  double err[3] = {0, 0, 0};
  double chi2() const { const int d = stereo ? 3 : 2; double s = 0; for (int k = 0; k < d; ++k) s += err[k] * info * err[k]; return s; }
  void computeError(const CamPose& P) {   // G2oTypes.h:401-405 / :477-481, G2oTypes.cc:171-186
    double Xc[3], uv[2];
    P.camPoint(Xw, Xc, cam);
    P.project(Xc, uv, cam);
    err[0] = obs[0] - uv[0]; err[1] = obs[1] - uv[1];
    if (stereo) { const double invZ = 1 / Xc[2]; err[2] = obs[2] - (uv[0] - P.bf * invZ); }
  }
  bool depthPositive(const CamPose& P) const { const double* R = P.R_cw(cam); return (R[6] * Xw[0] + R[7] * Xw[1] + R[8] * Xw[2] + P.t_cw(cam)[2]) > 0.0; }
  void jacobian(const CamPose& P, double* J /* d x 6 */) const {   // G2oTypes.cc:361-382 / :417-442
    double Xc[3], Xb[3];
    P.camPoint(Xw, Xc, cam);
    mul3v(P.R_bc(cam), Xc, Xb);
    for (int k = 0; k < 3; ++k) Xb[k] += P.t_bc(cam)[k];
    double pj[9];
    P.projectJac(Xc, pj, cam);
    int d = 2;
    if (stereo) {
      d = 3;
      const double inv_z2 = 1.0 / (Xc[2] * Xc[2]);
      pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + P.bf * inv_z2;
    }
    const double x = Xb[0], y = Xb[1], z = Xb[2];
    const double S[18] = {0.0, z, -y, 1.0, 0.0, 0.0, -z, 0.0, x, 0.0, 1.0, 0.0, y, -x, 0.0, 0.0, 0.0, 1.0};
    double PR[9];
    for (int r = 0; r < d; ++r)
      for (int c = 0; c < 3; ++c) PR[r * 3 + c] = pj[r * 3] * P.R_cb(cam)[c] + pj[r * 3 + 1] * P.R_cb(cam)[3 + c] + pj[r * 3 + 2] * P.R_cb(cam)[6 + c];
    for (int r = 0; r < d; ++r)
      for (int c = 0; c < 6; ++c) J[r * 6 + c] = PR[r * 3] * S[c] + PR[r * 3 + 1] * S[6 + c] + PR[r * 3 + 2] * S[12 + c];
  }
};

static void huber(double delta, double e2, double* rho) {   // robust_kernel_impl.cpp:65-91
  const double dsqr = delta * delta;
  if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.; rho[2] = 0.; }
  else {
    const double sqrte = std::sqrt(e2);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e2;
  }
}

}  // namespace imu
}  // namespace orc

using namespace orc::imu;

extern "C" void orc_imu_preintegrate(const float* bias6, const float* ngaDiag6, const float* walkDiag6, int n, const float* acc,
                                     const float* gyro, const float* dt, orc_imu_preintegrated* out) {
  initialize(out, bias6, ngaDiag6, walkDiag6);
  for (int i = 0; i < n; ++i) integrate(out, acc + 3 * i, gyro + 3 * i, dt[i]);
}

// GetDeltaRotation / GetDeltaVelocity / GetDeltaPosition at bias b1 (ImuTypes.cc:289-312), FP32 like the reference
extern "C" void orc_imu_delta(const orc_imu_preintegrated* P, const float* b1, float* dR, float* dV, float* dP) {
  const float dbg[3] = {b1[3] - P->b[3], b1[4] - P->b[4], b1[5] - P->b[5]};
  const float dba[3] = {b1[0] - P->b[0], b1[1] - P->b[1], b1[2] - P->b[2]};
  float w[3];
  mul3vf(P->JRg, dbg, w);
  // Sophus::SO3f::exp(w).matrix(): Rodrigues
  const float t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  float W[9], WW[9], E[9];
  hatf(w, W);
  mul33f(W, W, WW);
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (t < 1e-5f) for (int k = 0; k < 9; ++k) E[k] = I[k] + W[k] + 0.5f * WW[k];
  else { const float s = std::sin(t), c = std::cos(t); for (int k = 0; k < 9; ++k) E[k] = I[k] + W[k] * s / t + WW[k] * (1.0f - c) / t2; }
  mul33f(P->dR, E, dR);
  normalizeRotationF(dR);
  float g1[3], a1[3];
  mul3vf(P->JVg, dbg, g1); mul3vf(P->JVa, dba, a1);
  for (int k = 0; k < 3; ++k) dV[k] = P->dV[k] + g1[k] + a1[k];
  mul3vf(P->JPg, dbg, g1); mul3vf(P->JPa, dba, a1);
  for (int k = 0; k < 3; ++k) dP[k] = P->dP[k] + g1[k] + a1[k];
}

// Optimizer::PoseInertialOptimizationLastKeyFrame (Optimizer.cc:4391-4757) for one frame.
// state / kfState = Rwb (9, row-major), twb, velocity, gyro bias, acc bias (21 floats); Tbc12 = Rbc (9) + tbc (3).
// prior246 (optional) = ConstraintPoseImu: Rwb, twb, v, bg, ba (21 doubles) + H (15 x 15 row-major).
// Nleft / rig28: fisheye rig (pFrame->Nleft != -1): features [0, Nleft) are left-camera, the rest right-camera monocular edges
// (Optimizer.cc:4453-4528); rig28 = nullptr: one pinhole camera.
static int poseInertialLastKeyFrame(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2, const float* Xw,
                                    const uint8_t* closeFlag, float fx, float fy, float cx, float cy, float bf, const float* Tbc12,
                                    const float* kfState21, const orc_imu_preintegrated* pre, int bRecInit, float* state21,
                                    uint8_t* outlier, double* prior246, int Nleft, const float* rig28) {
  CamPose VP;
  for (int k = 0; k < 9; ++k) VP.Rwb[k] = state21[k];
  for (int k = 0; k < 3; ++k) VP.twb[k] = state21[9 + k];
  double v[3], bg[3], ba[3];
  for (int k = 0; k < 3; ++k) { v[k] = state21[12 + k]; bg[k] = state21[15 + k]; ba[k] = state21[18 + k]; }
  for (int k = 0; k < 9; ++k) VP.Rbc[k] = Tbc12[k];
  for (int k = 0; k < 3; ++k) VP.tbc[k] = Tbc12[9 + k];
  transpose33(VP.Rbc, VP.Rcb);                      // mTcb = mTbc.inverse()
  mul3v(VP.Rcb, VP.tbc, VP.tcb);
  for (double& c : VP.tcb) c = -c;
  VP.bf = bf; VP.fx = fx; VP.fy = fy; VP.cx = cx; VP.cy = cy;
  if (rig28) VP.setRig(rig28);
  VP.refreshCamera();

  const double thHuberMono = (double)(float)std::sqrt(5.991), thHuberStereo = (double)(float)std::sqrt(7.815);
  std::vector<VisEdge> mono, stereo;
  for (int i = 0; i < n; ++i) {
    if (!hasMP[i]) continue;
    VisEdge e;
    e.idx = i; e.stereo = !rig28 && !(obs[3 * i + 2] < 0);
    e.cam = rig28 && i >= Nleft ? 1 : 0;
    e.obs[0] = obs[3 * i]; e.obs[1] = obs[3 * i + 1]; e.obs[2] = obs[3 * i + 2];
    for (int k = 0; k < 3; ++k) e.Xw[k] = Xw[3 * i + k];
    const float unc2 = 1.0f;                        // Pinhole / KannalaBrandt8::uncertainty2
    const float is2 = invSigma2[i] / unc2;
    e.info = is2; e.close = closeFlag[i] != 0;
    outlier[i] = 0;
    (e.stereo ? stereo : mono).push_back(e);
  }
  const int nInitial = (int)(mono.size() + stereo.size());

  // fixed keyframe vertices and the inertial edge's constants
  double Rwb1[9], twb1[3], v1[3], bg1[3], ba1[3];
  for (int k = 0; k < 9; ++k) Rwb1[k] = kfState21[k];
  for (int k = 0; k < 3; ++k) { twb1[k] = kfState21[9 + k]; v1[k] = kfState21[12 + k]; bg1[k] = kfState21[15 + k]; ba1[k] = kfState21[18 + k]; }
  const float b1f[6] = {(float)ba1[0], (float)ba1[1], (float)ba1[2], (float)bg1[0], (float)bg1[1], (float)bg1[2]};   // IMU::Bias(float...)
  float dRf[9], dVf[3], dPf[3];
  orc_imu_delta(pre, b1f, dRf, dVf, dPf);
  double dR[9], dV[3], dP[3];
  for (int k = 0; k < 9; ++k) dR[k] = dRf[k];
  for (int k = 0; k < 3; ++k) { dV[k] = dVf[k]; dP[k] = dPf[k]; }
  const double dt = pre->dT;
  const double g[3] = {0, 0, -(double)9.81f};
  double InfoI[81], InfoG[9], InfoA[9];
  inertialInformation(pre->C, InfoI);
  {
    double Cg[9], Ca[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Cg[r * 3 + c] = pre->C[(9 + r) * 15 + 9 + c]; Ca[r * 3 + c] = pre->C[(12 + r) * 15 + 12 + c]; }
    invertN(Cg, 3, InfoG); invertN(Ca, 3, InfoA);
  }
  double Rbw1[9];
  transpose33(Rwb1, Rbw1);

  // inertial edge: error (9) and the Jacobians w.r.t. pose 2 (9 x 6) and velocity 2 (9 x 3)   G2oTypes.cc:494-585
  auto inertial = [&](double* err, double* J4, double* J5) {
    double dRt[9], M[9], eR[9], er[3];
    transpose33(dR, dRt);
    mul33(dRt, Rbw1, M);
    mul33(M, VP.Rwb, eR);
    logSO3(eR, er);
    double t[3], ev[3], ep[3];
    for (int k = 0; k < 3; ++k) t[k] = v[k] - v1[k] - g[k] * dt;
    mul3v(Rbw1, t, ev);
    for (int k = 0; k < 3; ++k) ev[k] -= dV[k];
    for (int k = 0; k < 3; ++k) t[k] = VP.twb[k] - twb1[k] - v1[k] * dt - g[k] * dt * dt / 2;
    mul3v(Rbw1, t, ep);
    for (int k = 0; k < 3; ++k) ep[k] -= dP[k];
    for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = ev[k]; err[6 + k] = ep[k]; }
    if (J4) {
      double invJr[9], RR[9];
      invRightJacobianSO3(er, invJr);
      mul33(Rbw1, VP.Rwb, RR);
      memset(J4, 0, sizeof(double) * 54); memset(J5, 0, sizeof(double) * 27);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { J4[r * 6 + c] = invJr[r * 3 + c]; J4[(6 + r) * 6 + 3 + c] = RR[r * 3 + c]; J5[(3 + r) * 3 + c] = Rbw1[r * 3 + c]; }
    }
  };

  const float chi2Mono[4] = {12, 7.5, 5.991, 5.991};
  const float chi2Stereo[4] = {15.6, 9.8, 7.815, 7.815};
  int nBad = 0, nInliers = 0;
  double xPrev[15];
  memset(xPrev, 0, sizeof xPrev);
  for (int it = 0; it < 4; ++it) {
    // optimizer.initializeOptimization(0); optimizer.optimize(10): Gauss-Newton over the level-0 edges
    bool ok = true;
    for (int iter = 0; iter < 10 && ok; ++iter) {
      double H[225], b[15];
      memset(H, 0, sizeof H); memset(b, 0, sizeof b);
      auto visual = [&](std::vector<VisEdge>& E, double delta) {
        for (VisEdge& e : E) {
          if (e.level != 0) continue;
          e.computeError(VP);
          const int d = e.stereo ? 3 : 2;
          double J[18];
          e.jacobian(VP, J);
          double w = 1.0;
          if (e.robust) { double rho[3]; huber(delta, e.chi2(), rho); w = rho[1]; }
          for (int r = 0; r < 6; ++r) {
            double bb = 0;
            for (int k = 0; k < d; ++k) bb += J[k * 6 + r] * (e.info * e.err[k]);
            b[r] -= w * bb;
            for (int c = 0; c < 6; ++c) {
              double h = 0;
              for (int k = 0; k < d; ++k) h += J[k * 6 + r] * (w * e.info) * J[k * 6 + c];
              H[r * 15 + c] += h;
            }
          }
        }
      };
      visual(mono, thHuberMono);
      visual(stereo, thHuberStereo);
      {  // EdgeInertial on (pose 2, velocity 2): columns 0..5 and 6..8
        double err[9], J4[54], J5[27], J[81];
        inertial(err, J4, J5);
        for (int r = 0; r < 9; ++r) { for (int c = 0; c < 6; ++c) J[r * 9 + c] = J4[r * 6 + c]; for (int c = 0; c < 3; ++c) J[r * 9 + 6 + c] = J5[r * 3 + c]; }
        double OJ[81], Oe[9];
        for (int r = 0; r < 9; ++r) {
          for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += InfoI[r * 9 + k] * J[k * 9 + c]; OJ[r * 9 + c] = s; }
          double s = 0; for (int k = 0; k < 9; ++k) s += InfoI[r * 9 + k] * err[k]; Oe[r] = s;
        }
        for (int r = 0; r < 9; ++r) {
          double s = 0; for (int k = 0; k < 9; ++k) s += J[k * 9 + r] * Oe[k];
          b[r] -= s;
          for (int c = 0; c < 9; ++c) { double h = 0; for (int k = 0; k < 9; ++k) h += J[k * 9 + r] * OJ[k * 9 + c]; H[r * 15 + c] += h; }
        }
      }
      for (int r = 0; r < 3; ++r) {  // EdgeGyroRW / EdgeAccRW: error = bias2 - bias1, Jacobian w.r.t. bias2 = I
        double sg = 0, sa = 0;
        for (int k = 0; k < 3; ++k) { sg += InfoG[r * 3 + k] * (bg[k] - bg1[k]); sa += InfoA[r * 3 + k] * (ba[k] - ba1[k]); }
        b[9 + r] -= sg; b[12 + r] -= sa;
        for (int c = 0; c < 3; ++c) { H[(9 + r) * 15 + 9 + c] += InfoG[r * 3 + c]; H[(12 + r) * 15 + 12 + c] += InfoA[r * 3 + c]; }
      }
      double x[15];
      memcpy(x, xPrev, sizeof x);   // a failed solve leaves the previous solution in the solver's x (block_solver.hpp)
      ok = ldltSolve(H, b, 15, x);
      memcpy(xPrev, x, sizeof x);
      VP.update(x);
      for (int k = 0; k < 3; ++k) { v[k] += x[6 + k]; bg[k] += x[9 + k]; ba[k] += x[12 + k]; }
    }

    nBad = 0; nInliers = 0;
    const float chi2close = 1.5 * chi2Mono[it];
    for (VisEdge& e : mono) {
      if (outlier[e.idx]) e.computeError(VP);
      const float chi2 = (float)e.chi2();
      const bool bClose = e.close;
      if ((chi2 > chi2Mono[it] && !bClose) || (bClose && chi2 > chi2close) || !e.depthPositive(VP)) { outlier[e.idx] = 1; e.level = 1; ++nBad; }
      else { outlier[e.idx] = 0; e.level = 0; ++nInliers; }
      if (it == 2) e.robust = false;
    }
    for (VisEdge& e : stereo) {
      if (outlier[e.idx]) e.computeError(VP);
      const float chi2 = (float)e.chi2();
      if (chi2 > chi2Stereo[it]) { outlier[e.idx] = 1; e.level = 1; ++nBad; }
      else { outlier[e.idx] = 0; e.level = 0; ++nInliers; }
      if (it == 2) e.robust = false;
    }
    if (nInitial + 3 < 10) break;   // optimizer.edges().size() < 10
  }

  if (nInliers < 30 && !bRecInit) {   // :4683-4707
    nBad = 0;
    for (VisEdge& e : mono) { e.computeError(VP); if ((float)e.chi2() < 18.f) outlier[e.idx] = 0; else ++nBad; }
    for (VisEdge& e : stereo) { e.computeError(VP); if ((float)e.chi2() < 24.f) outlier[e.idx] = 0; else ++nBad; }
  }

  // SetImuPoseVelocity + bias (:4709-4715): floats
  for (int k = 0; k < 9; ++k) state21[k] = (float)VP.Rwb[k];
  for (int k = 0; k < 3; ++k) { state21[9 + k] = (float)VP.twb[k]; state21[12 + k] = (float)v[k]; state21[15 + k] = (float)bg[k]; state21[18 + k] = (float)ba[k]; }

  if (prior246) {   // ConstraintPoseImu (:4717-4754)
    double Hm[225];
    memset(Hm, 0, sizeof Hm);
    {
      double err[9], J4[54], J5[27], J[81];
      inertial(err, J4, J5);
      for (int r = 0; r < 9; ++r) { for (int c = 0; c < 6; ++c) J[r * 9 + c] = J4[r * 6 + c]; for (int c = 0; c < 3; ++c) J[r * 9 + 6 + c] = J5[r * 3 + c]; }
      for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
          double h = 0;
          for (int k = 0; k < 9; ++k) for (int l = 0; l < 9; ++l) h += J[k * 9 + r] * InfoI[k * 9 + l] * J[l * 9 + c];
          Hm[r * 15 + c] += h;
        }
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Hm[(9 + r) * 15 + 9 + c] += InfoG[r * 3 + c]; Hm[(12 + r) * 15 + 12 + c] += InfoA[r * 3 + c]; }
    auto add = [&](std::vector<VisEdge>& E) {
      for (VisEdge& e : E) {
        if (outlier[e.idx]) continue;
        const int d = e.stereo ? 3 : 2;
        double J[18];
        e.jacobian(VP, J);
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double h = 0; for (int k = 0; k < d; ++k) h += J[k * 6 + r] * e.info * J[k * 6 + c]; Hm[r * 15 + c] += h; }
      }
    };
    add(mono); add(stereo);
    for (int k = 0; k < 9; ++k) prior246[k] = VP.Rwb[k];
    for (int k = 0; k < 3; ++k) { prior246[9 + k] = VP.twb[k]; prior246[12 + k] = v[k]; prior246[15 + k] = bg[k]; prior246[18 + k] = ba[k]; }
    clampConstraintH(Hm);
    memcpy(prior246 + 21, Hm, sizeof Hm);
  }
  return nInitial - nBad;
}

extern "C" int orc_pose_inertial_optimization_last_keyframe(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2,
                                                            const float* Xw, const uint8_t* closeFlag, float fx, float fy, float cx,
                                                            float cy, float bf, const float* Tbc12, const float* kfState21,
                                                            const orc_imu_preintegrated* pre, int bRecInit, float* state21,
                                                            uint8_t* outlier, double* prior246) {
  return poseInertialLastKeyFrame(n, hasMP, obs, invSigma2, Xw, closeFlag, fx, fy, cx, cy, bf, Tbc12, kfState21, pre, bRecInit, state21,
                                  outlier, prior246, -1, nullptr);
}
extern "C" int orc_pose_inertial_optimization_last_keyframe_fisheye(int n, int Nleft, const uint8_t* hasMP, const float* obs,
                                                                    const float* invSigma2, const float* Xw, const uint8_t* closeFlag,
                                                                    const float* rig28, const float* Tbc12, const float* kfState21,
                                                                    const orc_imu_preintegrated* pre, int bRecInit, float* state21,
                                                                    uint8_t* outlier, double* prior246) {
  return poseInertialLastKeyFrame(n, hasMP, obs, invSigma2, Xw, closeFlag, 0, 0, 0, 0, 0, Tbc12, kfState21, pre, bRecInit, state21,
                                  outlier, prior246, Nleft, rig28);
}

// Optimizer::PoseInertialOptimizationLastFrame (Optimizer.cc:4761-5161) for one frame: the frame's 15 states and the previous
// frame's 15 states are free, tied by EdgeInertial(mpImuPreintegratedFrame), the two random-walk edges (information from
// mpImuPreintegrated, i.e. since the last keyframe) and EdgePriorPoseImu(pFp->mpcpi) with a Huber kernel (delta 5).
// prevPrior246 = pFp->mpcpi (21 state doubles + 15 x 15 H); prior246 (out) = the frame's new mpcpi after Marginalize(H, 0, 14).
static int poseInertialLastFrame(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2, const float* Xw,
                                 const uint8_t* closeFlag, float fx, float fy, float cx, float cy, float bf, const float* Tbc12,
                                 const float* prevState21, const orc_imu_preintegrated* preFrame, const orc_imu_preintegrated* preKF,
                                 const double* prevPrior246, int bRecInit, float* state21, uint8_t* outlier, double* prior246, int Nleft,
                                 const float* rig28) {
  CamPose VP, VPk;
  auto load = [&](CamPose& P, const float* s, double* v, double* bg, double* ba) {
    for (int k = 0; k < 9; ++k) P.Rwb[k] = s[k];
    for (int k = 0; k < 3; ++k) { P.twb[k] = s[9 + k]; v[k] = s[12 + k]; bg[k] = s[15 + k]; ba[k] = s[18 + k]; }
    for (int k = 0; k < 9; ++k) P.Rbc[k] = Tbc12[k];
    for (int k = 0; k < 3; ++k) P.tbc[k] = Tbc12[9 + k];
    transpose33(P.Rbc, P.Rcb);
    mul3v(P.Rcb, P.tbc, P.tcb);
    for (double& c : P.tcb) c = -c;
    P.bf = bf; P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy;
    if (rig28) P.setRig(rig28);
    P.refreshCamera();
  };
  double v[3], bg[3], ba[3], vk[3], bgk[3], bak[3];
  load(VP, state21, v, bg, ba);
  load(VPk, prevState21, vk, bgk, bak);

  const double thHuberMono = (double)(float)std::sqrt(5.991), thHuberStereo = (double)(float)std::sqrt(7.815);
  std::vector<VisEdge> mono, stereo;
  for (int i = 0; i < n; ++i) {
    if (!hasMP[i]) continue;
    VisEdge e;
    e.idx = i; e.stereo = !rig28 && !(obs[3 * i + 2] < 0);
    e.cam = rig28 && i >= Nleft ? 1 : 0;
    e.obs[0] = obs[3 * i]; e.obs[1] = obs[3 * i + 1]; e.obs[2] = obs[3 * i + 2];
    for (int k = 0; k < 3; ++k) e.Xw[k] = Xw[3 * i + k];
    const float is2 = invSigma2[i] / 1.0f;
    e.info = is2; e.close = closeFlag[i] != 0;
    outlier[i] = 0;
    (e.stereo ? stereo : mono).push_back(e);
  }
  const int nInitial = (int)(mono.size() + stereo.size());

  double JRg[9], JVg[9], JPg[9], JVa[9], JPa[9];
  for (int k = 0; k < 9; ++k) { JRg[k] = preFrame->JRg[k]; JVg[k] = preFrame->JVg[k]; JPg[k] = preFrame->JPg[k]; JVa[k] = preFrame->JVa[k]; JPa[k] = preFrame->JPa[k]; }
  const double dt = preFrame->dT;
  const double g[3] = {0, 0, -(double)9.81f};
  double InfoI[81], InfoG[9], InfoA[9];
  inertialInformation(preFrame->C, InfoI);
  {
    double Cg[9], Ca[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Cg[r * 3 + c] = preKF->C[(9 + r) * 15 + 9 + c]; Ca[r * 3 + c] = preKF->C[(12 + r) * 15 + 12 + c]; }
    invertN(Cg, 3, InfoG); invertN(Ca, 3, InfoA);
  }
  // prior (EdgePriorPoseImu, G2oTypes.cc:729-766)
  double pR[9], pt[3], pv[3], pbg[3], pba[3], pH[225];
  for (int k = 0; k < 9; ++k) pR[k] = prevPrior246[k];
  for (int k = 0; k < 3; ++k) { pt[k] = prevPrior246[9 + k]; pv[k] = prevPrior246[12 + k]; pbg[k] = prevPrior246[15 + k]; pba[k] = prevPrior246[18 + k]; }
  memcpy(pH, prevPrior246 + 21, sizeof pH);

  // inertial edge between (VPk, vk, bgk, bak) and (VP, v): error (9) and the 9 x 24 Jacobian in edge order [P1 V1 G1 A1 P2 V2]
  auto inertial = [&](double* err, double* J) {
    const float b1f[6] = {(float)bak[0], (float)bak[1], (float)bak[2], (float)bgk[0], (float)bgk[1], (float)bgk[2]};
    float dRf[9], dVf[3], dPf[3];
    orc_imu_delta(preFrame, b1f, dRf, dVf, dPf);
    double dR[9], dV[3], dP[3];
    for (int k = 0; k < 9; ++k) dR[k] = dRf[k];
    for (int k = 0; k < 3; ++k) { dV[k] = dVf[k]; dP[k] = dPf[k]; }
    double Rbw1[9], dRt[9], M[9], eR[9], er[3];
    transpose33(VPk.Rwb, Rbw1);
    transpose33(dR, dRt);
    mul33(dRt, Rbw1, M);
    mul33(M, VP.Rwb, eR);
    logSO3(eR, er);
    double t1[3], t2[3], a1[3], a2[3];
    for (int k = 0; k < 3; ++k) t1[k] = v[k] - vk[k] - g[k] * dt;
    mul3v(Rbw1, t1, a1);
    for (int k = 0; k < 3; ++k) t2[k] = VP.twb[k] - VPk.twb[k] - vk[k] * dt - g[k] * dt * dt / 2;
    mul3v(Rbw1, t2, a2);
    for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = a1[k] - dV[k]; err[6 + k] = a2[k] - dP[k]; }
    if (!J) return;
    memset(J, 0, sizeof(double) * 9 * 24);
    double invJr[9], Rwb2t[9], T[9], T2[9];
    invRightJacobianSO3(er, invJr);
    auto put = [&](int r0, int c0, const double* B, double sgn) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[(r0 + r) * 24 + c0 + c] = sgn * B[r * 3 + c]; };
    // pose 1 (cols 0..5)
    transpose33(VP.Rwb, Rwb2t);
    mul33(invJr, Rwb2t, T); mul33(T, VPk.Rwb, T2);
    put(0, 0, T2, -1.0);
    {
      // hat(Rbw1 * (v2 - v1 - g dt)); the position row uses 0.5 * g * dt * dt (G2oTypes.cc:549-552)
      double W[9];
      hat(a1, W); put(3, 0, W, 1.0);
      double t3[3], a3[3];
      for (int k = 0; k < 3; ++k) t3[k] = VP.twb[k] - VPk.twb[k] - vk[k] * dt - 0.5 * g[k] * dt * dt;
      mul3v(Rbw1, t3, a3);
      hat(a3, W); put(6, 0, W, 1.0);
    }
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    put(6, 3, I3, -1.0);
    // velocity 1 (cols 6..8)
    put(3, 6, Rbw1, -1.0);
    { double B[9]; for (int k = 0; k < 9; ++k) B[k] = Rbw1[k] * dt; put(6, 6, B, -1.0); }
    // gyro bias 1 (cols 9..11): -invJr * eR^T * RightJacobianSO3(JRg * dbg) * JRg, -JVg, -JPg
    {
      const float dbgf[3] = {b1f[3] - preFrame->b[3], b1f[4] - preFrame->b[4], b1f[5] - preFrame->b[5]};   // GetDeltaBias (float)
      const double dbg[3] = {dbgf[0], dbgf[1], dbgf[2]};
      double w3[3], rj[9], eRt[9];
      mul3v(JRg, dbg, w3);
      rightJacobianSO3(w3, rj);
      transpose33(eR, eRt);
      mul33(invJr, eRt, T); mul33(T, rj, T2); mul33(T2, JRg, T);
      put(0, 9, T, -1.0);
      put(3, 9, JVg, -1.0); put(6, 9, JPg, -1.0);
    }
    // acc bias 1 (cols 12..14)
    put(3, 12, JVa, -1.0); put(6, 12, JPa, -1.0);
    // pose 2 (cols 15..20)
    put(0, 15, invJr, 1.0);
    mul33(Rbw1, VP.Rwb, T); put(6, 18, T, 1.0);
    // velocity 2 (cols 21..23)
    put(3, 21, Rbw1, 1.0);
  };
  auto priorEdge = [&](double* err, double* J /* 15 x 15 */) {
    double pRt[9], E[9], er[3], d[3], et[3];
    transpose33(pR, pRt);
    mul33(pRt, VPk.Rwb, E);
    logSO3(E, er);
    for (int k = 0; k < 3; ++k) d[k] = VPk.twb[k] - pt[k];
    mul3v(pRt, d, et);
    for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = et[k]; err[6 + k] = vk[k] - pv[k]; err[9 + k] = bgk[k] - pbg[k]; err[12 + k] = bak[k] - pba[k]; }
    if (!J) return;
    memset(J, 0, sizeof(double) * 225);
    double invJr[9];
    invRightJacobianSO3(er, invJr);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { J[r * 15 + c] = invJr[r * 3 + c]; J[(3 + r) * 15 + 3 + c] = E[r * 3 + c]; }
    for (int k = 6; k < 15; ++k) J[k * 15 + k] = 1.0;
  };
  // system order (vertex ids): frame [P 0..5, V 6..8, G 9..11, A 12..14], previous frame 15 + the same
  const int edgeCol[24] = {15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 0, 1, 2, 3, 4, 5, 6, 7, 8};

  const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
  const float chi2Stereo[4] = {15.6f, 9.8f, 7.815f, 7.815f};
  int nBad = 0, nInliers = 0;
  double xPrev[30];
  memset(xPrev, 0, sizeof xPrev);
  for (int it = 0; it < 4; ++it) {
    bool ok = true;
    for (int iter = 0; iter < 10 && ok; ++iter) {
      double H[900], b[30];
      memset(H, 0, sizeof H); memset(b, 0, sizeof b);
      auto visual = [&](std::vector<VisEdge>& E, double delta) {
        for (VisEdge& e : E) {
          if (e.level != 0) continue;
          e.computeError(VP);
          const int d = e.stereo ? 3 : 2;
          double J[18];
          e.jacobian(VP, J);
          double w = 1.0;
          if (e.robust) { double rho[3]; huber(delta, e.chi2(), rho); w = rho[1]; }
          for (int r = 0; r < 6; ++r) {
            double bb = 0;
            for (int k = 0; k < d; ++k) bb += J[k * 6 + r] * (e.info * e.err[k]);
            b[r] -= w * bb;
            for (int c = 0; c < 6; ++c) { double h = 0; for (int k = 0; k < d; ++k) h += J[k * 6 + r] * (w * e.info) * J[k * 6 + c]; H[r * 30 + c] += h; }
          }
        }
      };
      visual(mono, thHuberMono);
      visual(stereo, thHuberStereo);
      {
        double err[9], J[216];
        inertial(err, J);
        for (int a = 0; a < 24; ++a) {
          double JtO[9];
          for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += J[k * 24 + a] * InfoI[k * 9 + c]; JtO[c] = s; }
          double s = 0; for (int k = 0; k < 9; ++k) s += JtO[k] * err[k];
          b[edgeCol[a]] -= s;
          for (int c = 0; c < 24; ++c) { double h = 0; for (int k = 0; k < 9; ++k) h += JtO[k] * J[k * 24 + c]; H[edgeCol[a] * 30 + edgeCol[c]] += h; }
        }
      }
      for (int r = 0; r < 3; ++r) {   // EdgeGyroRW(VGk, VG) / EdgeAccRW(VAk, VA): e = x2 - x1, J1 = -I, J2 = I
        double sg = 0, sa = 0;
        for (int k = 0; k < 3; ++k) { sg += InfoG[r * 3 + k] * (bg[k] - bgk[k]); sa += InfoA[r * 3 + k] * (ba[k] - bak[k]); }
        b[9 + r] -= sg; b[24 + r] += sg; b[12 + r] -= sa; b[27 + r] += sa;
        for (int c = 0; c < 3; ++c) {
          const double ig = InfoG[r * 3 + c], ia = InfoA[r * 3 + c];
          H[(9 + r) * 30 + 9 + c] += ig; H[(24 + r) * 30 + 24 + c] += ig; H[(9 + r) * 30 + 24 + c] -= ig; H[(24 + r) * 30 + 9 + c] -= ig;
          H[(12 + r) * 30 + 12 + c] += ia; H[(27 + r) * 30 + 27 + c] += ia; H[(12 + r) * 30 + 27 + c] -= ia; H[(27 + r) * 30 + 12 + c] -= ia;
        }
      }
      {  // EdgePriorPoseImu with Huber(5) on the previous frame's block (system rows 15..29)
        double err[15], J[225], Oe[15];
        priorEdge(err, J);
        double chi2 = 0;
        for (int r = 0; r < 15; ++r) { double s = 0; for (int k = 0; k < 15; ++k) s += pH[r * 15 + k] * err[k]; Oe[r] = s; chi2 += err[r] * s; }
        double rho[3];
        huber(5.0, chi2, rho);
        for (int a = 0; a < 15; ++a) {
          double s = 0; for (int k = 0; k < 15; ++k) s += J[k * 15 + a] * Oe[k];
          b[15 + a] -= rho[1] * s;
          for (int c = 0; c < 15; ++c) {
            double h = 0;
            for (int k = 0; k < 15; ++k) for (int l = 0; l < 15; ++l) h += J[k * 15 + a] * (rho[1] * pH[k * 15 + l]) * J[l * 15 + c];
            H[(15 + a) * 30 + 15 + c] += h;
          }
        }
      }
      double x[30];
      memcpy(x, xPrev, sizeof x);
      ok = ldltSolve(H, b, 30, x);
      memcpy(xPrev, x, sizeof x);
      VP.update(x);
      for (int k = 0; k < 3; ++k) { v[k] += x[6 + k]; bg[k] += x[9 + k]; ba[k] += x[12 + k]; }
      VPk.update(x + 15);
      for (int k = 0; k < 3; ++k) { vk[k] += x[21 + k]; bgk[k] += x[24 + k]; bak[k] += x[27 + k]; }
    }
    nBad = 0; nInliers = 0;
    const float chi2close = 1.5 * chi2Mono[it];
    for (VisEdge& e : mono) {
      if (outlier[e.idx]) e.computeError(VP);
      const float chi2 = (float)e.chi2();
      const bool bClose = e.close;
      if ((chi2 > chi2Mono[it] && !bClose) || (bClose && chi2 > chi2close) || !e.depthPositive(VP)) { outlier[e.idx] = 1; e.level = 1; ++nBad; }
      else { outlier[e.idx] = 0; e.level = 0; ++nInliers; }
      if (it == 2) e.robust = false;
    }
    for (VisEdge& e : stereo) {
      if (outlier[e.idx]) e.computeError(VP);
      const float chi2 = (float)e.chi2();
      if (chi2 > chi2Stereo[it]) { outlier[e.idx] = 1; e.level = 1; ++nBad; }
      else { outlier[e.idx] = 0; e.level = 0; ++nInliers; }
      if (it == 2) e.robust = false;
    }
    if (nInitial + 4 < 10) break;   // optimizer.edges().size() < 10
  }
  if (nInliers < 30 && !bRecInit) {
    nBad = 0;
    for (VisEdge& e : mono) { e.computeError(VP); if ((float)e.chi2() < 18.f) outlier[e.idx] = 0; else ++nBad; }
    for (VisEdge& e : stereo) { e.computeError(VP); if ((float)e.chi2() < 24.f) outlier[e.idx] = 0; else ++nBad; }
  }
  for (int k = 0; k < 9; ++k) state21[k] = (float)VP.Rwb[k];
  for (int k = 0; k < 3; ++k) { state21[9 + k] = (float)VP.twb[k]; state21[12 + k] = (float)v[k]; state21[15 + k] = (float)bg[k]; state21[18 + k] = (float)ba[k]; }

  if (prior246) {   // :5094-5150, H in the reference's order [previous frame 0..14 | frame 15..29]
    double H[900];
    memset(H, 0, sizeof H);
    {
      double err[9], J[216];
      inertial(err, J);   // edge order = the first 24 rows / columns of H
      for (int a = 0; a < 24; ++a)
        for (int c = 0; c < 24; ++c) {
          double h = 0;
          for (int k = 0; k < 9; ++k) for (int l = 0; l < 9; ++l) h += J[k * 24 + a] * InfoI[k * 9 + l] * J[l * 24 + c];
          H[a * 30 + c] += h;
        }
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        const double ig = InfoG[r * 3 + c], ia = InfoA[r * 3 + c];
        H[(9 + r) * 30 + 9 + c] += ig; H[(9 + r) * 30 + 24 + c] -= ig; H[(24 + r) * 30 + 9 + c] -= ig; H[(24 + r) * 30 + 24 + c] += ig;
        H[(12 + r) * 30 + 12 + c] += ia; H[(12 + r) * 30 + 27 + c] -= ia; H[(27 + r) * 30 + 12 + c] -= ia; H[(27 + r) * 30 + 27 + c] += ia;
      }
    {
      double err[15], J[225];
      priorEdge(err, J);
      for (int a = 0; a < 15; ++a)
        for (int c = 0; c < 15; ++c) {
          double h = 0;
          for (int k = 0; k < 15; ++k) for (int l = 0; l < 15; ++l) h += J[k * 15 + a] * pH[k * 15 + l] * J[l * 15 + c];
          H[a * 30 + c] += h;
        }
    }
    auto add = [&](std::vector<VisEdge>& E) {
      for (VisEdge& e : E) {
        if (outlier[e.idx]) continue;
        const int d = e.stereo ? 3 : 2;
        double J[18];
        e.jacobian(VP, J);
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double h = 0; for (int k = 0; k < d; ++k) h += J[k * 6 + r] * e.info * J[k * 6 + c]; H[(15 + r) * 30 + 15 + c] += h; }
      }
    };
    add(mono); add(stereo);
    double Hm[225];
    marginalizeFirst15(H, Hm);
    clampConstraintH(Hm);
    for (int k = 0; k < 9; ++k) prior246[k] = VP.Rwb[k];
    for (int k = 0; k < 3; ++k) { prior246[9 + k] = VP.twb[k]; prior246[12 + k] = v[k]; prior246[15 + k] = bg[k]; prior246[18 + k] = ba[k]; }
    memcpy(prior246 + 21, Hm, sizeof Hm);
  }
  return nInitial - nBad;
}

extern "C" int orc_pose_inertial_optimization_last_frame(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2,
                                                         const float* Xw, const uint8_t* closeFlag, float fx, float fy, float cx,
                                                         float cy, float bf, const float* Tbc12, const float* prevState21,
                                                         const orc_imu_preintegrated* preFrame, const orc_imu_preintegrated* preKF,
                                                         const double* prevPrior246, int bRecInit, float* state21,
                                                         uint8_t* outlier, double* prior246) {
  return poseInertialLastFrame(n, hasMP, obs, invSigma2, Xw, closeFlag, fx, fy, cx, cy, bf, Tbc12, prevState21, preFrame, preKF,
                               prevPrior246, bRecInit, state21, outlier, prior246, -1, nullptr);
}
extern "C" int orc_pose_inertial_optimization_last_frame_fisheye(int n, int Nleft, const uint8_t* hasMP, const float* obs,
                                                                 const float* invSigma2, const float* Xw, const uint8_t* closeFlag,
                                                                 const float* rig28, const float* Tbc12, const float* prevState21,
                                                                 const orc_imu_preintegrated* preFrame, const orc_imu_preintegrated* preKF,
                                                                 const double* prevPrior246, int bRecInit, float* state21,
                                                                 uint8_t* outlier, double* prior246) {
  return poseInertialLastFrame(n, hasMP, obs, invSigma2, Xw, closeFlag, 0, 0, 0, 0, 0, Tbc12, prevState21, preFrame, preKF, prevPrior246,
                               bRecInit, state21, outlier, prior246, Nleft, rig28);
}

// =========================================================================================================================
// Optimizer::LocalInertialBA (Optimizer.cc:2324-2897) on the flattened graph the reference assembles at :2337-2768:
//   kfKind[k]: 0 = temporal optimizable keyframe (pose, velocity, gyro bias, acc bias free), 1 = the fixed keyframe before the
//   window (its IMU state enters the last inertial edge), 2 = fixed keyframe that only observes points.
//   nI inertial links (iKF1 = mPrevKF, iKF2 = the keyframe whose mpImuPreintegrated is iPre[i]); iRobust[i] = Huber
//   sqrt(16.92) (i == N - 1 || bRecInit, :2553-2563), iInfoScale[i] = 1e-2 on the link to the fixed keyframe else 1.
//   EdgeGyroRW / EdgeAccRW accompany every link (:2566-2583).  Points are marginalised (Schur), EdgeMono / EdgeStereo
//   (G2oTypes.h:352-463, G2oTypes.cc:334-415) with Huber; g2o Levenberg-Marquardt with a user lambda (1 or 1e-2), opt_it = 10
//   (4 when bLarge) iterations (optimization_algorithm_levenberg.cpp:61-194, block_solver.hpp Schur path).
// Returns 1, or 0 for "FAIL LOCAL-INERTIAL BA" (:2808-2813: nothing is written back).  eraseFlag[e] = observation erased.
// =========================================================================================================================
namespace orc {
namespace imu {
namespace {

struct IbaKF { CamPose P; double v[3], bg[3], ba[3]; int col; };
struct IbaState { std::vector<IbaKF> kf; std::vector<double> pts; };

static void inertialFull(const orc_imu_preintegrated* pre, const IbaKF& K1, const IbaKF& K2, double* err, double* J) {
  const float b1f[6] = {(float)K1.ba[0], (float)K1.ba[1], (float)K1.ba[2], (float)K1.bg[0], (float)K1.bg[1], (float)K1.bg[2]};
  float dRf[9], dVf[3], dPf[3];
  orc_imu_delta(pre, b1f, dRf, dVf, dPf);
  double dR[9], dV[3], dP[3], JRg[9], JVg[9], JPg[9], JVa[9], JPa[9];
  for (int k = 0; k < 9; ++k) { dR[k] = dRf[k]; JRg[k] = pre->JRg[k]; JVg[k] = pre->JVg[k]; JPg[k] = pre->JPg[k]; JVa[k] = pre->JVa[k]; JPa[k] = pre->JPa[k]; }
  for (int k = 0; k < 3; ++k) { dV[k] = dVf[k]; dP[k] = dPf[k]; }
  const double dt = pre->dT;
  const double g[3] = {0, 0, -(double)9.81f};
  double Rbw1[9], dRt[9], M[9], eR[9], er[3];
  transpose33(K1.P.Rwb, Rbw1);
  transpose33(dR, dRt);
  mul33(dRt, Rbw1, M);
  mul33(M, K2.P.Rwb, eR);
  logSO3(eR, er);
  double t1[3], t2[3], a1[3], a2[3];
  for (int k = 0; k < 3; ++k) t1[k] = K2.v[k] - K1.v[k] - g[k] * dt;
  mul3v(Rbw1, t1, a1);
  for (int k = 0; k < 3; ++k) t2[k] = K2.P.twb[k] - K1.P.twb[k] - K1.v[k] * dt - g[k] * dt * dt / 2;
  mul3v(Rbw1, t2, a2);
  for (int k = 0; k < 3; ++k) { err[k] = er[k]; err[3 + k] = a1[k] - dV[k]; err[6 + k] = a2[k] - dP[k]; }
  if (!J) return;
  memset(J, 0, sizeof(double) * 9 * 24);
  double invJr[9], Rwb2t[9], T[9], T2[9], W[9];
  invRightJacobianSO3(er, invJr);
  auto put = [&](int r0, int c0, const double* B, double sgn) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[(r0 + r) * 24 + c0 + c] = sgn * B[r * 3 + c]; };
  transpose33(K2.P.Rwb, Rwb2t);
  mul33(invJr, Rwb2t, T); mul33(T, K1.P.Rwb, T2);
  put(0, 0, T2, -1.0);
  hat(a1, W); put(3, 0, W, 1.0);
  {
    double t3[3], a3[3];
    for (int k = 0; k < 3; ++k) t3[k] = K2.P.twb[k] - K1.P.twb[k] - K1.v[k] * dt - 0.5 * g[k] * dt * dt;
    mul3v(Rbw1, t3, a3);
    hat(a3, W); put(6, 0, W, 1.0);
  }
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  put(6, 3, I3, -1.0);
  put(3, 6, Rbw1, -1.0);
  { double B[9]; for (int k = 0; k < 9; ++k) B[k] = Rbw1[k] * dt; put(6, 6, B, -1.0); }
  {
    const float dbgf[3] = {b1f[3] - pre->b[3], b1f[4] - pre->b[4], b1f[5] - pre->b[5]};
    const double dbg[3] = {dbgf[0], dbgf[1], dbgf[2]};
    double w3[3], rj[9], eRt[9];
    mul3v(JRg, dbg, w3);
    rightJacobianSO3(w3, rj);
    transpose33(eR, eRt);
    mul33(invJr, eRt, T); mul33(T, rj, T2); mul33(T2, JRg, T);
    put(0, 9, T, -1.0);
    put(3, 9, JVg, -1.0); put(6, 9, JPg, -1.0);
  }
  put(3, 12, JVa, -1.0); put(6, 12, JPa, -1.0);
  put(0, 15, invJr, 1.0);
  mul33(Rbw1, K2.P.Rwb, T); put(6, 18, T, 1.0);
  put(3, 21, Rbw1, 1.0);
}

}  // namespace
}  // namespace imu
}  // namespace orc

// eRight / rig28: fisheye rig (pKFi->mpCamera2): eRight[e] != 0 = EdgeMono(1) on the right camera (:2722-2754), all edges monocular
static int localInertialBA(int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos, const uint8_t* mpClose, int nE,
                           const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2, int nI, const int* iKF1,
                           const int* iKF2, const orc_imu_preintegrated* iPre, const uint8_t* iRobust, const float* iInfoScale, float fx,
                           float fy, float cx, float cy, float bf, const float* Tbc12, int bLarge, uint8_t* eraseFlag, int* stats2,
                           const uint8_t* eRight, const float* rig28) {
  using namespace orc::imu;
  auto isStereo = [&](int e) { return !rig28 && !(eObs[3 * e + 2] < 0); };
  auto camOf = [&](int e) { return rig28 && eRight[e] ? 1 : 0; };
  IbaState S;
  S.kf.resize(nKF);
  int nOpt = 0;
  for (int k = 0; k < nKF; ++k) {
    IbaKF& K = S.kf[k];
    const float* s = kfState21 + 21 * k;
    for (int i = 0; i < 9; ++i) K.P.Rwb[i] = s[i];
    for (int i = 0; i < 3; ++i) { K.P.twb[i] = s[9 + i]; K.v[i] = s[12 + i]; K.bg[i] = s[15 + i]; K.ba[i] = s[18 + i]; }
    for (int i = 0; i < 9; ++i) K.P.Rbc[i] = Tbc12[i];
    for (int i = 0; i < 3; ++i) K.P.tbc[i] = Tbc12[9 + i];
    transpose33(K.P.Rbc, K.P.Rcb);
    mul3v(K.P.Rcb, K.P.tbc, K.P.tcb);
    for (double& c : K.P.tcb) c = -c;
    K.P.bf = bf; K.P.fx = fx; K.P.fy = fy; K.P.cx = cx; K.P.cy = cy;
    if (rig28) K.P.setRig(rig28);
    K.P.refreshCamera();
    K.col = kfKind[k] == 0 ? nOpt++ : -1;
  }
  S.pts.resize((size_t)3 * nMP);
  for (int k = 0; k < 3 * nMP; ++k) S.pts[k] = mpPos[k];
  const int P = 15 * nOpt;
  const double thMono = (double)(float)std::sqrt(5.991), thStereo = (double)(float)std::sqrt(7.815), thInertial = std::sqrt(16.92);

  // informations
  std::vector<double> InfoI((size_t)nI * 81), InfoG((size_t)nI * 9), InfoA((size_t)nI * 9);
  for (int i = 0; i < nI; ++i) {
    inertialInformation(iPre[i].C, &InfoI[(size_t)i * 81]);
    for (int k = 0; k < 81; ++k) InfoI[(size_t)i * 81 + k] *= (double)iInfoScale[i];
    double Cg[9], Ca[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Cg[r * 3 + c] = iPre[i].C[(9 + r) * 15 + 9 + c]; Ca[r * 3 + c] = iPre[i].C[(12 + r) * 15 + 12 + c]; }
    invertN(Cg, 3, &InfoG[(size_t)i * 9]); invertN(Ca, 3, &InfoA[(size_t)i * 9]);
  }

  // error storage (the values of the last computeActiveErrors)
  std::vector<double> vErr((size_t)nE * 3), iErr((size_t)nI * 9), gErr((size_t)nI * 3), aErr((size_t)nI * 3);
  auto visChi2 = [&](int e) { const bool st = isStereo(e); const double info = eInvSigma2[e]; double s = 0; for (int k = 0; k < (st ? 3 : 2); ++k) s += vErr[3 * e + k] * info * vErr[3 * e + k]; return s; };
  auto inertialChi2 = [&](int i) { double s = 0; for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) s += iErr[9 * i + r] * InfoI[(size_t)i * 81 + r * 9 + c] * iErr[9 * i + c]; return s; };
  auto rwChi2 = [&](const std::vector<double>& E, const std::vector<double>& I3, int i) { double s = 0; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) s += E[3 * i + r] * I3[(size_t)i * 9 + r * 3 + c] * E[3 * i + c]; return s; };
  auto computeActiveErrors = [&]() {
    for (int e = 0; e < nE; ++e) {
      VisEdge ve;
      ve.stereo = isStereo(e); ve.cam = camOf(e);
      for (int k = 0; k < 3; ++k) { ve.obs[k] = eObs[3 * e + k]; ve.Xw[k] = S.pts[3 * eMP[e] + k]; }
      ve.err[2] = 0;
      ve.computeError(S.kf[eKF[e]].P);
      for (int k = 0; k < 3; ++k) vErr[3 * e + k] = ve.stereo || k < 2 ? ve.err[k] : 0.0;
    }
    for (int i = 0; i < nI; ++i) {
      inertialFull(&iPre[i], S.kf[iKF1[i]], S.kf[iKF2[i]], &iErr[9 * i], nullptr);
      for (int k = 0; k < 3; ++k) { gErr[3 * i + k] = S.kf[iKF2[i]].bg[k] - S.kf[iKF1[i]].bg[k]; aErr[3 * i + k] = S.kf[iKF2[i]].ba[k] - S.kf[iKF1[i]].ba[k]; }
    }
  };
  auto activeRobustChi2 = [&]() {
    double s = 0;
    for (int e = 0; e < nE; ++e) { double rho[3]; huber(isStereo(e) ? thStereo : thMono, visChi2(e), rho); s += rho[0]; }
    for (int i = 0; i < nI; ++i) {
      const double c = inertialChi2(i);
      if (iRobust[i]) { double rho[3]; huber(thInertial, c, rho); s += rho[0]; } else s += c;
      s += rwChi2(gErr, InfoG, i) + rwChi2(aErr, InfoA, i);
    }
    return s;
  };

  // system
  std::vector<double> H((size_t)P * P), b((size_t)P + 3 * nMP), Hll((size_t)nMP * 9), Hpl((size_t)nE * 18), x((size_t)P + 3 * nMP);
  std::vector<std::vector<int>> byPoint(nMP);
  for (int e = 0; e < nE; ++e) byPoint[eMP[e]].push_back(e);
  auto buildSystem = [&]() {
    std::fill(H.begin(), H.end(), 0.0); std::fill(b.begin(), b.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(Hpl.begin(), Hpl.end(), 0.0);
    for (int e = 0; e < nE; ++e) {
      const IbaKF& K = S.kf[eKF[e]];
      VisEdge ve;
      ve.stereo = isStereo(e); ve.cam = camOf(e);
      for (int k = 0; k < 3; ++k) ve.Xw[k] = S.pts[3 * eMP[e] + k];
      const int d = ve.stereo ? 3 : 2;
      const double info = eInvSigma2[e];
      double rho[3];
      huber(ve.stereo ? thStereo : thMono, visChi2(e), rho);
      const double w = rho[1];
      double Jp[18], Jl[9];
      ve.jacobian(K.P, Jp);
      {  // EdgeMono / EdgeStereo::linearizeOplus: _jacobianOplusXi = -proj_jac * Rcw (G2oTypes.cc:334-415)
        double Xc[3], pj[9];
        K.P.camPoint(ve.Xw, Xc, ve.cam);
        K.P.projectJac(Xc, pj, ve.cam);
        if (ve.stereo) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + K.P.bf * (1.0 / (Xc[2] * Xc[2])); }
        const double* Rc = K.P.R_cw(ve.cam);
        for (int r = 0; r < d; ++r) for (int c = 0; c < 3; ++c) Jl[r * 3 + c] = -(pj[r * 3] * Rc[c] + pj[r * 3 + 1] * Rc[3 + c] + pj[r * 3 + 2] * Rc[6 + c]);
      }
      const double* er = &vErr[3 * e];
      const int l = eMP[e];
      for (int r = 0; r < 3; ++r) {
        double s = 0; for (int k = 0; k < d; ++k) s += Jl[k * 3 + r] * info * er[k];
        b[P + 3 * l + r] -= w * s;
        for (int c = 0; c < 3; ++c) { double h = 0; for (int k = 0; k < d; ++k) h += Jl[k * 3 + r] * (w * info) * Jl[k * 3 + c]; Hll[(size_t)l * 9 + r * 3 + c] += h; }
      }
      if (K.col < 0) continue;
      const int o = 15 * K.col;
      for (int r = 0; r < 6; ++r) {
        double s = 0; for (int k = 0; k < d; ++k) s += Jp[k * 6 + r] * info * er[k];
        b[o + r] -= w * s;
        for (int c = 0; c < 6; ++c) { double h = 0; for (int k = 0; k < d; ++k) h += Jp[k * 6 + r] * (w * info) * Jp[k * 6 + c]; H[(size_t)(o + r) * P + o + c] += h; }
        for (int c = 0; c < 3; ++c) { double h = 0; for (int k = 0; k < d; ++k) h += Jp[k * 6 + r] * (w * info) * Jl[k * 3 + c]; Hpl[(size_t)e * 18 + r * 3 + c] = h; }
      }
    }
    for (int i = 0; i < nI; ++i) {
      const IbaKF &K1 = S.kf[iKF1[i]], &K2 = S.kf[iKF2[i]];
      double err[9], J[216];
      inertialFull(&iPre[i], K1, K2, err, J);
      double w = 1.0;
      if (iRobust[i]) { double rho[3]; huber(thInertial, inertialChi2(i), rho); w = rho[1]; }
      // edge column -> system column (-1: fixed vertex): P1 V1 G1 A1 of K1, P2 V2 of K2
      int colOf[24];
      for (int a = 0; a < 15; ++a) colOf[a] = K1.col >= 0 ? 15 * K1.col + a : -1;
      for (int a = 0; a < 9; ++a) colOf[15 + a] = K2.col >= 0 ? 15 * K2.col + a : -1;
      const double* Om = &InfoI[(size_t)i * 81];
      const double* er = &iErr[9 * i];
      for (int a = 0; a < 24; ++a) {
        if (colOf[a] < 0) continue;
        double JtO[9];
        for (int c = 0; c < 9; ++c) { double s = 0; for (int k = 0; k < 9; ++k) s += J[k * 24 + a] * Om[k * 9 + c]; JtO[c] = s; }
        double s = 0; for (int k = 0; k < 9; ++k) s += JtO[k] * er[k];
        b[colOf[a]] -= w * s;
        for (int c = 0; c < 24; ++c) {
          if (colOf[c] < 0) continue;
          double h = 0; for (int k = 0; k < 9; ++k) h += JtO[k] * J[k * 24 + c];
          H[(size_t)colOf[a] * P + colOf[c]] += w * h;
        }
      }
      // random walks: e = bias2 - bias1
      const int c1g = K1.col >= 0 ? 15 * K1.col + 9 : -1, c2g = K2.col >= 0 ? 15 * K2.col + 9 : -1;
      for (int t = 0; t < 2; ++t) {
        const double* I3 = t == 0 ? &InfoG[(size_t)i * 9] : &InfoA[(size_t)i * 9];
        const double* er3 = t == 0 ? &gErr[3 * i] : &aErr[3 * i];
        const int o1 = c1g < 0 ? -1 : c1g + 3 * t, o2 = c2g < 0 ? -1 : c2g + 3 * t;
        for (int r = 0; r < 3; ++r) {
          double s = 0; for (int k = 0; k < 3; ++k) s += I3[r * 3 + k] * er3[k];
          if (o2 >= 0) b[o2 + r] -= s;
          if (o1 >= 0) b[o1 + r] += s;
          for (int c = 0; c < 3; ++c) {
            const double v = I3[r * 3 + c];
            if (o2 >= 0) H[(size_t)(o2 + r) * P + o2 + c] += v;
            if (o1 >= 0) H[(size_t)(o1 + r) * P + o1 + c] += v;
            if (o1 >= 0 && o2 >= 0) { H[(size_t)(o1 + r) * P + o2 + c] -= v; H[(size_t)(o2 + r) * P + o1 + c] -= v; }
          }
        }
      }
    }
  };
  auto solveSystem = [&](double lam) -> bool {   // block_solver.hpp: Schur complement on the points, dense solve, back-substitution
    std::vector<double> Hs = H, bs(b.begin(), b.begin() + P), Dinv((size_t)nMP * 9);
    for (int i = 0; i < P; ++i) Hs[(size_t)i * P + i] += lam;
    for (int l = 0; l < nMP; ++l) {
      double D[9];
      memcpy(D, &Hll[(size_t)l * 9], sizeof D);
      D[0] += lam; D[4] += lam; D[8] += lam;
      if (!invertN(D, 3, &Dinv[(size_t)l * 9])) memset(&Dinv[(size_t)l * 9], 0, sizeof D);
    }
    for (int l = 0; l < nMP; ++l) {
      const double* Di = &Dinv[(size_t)l * 9];
      double db[3];
      for (int r = 0; r < 3; ++r) db[r] = Di[r * 3] * b[P + 3 * l] + Di[r * 3 + 1] * b[P + 3 * l + 1] + Di[r * 3 + 2] * b[P + 3 * l + 2];
      for (int e1 : byPoint[l]) {
        const int c1 = S.kf[eKF[e1]].col;
        if (c1 < 0) continue;
        const double* B1 = &Hpl[(size_t)e1 * 18];
        double BD[18];
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) BD[r * 3 + c] = B1[r * 3] * Di[c] + B1[r * 3 + 1] * Di[3 + c] + B1[r * 3 + 2] * Di[6 + c];
        for (int r = 0; r < 6; ++r) bs[15 * c1 + r] -= B1[r * 3] * db[0] + B1[r * 3 + 1] * db[1] + B1[r * 3 + 2] * db[2];
        for (int e2 : byPoint[l]) {
          const int c2 = S.kf[eKF[e2]].col;
          if (c2 < 0) continue;
          const double* B2 = &Hpl[(size_t)e2 * 18];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c)
              Hs[(size_t)(15 * c1 + r) * P + 15 * c2 + c] -= BD[r * 3] * B2[c * 3] + BD[r * 3 + 1] * B2[c * 3 + 1] + BD[r * 3 + 2] * B2[c * 3 + 2];
        }
      }
    }
    if (!ldltSolve(Hs.data(), bs.data(), P, x.data())) return false;
    for (int l = 0; l < nMP; ++l) {
      double cl[3] = {b[P + 3 * l], b[P + 3 * l + 1], b[P + 3 * l + 2]};
      for (int e : byPoint[l]) {
        const int c1 = S.kf[eKF[e]].col;
        if (c1 < 0) continue;
        const double* B = &Hpl[(size_t)e * 18];
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 6; ++r) cl[c] -= B[r * 3 + c] * x[15 * c1 + r];
      }
      const double* Di = &Dinv[(size_t)l * 9];
      for (int r = 0; r < 3; ++r) x[P + 3 * l + r] = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
    }
    return true;
  };
  auto update = [&]() {
    for (IbaKF& K : S.kf) {
      if (K.col < 0) continue;
      const double* dx = &x[15 * K.col];
      K.P.update(dx);
      for (int k = 0; k < 3; ++k) { K.v[k] += dx[6 + k]; K.bg[k] += dx[9 + k]; K.ba[k] += dx[12 + k]; }
    }
    for (int k = 0; k < 3 * nMP; ++k) S.pts[k] += x[P + k];
  };

  computeActiveErrors();
  const float err0 = (float)activeRobustChi2();
  // optimizer.optimize(opt_it): Levenberg-Marquardt with the user lambda
  double lambda = bLarge ? 1e-2 : 1e0, ni = 2;
  int nBadIts = 0, trials = 0, outer = 0;
  const int optIt = bLarge ? 4 : 10;
  for (int it = 0; it < optIt; ++it) {
    ++outer;
    computeActiveErrors();
    double currentChi = activeRobustChi2(), tempChi = currentChi;
    const double iniChi = currentChi;
    buildSystem();
    if (it == 0) { ni = 2; nBadIts = 0; }
    double rho = 0;
    int qmax = 0;
    do {
      const IbaState backup = S;
      const bool ok2 = solveSystem(lambda);
      update();
      computeActiveErrors();
      tempChi = activeRobustChi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (size_t j = 0; j < x.size(); ++j) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        S = backup;
      }
      ++qmax; ++trials;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBadIts++; else nBadIts = 0;
    if (nBadIts >= 3) break;
  }
  if (stats2) { stats2[0] = outer; stats2[1] = trials; }
  const float errEnd = (float)activeRobustChi2();
  memset(eraseFlag, 0, nE);
  if ((2 * err0 < errEnd || std::isnan(err0) || std::isnan(errEnd)) && !bLarge) return 0;   // "FAIL LOCAL-INERTIAL BA"
  for (int e = 0; e < nE; ++e) {   // :2773-2801 with the errors of the last computeActiveErrors
    const bool st = isStereo(e);
    const double c = visChi2(e);
    if (st) { if (c > 7.815f) eraseFlag[e] = 1; }
    else {
      const bool bClose = mpClose[eMP[e]] != 0;
      const IbaKF& K = S.kf[eKF[e]];
      const double* X = &S.pts[3 * eMP[e]];
      const double* Rc = K.P.R_cw(camOf(e));
      const bool depthPos = (Rc[6] * X[0] + Rc[7] * X[1] + Rc[8] * X[2] + K.P.t_cw(camOf(e))[2]) > 0.0;
      if ((c > 5.991f && !bClose) || (c > 1.5f * 5.991f && bClose) || !depthPos) eraseFlag[e] = 1;
    }
  }
  for (int k = 0; k < nKF; ++k) {
    if (S.kf[k].col < 0) continue;
    float* s = kfState21 + 21 * k;
    const IbaKF& K = S.kf[k];
    for (int i = 0; i < 9; ++i) s[i] = (float)K.P.Rwb[i];
    for (int i = 0; i < 3; ++i) { s[9 + i] = (float)K.P.twb[i]; s[12 + i] = (float)K.v[i]; s[15 + i] = (float)K.bg[i]; s[18 + i] = (float)K.ba[i]; }
  }
  for (int k = 0; k < 3 * nMP; ++k) mpPos[k] = (float)S.pts[k];
  return 1;
}

extern "C" int orc_local_inertial_ba(int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos, const uint8_t* mpClose,
                                     int nE, const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2, int nI,
                                     const int* iKF1, const int* iKF2, const orc_imu_preintegrated* iPre, const uint8_t* iRobust,
                                     const float* iInfoScale, float fx, float fy, float cx, float cy, float bf, const float* Tbc12,
                                     int bLarge, uint8_t* eraseFlag, int* stats2) {
  return localInertialBA(nKF, kfState21, kfKind, nMP, mpPos, mpClose, nE, eKF, eMP, eObs, eInvSigma2, nI, iKF1, iKF2, iPre, iRobust,
                         iInfoScale, fx, fy, cx, cy, bf, Tbc12, bLarge, eraseFlag, stats2, nullptr, nullptr);
}
extern "C" int orc_local_inertial_ba_fisheye(int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos,
                                             const uint8_t* mpClose, int nE, const int* eKF, const int* eMP, const float* eObs,
                                             const uint8_t* eRight, const float* eInvSigma2, int nI, const int* iKF1, const int* iKF2,
                                             const orc_imu_preintegrated* iPre, const uint8_t* iRobust, const float* iInfoScale,
                                             const float* rig28, const float* Tbc12, int bLarge, uint8_t* eraseFlag, int* stats2) {
  return localInertialBA(nKF, kfState21, kfKind, nMP, mpPos, mpClose, nE, eKF, eMP, eObs, eInvSigma2, nI, iKF1, iKF2, iPre, iRobust,
                         iInfoScale, 0, 0, 0, 0, 0, Tbc12, bLarge, eraseFlag, stats2, eRight, rig28);
}
