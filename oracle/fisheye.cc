// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
//
// CPU restatement of the KannalaBrandt8 (fisheye) pieces of the hot path:
//   KannalaBrandt8::project (f32 and f64 variants), unproject, projectJac   /root/reference/src/CameraModels/KannalaBrandt8.cpp:31-184
//   KannalaBrandt8::TriangulateMatches / Triangulate                        :323-395, :415-428
//   Frame::ComputeStereoFishEyeMatches                                      src/Frame.cc:1222-1274
//   Optimizer::PoseOptimization with a second camera (EdgeSE3ProjectXYZOnlyPose on the left KB8 camera,
//   EdgeSE3ProjectXYZOnlyPoseToBody on the right one)                        src/Optimizer.cc:880-946, OptimizableTypes.cpp:49-104
// Eigen::JacobiSVD (not vendored, absent) is replaced by a cyclic Jacobi eigen-decomposition of A^T A in
// double precision (the null vector of A = the eigenvector of the smallest eigenvalue); vs the reference's
// float JacobiSVD the triangulated point agrees to float rounding (PARITY UNPINNED there).  libm supplies
// atan2f / sqrtf / tan / cos / sin exactly as in the reference.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "matcher.h"
#include "orb_oracle.h"

namespace orc {
void knnMatch2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist);

namespace kb8 {


// KannalaBrandt8::project(const Eigen::Vector3f&) (:66-92)
void projectF(const Cam& c, const float* v3D, float* uv) {
  const float x2_plus_y2 = v3D[0] * v3D[0] + v3D[1] * v3D[1];
  const float theta = atan2f(sqrtf(x2_plus_y2), v3D[2]);
  const float psi = atan2f(v3D[1], v3D[0]);
  const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
              theta9 = theta7 * theta2;
  const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
  uv[0] = c.p[0] * r * std::cos(psi) + c.p[2];
  uv[1] = c.p[1] * r * std::sin(psi) + c.p[3];
}
// KannalaBrandt8::project(const Eigen::Vector3d&) (:48-66): atan2f / sqrtf on doubles (float leak), double polynomial
void projectD(const Cam& c, const double* v3D, double* uv) {
  const double x2_plus_y2 = v3D[0] * v3D[0] + v3D[1] * v3D[1];
  const double theta = atan2f(sqrtf((float)x2_plus_y2), (float)v3D[2]);
  const double psi = atan2f((float)v3D[1], (float)v3D[0]);
  const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
               theta9 = theta7 * theta2;
  const double r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
  uv[0] = c.p[0] * r * std::cos(psi) + c.p[2];
  uv[1] = c.p[1] * r * std::sin(psi) + c.p[3];
}
// KannalaBrandt8::unproject (:115-147), precision = 1e-6
void unproject(const Cam& c, float px, float py, float* ray) {
  const float pwx = (px - c.p[2]) / c.p[0], pwy = (py - c.p[3]) / c.p[1];
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  theta_d = fminf(fmaxf((float)(-3.14159265358979323846 / 2.f), theta_d), (float)(3.14159265358979323846 / 2.f));
  if (theta_d > 1e-8) {
    float theta = theta_d;
    for (int j = 0; j < 10; j++) {
      float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      float k0_theta2 = c.p[4] * theta2, k1_theta4 = c.p[5] * theta4;
      float k2_theta6 = c.p[6] * theta6, k3_theta8 = c.p[7] * theta8;
      float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                        (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
      theta = theta - theta_fix;
      if (fabsf(theta_fix) < 1e-6f) break;
    }
    scale = std::tan(theta) / theta_d;
  }
  ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}
// KannalaBrandt8::projectJac (:149-184)
void projectJac(const Cam& c, const double* v, double* J /*2x3*/) {
  double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
  double r2 = x2 + y2, r = std::sqrt(r2), r3 = r2 * r;
  double theta = std::atan2(r, v[2]);
  double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta;
  double theta6 = theta2 * theta4, theta7 = theta6 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
  double f = theta + theta3 * c.p[4] + theta5 * c.p[5] + theta7 * c.p[6] + theta9 * c.p[7];
  double fd = 1 + 3 * c.p[4] * theta2 + 5 * c.p[5] * theta4 + 7 * c.p[6] * theta6 + 9 * c.p[7] * theta8;
  J[0] = c.p[0] * (fd * v[2] * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
  J[3] = c.p[1] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
  J[1] = c.p[0] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
  J[4] = c.p[1] * (fd * v[2] * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
  J[2] = -c.p[0] * fd * v[0] / (r2 + z2);
  J[5] = -c.p[1] * fd * v[1] / (r2 + z2);
}

// null vector of the 4x4 matrix A: eigenvector of A^T A for the smallest eigenvalue, cyclic Jacobi, double
static void nullVector4(const float A[16], double out[4]) {
  double M[16], V[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += (double)A[k * 4 + i] * (double)A[k * 4 + j];
      M[i * 4 + j] = s;
      V[i * 4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = M[p * 4 + q];
        if (apq == 0.0) continue;
        const double tau = (M[q * 4 + q] - M[p * 4 + p]) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
        for (int k = 0; k < 4; ++k) {  // rotate columns p, q
          const double mkp = M[k * 4 + p], mkq = M[k * 4 + q];
          M[k * 4 + p] = cs * mkp - sn * mkq;
          M[k * 4 + q] = sn * mkp + cs * mkq;
        }
        for (int k = 0; k < 4; ++k) {  // rotate rows p, q
          const double mpk = M[p * 4 + k], mqk = M[q * 4 + k];
          M[p * 4 + k] = cs * mpk - sn * mqk;
          M[q * 4 + k] = sn * mpk + cs * mqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k * 4 + p], vkq = V[k * 4 + q];
          V[k * 4 + p] = cs * vkp - sn * vkq;
          V[k * 4 + q] = sn * vkp + cs * vkq;
        }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i) if (M[i * 4 + i] < M[best * 4 + best]) best = i;
  for (int k = 0; k < 4; ++k) out[k] = V[k * 4 + best];
}

// KannalaBrandt8::TriangulateMatches (:323-395) — both cameras KB8
static float TriangulateMatches(const Cam& c1, const Cam& c2, float x1, float y1, float x2, float y2, const float* R12,
                                const float* t12, float sigmaLevel, float unc, float* p3D) {
  float r1[3], r2[3];
  unproject(c1, x1, y1, r1);
  unproject(c2, x2, y2, r2);
  float r21[3];
  for (int i = 0; i < 3; ++i) r21[i] = (R12[i * 3] * r2[0] + R12[i * 3 + 1] * r2[1]) + R12[i * 3 + 2] * r2[2];
  const float n1 = std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  const float n21 = std::sqrt(r21[0] * r21[0] + r21[1] * r21[1] + r21[2] * r21[2]);
  const float cosParallaxRays = (r1[0] * r21[0] + r1[1] * r21[1] + r1[2] * r21[2]) / (n1 * n21);
  if (cosParallaxRays > 0.9998) return -1;
  float R21[9], t2[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R21[i * 3 + j] = R12[j * 3 + i];
  for (int i = 0; i < 3; ++i) t2[i] = -((R21[i * 3] * t12[0] + R21[i * 3 + 1] * t12[1]) + R21[i * 3 + 2] * t12[2]);
  // Triangulate (:415-428): rows p.x*T.row(2) - T.row(0) ...
  float A[16];
  const float T1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  float T2[12];
  for (int i = 0; i < 3; ++i) { T2[i * 4] = R21[i * 3]; T2[i * 4 + 1] = R21[i * 3 + 1]; T2[i * 4 + 2] = R21[i * 3 + 2]; T2[i * 4 + 3] = t2[i]; }
  for (int k = 0; k < 4; ++k) {
    A[k] = r1[0] * T1[8 + k] - T1[k];
    A[4 + k] = r1[1] * T1[8 + k] - T1[4 + k];
    A[8 + k] = r2[0] * T2[8 + k] - T2[k];
    A[12 + k] = r2[1] * T2[8 + k] - T2[4 + k];
  }
  double xh[4];
  nullVector4(A, xh);
  float x3D[3] = {(float)(xh[0] / xh[3]), (float)(xh[1] / xh[3]), (float)(xh[2] / xh[3])};
  const float z1 = x3D[2];
  if (z1 <= 0) return -2;
  const float z2 = (R21[6] * x3D[0] + R21[7] * x3D[1] + R21[8] * x3D[2]) + t2[2];
  if (z2 <= 0) return -3;
  float uv1[2];
  projectF(c1, x3D, uv1);
  const float errX1 = uv1[0] - x1, errY1 = uv1[1] - y1;
  if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -4;
  float x3D2[3];
  for (int i = 0; i < 3; ++i) x3D2[i] = (R21[i * 3] * x3D[0] + R21[i * 3 + 1] * x3D[1]) + R21[i * 3 + 2] * x3D[2] + t2[i];
  float uv2[2];
  projectF(c2, x3D2, uv2);
  const float errX2 = uv2[0] - x2, errY2 = uv2[1] - y2;
  if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -5;
  p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
  return z1;
}

}  // namespace kb8
}  // namespace orc

using namespace orc;
extern "C" {

float orc_kb8_triangulate_matches(const float* cam1_8, const float* cam2_8, float x1, float y1, float x2, float y2, const float* R12,
                                  const float* t12, float sigmaLevel, float unc, float* p3D) {
  kb8::Cam a, b; memcpy(a.p, cam1_8, 32); memcpy(b.p, cam2_8, 32);
  return kb8::TriangulateMatches(a, b, x1, y1, x2, y2, R12, t12, sigmaLevel, unc, p3D);
}
void orc_kb8_project_f(const float* cam8, const float* v3, float* uv) { kb8::Cam c; memcpy(c.p, cam8, 32); kb8::projectF(c, v3, uv); }
void orc_kb8_project_d(const float* cam8, const double* v3, double* uv) { kb8::Cam c; memcpy(c.p, cam8, 32); kb8::projectD(c, v3, uv); }
void orc_kb8_unproject(const float* cam8, float x, float y, float* ray) { kb8::Cam c; memcpy(c.p, cam8, 32); kb8::unproject(c, x, y, ray); }
void orc_kb8_project_jac(const float* cam8, const double* v3, double* J6) { kb8::Cam c; memcpy(c.p, cam8, 32); kb8::projectJac(c, v3, J6); }

// Frame::ComputeStereoFishEyeMatches (Frame.cc:1222-1274).  Left keypoints [0, Nleft), stereo area from monoLeft;
// right likewise.  Rlr/tlr = mRlr / mtlr.  Outputs: mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvStereo3Dpoints.
int orc_stereo_fisheye_matches(int Nleft, int monoLeft, const orc_keypoint* kpsL, const uint8_t* descL, int Nright,
                               int monoRight, const orc_keypoint* kpsR, const uint8_t* descR, const float* camL8,
                               const float* camR8, const float* Rlr, const float* tlr, const float* levelSigma2,
                               int* leftToRight, int* rightToLeft, float* depth, float* p3D) {
  kb8::Cam c1, c2;
  memcpy(c1.p, camL8, 32); memcpy(c2.p, camR8, 32);
  for (int i = 0; i < Nleft; ++i) { leftToRight[i] = -1; depth[i] = -1.0f; p3D[3 * i] = p3D[3 * i + 1] = p3D[3 * i + 2] = 0; }
  for (int i = 0; i < Nright; ++i) rightToLeft[i] = -1;
  const int nq = Nleft - monoLeft, nt = Nright - monoRight;
  if (nq <= 0) return 0;
  std::vector<int> idx((size_t)nq * 2), dist((size_t)nq * 2);
  knnMatch2(descL + (size_t)monoLeft * 32, nq, descR + (size_t)monoRight * 32, nt, idx.data(), dist.data());
  int nMatches = 0;
  const KeyPoint* kl = (const KeyPoint*)kpsL; const KeyPoint* kr = (const KeyPoint*)kpsR;
  for (int q = 0; q < nq; ++q) {
    if (idx[2 * q + 1] < 0) continue;  // (*it).size() >= 2
    if ((float)dist[2 * q] < (float)dist[2 * q + 1] * 0.7) {
      const int iL = q + monoLeft, iR = idx[2 * q] + monoRight;
      float P[3];
      const float sigma1 = levelSigma2[kl[iL].octave], sigma2 = levelSigma2[kr[iR].octave];
      const float d = kb8::TriangulateMatches(c1, c2, kl[iL].x, kl[iL].y, kr[iR].x, kr[iR].y, Rlr, tlr, sigma1, sigma2, P);
      if (d > 0.0001f) {
        leftToRight[iL] = iR;
        rightToLeft[iR] = iL;
        p3D[3 * iL] = P[0]; p3D[3 * iL + 1] = P[1]; p3D[3 * iL + 2] = P[2];
        depth[iL] = d;
        nMatches++;
      }
    }
  }
  return nMatches;
}
}
