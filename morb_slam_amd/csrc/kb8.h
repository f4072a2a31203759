// KannalaBrandt8 camera pieces shared by the fisheye stereo matcher (fisheye.hip) and the fisheye branch of
// SearchForTriangulation (projection.hip): project / unproject (KannalaBrandt8.cpp:49-67, :100-137), the null vector
// of the 4x4 triangulation system (Eigen::JacobiSVD restated as an FP64 Jacobi eigen-decomposition of A^T A) and
// TriangulateMatches (:323-395).
#pragma once
#include <hip/hip_runtime.h>

#include "libm_f32.h"   // glibc's atan2f / tanf / cosf / sinf, bit for bit: the reference runs on the CPU's libm

namespace morbkb8 {

struct KB8 { float p[8]; };

__device__ __forceinline__ void kb8_project_f(const KB8& c, const float* v, float* uv) {
  const float x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
  const float theta = morbm::atan2f_glibc(sqrtf(x2_plus_y2), v[2]);
  const float psi = morbm::atan2f_glibc(v[1], v[0]);
  const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
              theta9 = theta7 * theta2;
  const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
  uv[0] = c.p[0] * r * morbm::cosf_glibc(psi) + c.p[2];
  uv[1] = c.p[1] * r * morbm::sinf_glibc(psi) + c.p[3];
}
__device__ __forceinline__ void kb8_unproject(const KB8& c, float px, float py, float* ray) {
  const float pwx = (px - c.p[2]) / c.p[0], pwy = (py - c.p[3]) / c.p[1];
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  theta_d = fminf(fmaxf((float)(-3.14159265358979323846 / 2.f), theta_d), (float)(3.14159265358979323846 / 2.f));
  if (theta_d > 1e-8f) {
    float theta = theta_d;
    for (int j = 0; j < 10; j++) {
      const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      const float k0 = c.p[4] * theta2, k1 = c.p[5] * theta4, k2 = c.p[6] * theta6, k3 = c.p[7] * theta8;
      const float fix = (theta * (1 + k0 + k1 + k2 + k3) - theta_d) / (1 + 3 * k0 + 5 * k1 + 7 * k2 + 9 * k3);
      theta = theta - fix;
      if (fabsf(fix) < 1e-6f) break;
    }
    scale = morbm::tanf_glibc(theta) / theta_d;
  }
  ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}
// (the (p, q) loops and every k loop are unrolled: with run-time indices M and V live in scratch memory and each of the 180 rotations is
// ~50 dependent memory accesses; with constant indices they are 32 FP64 registers)
__device__ inline void null_vector4(const float* A, double* out) {
  double M[16], V[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += (double)A[k * 4 + i] * (double)A[k * 4 + j];
      M[i * 4 + j] = s;
      V[i * 4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double apq = M[p * 4 + q];
        if (apq == 0.0) continue;
        const double tau = (M[q * 4 + q] - M[p * 4 + p]) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double a = M[k * 4 + p], b = M[k * 4 + q]; M[k * 4 + p] = cs * a - sn * b; M[k * 4 + q] = sn * a + cs * b; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double a = M[p * 4 + k], b = M[q * 4 + k]; M[p * 4 + k] = cs * a - sn * b; M[q * 4 + k] = sn * a + cs * b; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double a = V[k * 4 + p], b = V[k * 4 + q]; V[k * 4 + p] = cs * a - sn * b; V[k * 4 + q] = sn * a + cs * b; }
      }
  }
  // the eigenvector of the smallest eigenvalue (first of equal ones), selected without a run-time column index
  double bestVal = M[0];
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = V[k * 4];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (M[i * 4 + i] < bestVal) {
      bestVal = M[i * 4 + i];
#pragma unroll
      for (int k = 0; k < 4; ++k) out[k] = V[k * 4 + i];
    }
}
// KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:323-395): this = c1, pCamera2 = c2; returns the depth in c1 (> 0) or a
// negative reject code
__device__ inline float triangulate_matches(const KB8& c1, const KB8& c2, const float* R12, const float* t12, float x1, float y1,
                                            float x2, float y2, float sigma1, float unc, float* p3D) {
  float r1[3], r2[3], r21[3];
  kb8_unproject(c1, x1, y1, r1);
  kb8_unproject(c2, x2, y2, r2);
  for (int i = 0; i < 3; ++i) r21[i] = (R12[i * 3] * r2[0] + R12[i * 3 + 1] * r2[1]) + R12[i * 3 + 2] * r2[2];
  const float n1 = sqrtf(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  const float n21 = sqrtf(r21[0] * r21[0] + r21[1] * r21[1] + r21[2] * r21[2]);
  const float cosParallaxRays = (r1[0] * r21[0] + r1[1] * r21[1] + r1[2] * r21[2]) / (n1 * n21);
  if ((double)cosParallaxRays > 0.9998) return -1;
  float R21[9], t2[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R21[i * 3 + j] = R12[j * 3 + i];
  for (int i = 0; i < 3; ++i) t2[i] = -((R21[i * 3] * t12[0] + R21[i * 3 + 1] * t12[1]) + R21[i * 3 + 2] * t12[2]);
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
  null_vector4(A, xh);
  const float x3D[3] = {(float)(xh[0] / xh[3]), (float)(xh[1] / xh[3]), (float)(xh[2] / xh[3])};
  const float z1 = x3D[2];
  if (z1 <= 0) return -2;
  const float z2 = (R21[6] * x3D[0] + R21[7] * x3D[1] + R21[8] * x3D[2]) + t2[2];
  if (z2 <= 0) return -3;
  float uv1[2];
  kb8_project_f(c1, x3D, uv1);
  const float ex1 = uv1[0] - x1, ey1 = uv1[1] - y1;
  if ((double)(ex1 * ex1 + ey1 * ey1) > 5.991 * (double)sigma1) return -4;
  float x3D2[3];
  for (int i = 0; i < 3; ++i) x3D2[i] = (R21[i * 3] * x3D[0] + R21[i * 3 + 1] * x3D[1]) + R21[i * 3 + 2] * x3D[2] + t2[i];
  float uv2[2];
  kb8_project_f(c2, x3D2, uv2);
  const float ex2 = uv2[0] - x2, ey2 = uv2[1] - y2;
  if ((double)(ex2 * ex2 + ey2 * ey2) > 5.991 * (double)unc) return -5;
  p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
  return z1;
}

// FP64 forms used by the optimisers: KannalaBrandt8::project(Vector3d) (KannalaBrandt8.cpp:74-92, with its float atan2f leak) and
// projectJac (:164-199).  c = fx fy cx cy k0 k1 k2 k3.
__device__ __forceinline__ void kb8_project_d(const float* c, const double* v, double* uv) {
  const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
  const double theta = (double)morbm::atan2f_glibc(sqrtf((float)x2_plus_y2), (float)v[2]);   // the reference's float leak
  const double psi = (double)morbm::atan2f_glibc((float)v[1], (float)v[0]);
  const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
               theta9 = theta7 * theta2;
  const double r = theta + c[4] * theta3 + c[5] * theta5 + c[6] * theta7 + c[7] * theta9;
  uv[0] = c[0] * r * cos(psi) + c[2];
  uv[1] = c[1] * r * sin(psi) + c[3];
}
__device__ __forceinline__ void kb8_project_jac(const float* c, const double* v, double* J) {
  const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
  const double r2 = x2 + y2, r = sqrt(r2), r3 = r2 * r;
  const double theta = atan2(r, v[2]);
  const double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta;
  const double theta6 = theta2 * theta4, theta7 = theta6 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
  const double f = theta + theta3 * c[4] + theta5 * c[5] + theta7 * c[6] + theta9 * c[7];
  const double fd = 1 + 3 * c[4] * theta2 + 5 * c[5] * theta4 + 7 * c[6] * theta6 + 9 * c[7] * theta8;
  J[0] = c[0] * (fd * v[2] * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
  J[3] = c[1] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
  J[1] = c[0] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
  J[4] = c[1] * (fd * v[2] * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
  J[2] = -c[0] * fd * v[0] / (r2 + z2);
  J[5] = -c[1] * fd * v[1] / (r2 + z2);
}

}  // namespace morbkb8
