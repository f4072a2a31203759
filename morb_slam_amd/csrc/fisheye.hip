// Fisheye stereo front end for MI355X (gfx950): Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:1222-1274)
// = cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) over the two lapping areas (k_knn2 in matcher.hip), Lowe's ratio 0.7,
// then KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:323-395): unproject (Newton on the KB8
// polynomial), parallax test, linear triangulation, depth and reprojection gates.  One thread per left query.
// Eigen::JacobiSVD of the 4x4 system is replaced by a cyclic Jacobi eigen-decomposition of A^T A in FP64 (same null
// vector; see oracle/fisheye.cc).  Matches of several left features to one right feature resolve like the reference's
// sequential loop: the highest left index wins (atomicMax).
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.h"

using namespace morb;

struct morb_matcher;
extern "C" {
int morb_matcher_device(const morb_matcher*);
void* morb_matcher_stream(const morb_matcher*);
int morb_matcher_workspace(morb_matcher*, int which, size_t bytes, void** out);
int morb_hamming_knn2_batch(morb_matcher*, int nprob, const uint8_t* d_query, const int* d_nq, int qPitch, const int* d_qOff,
                            const uint8_t* d_train, const int* d_nt, int tPitch, const int* d_tOff, int* d_idx, int* d_dist,
                            void* stream);
}

namespace {

struct KB8 { float p[8]; };
struct RigF { KB8 cl, cr; float Rlr[9], tlr[3]; float sigma2[16]; };

__device__ __forceinline__ void kb8_project_f(const KB8& c, const float* v, float* uv) {
  const float x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
  const float theta = atan2f(sqrtf(x2_plus_y2), v[2]);
  const float psi = atan2f(v[1], v[0]);
  const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
              theta9 = theta7 * theta2;
  const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
  uv[0] = c.p[0] * r * cosf(psi) + c.p[2];
  uv[1] = c.p[1] * r * sinf(psi) + c.p[3];
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
    scale = tanf(theta) / theta_d;
  }
  ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}
__device__ void null_vector4(const float* A, double* out) {
  double M[16], V[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += (double)A[k * 4 + i] * (double)A[k * 4 + j];
      M[i * 4 + j] = s;
      V[i * 4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; ++sweep)
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = M[p * 4 + q];
        if (apq == 0.0) continue;
        const double tau = (M[q * 4 + q] - M[p * 4 + p]) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
        for (int k = 0; k < 4; ++k) { const double a = M[k * 4 + p], b = M[k * 4 + q]; M[k * 4 + p] = cs * a - sn * b; M[k * 4 + q] = sn * a + cs * b; }
        for (int k = 0; k < 4; ++k) { const double a = M[p * 4 + k], b = M[q * 4 + k]; M[p * 4 + k] = cs * a - sn * b; M[q * 4 + k] = sn * a + cs * b; }
        for (int k = 0; k < 4; ++k) { const double a = V[k * 4 + p], b = V[k * 4 + q]; V[k * 4 + p] = cs * a - sn * b; V[k * 4 + q] = sn * a + cs * b; }
      }
  int best = 0;
  for (int i = 1; i < 4; ++i) if (M[i * 4 + i] < M[best * 4 + best]) best = i;
  for (int k = 0; k < 4; ++k) out[k] = V[k * 4 + best];
}
__device__ float triangulate_matches(const RigF& g, float x1, float y1, float x2, float y2, float sigma1, float unc, float* p3D) {
  float r1[3], r2[3], r21[3];
  kb8_unproject(g.cl, x1, y1, r1);
  kb8_unproject(g.cr, x2, y2, r2);
  const float* R12 = g.Rlr;
  for (int i = 0; i < 3; ++i) r21[i] = (R12[i * 3] * r2[0] + R12[i * 3 + 1] * r2[1]) + R12[i * 3 + 2] * r2[2];
  const float n1 = sqrtf(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  const float n21 = sqrtf(r21[0] * r21[0] + r21[1] * r21[1] + r21[2] * r21[2]);
  const float cosParallaxRays = (r1[0] * r21[0] + r1[1] * r21[1] + r1[2] * r21[2]) / (n1 * n21);
  if ((double)cosParallaxRays > 0.9998) return -1;
  float R21[9], t2[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R21[i * 3 + j] = R12[j * 3 + i];
  for (int i = 0; i < 3; ++i) t2[i] = -((R21[i * 3] * g.tlr[0] + R21[i * 3 + 1] * g.tlr[1]) + R21[i * 3 + 2] * g.tlr[2]);
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
  kb8_project_f(g.cl, x3D, uv1);
  const float ex1 = uv1[0] - x1, ey1 = uv1[1] - y1;
  if ((double)(ex1 * ex1 + ey1 * ey1) > 5.991 * (double)sigma1) return -4;
  float x3D2[3];
  for (int i = 0; i < 3; ++i) x3D2[i] = (R21[i * 3] * x3D[0] + R21[i * 3 + 1] * x3D[1]) + R21[i * 3 + 2] * x3D[2] + t2[i];
  float uv2[2];
  kb8_project_f(g.cr, x3D2, uv2);
  const float ex2 = uv2[0] - x2, ey2 = uv2[1] - y2;
  if ((double)(ex2 * ex2 + ey2 * ey2) > 5.991 * (double)unc) return -5;
  p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
  return z1;
}

__global__ void k_fe_prepare(const int* __restrict__ count, const int* __restrict__ mono, int nframes, int* __restrict__ nq,
                             int* __restrict__ nt, int* __restrict__ qOff, int* __restrict__ tOff) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= nframes) return;
  nq[f] = count[2 * f]; nt[f] = count[2 * f + 1]; qOff[f] = mono[2 * f]; tOff[f] = mono[2 * f + 1];
}
__global__ void k_fe_init(int n, int* __restrict__ l2r, int* __restrict__ r2l, float* __restrict__ depth, float* __restrict__ p3D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  l2r[i] = -1; r2l[i] = -1; depth[i] = -1.0f; p3D[3 * i] = 0; p3D[3 * i + 1] = 0; p3D[3 * i + 2] = 0;
}
__global__ __launch_bounds__(256) void k_fe_triangulate(RigF g, int cap, const int* __restrict__ count, const int* __restrict__ mono,
                                                        const morb_keypoint* __restrict__ kps, const int* __restrict__ idx,
                                                        const int* __restrict__ dist, int* __restrict__ l2r,
                                                        int* __restrict__ r2l, float* __restrict__ depth,
                                                        float* __restrict__ p3D, int* __restrict__ nMatches) {
  const int f = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  const int monoL = mono[2 * f], monoR = mono[2 * f + 1];
  const int nq = count[2 * f] - monoL;
  if (q >= nq) return;
  const size_t o = ((size_t)2 * f * cap + q) * 2;   // knn outputs are laid out with the query pitch (2*cap rows per problem)
  const int i1 = idx[o + 1];
  if (i1 < 0) return;                                // (*it).size() >= 2
  if (!((double)(float)dist[o] < (double)(float)dist[o + 1] * 0.7)) return;
  const int iL = q + monoL, iR = idx[o] + monoR;
  const morb_keypoint kl = kps[(size_t)(2 * f) * cap + iL], kr = kps[(size_t)(2 * f + 1) * cap + iR];
  float P[3];
  const float d = triangulate_matches(g, kl.x, kl.y, kr.x, kr.y, g.sigma2[kl.octave], g.sigma2[kr.octave], P);
  if (d > 0.0001f) {
    const size_t of = (size_t)f * cap;
    l2r[of + iL] = iR;
    atomicMax(&r2l[of + iR], iL);   // sequential loop: the last (highest) left index overwrites
    p3D[(of + iL) * 3] = P[0]; p3D[(of + iL) * 3 + 1] = P[1]; p3D[(of + iL) * 3 + 2] = P[2];
    depth[of + iL] = d;
    atomicAdd(&nMatches[f], 1);
  }
}

}  // namespace

extern "C" int morb_stereo_fisheye_match_batch(morb_matcher* m, int nframes, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                               const int* d_count, const int* d_mono, int cap, const float* camL8,
                                               const float* camR8, const float* Rlr9, const float* tlr3,
                                               const float* levelSigma2, int nlevels, int* d_leftToRight, int* d_rightToLeft,
                                               float* d_depth, float* d_p3D, int* d_nMatches, void* stream) {
  MORB_REQUIRE(m && d_kps && d_desc && d_count && d_mono && camL8 && camR8 && Rlr9 && tlr3 && levelSigma2 && d_leftToRight &&
                   d_rightToLeft && d_depth && d_p3D && d_nMatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && nlevels >= 1 && nlevels <= 16, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  RigF g;
  memset(&g, 0, sizeof g);
  memcpy(g.cl.p, camL8, 32); memcpy(g.cr.p, camR8, 32);
  memcpy(g.Rlr, Rlr9, 36); memcpy(g.tlr, tlr3, 12);
  memcpy(g.sigma2, levelSigma2, sizeof(float) * nlevels);
  void *aux = nullptr, *idx = nullptr, *dist = nullptr;
  int rc = morb_matcher_workspace(m, 5, sizeof(int) * 4 * (size_t)nframes, &aux);
  // knn outputs indexed by (problem, query row) with the query pitch = 2*cap rows (left image of frame f = row block 2f)
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 0, sizeof(int) * 2 * (size_t)nframes * 2 * cap, &idx);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 1, sizeof(int) * 2 * (size_t)nframes * 2 * cap, &dist);
  if (rc != MORB_OK) return rc;
  int* nq = (int*)aux; int* nt = nq + nframes; int* qOff = nt + nframes; int* tOff = qOff + nframes;
  hipLaunchKernelGGL(k_fe_prepare, dim3(div_up(nframes, 256)), dim3(256), 0, st, d_count, d_mono, nframes, nq, nt, qOff, tOff);
  hipLaunchKernelGGL(k_fe_init, dim3(div_up(nframes * cap, 256)), dim3(256), 0, st, nframes * cap, d_leftToRight, d_rightToLeft, d_depth, d_p3D);
  MORB_HIP_CHECK(hipMemsetAsync(d_nMatches, 0, sizeof(int) * nframes, st));
  rc = morb_hamming_knn2_batch(m, nframes, d_desc, nq, 2 * cap, qOff, d_desc + (size_t)cap * 32, nt, 2 * cap, tOff, (int*)idx, (int*)dist, st);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_fe_triangulate, dim3(div_up(cap, 256), nframes), dim3(256), 0, st, g, cap, d_count, d_mono, d_kps,
                     (const int*)idx, (const int*)dist, d_leftToRight, d_rightToLeft, d_depth, d_p3D, d_nMatches);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}
