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
#include "internal_abi.h"
#include "kb8.h"

using namespace morb;

struct morb_matcher;
extern "C" {
void* morb_matcher_stream(const morb_matcher*);
int morb_hamming_knn2_batch(morb_matcher*, int nprob, const uint8_t* d_query, const int* d_nq, int qPitch, const int* d_qOff,
                            const uint8_t* d_train, const int* d_nt, int tPitch, const int* d_tOff, int* d_idx, int* d_dist,
                            void* stream);
}

namespace {

using morbkb8::KB8;
using morbkb8::kb8_project_f;
using morbkb8::kb8_unproject;
struct RigF { KB8 cl, cr; float Rlr[9], tlr[3]; float sigma2[16]; };

__device__ __forceinline__ float triangulate_matches(const RigF& g, float x1, float y1, float x2, float y2, float sigma1, float unc,
                                                      float* p3D) {
  return morbkb8::triangulate_matches(g.cl, g.cr, g.Rlr, g.tlr, x1, y1, x2, y2, sigma1, unc, p3D);
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
