// Frame-side helpers of the non-rectified / RGB-D input paths for MI355X (gfx950), SURVEY 8(f) row N4 (second half):
//   Frame::UndistortKeyPoints     (reference src/Frame.cc:829-857)  = cv::undistortPoints(pts, K, distCoef, R = I, P = mK)
//   Frame::ComputeStereoFromRGBD  (reference src/Frame.cc:1049-1067)
//   Frame::ComputeImageBounds     (reference src/Frame.cc:859-887), host
// cv::undistortPoints (OpenCV 4.x calib3d, cvUndistortPointsInternal; not vendored: restated, parity unpinned): normalise with K,
// five fixed-point iterations of the inverse Brown-Conrady model in FP64 (the 6-argument overload's TermCriteria(MAX_ITER, 5,
// 0.01)), re-project with P.  Only +, *, / in FP64 without contraction: bit-identical to the oracle's restatement.
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.h"
#include "internal_abi.h"

using namespace morb;

struct morb_matcher;
extern "C" {
void* morb_matcher_stream(const morb_matcher*);
}

namespace {

struct Distortion { double fx, fy, cx, cy, ifx, ify; double k[5]; };   // k1 k2 p1 p2 k3

__host__ __device__ inline void undistort_point(const Distortion& D, float px, float py, float* ox, float* oy) {
  const double u = (double)px, v = (double)py;
  double x = (u - D.cx) * D.ifx, y = (v - D.cy) * D.ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((D.k[4] * r2 + D.k[1]) * r2 + D.k[0]) * r2);   // k4..k6 = 0
    if (icdist < 0) { x = (u - D.cx) * D.ifx; y = (v - D.cy) * D.ify; break; }
    const double deltaX = 2 * D.k[2] * x * y + D.k[3] * (r2 + 2 * x * x) + 0.0 * r2 + 0.0 * r2 * r2;   // thin-prism terms absent
    const double deltaY = D.k[2] * (r2 + 2 * y * y) + 2 * D.k[3] * x * y + 0.0 * r2 + 0.0 * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  // RR = P * I with P = mK: the third row is (0 0 1)
  const double xx = D.fx * x + 0.0 * y + D.cx, yy = 0.0 * x + D.fy * y + D.cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
  *ox = (float)(xx * ww);
  *oy = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort(Distortion D, int passthrough, int cap, const int* __restrict__ count,
                                                   const morb_keypoint* __restrict__ in, morb_keypoint* __restrict__ out) {
  const int img = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int n = count ? count[img] : cap;
  if (i >= n) return;
  morb_keypoint kp = in[(size_t)img * cap + i];
  if (!passthrough) undistort_point(D, kp.x, kp.y, &kp.x, &kp.y);
  out[(size_t)img * cap + i] = kp;
}

__global__ __launch_bounds__(256) void k_rgbd(int cap, const int* __restrict__ count, const morb_keypoint* __restrict__ kps,
                                              const morb_keypoint* __restrict__ kpsUn, const float* __restrict__ depth, int W, int H,
                                              size_t pitch, size_t imgPitch, float bf, float* __restrict__ uRight,
                                              float* __restrict__ depthOut) {
  const int img = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int n = count ? count[img] : cap;
  if (i >= cap) return;
  float ur = -1.f, d = -1.f;
  if (i < n) {
    const morb_keypoint kp = kps[(size_t)img * cap + i];
    const int v = (int)kp.y, u = (int)kp.x;   // imDepth.at<float>(v, u): float -> int conversion truncates
    if (u >= 0 && u < W && v >= 0 && v < H) {
      const float z = depth[(size_t)img * imgPitch + (size_t)v * pitch + u];
      if (z > 0) { d = z; ur = kpsUn[(size_t)img * cap + i].x - bf / z; }
    }
  }
  uRight[(size_t)img * cap + i] = ur;
  depthOut[(size_t)img * cap + i] = d;
}

// TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)'s BowVector (Thirdparty/DBoW2/DBoW2/
// TemplatedVocabulary.h:1127-1190, BowVector.cpp:34-84): a std::map<WordId, double>, i.e. the distinct words in ascending id
// order.  One workgroup per image: (word, feature index) keys are sorted in LDS, a run head adds the word's weight once per
// feature of the run IN ORDER (addWeight's repeated +=; addIfNotExist keeps the first), then TF normalisation by the number
// of words or the scoring object's L1 / L2 norm, accumulated by one thread in word order like the reference's loop.
__global__ __launch_bounds__(256) void k_bow_vector(const int* __restrict__ leaf, const int* __restrict__ count, int cap, int P,
                                                    const int* __restrict__ nodeWordId, const double* __restrict__ nodeWeight,
                                                    int weighting, int scoring, int* __restrict__ outWord,
                                                    double* __restrict__ outValue, int* __restrict__ outCount) {
  extern __shared__ unsigned long long skeys[];   // P keys, then P doubles (values of the run heads), then P ints (output slots)
  double* sval = reinterpret_cast<double*>(skeys + P);
  int* sslot = reinterpret_cast<int*>(sval + P);
  __shared__ int sN, sWords;
  __shared__ double sNorm;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int n = count ? count[img] : cap;
  for (int i = tid; i < P; i += 256) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const int lf = leaf[(size_t)img * cap + i];
      if (lf >= 0 && nodeWeight[lf] > 0) {   // w > 0: "not stopped"
        const unsigned w = (unsigned)(nodeWordId ? nodeWordId[lf] : lf);
        k = ((unsigned long long)w << 32) | (unsigned)i;
      }
    }
    skeys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = skeys[i], b = skeys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { skeys[i] = b; skeys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  // run heads and their values
  for (int i = tid; i < P; i += 256) {
    const unsigned long long k = skeys[i];
    int head = 0;
    if (k != ~0ull && (i == 0 || (skeys[i - 1] >> 32) != (k >> 32))) {
      head = 1;
      const int lf = leaf[(size_t)img * cap + (int)(k & 0xFFFFFFFFu)];
      const double w = nodeWeight[lf];
      double v = w;
      if (weighting == 0 || weighting == 1) {   // TF_IDF / TF: addWeight
        for (int q = i + 1; q < P && (skeys[q] >> 32) == (k >> 32); ++q) v += w;
      }
      sval[i] = v;
    }
    sslot[i] = head;
  }
  __syncthreads();
  if (tid == 0) {   // exclusive scan of the head flags (<= P entries; the map's iteration order)
    int acc = 0;
    for (int i = 0; i < P; ++i) { const int h = sslot[i]; sslot[i] = h ? acc : -1; acc += h; }
    sWords = acc;
    const bool must = scoring != 5;   // DotProductScoring is the only object that does not normalise (ScoringObject.h:74-90)
    double norm = 0.0;
    if (must) {
      for (int i = 0; i < P; ++i) if (sslot[i] >= 0) norm += (scoring == 1) ? sval[i] * sval[i] : fabs(sval[i]);   // L2_NORM : L1
      if (scoring == 1) norm = sqrt(norm);
    }
    sNorm = must ? norm : -1.0;
    sN = acc;
  }
  __syncthreads();
  const int nWords = sWords;
  const double norm = sNorm;
  for (int i = tid; i < P; i += 256) {
    const int slot = sslot[i];
    if (slot < 0) continue;
    double v = sval[i];
    if (norm < 0) { if ((weighting == 0 || weighting == 1) && nWords > 0) v /= (double)nWords; }   // !must: divide by v.size()
    else if (norm > 0.0) v /= norm;
    outWord[(size_t)img * cap + slot] = (int)(skeys[i] >> 32);
    outValue[(size_t)img * cap + slot] = v;
  }
  if (tid == 0) outCount[img] = sN;
}

static void make_distortion(float fx, float fy, float cx, float cy, const float* dist5, Distortion& D) {
  D.fx = fx; D.fy = fy; D.cx = cx; D.cy = cy; D.ifx = 1. / D.fx; D.ify = 1. / D.fy;
  for (int k = 0; k < 5; ++k) D.k[k] = dist5 ? (double)dist5[k] : 0.0;
}

}  // namespace

extern "C" {

int morb_undistort_keypoints_batch(morb_matcher* m, int nimg, int cap, const int* d_count, const morb_keypoint* d_kps, float fx,
                                   float fy, float cx, float cy, const float* dist5, morb_keypoint* d_kpsUn, void* stream) {
  MORB_REQUIRE(m && d_kps && d_kpsUn && dist5, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nimg > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  Distortion D;
  make_distortion(fx, fy, cx, cy, dist5, D);
  hipLaunchKernelGGL(k_undistort, dim3(div_up(cap, 256), nimg), dim3(256), 0, st, D, dist5[0] == 0.0f ? 1 : 0, cap, d_count, d_kps, d_kpsUn);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_stereo_from_rgbd_batch(morb_matcher* m, int nimg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                const morb_keypoint* d_kpsUn, const float* d_depth, int width, int height, size_t rowPitchFloats,
                                size_t imagePitchFloats, float bf, float* d_uRight, float* d_depthOut, void* stream) {
  MORB_REQUIRE(m && d_kps && d_kpsUn && d_depth && d_uRight && d_depthOut, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nimg > 0 && cap > 0 && width > 0 && height > 0 && rowPitchFloats >= (size_t)width &&
                   imagePitchFloats >= rowPitchFloats * (size_t)height, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  hipLaunchKernelGGL(k_rgbd, dim3(div_up(cap, 256), nimg), dim3(256), 0, st, cap, d_count, d_kps, d_kpsUn, d_depth, width, height,
                     rowPitchFloats, imagePitchFloats, bf, d_uRight, d_depthOut);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_bow_vector_batch(morb_matcher* m, int nimg, const int* d_leaf, const int* d_count, int cap, const int* d_nodeWordId,
                          const double* d_nodeWeight, int weighting, int scoring, int* d_bowWord, double* d_bowValue, int* d_bowCount,
                          void* stream) {
  MORB_REQUIRE(m && d_leaf && d_nodeWeight && d_bowWord && d_bowValue && d_bowCount, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nimg > 0 && cap > 0 && weighting >= 0 && weighting <= 3 && scoring >= 0 && scoring <= 5, MORB_ERR_INVALID, "bad argument");
  int P = 1;
  while (P < cap) P <<= 1;
  const size_t lds = (size_t)P * (8 + 8 + 4);
  MORB_REQUIRE(lds <= 150 * 1024, MORB_ERR_CAPACITY, "cap too large for the BowVector kernel");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  if (lds > 48 * 1024) MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_vector), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_bow_vector, dim3(nimg), dim3(256), lds, st, d_leaf, d_count, cap, P, d_nodeWordId, d_nodeWeight, weighting, scoring,
                     d_bowWord, d_bowValue, d_bowCount);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float* dist5, float* bounds4) {
  MORB_REQUIRE(dist5 && bounds4 && width > 0 && height > 0, MORB_ERR_INVALID, "bad argument");
  if (dist5[0] == 0.0f) { bounds4[0] = 0.0f; bounds4[1] = (float)width; bounds4[2] = 0.0f; bounds4[3] = (float)height; return MORB_OK; }
  Distortion D;
  make_distortion(fx, fy, cx, cy, dist5, D);
  const float corners[4][2] = {{0.f, 0.f}, {(float)width, 0.f}, {0.f, (float)height}, {(float)width, (float)height}};
  float ux[4], uy[4];
  for (int k = 0; k < 4; ++k) undistort_point(D, corners[k][0], corners[k][1], &ux[k], &uy[k]);
  bounds4[0] = ux[0] < ux[2] ? ux[0] : ux[2];   // mnMinX = min(corner 0, corner 2)
  bounds4[1] = ux[1] > ux[3] ? ux[1] : ux[3];   // mnMaxX
  bounds4[2] = uy[0] < uy[1] ? uy[0] : uy[1];   // mnMinY
  bounds4[3] = uy[2] > uy[3] ? uy[2] : uy[3];   // mnMaxY
  return MORB_OK;
}

}  // extern "C"
