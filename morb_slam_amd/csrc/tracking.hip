// Marshalling between the tracking-side matchers and PoseOptimization for batches of frames whose data stays in HBM.
// In the reference these are the few host loops Tracking.cc runs between its calls into ORBmatcher / Optimizer — on one
// frame, over std::vector<MapPoint*>:
//   Frame::SetPose -> UpdatePoseMatrices                         (src/Frame.cc:541-585)
//   Optimizer::PoseOptimization's edge fill                      (src/Optimizer.cc:803-905)
//   Tracking::TrackWithMotionModel's outlier discard             (src/Tracking.cc:2716-2740)
//   Tracking::SearchLocalPoints' "already matched" marking       (src/Tracking.cc:3117-3133)
//   Tracking::TrackLocalMap's inlier count                       (src/Tracking.cc:2779-2806)
// The drop-in adapters (include/morb/*_reference.h) do them on the host exactly where the reference does; a batch of
// device-resident frames (bench.py's tracking chain, morb_slam_amd/tracking.py) needs them on the device so that
// SearchByProjection -> PoseOptimization -> isInFrustum -> SearchByProjection -> PoseOptimization runs without a host round trip.
// A frame's map points are rows of a per-frame table [mpCap]; "mvpMapPoints" is an index into it (-1 = NULL).
#include <hip/hip_runtime.h>

#include "common.h"
#include "internal_abi.h"

using namespace morb;

struct morb_matcher;
extern "C" {
void* morb_matcher_stream(const morb_matcher*);
}

namespace {

// Sophus::SE3f::rotationMatrix() = Eigen::Quaternionf::toRotationMatrix(); Twc = Tcw.inverse(): conjugate quaternion, translation =
// conj * (t * -1) through Eigen's _transformVector (uv = 2 vec x v; v + w uv + vec x uv)
__global__ __launch_bounds__(64) void k_set_pose(int nframes, const float* __restrict__ pose7, float* __restrict__ Rcw,
                                                 float* __restrict__ tcw, float* __restrict__ Ow) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= nframes) return;
  const float* p = pose7 + 7 * f;
  const float x = p[0], y = p[1], z = p[2], w = p[3];
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  float* R = Rcw + 9 * f;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0f - (txx + tyy);
  tcw[3 * f] = p[4]; tcw[3 * f + 1] = p[5]; tcw[3 * f + 2] = p[6];
  const float ux = -x, uy = -y, uz = -z;            // conjugate
  const float a = p[4] * -1.0f, b = p[5] * -1.0f, c = p[6] * -1.0f;
  float cx = uy * c - uz * b, cy = uz * a - ux * c, cz = ux * b - uy * a;
  cx += cx; cy += cy; cz += cz;
  Ow[3 * f] = a + w * cx + (uy * cz - uz * cy);
  Ow[3 * f + 1] = b + w * cy + (uz * cx - ux * cz);
  Ow[3 * f + 2] = c + w * cz + (ux * cy - uy * cx);
}

// one thread per feature: the unary edge PoseOptimization builds for it (Optimizer.cc:815-900; mono when mvuRight < 0)
__global__ __launch_bounds__(256) void k_pose_edges(morb_frame_params P, int cap, const int* __restrict__ fImg,
                                                    const int* __restrict__ count, const morb_keypoint* __restrict__ kps,
                                                    const float* __restrict__ uRight, const int* __restrict__ match,
                                                    const int* __restrict__ remap, int remapCap, int mpCap,
                                                    const float* __restrict__ mpXw, int* __restrict__ frameMP,
                                                    uint8_t* __restrict__ hasMP, float* __restrict__ obs,
                                                    float* __restrict__ invSigma2, float* __restrict__ Xw) {
  const int f = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cap) return;
  const size_t o = (size_t)f * cap + j;
  const int img = fImg[f];
  int mp = -1;
  if (j < count[img]) {
    mp = match ? match[o] : frameMP[o];
    if (remap && mp >= 0) mp = mp < remapCap ? remap[(size_t)f * remapCap + mp] : -1;
    if (mp >= mpCap) mp = -1;
  }
  if (match) frameMP[o] = mp;
  float ox = 0.f, oy = 0.f, our = -1.f, is2 = 0.f, X0 = 0.f, X1 = 0.f, X2 = 0.f;
  if (mp >= 0) {
    const morb_keypoint kp = kps[(size_t)img * cap + j];
    ox = kp.x; oy = kp.y; our = uRight ? uRight[o] : -1.f;
    is2 = 1.0f / P.levelSigma2[kp.octave & 15];   // mvInvLevelSigma2 (ORBextractor.cc:421)
    const float* X = mpXw + ((size_t)f * mpCap + mp) * 3;
    X0 = X[0]; X1 = X[1]; X2 = X[2];
  }
  hasMP[o] = mp >= 0 ? 1 : 0;
  obs[o * 3] = ox; obs[o * 3 + 1] = oy; obs[o * 3 + 2] = our;
  invSigma2[o] = is2;
  Xw[o * 3] = X0; Xw[o * 3 + 1] = X1; Xw[o * 3 + 2] = X2;
}

// one workgroup per frame.  Features whose map point PoseOptimization flagged lose it (stereo frames: TrackLocalMap does the same
// after its optimisation); every map point the frame still holds — or just lost — is marked "seen in this frame", which is what
// keeps SearchLocalPoints from projecting it again; blocked = the feature holds a map point with observations.
__global__ __launch_bounds__(1024) void k_discard(int cap, const int* __restrict__ fImg, const int* __restrict__ count,
                                                 int* __restrict__ frameMP, uint8_t* __restrict__ outlier, int mpCap,
                                                 const uint8_t* __restrict__ mpHasObs, uint8_t* __restrict__ blocked,
                                                 uint8_t* __restrict__ mpSeen, int* __restrict__ nmatches,
                                                 int* __restrict__ nmatchesMap) {
  const int f = blockIdx.x;
  const int N = count[fImg[f]];
  if (mpSeen) {
    for (int i = threadIdx.x; i < mpCap; i += blockDim.x) mpSeen[(size_t)f * mpCap + i] = 0;
    __syncthreads();
  }
  int nm = 0, nmap = 0;
  for (int j = threadIdx.x; j < cap; j += blockDim.x) {
    const size_t o = (size_t)f * cap + j;
    uint8_t blk = 0;
    if (j < N) {
      int mp = frameMP[o];
      if (mp >= 0) {
        if (mpSeen) mpSeen[(size_t)f * mpCap + mp] = 1;
        const uint8_t ho = mpHasObs ? mpHasObs[(size_t)f * mpCap + mp] : 1;
        if (outlier[o]) { frameMP[o] = -1; outlier[o] = 0; }
        else { ++nm; nmap += ho ? 1 : 0; blk = ho; }
      }
    }
    if (blocked) blocked[o] = blk;
  }
  __shared__ int s[2];
  if (threadIdx.x == 0) { s[0] = 0; s[1] = 0; }
  __syncthreads();
  if (nm) atomicAdd(&s[0], nm);
  if (nmap) atomicAdd(&s[1], nmap);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (nmatches) nmatches[f] = s[0];
    if (nmatchesMap) nmatchesMap[f] = s[1];
  }
}

}  // namespace

extern "C" {

int morb_frame_set_pose_batch(morb_matcher* m, int nframes, const float* d_pose7, float* d_Rcw, float* d_tcw, float* d_Ow,
                              void* stream) {
  MORB_REQUIRE(m && d_pose7 && d_Rcw && d_tcw && d_Ow, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  hipLaunchKernelGGL(k_set_pose, dim3(div_up(nframes, 64)), dim3(64), 0, st, nframes, d_pose7, d_Rcw, d_tcw, d_Ow);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_pose_edges_batch(morb_matcher* m, const morb_frame_params* P, int nframes, const int* d_fImg, int cap, const int* d_count,
                          const morb_keypoint* d_kps, const float* d_uRight, const int* d_match, const int* d_remap, int remapCap,
                          int mpCap, const float* d_mpXw, int* d_frameMP, uint8_t* d_hasMP, float* d_obs, float* d_invSigma2,
                          float* d_Xw, void* stream) {
  MORB_REQUIRE(m && P && d_fImg && d_count && d_kps && d_mpXw && d_frameMP && d_hasMP && d_obs && d_invSigma2 && d_Xw,
               MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && mpCap > 0 && (!d_remap || (d_match && remapCap > 0)), MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  hipLaunchKernelGGL(k_pose_edges, dim3(div_up(cap, 256), nframes), dim3(256), 0, st, *P, cap, d_fImg, d_count, d_kps, d_uRight,
                     d_match, d_remap, remapCap, mpCap, d_mpXw, d_frameMP, d_hasMP, d_obs, d_invSigma2, d_Xw);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_track_discard_outliers_batch(morb_matcher* m, int nframes, const int* d_fImg, int cap, const int* d_count, int* d_frameMP,
                                      uint8_t* d_outlier, int mpCap, const uint8_t* d_mpHasObs, uint8_t* d_blocked,
                                      uint8_t* d_mpSeen, int* d_nmatches, int* d_nmatchesMap, void* stream) {
  MORB_REQUIRE(m && d_fImg && d_count && d_frameMP && d_outlier, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && mpCap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  hipLaunchKernelGGL(k_discard, dim3(nframes), dim3(1024), 0, st,   // (one frame's 2048 map points and ~1200 features in two trips each)
                     cap, d_fImg, d_count, d_frameMP, d_outlier, mpCap, d_mpHasObs,
                     d_blocked, d_mpSeen, d_nmatches, d_nmatchesMap);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

}  // extern "C"
