// Drop-in adapter: ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:36-129) over libmorb_hip.so.
//
// The reference's methods take Frame / KeyFrame / MapPoint objects; what they READ from them is a handful of arrays.  The adapter
// takes exactly those arrays as plain views (FrameView, MapPointView below, each member named after the reference member it
// mirrors), so that inside the reference tree the glue is one function per class that fills a view from the object (INTEGRATION.md
// shows it), and outside it (this repository has no OpenCV / Eigen / Sophus) the header compiles as is.  Results come back in the
// containers the reference methods fill (match index vectors), the bookkeeping on live map state (AddObservation, Replace, ...)
// stays with the caller, as DESIGN.md states for the whole matcher family.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../morb_hip.h"
#include "device_buffer.h"

namespace ORB_SLAM3 {

// Frame members the projection-guided searches read (include/Frame.h): N, mvKeysUn, mDescriptors, mvuRight, the grid / camera /
// pyramid constants (morb_frame_params), the pose (mRcw, mtcw, mOw) and "feature i already holds a tracked map point"
// (mvpMapPoints[i] && Observations() > 0).
struct FrameView {
  int N = 0;
  const morb_keypoint* mvKeysUn = nullptr;      // [N] (cv::KeyPoint has the same 28-byte layout)
  const uint8_t* mDescriptors = nullptr;        // [N][32]
  const float* mvuRight = nullptr;              // [N] or nullptr (monocular)
  const uint8_t* hasTrackedMapPoint = nullptr;  // [N] or nullptr
  morb_frame_params params{};
  float mRcw[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
  float mtcw[3] = {0, 0, 0};
  float mOw[3] = {0, 0, 0};
};
// MapPoint members read by isInFrustum / SearchByProjection (include/MapPoint.h): GetWorldPos, GetNormal, mfMaxDistance,
// mfMinDistance, GetDescriptor, isBad, Observations() > 0.
struct MapPointView {
  int n = 0;
  const float* worldPos = nullptr;      // [n][3]
  const float* normal = nullptr;        // [n][3]
  const float* maxDistance = nullptr;   // [n]
  const float* minDistance = nullptr;   // [n]
  const uint8_t* descriptor = nullptr;  // [n][32]
  const uint8_t* isBad = nullptr;       // [n]
  const uint8_t* hasObservations = nullptr;  // [n]
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50;      // ORBmatcher.cc:33-35
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;

  ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
    if (morb_matcher_create(&h_, device) != MORB_OK) throw std::runtime_error(std::string("morb_matcher_create: ") + morb_last_error());
  }
  ~ORBmatcher() { morb_matcher_destroy(h_); }
  ORBmatcher(const ORBmatcher&) = delete;
  ORBmatcher& operator=(const ORBmatcher&) = delete;

  // static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b)  (ORBmatcher.cc:1880-1894): the bit-parallel popcount of the
  // reference on two 32-byte rows.  A host helper of the class surface (callers use it on single pairs); the matchers themselves
  // compute their distances on the device.
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
      uint32_t pa, pb;
      std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
      dist += __builtin_popcount(pa ^ pb);
    }
    return dist;
  }

  // Tracking::SearchLocalPoints' pair (Tracking.cc:3222-3285): Frame::isInFrustum(pMP, 0.5) for every candidate, then
  // int SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)  (ORBmatcher.h:49-51,
  // ORBmatcher.cc:42-209).  matchF[i] = index of the map point assigned to feature i, or -1 (in/out: earlier assignments are kept
  // if the vector already has N entries).  Returns the reference's return value.
  int SearchByProjection(const FrameView& F, const MapPointView& mps, std::vector<int>& matchF, float th = 3.f, bool bFarPoints = false,
                         float thFarPoints = 50.f, float viewingCosLimit = 0.5f) {
    using morb_adapter::DeviceBuffer;
    const int N = F.N, M = mps.n;
    if (N <= 0 || M <= 0) { matchF.assign(N > 0 ? N : 0, -1); return 0; }
    // per-thread staging buffers that only grow (Tracking calls this every frame: ~25 hipMalloc / hipFree pairs per call otherwise)
    static thread_local DeviceBuffer<morb_keypoint> kps;
    static thread_local DeviceBuffer<uint8_t> desc, blocked, inView, isBad, hasObs, mpDesc;
    static thread_local DeviceBuffer<float> uR, R, t, Ow, Pw, nrm, maxD, minD, projX, projY, projXR, depth, viewCos;
    static thread_local DeviceBuffer<int> fImg, count, nMP, nmatch, level, match;
    kps.assign(F.mvKeysUn, N);
    desc.assign(F.mDescriptors, (size_t)N * 32);
    if (F.mvuRight) uR.assign(F.mvuRight, N);
    blocked.resize(N);
    if (F.hasTrackedMapPoint) blocked.upload(F.hasTrackedMapPoint, N); else blocked.fill_bytes(0);
    const int one = 0, cnt = N, nmp = M;
    fImg.assign(&one, 1); count.assign(&cnt, 1); nMP.assign(&nmp, 1); nmatch.resize(1);
    R.assign(F.mRcw, 9); t.assign(F.mtcw, 3); Ow.assign(F.mOw, 3); Pw.assign(mps.worldPos, (size_t)M * 3); nrm.assign(mps.normal, (size_t)M * 3);
    maxD.assign(mps.maxDistance, M); minD.assign(mps.minDistance, M);
    projX.resize(M); projY.resize(M); projXR.resize(M); depth.resize(M); viewCos.resize(M);
    inView.resize(M); isBad.assign(mps.isBad, M); hasObs.assign(mps.hasObservations, M); mpDesc.assign(mps.descriptor, (size_t)M * 32);
    level.resize(M); match.resize(N);
    if ((int)matchF.size() == N) match.upload(matchF.data(), N); else match.fill_bytes(0xFF);   // -1
    check(morb_is_in_frustum_batch(h_, &F.params, 1, R.get(), t.get(), Ow.get(), M, nMP.get(), Pw.get(), nrm.get(), maxD.get(), minD.get(),
                                   viewingCosLimit, inView.get(), projX.get(), projY.get(), projXR.get(), depth.get(), level.get(),
                                   viewCos.get(), nullptr));
    check(morb_search_by_projection_mps_batch(h_, &F.params, 1, fImg.get(), N, count.get(), kps.get(), desc.get(), F.mvuRight ? uR.get() : nullptr,
                                              blocked.get(), M, nMP.get(), inView.get(), isBad.get(), depth.get(), projX.get(), projY.get(),
                                              projXR.get(), level.get(), viewCos.get(), mpDesc.get(), hasObs.get(), th, bFarPoints ? 1 : 0,
                                              thFarPoints, mfNNratio, match.get(), nmatch.get(), nullptr));
    morb_adapter::hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    matchF = match.to_host();
    return nmatch.to_host()[0];
  }

  morb_matcher* handle() { return h_; }   // for the batched / device-resident entry points (SearchByBoW, SearchForTriangulation, ...)

 protected:
  static void check(int rc) { if (rc < 0) throw std::runtime_error(morb_last_error()); }
  float mfNNratio;
  bool mbCheckOrientation;
  morb_matcher* h_ = nullptr;
};

}  // namespace ORB_SLAM3
