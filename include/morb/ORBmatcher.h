// Drop-in adapter: ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:36-129) over libmorb_hip.so — all 13 public methods.
//
// The reference's methods take Frame / KeyFrame / MapPoint objects; what they READ from them is a handful of arrays.  The adapter
// takes exactly those arrays as plain views (FrameView / KeyFrameView, MapPointView below, each member named after the reference
// member it mirrors), so that inside the reference tree the glue is one function per class that fills a view from the object
// (include/morb/reference_glue.h does it under the reference's locks; INTEGRATION.md shows the call sites), and outside it (this
// repository has no OpenCV / Eigen / Sophus) the header compiles as is.  Results come back as index tables with the meaning of the
// containers the reference methods fill (vpMapPointMatches[i] = map point of keyframe feature table[i], ...); the bookkeeping on live
// map state (AddObservation, Replace, ...) stays with the caller, as DESIGN.md states for the whole matcher family.
// Every method: hipSetDevice(device), uploads into per-thread, per-device staging buffers that only grow, one C-ABI call per
// reference call, synchronised on the handle's own stream (never the device: Tracking's matcher calls do not wait for the LocalBundleAdjustment
// trials the mapping thread has in flight on its optimizer handle).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <set>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../morb_hip.h"
#include "device_buffer.h"

namespace ORB_SLAM3 {

// Frame / KeyFrame members the searches read (include/Frame.h, include/KeyFrame.h).  Unused members stay NULL.
struct FrameView {
  int N = 0;                                    // N (for a fisheye rig: left + right features, left first)
  const morb_keypoint* mvKeysUn = nullptr;      // [N] (cv::KeyPoint has the same 28-byte layout)
  const uint8_t* mDescriptors = nullptr;        // [N][32]
  const float* mvuRight = nullptr;              // [N] or NULL (monocular)
  const uint8_t* hasTrackedMapPoint = nullptr;  // [N] mvpMapPoints[i] && mvpMapPoints[i]->Observations() > 0 (SearchByProjection's "already matched")
  morb_frame_params params{};
  float mRcw[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
  float mtcw[3] = {0, 0, 0};
  float mOw[3] = {0, 0, 0};                     // camera centre = Tcw.inverse().translation()
  float Tcw[7] = {0, 0, 0, 1, 0, 0, 0};         // GetPose() as unit quaternion xyzw + translation
  // BoW: mFeatVec inverted — the vocabulary node (at levelsup) feature i is listed under, -1 if none
  const int* featNode = nullptr;                // [N]
  int nValid = -1;                              // KeyFrame: mvKeysUn.size() where it is smaller than N (fisheye, ORBmatcher.cc:734); -1 = N
  // the map point held by feature i (mvpMapPoints[i] / GetMapPointMatches()[i]):
  const uint8_t* hasMapPoint = nullptr;         // [N] non-NULL && !isBad() (for a last frame: && !mvbOutlier[i])
  const float* mpWorldPos = nullptr;            // [N][3] GetWorldPos()
  const float* mpMaxDistance = nullptr;         // [N] mfMaxDistance (the 1.2 / 0.8 invariance factors are applied inside)
  const float* mpMinDistance = nullptr;         // [N] mfMinDistance
  const uint8_t* mpDescriptor = nullptr;        // [N][32] GetDescriptor()
  const uint8_t* mpHasObservations = nullptr;   // [N] Observations() > 0
};
struct KeyFrameView : FrameView {   // the same members read from a KeyFrame (GetMapPointMatches(), mFeatVec, GetPose(), ...)
  // KannalaBrandt8 rig keyframe (KeyFrame::NLeft != -1): the rows hold mvKeys | mvKeysRight; the loop-closing searches (no rig branch in the reference) then
  // look among the first NLeft features only and project with mpCamera (rigCam8: fx fy cx cy k0..k3) where the reference calls mpCamera->project
  int NLeft = -1;
  float rigCam8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
// MapPoint members read by isInFrustum / SearchByProjection / Fuse (include/MapPoint.h)
struct MapPointView {
  int n = 0;
  const float* worldPos = nullptr;      // [n][3]
  const float* normal = nullptr;        // [n][3] GetNormal()
  const float* maxDistance = nullptr;   // [n]
  const float* minDistance = nullptr;   // [n]
  const uint8_t* descriptor = nullptr;  // [n][32]
  const uint8_t* isBad = nullptr;       // [n]
  const uint8_t* hasObservations = nullptr;  // [n]
  // the state checks at the top of the Sim3 / Fuse loops: non-NULL && !isBad() && !IsInKeyFrame(pKF) (or && !spAlreadyFound.count(pMP))
  const uint8_t* valid = nullptr;       // [n]
};
// Sophus::Sim3f Scw as the searches use it: Tcw = SE3(Scw.rotationMatrix(), Scw.translation() / Scw.scale()), Ow = Tcw.inverse().translation()
struct Sim3View {
  float Tcw[7] = {0, 0, 0, 1, 0, 0, 0};
  float Ow[3] = {0, 0, 0};
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50;      // ORBmatcher.cc:33-35
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;

  ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {
    if (device < 0 || device >= kMaxDevices) throw std::runtime_error("ORBmatcher: bad device");
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

  // Tracking::SearchLocalPoints' pair (Tracking.cc:3117-3183): Frame::isInFrustum(pMP, 0.5) for every candidate — that host loop (IncreaseVisible,
  // mnLastFrameSeen, mmProjectPoints) stays in the integrated tree, see the reference-typed member below — then
  // int SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)  (ORBmatcher.h:45-47,
  // ORBmatcher.cc:42-209).  matchF[i] = index of the map point assigned to feature i, or -1 (in/out: earlier assignments are kept
  // if the vector already has N entries).  Returns the reference's return value.
  int SearchByProjection(const FrameView& F, const MapPointView& mps, std::vector<int>& matchF, float th = 3.f, bool bFarPoints = false,
                         float thFarPoints = 50.f, float viewingCosLimit = 0.5f) {
    const int N = F.N, M = mps.n;
    if (N <= 0 || M <= 0) { matchF.assign(N > 0 ? N : 0, -1); return 0; }
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    s.kp[0].assign(F.mvKeysUn, N);
    s.u8[0].assign(F.mDescriptors, (size_t)N * 32);
    if (F.mvuRight) s.f32[0].assign(F.mvuRight, N);
    s.u8[1].resize(N);
    if (F.hasTrackedMapPoint) s.u8[1].upload(F.hasTrackedMapPoint, N); else s.u8[1].fill_bytes(0);
    const int one = 0, cnt = N, nmp = M;
    s.i32[0].assign(&one, 1); s.i32[1].assign(&cnt, 1); s.i32[2].assign(&nmp, 1); s.i32[3].resize(1); s.i32[3].fill_bytes(0);
    s.f32[1].assign(F.mRcw, 9); s.f32[2].assign(F.mtcw, 3); s.f32[3].assign(F.mOw, 3); s.f32[4].assign(mps.worldPos, (size_t)M * 3);
    s.f32[5].assign(mps.normal, (size_t)M * 3); s.f32[6].assign(mps.maxDistance, M); s.f32[7].assign(mps.minDistance, M);
    for (int k = 8; k <= 12; ++k) s.f32[k].resize(M);   // projX, projY, projXR, depth, viewCos
    s.u8[2].resize(M); s.u8[3].assign(mps.isBad, M); s.u8[4].assign(mps.hasObservations, M); s.u8[5].assign(mps.descriptor, (size_t)M * 32);
    s.i32[4].resize(M); s.i32[5].resize(N);
    if ((int)matchF.size() == N) s.i32[5].upload(matchF.data(), N); else s.i32[5].fill_bytes(0xFF);   // -1
    check(morb_is_in_frustum_batch(h_, &F.params, 1, s.f32[1].get(), s.f32[2].get(), s.f32[3].get(), M, s.i32[2].get(), s.f32[4].get(), s.f32[5].get(),
                                   s.f32[6].get(), s.f32[7].get(), viewingCosLimit, s.u8[2].get(), s.f32[8].get(), s.f32[9].get(), s.f32[10].get(),
                                   s.f32[11].get(), s.i32[4].get(), s.f32[12].get(), nullptr));
    check(morb_search_by_projection_mps_batch(h_, &F.params, 1, s.i32[0].get(), N, s.i32[1].get(), s.kp[0].get(), s.u8[0].get(),
                                              F.mvuRight ? s.f32[0].get() : nullptr, s.u8[1].get(), M, s.i32[2].get(), s.u8[2].get(), s.u8[3].get(),
                                              s.f32[11].get(), s.f32[8].get(), s.f32[9].get(), s.f32[10].get(), s.i32[4].get(), s.f32[12].get(),
                                              s.u8[5].get(), s.u8[4].get(), th, bFarPoints ? 1 : 0, thFarPoints, mfNNratio, s.i32[5].get(),
                                              s.i32[3].get(), nullptr));
    sync();
    matchF = s.i32[5].to_host();
    return s.i32[3].to_host()[0];
  }

  // int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)  (ORBmatcher.h:51-52,
  // ORBmatcher.cc:1521-1733; Tracking::TrackWithMotionModel).  Last.hasMapPoint[i] = LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i];
  // Cur.hasTrackedMapPoint[i2] = CurrentFrame.mvpMapPoints[i2] && Observations() > 0.  matchCur[i2] = index of the LAST-frame feature
  // whose map point is assigned to current feature i2, -1 otherwise (in/out like matchF above).
  int SearchByProjection(const FrameView& Cur, const FrameView& Last, std::vector<int>& matchCur, float th, bool bMono) {
    const int N = Cur.N, NL = Last.N;
    if (N <= 0 || NL <= 0) { matchCur.assign(N > 0 ? N : 0, -1); return 0; }
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&Cur, &Last});
    // :1536-1539: tlc = Tlw * twc; forward / backward motion widens the octave range
    float tlc2 = Last.mtcw[2];
    for (int k = 0; k < 3; ++k) tlc2 += Last.mRcw[6 + k] * Cur.mOw[k];
    const uint8_t fwd = (tlc2 > Cur.params.mb && !bMono) ? 1 : 0, bwd = (-tlc2 > Cur.params.mb && !bMono) ? 1 : 0;
    const int cur = 0, last = 1;
    s.i32[2].assign(&cur, 1); s.i32[3].assign(&last, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0);
    s.u8[2].assign(&fwd, 1); s.u8[3].assign(&bwd, 1);
    s.f32[1].assign(Cur.Tcw, 7);
    up_row(s.u8[4], Cur.hasTrackedMapPoint, N, cap, 1);                  // curBlocked
    up_row(s.u8[5], Last.hasMapPoint, NL, cap, 1);                       // lastValid
    up_row(s.f32[2], Last.mpWorldPos, NL, cap, 3);
    up_row(s.u8[6], Last.mpDescriptor, NL, cap, 32);
    up_row(s.u8[7], Last.mpHasObservations, NL, cap, 1);
    init_match(s.i32[5], matchCur, N, cap);
    check(morb_search_by_projection_last_batch(h_, &Cur.params, 1, s.i32[2].get(), s.i32[3].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(),
                                               Cur.mvuRight ? s.f32[0].get() : nullptr, s.u8[4].get(), s.f32[1].get(), s.u8[5].get(), s.f32[2].get(),
                                               s.u8[6].get(), s.u8[7].get(), th, s.u8[2].get(), s.u8[3].get(), mbCheckOrientation ? 1 : 0,
                                               s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    matchCur = s.i32[5].to_host(); matchCur.resize(N);
    return s.i32[4].to_host()[0];
  }

  // int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist)
  // (ORBmatcher.h:56-58, ORBmatcher.cc:1735-1842; relocalisation).  alreadyFound[i] != 0 <=> sAlreadyFound.count(pKF's map point i);
  // Cur.hasMapPoint[i2] = CurrentFrame.mvpMapPoints[i2] != NULL.  matchCur[i2] = keyframe feature whose map point goes to current feature i2.
  // curCam8 / curNLeft: CurrentFrame is a KannalaBrandt8 rig frame (rows hold left | right features): projection with mpCamera (the left
  // camera: fx fy cx cy k0..k3), search among the left features only — the reference has no rig branch in this member, this is what its
  // code does on such a frame (GetFeaturesInArea's bRight defaults to false; morb_search_by_projection_kf_rig_batch).
  int SearchByProjection(const FrameView& Cur, const KeyFrameView& KF, const std::vector<uint8_t>& alreadyFound, std::vector<int>& matchCur,
                         float th, int ORBdist, const float* curCam8 = nullptr, int curNLeft = -1) {
    const int N = Cur.N, NK = KF.N;
    if (N <= 0 || NK <= 0) { matchCur.assign(N > 0 ? N : 0, -1); return 0; }
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&Cur, &KF});
    std::vector<uint8_t> kfValid(NK);
    for (int i = 0; i < NK; ++i) kfValid[i] = (KF.hasMapPoint && KF.hasMapPoint[i] && !(i < (int)alreadyFound.size() && alreadyFound[i])) ? 1 : 0;
    const int cur = 0, kf = 1;
    s.i32[2].assign(&cur, 1); s.i32[3].assign(&kf, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0);
    s.f32[1].assign(Cur.Tcw, 7); s.f32[2].assign(Cur.mOw, 3);
    up_row(s.u8[2], Cur.hasMapPoint, N, cap, 1);
    up_row(s.u8[3], kfValid.data(), NK, cap, 1);
    up_row(s.f32[3], KF.mpWorldPos, NK, cap, 3); up_row(s.f32[4], KF.mpMaxDistance, NK, cap, 1); up_row(s.f32[5], KF.mpMinDistance, NK, cap, 1);
    up_row(s.u8[4], KF.mpDescriptor, NK, cap, 32);
    init_match(s.i32[5], matchCur, N, cap);
    if (curCam8 && curNLeft >= 0) {
      s.i32[6].assign(&curNLeft, 1);
      check(morb_search_by_projection_kf_rig_batch(h_, &Cur.params, curCam8, 1, s.i32[2].get(), s.i32[3].get(), s.i32[6].get(), cap, s.i32[0].get(),
                                                   s.kp[0].get(), s.u8[0].get(), s.u8[2].get(), s.f32[1].get(), s.f32[2].get(), s.u8[3].get(),
                                                   s.f32[3].get(), s.f32[4].get(), s.f32[5].get(), s.u8[4].get(), th, ORBdist,
                                                   mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    } else
    check(morb_search_by_projection_kf_batch(h_, &Cur.params, 1, s.i32[2].get(), s.i32[3].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(),
                                             s.u8[2].get(), s.f32[1].get(), s.f32[2].get(), s.u8[3].get(), s.f32[3].get(), s.f32[4].get(), s.f32[5].get(),
                                             s.u8[4].get(), th, ORBdist, mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    matchCur = s.i32[5].to_host(); matchCur.resize(N);
    return s.i32[4].to_host()[0];
  }

  // int SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th,
  // float ratioHamming)  (ORBmatcher.h:62-65, ORBmatcher.cc:397-494; loop detection).  vpMatched[idx] = index into vpPoints of the map point
  // matched to keyframe feature idx, -1 = none; entries >= 0 on entry are "already matched" and block their feature (the reference's
  // spAlreadyFound is folded into vpPoints.valid by the caller).  Returns the number of NEW matches, as the reference does.
  int SearchByProjection(const KeyFrameView& KF, const Sim3View& Scw, const MapPointView& vpPoints, std::vector<int>& vpMatched, int th,
                         float ratioHamming = 1.0f) {
    return sim3_projection(KF, Scw, vpPoints, vpMatched, th, ratioHamming, 0);
  }
  // ... the twin with vpPointsKFs / vpMatchedKF (ORBmatcher.h:69-74, ORBmatcher.cc:496-601; place recognition): vpMatchedKF[idx] is
  // vpPointsKFs[vpMatched[idx]] — the keyframe list is parallel to the point list, so the caller reads it through the same index.
  int SearchByProjection(const KeyFrameView& KF, const Sim3View& Scw, const MapPointView& vpPoints, std::vector<int>& vpMatched,
                         std::vector<int>& vpMatchedKF, int th, float ratioHamming = 1.0f) {
    const int n = sim3_projection(KF, Scw, vpPoints, vpMatched, th, ratioHamming, 1);
    vpMatchedKF = vpMatched;
    return n;
  }

  // int SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  (ORBmatcher.h:79, ORBmatcher.cc:218-395; tracking the
  // reference keyframe, relocalisation).  vpMapPointMatches[j] = keyframe feature whose map point is matched to frame feature j, or -1.
  int SearchByBoW(const KeyFrameView& KF, const FrameView& F, std::vector<int>& vpMapPointMatches) {
    if (KF.N <= 0 || F.N <= 0) { vpMapPointMatches.assign(F.N > 0 ? F.N : 0, -1); return 0; }
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&KF, &F});
    const int kf = 0, fr = 1;
    s.i32[2].assign(&kf, 1); s.i32[3].assign(&fr, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap);
    check(morb_search_by_bow_batch(h_, 1, s.i32[2].get(), s.i32[3].get(), 2, s.kp[0].get(), s.u8[0].get(), s.i32[1].get(), s.i32[0].get(), s.u8[1].get(),
                                   cap, mfNNratio, mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    vpMapPointMatches = s.i32[5].to_host(); vpMapPointMatches.resize(F.N);
    return s.i32[4].to_host()[0];
  }
  // int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)  (ORBmatcher.h:80-81, ORBmatcher.cc:702-819; loop closing).
  // vpMatches12[i1] = feature of pKF2 whose map point is matched to feature i1 of pKF1, or -1.
  int SearchByBoW(const KeyFrameView& KF1, const KeyFrameView& KF2, std::vector<int>& vpMatches12) {
    if (KF1.N <= 0 || KF2.N <= 0) { vpMatches12.assign(KF1.N > 0 ? KF1.N : 0, -1); return 0; }
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&KF1, &KF2});
    const int a = 0, b = 1;
    const int nv[2] = {KF1.nValid >= 0 ? KF1.nValid : KF1.N, KF2.nValid >= 0 ? KF2.nValid : KF2.N};
    s.i32[2].assign(&a, 1); s.i32[3].assign(&b, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap); s.i32[6].assign(nv, 2);
    check(morb_search_by_bow_kfkf_batch(h_, 1, s.i32[2].get(), s.i32[3].get(), s.i32[6].get(), 2, s.kp[0].get(), s.u8[0].get(), s.i32[1].get(),
                                        s.i32[0].get(), s.u8[1].get(), cap, mfNNratio, mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    vpMatches12 = s.i32[5].to_host(); vpMatches12.resize(KF1.N);
    return s.i32[4].to_host()[0];
  }

  // int SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize)
  // (ORBmatcher.h:84-87, ORBmatcher.cc:603-700; monocular initialisation).  vbPrevMatched = [N1][2] floats (x, y), updated like the reference.
  int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<float>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10) {
    if (F1.N <= 0 || F2.N <= 0) { vnMatches12.assign(F1.N > 0 ? F1.N : 0, -1); return 0; }
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&F1, &F2});
    const int a = 0, b = 1;
    s.i32[2].assign(&a, 1); s.i32[3].assign(&b, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap);
    std::vector<float> prev((size_t)cap * 2, 0.f);
    std::copy(vbPrevMatched.begin(), vbPrevMatched.begin() + std::min(vbPrevMatched.size(), (size_t)F1.N * 2), prev.begin());
    s.f32[1].assign(prev.data(), prev.size());
    check(morb_search_for_initialization_batch(h_, &F2.params, 1, s.i32[2].get(), s.i32[3].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(),
                                               s.f32[1].get(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    vnMatches12 = s.i32[5].to_host(); vnMatches12.resize(F1.N);
    prev = s.f32[1].to_host(); vbPrevMatched.assign(prev.begin(), prev.begin() + (size_t)F1.N * 2);
    return s.i32[4].to_host()[0];
  }

  // int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<pair<size_t, size_t>>& vMatchedPairs, bool bOnlyStereo, bool bCoarse)
  // (ORBmatcher.h:90-93, ORBmatcher.cc:821-1042; LocalMapping::CreateNewMapPoints), pinhole keyframes.  R12 / t12 = (T1w * Tw2) as rotation
  // matrix (row-major) and translation, ep = pKF2->mpCamera->project(T2w * pKF1->GetCameraCenter()) (:829-838): Sophus expressions the glue
  // evaluates exactly as the reference does.
  int SearchForTriangulation(const KeyFrameView& KF1, const KeyFrameView& KF2, const float R12[9], const float t12[3], const float ep[2],
                             std::vector<std::pair<size_t, size_t>>& vMatchedPairs, bool bOnlyStereo, bool bCoarse = false) {
    vMatchedPairs.clear();
    if (KF1.N <= 0 || KF2.N <= 0) return 0;
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&KF1, &KF2});
    const int a = 0, b = 1;
    s.i32[2].assign(&a, 1); s.i32[3].assign(&b, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap);
    check(morb_search_for_triangulation_batch(h_, &KF1.params, 1, s.i32[2].get(), s.i32[3].get(), 2, cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(),
                                              s.i32[1].get(), s.u8[1].get(), (KF1.mvuRight || KF2.mvuRight) ? s.f32[0].get() : nullptr, R12, t12, ep,
                                              bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0, mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    const std::vector<int> m12 = s.i32[5].to_host();
    for (int i = 0; i < KF1.N; ++i)
      if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);   // :1030-1036: ascending feature of pKF1
    return s.i32[4].to_host()[0];
  }

  // int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th)
  // (ORBmatcher.h:102-104, ORBmatcher.cc:1323-1519; loop closing).  S12 / S21 = S12 and S12.inverse() as 7 floats each: RxSO3 quaternion xyzw
  // whose squared norm is the scale, then the translation.  vpMatches12[i1] = feature of pKF2 (index) matched to feature i1 of pKF1 or -1;
  // entries >= 0 on entry are the reference's vbAlreadyMatched1 / vbAlreadyMatched2 (:1356-1369) and are kept.
  int SearchBySim3(const KeyFrameView& KF1, const KeyFrameView& KF2, std::vector<int>& vpMatches12, const float S12[7], const float S21[7], float th) {
    const int N1 = KF1.N, N2 = KF2.N;
    if (N1 <= 0 || N2 <= 0) return 0;
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&KF1, &KF2});
    if ((int)vpMatches12.size() != N1) vpMatches12.assign(N1, -1);
    std::vector<uint8_t> v1(N1), v2(N2);
    std::vector<uint8_t> already2(N2, 0);
    for (int i = 0; i < N1; ++i) if (vpMatches12[i] >= 0 && vpMatches12[i] < N2) already2[vpMatches12[i]] = 1;
    for (int i = 0; i < N1; ++i) v1[i] = (KF1.hasMapPoint && KF1.hasMapPoint[i] && vpMatches12[i] < 0) ? 1 : 0;
    for (int i = 0; i < N2; ++i) v2[i] = (KF2.hasMapPoint && KF2.hasMapPoint[i] && !already2[i]) ? 1 : 0;
    const int a = 0, b = 1;
    s.i32[2].assign(&a, 1); s.i32[3].assign(&b, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap); s.i32[6].resize(cap); s.i32[7].resize(cap);
    s.f32[1].assign(KF1.Tcw, 7); s.f32[2].assign(KF2.Tcw, 7); s.f32[3].assign(S12, 7); s.f32[4].assign(S21, 7);
    up_row(s.u8[2], v1.data(), N1, cap, 1); up_row(s.f32[5], KF1.mpWorldPos, N1, cap, 3); up_row(s.f32[6], KF1.mpMaxDistance, N1, cap, 1);
    up_row(s.f32[7], KF1.mpMinDistance, N1, cap, 1); up_row(s.u8[3], KF1.mpDescriptor, N1, cap, 32);
    up_row(s.u8[4], v2.data(), N2, cap, 1); up_row(s.f32[8], KF2.mpWorldPos, N2, cap, 3); up_row(s.f32[9], KF2.mpMaxDistance, N2, cap, 1);
    up_row(s.f32[10], KF2.mpMinDistance, N2, cap, 1); up_row(s.u8[5], KF2.mpDescriptor, N2, cap, 32);
    if (KF1.NLeft >= 0 || KF2.NLeft >= 0) {   // rig keyframes: the candidates are the left features (a keyframe without a right camera: all of them)
      const int nl1 = KF1.NLeft >= 0 ? KF1.NLeft : N1, nl2 = KF2.NLeft >= 0 ? KF2.NLeft : N2;
      rigI32(0).assign(&nl1, 1); rigI32(1).assign(&nl2, 1);
      check(morb_search_by_sim3_rig_batch(h_, &KF1.params, 1, s.i32[2].get(), s.i32[3].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(), s.f32[1].get(),
                                          s.f32[2].get(), s.f32[3].get(), s.f32[4].get(), s.u8[2].get(), s.f32[5].get(), s.f32[6].get(), s.f32[7].get(),
                                          s.u8[3].get(), s.u8[4].get(), s.f32[8].get(), s.f32[9].get(), s.f32[10].get(), s.u8[5].get(), th, rigI32(0).get(),
                                          rigI32(1).get(), s.i32[5].get(), s.i32[6].get(), s.i32[7].get(), s.i32[4].get(), nullptr));
    } else
    check(morb_search_by_sim3_batch(h_, &KF1.params, 1, s.i32[2].get(), s.i32[3].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(), s.f32[1].get(),
                                    s.f32[2].get(), s.f32[3].get(), s.f32[4].get(), s.u8[2].get(), s.f32[5].get(), s.f32[6].get(), s.f32[7].get(),
                                    s.u8[3].get(), s.u8[4].get(), s.f32[8].get(), s.f32[9].get(), s.f32[10].get(), s.u8[5].get(), th, s.i32[5].get(),
                                    s.i32[6].get(), s.i32[7].get(), s.i32[4].get(), nullptr));
    sync();
    const std::vector<int> m12 = s.i32[7].to_host();
    for (int i = 0; i < N1; ++i) if (m12[i] >= 0) vpMatches12[i] = m12[i];   // :1506-1515
    return s.i32[4].to_host()[0];
  }

  // int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th, const bool bRight)  (ORBmatcher.h:107-108,
  // ORBmatcher.cc:1044-1213; LocalMapping::SearchInNeighbors) — the search: bestIdx[i] = feature of pKF chosen for map point i
  // (bestDist <= TH_LOW) or -1.  What follows a hit (Replace / AddObservation / AddMapPoint, :1196-1208) depends on live map state and is
  // replayed by the caller over bestIdx in map-point order; the return value here is the number of hits.
  // On a KannalaBrandt8 rig (KeyFrame::NLeft != -1; KF holds mvKeys | mvKeysRight in one row) `side` names the camera searched: its pose and
  // centre (GetPose / GetCameraCenter, or GetRightPose / GetRightCameraCenter with bRight), its KB8 parameters and its feature range
  // ([0, NLeft) or [NLeft, N)); the indices returned count from the start of the row (:1050-1054, :1130-1177).
  struct RigSide { float Tcw[7]; float Ow[3]; float cam8[8]; int jLo, jHi; };
  int Fuse(const KeyFrameView& KF, const MapPointView& vpMapPoints, std::vector<int>& bestIdx, std::vector<int>& bestDist, float th = 3.0f,
           const RigSide* side = nullptr) {
    Sim3View T;
    if (side) { std::memcpy(T.Tcw, side->Tcw, sizeof T.Tcw); std::memcpy(T.Ow, side->Ow, sizeof T.Ow); }
    else { std::memcpy(T.Tcw, KF.Tcw, sizeof T.Tcw); std::memcpy(T.Ow, KF.mOw, sizeof T.Ow); }
    return fuse(KF, T, vpMapPoints, bestIdx, bestDist, th, 0, side);
  }
  // int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint)
  // (ORBmatcher.h:112-114, ORBmatcher.cc:1215-1321; loop closing): vpReplacePoint[i] = pKF->GetMapPoint(bestIdx[i]) where that is non-NULL.
  int Fuse(const KeyFrameView& KF, const Sim3View& Scw, const MapPointView& vpPoints, float th, std::vector<int>& bestIdx, std::vector<int>& bestDist) {
    if (KF.NLeft >= 0) {   // rig keyframe (no rig branch in the reference: pCamera = pKF->mpCamera, the left features)
      RigSide side;
      std::memcpy(side.Tcw, Scw.Tcw, sizeof side.Tcw); std::memcpy(side.Ow, Scw.Ow, sizeof side.Ow); std::memcpy(side.cam8, KF.rigCam8, sizeof side.cam8);
      side.jLo = 0; side.jHi = KF.NLeft;
      return fuse(KF, Scw, vpPoints, bestIdx, bestDist, th, 1, &side);
    }
    return fuse(KF, Scw, vpPoints, bestIdx, bestDist, th, 1);
  }

  morb_matcher* handle() { return h_; }   // for the batched / device-resident entry points

  // ---- the reference's own signatures (include/ORBmatcher.h:41-114) as member templates: a call site of src/Tracking.cc / src/LocalMapping.cc /
  // src/LoopClosing.cc compiles unchanged and instantiates them with the reference's Frame / KeyFrame / MapPoint / Sophus types
  // (definitions: ORBmatcher_reference.h, included below) ----
  template <class Mat, class = typename std::enable_if<!std::is_pointer<Mat>::value>::type>
  static int DescriptorDistance(const Mat& a, const Mat& b);
  template <class FrameT, class MP>
  int SearchByProjection(FrameT& F, const std::vector<MP*>& vpMapPoints, const float th = 3, const bool bFarPoints = false, const float thFarPoints = 50.0f);
  template <class FrameT>
  int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono);
  template <class FrameT, class KF, class MP>
  int SearchByProjection(FrameT& CurrentFrame, KF* pKF, const std::set<MP*>& sAlreadyFound, const float th, const int ORBdist);
  template <class KF, class Sim3, class MP>
  int SearchByProjection(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, std::vector<MP*>& vpMatched, int th, float ratioHamming = 1.0);
  template <class KF, class Sim3, class MP>
  int SearchByProjection(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, const std::vector<KF*>& vpPointsKFs, std::vector<MP*>& vpMatched,
                         std::vector<KF*>& vpMatchedKF, int th, float ratioHamming = 1.0);
  template <class KF, class FrameT, class MP, class = typename std::enable_if<!std::is_pointer<FrameT>::value>::type>
  int SearchByBoW(KF* pKF, FrameT& F, std::vector<MP*>& vpMapPointMatches);
  template <class KF, class MP>
  int SearchByBoW(KF* pKF1, KF* pKF2, std::vector<MP*>& vpMatches12);
  template <class FrameT, class Pt, class = typename std::enable_if<!std::is_base_of<FrameView, FrameT>::value>::type>
  int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<Pt>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
  template <class KF>
  int SearchForTriangulation(KF* pKF1, KF* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse = false);
  template <class KF, class MP, class Sim3>
  int SearchBySim3(KF* pKF1, KF* pKF2, std::vector<MP*>& vpMatches12, const Sim3& S12, const float th);
  template <class KF, class MP>
  int Fuse(KF* pKF, const std::vector<MP*>& vpMapPoints, const float th = 3.0, const bool bRight = false);
  template <class KF, class Sim3, class MP>
  int Fuse(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, float th, std::vector<MP*>& vpReplacePoint);

 protected:
  static constexpr int kMaxDevices = 16;
  // grow-only device staging, one set per host thread and device (the reference constructs a matcher per call; the buffers outlive it)
  struct Staging {
    morb_adapter::DeviceBuffer<morb_keypoint> kp[1];
    morb_adapter::DeviceBuffer<uint8_t> u8[8];
    morb_adapter::DeviceBuffer<float> f32[13];
    morb_adapter::DeviceBuffer<int> i32[8];
    morb_adapter::DeviceBuffer<int> rig[2];   // fisheye forms: mvRightToLeftMatch, mnTrackScaleLevelR
  };
  morb_adapter::DeviceBuffer<int>& rigI32(int k) { return staging().rig[k]; }
  template <class KF, class Sim3, class MP>
  int sim3_projection_ref(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, const std::vector<KF*>* vpPointsKFs, std::vector<MP*>& vpMatched,
                          std::vector<KF*>* vpMatchedKF, int th, float ratioHamming);
  Staging& staging() {
    morb_adapter::hip_check(hipSetDevice(device_), "hipSetDevice");   // the buffers below and the handle's kernels live on device_
    static thread_local Staging per_device[kMaxDevices];
    return per_device[device_];
  }
  static void sync() { morb_adapter::sync_current_stream(); }   // the handle's stream only (a device-wide wait would also wait for LocalMapping's optimizer)
  static void check(int rc) { if (rc < 0) throw std::runtime_error(morb_last_error()); }
  // row r (of capacity cap elements x width) of a pooled device array <- n elements of a host array (NULL = zeros)
  template <typename T>
  static void up_row(morb_adapter::DeviceBuffer<T>& d, const T* host, int n, int cap, int width) {
    std::vector<T> tmp((size_t)cap * width, T());
    if (host) std::copy(host, host + (size_t)n * width, tmp.begin());
    d.assign(tmp.data(), tmp.size());
  }
  static void init_match(morb_adapter::DeviceBuffer<int>& d, const std::vector<int>& m, int N, int cap) {
    std::vector<int> tmp(cap, -1);
    if ((int)m.size() == N) std::copy(m.begin(), m.end(), tmp.begin());
    d.assign(tmp.data(), tmp.size());
  }
  // the pool the batched entry points index: image k = views[k]; kps, descriptors, counts, BoW nodes, hasMapPoint, uRight, [nimg][cap]
  static int load_pool(Staging& s, std::initializer_list<const FrameView*> views) {
    int cap = 1;
    for (const FrameView* v : views) cap = std::max(cap, v->N);
    const int nimg = (int)views.size();
    std::vector<morb_keypoint> kps((size_t)nimg * cap);
    std::vector<uint8_t> desc((size_t)nimg * cap * 32, 0), has((size_t)nimg * cap, 0);
    std::vector<int> node((size_t)nimg * cap, -1), count(nimg);
    std::vector<float> ur((size_t)nimg * cap, -1.f);
    std::memset(static_cast<void*>(kps.data()), 0, kps.size() * sizeof(morb_keypoint));
    int k = 0;
    for (const FrameView* v : views) {
      count[k] = v->N;
      std::copy(v->mvKeysUn, v->mvKeysUn + v->N, kps.begin() + (size_t)k * cap);
      std::copy(v->mDescriptors, v->mDescriptors + (size_t)v->N * 32, desc.begin() + (size_t)k * cap * 32);
      if (v->featNode) std::copy(v->featNode, v->featNode + v->N, node.begin() + (size_t)k * cap);
      if (v->hasMapPoint) std::copy(v->hasMapPoint, v->hasMapPoint + v->N, has.begin() + (size_t)k * cap);
      if (v->mvuRight) std::copy(v->mvuRight, v->mvuRight + v->N, ur.begin() + (size_t)k * cap);
      ++k;
    }
    s.kp[0].assign(kps.data(), kps.size()); s.u8[0].assign(desc.data(), desc.size()); s.u8[1].assign(has.data(), has.size());
    s.i32[0].assign(count.data(), count.size()); s.i32[1].assign(node.data(), node.size()); s.f32[0].assign(ur.data(), ur.size());
    return cap;
  }
  void load_points(Staging& s, const MapPointView& P) {   // u8[2] valid, f32[3] Pw, [4] normal, [5] maxDist, [6] minDist, u8[3] descriptors
    const int M = P.n;
    s.u8[2].assign(P.valid, M); s.f32[3].assign(P.worldPos, (size_t)M * 3); s.f32[4].assign(P.normal, (size_t)M * 3);
    s.f32[5].assign(P.maxDistance, M); s.f32[6].assign(P.minDistance, M); s.u8[3].assign(P.descriptor, (size_t)M * 32);
  }
  int sim3_projection(const KeyFrameView& KF, const Sim3View& Scw, const MapPointView& P, std::vector<int>& vpMatched, int th, float ratioHamming,
                      int manual) {
    const int N = KF.N, M = P.n;
    if ((int)vpMatched.size() != N) vpMatched.assign(N > 0 ? N : 0, -1);
    if (N <= 0 || M <= 0) return 0;
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&KF});
    load_points(s, P);
    std::vector<uint8_t> matched(cap, 0);
    for (int i = 0; i < N; ++i) matched[i] = vpMatched[i] >= 0 ? 1 : 0;
    const int kf = 0, nmp = M;
    s.i32[2].assign(&kf, 1); s.i32[3].assign(&nmp, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap);
    s.f32[1].assign(Scw.Tcw, 7); s.f32[2].assign(Scw.Ow, 3); s.u8[4].assign(matched.data(), cap);
    if (KF.NLeft >= 0) {   // rig keyframe: left features, the left KB8 camera where the reference calls mpCamera->project
      rigI32(0).assign(&KF.NLeft, 1);
      check(morb_search_by_projection_sim3_rig_batch(h_, &KF.params, 1, s.i32[2].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(), s.f32[1].get(),
                                                     s.f32[2].get(), M, s.i32[3].get(), s.u8[2].get(), s.f32[3].get(), s.f32[4].get(), s.f32[5].get(),
                                                     s.f32[6].get(), s.u8[3].get(), s.u8[4].get(), th, ratioHamming, manual, KF.rigCam8, rigI32(0).get(),
                                                     s.i32[5].get(), s.i32[4].get(), nullptr));
    } else
    check(morb_search_by_projection_sim3_batch(h_, &KF.params, 1, s.i32[2].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(), s.f32[1].get(),
                                               s.f32[2].get(), M, s.i32[3].get(), s.u8[2].get(), s.f32[3].get(), s.f32[4].get(), s.f32[5].get(),
                                               s.f32[6].get(), s.u8[3].get(), s.u8[4].get(), th, ratioHamming, manual, s.i32[5].get(), s.i32[4].get(),
                                               nullptr));
    sync();
    const std::vector<int> mf = s.i32[5].to_host();
    for (int i = 0; i < N; ++i) if (mf[i] >= 0) vpMatched[i] = mf[i];
    return s.i32[4].to_host()[0];
  }
  int fuse(const KeyFrameView& KF, const Sim3View& T, const MapPointView& P, std::vector<int>& bestIdx, std::vector<int>& bestDist, float th, int sim3Form,
           const RigSide* side = nullptr) {
    const int N = KF.N, M = P.n;
    bestIdx.assign(M > 0 ? M : 0, -1); bestDist.assign(M > 0 ? M : 0, 256);
    if (N <= 0 || M <= 0) return 0;
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));   // uploads, kernels, downloads: the handle's stream, never the null stream
    Staging& s = staging();
    const int cap = load_pool(s, {&KF});
    load_points(s, P);
    const int kf = 0, nmp = M;
    s.i32[2].assign(&kf, 1); s.i32[3].assign(&nmp, 1); s.i32[5].resize(M); s.i32[6].resize(M);
    s.f32[1].assign(T.Tcw, 7); s.f32[2].assign(T.Ow, 3);
    if (side) { s.i32[4].assign(&side->jLo, 1); s.i32[7].assign(&side->jHi, 1); }
    check(morb_fuse_batch(h_, &KF.params, 1, s.i32[2].get(), cap, s.i32[0].get(), s.kp[0].get(), s.u8[0].get(), (KF.mvuRight && !side) ? s.f32[0].get() : nullptr,
                          s.f32[1].get(), s.f32[2].get(), side ? side->cam8 : nullptr, side ? s.i32[4].get() : nullptr, side ? s.i32[7].get() : nullptr, M,
                          s.i32[3].get(), s.u8[2].get(), s.f32[3].get(), s.f32[4].get(),
                          s.f32[5].get(), s.f32[6].get(), s.u8[3].get(), th, sim3Form, s.i32[5].get(), s.i32[6].get(), nullptr));
    sync();
    bestIdx = s.i32[5].to_host(); bestDist = s.i32[6].to_host();
    int hits = 0;
    for (int i = 0; i < M; ++i) hits += bestIdx[i] >= 0 ? 1 : 0;
    return hits;
  }

  float mfNNratio;
  bool mbCheckOrientation;
  int device_ = 0;
  morb_matcher* h_ = nullptr;
};

}  // namespace ORB_SLAM3

#include "ORBmatcher_reference.h"
