// Gathering helpers behind the reference-typed members of ORB_SLAM3::ORBmatcher / ORB_SLAM3::Optimizer (ORBmatcher_reference.h,
// Optimizer_reference.h): they fill the adapters' views (FrameView, KeyFrameView, MapPointView, ...) from the reference's own
// Frame / KeyFrame / MapPoint objects, through the getters — i.e. under the locks — the reference's methods use.
//
// Everything here is a template on the reference's types: the header names no OpenCV / Eigen / Sophus / DBoW2 type and includes no
// reference header, so it is part of ORBmatcher.h / Optimizer.h everywhere and is instantiated only in a translation unit that calls
// a member with the reference's objects (src/Tracking.cc, src/LocalMapping.cc, ...), where those types are complete.
//
// THIS FILE HAS NEVER BEEN COMPILED AGAINST THE REFERENCE: the build container has no OpenCV / Eigen.  What is checked here
// (tests/test_oracle_cpu.py::test_reference_call_sites_compile_unchanged) is that the call expressions of src/Tracking.cc and
// src/LocalMapping.cc, pasted verbatim, compile against mock declarations of the members touched (tests/native/mock_ref: names and types
// read off the reference headers), and (tests/test_adapter_gpu.py, on the GPU box) that the gather / write-back logic, driven with
// mock objects that carry data, reproduces what the view-taking adapters and the oracle give.
#pragma once
#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../morb_hip.h"

namespace ORB_SLAM3 {
struct FrameView;
struct KeyFrameView;
struct MapPointView;
struct Sim3View;

namespace morb_glue {

// MapPoint::mfMaxDistance / mfMinDistance are protected (include/MapPoint.h:242-243); the public getters return them times 1.2f / 0.8f
// (MapPoint.cc:526-534), which cannot be undone exactly, and PredictScale (MapPoint.cc:536-570) needs the raw value.  A derived class may
// form a pointer to a protected member of its base: the raw values are read, under mMutexPos like the getters, without touching the header.
template <class MP>
struct MapPointAccess : MP {
  static void distances(MP* p, float& maxD, float& minD) {
    std::unique_lock<std::mutex> lock(p->*(&MapPointAccess::mMutexPos));
    maxD = p->*(&MapPointAccess::mfMaxDistance);
    minD = p->*(&MapPointAccess::mfMinDistance);
  }
};

template <class SE3>
inline void pose7(const SE3& T, float out[7]) {   // unit quaternion xyzw + translation
  const auto q = T.unit_quaternion();
  const auto t = T.translation();
  out[0] = q.x(); out[1] = q.y(); out[2] = q.z(); out[3] = q.w();
  for (int k = 0; k < 3; ++k) out[4 + k] = t(k);
}
template <class SE3>
inline SE3 make_se3(const float p[7]) {   // Sophus::SE3f(Eigen::Quaternionf(w, x, y, z), Eigen::Vector3f(...))
  return SE3(typename SE3::QuaternionType(p[3], p[0], p[1], p[2]), typename SE3::Point(p[4], p[5], p[6]));
}
template <class View, class SE3>
inline void fill_pose(View& v, const SE3& Tcw) {
  const auto R = Tcw.rotationMatrix();
  const auto t = Tcw.translation();
  const auto Ow = Tcw.inverse().translation();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) v.mRcw[3 * r + c] = R(r, c);
  for (int k = 0; k < 3; ++k) { v.mtcw[k] = t(k); v.mOw[k] = Ow(k); }
  pose7(Tcw, v.Tcw);
}
template <class F>   // Frame or KeyFrame: the members have the same names
inline void fill_params(morb_frame_params& P, const F& f, float gridInvW, float gridInvH) {
  P.minX = (float)f.mnMinX; P.minY = (float)f.mnMinY; P.maxX = (float)f.mnMaxX; P.maxY = (float)f.mnMaxY;
  P.gridInvW = gridInvW; P.gridInvH = gridInvH;
  P.fx = f.fx; P.fy = f.fy; P.cx = f.cx; P.cy = f.cy; P.mbf = f.mbf; P.mb = f.mb; P.logScaleFactor = f.mfLogScaleFactor;
  P.nlevels = f.mnScaleLevels;
  for (int l = 0; l < f.mnScaleLevels && l < 16; ++l) { P.scaleFactors[l] = f.mvScaleFactors[l]; P.levelSigma2[l] = f.mvLevelSigma2[l]; }
}
template <class FeatVec>
inline void node_table(const FeatVec& fv, int N, std::vector<int>& node) {   // mFeatVec: node -> feature indices
  node.assign(N, -1);
  for (const auto& kv : fv) for (unsigned int i : kv.second) if ((int)i < N) node[i] = (int)kv.first;
}
template <class KP>
inline morb_keypoint to_kp(const KP& k) {   // cv::KeyPoint -> the 28-byte record (same member order; copied field by field, no layout assumption)
  morb_keypoint o;
  o.x = k.pt.x; o.y = k.pt.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id;
  return o;
}

// owns the arrays a FrameView / KeyFrameView points at
template <class View>
struct Store {
  View v;
  std::vector<morb_keypoint> kps;
  std::vector<uint8_t> desc, tracked, hasMP, mpDesc, mpObs;
  std::vector<float> ur, mpPos, mpMax, mpMin;
  std::vector<int> node;
};
enum PointRule { kSkipBad = 0, kLastFrame = 1, kAny = 2 };
// the map point of every feature, read under the locks the getters take.  hasMapPoint: kSkipBad = non-NULL && !isBad() (keyframes, BoW);
// kLastFrame = non-NULL && !mvbOutlier[i] (ORBmatcher.cc:1540-1542 tests nothing else); kAny = non-NULL
template <class View, class MP>
inline void fill_points(Store<View>& s, const std::vector<MP*>& mps, PointRule rule, const std::vector<bool>* outlier) {
  const int N = s.v.N;
  s.hasMP.assign(N, 0); s.tracked.assign(N, 0); s.mpObs.assign(N, 0); s.mpDesc.assign((size_t)N * 32, 0);
  s.mpPos.assign((size_t)N * 3, 0.f); s.mpMax.assign(N, 1.f); s.mpMin.assign(N, 1.f);
  for (int i = 0; i < N && i < (int)mps.size(); ++i) {
    MP* p = mps[i];
    if (!p) continue;
    const bool obs = p->Observations() > 0;
    s.tracked[i] = obs ? 1 : 0; s.mpObs[i] = obs ? 1 : 0;
    if (rule == kSkipBad && p->isBad()) continue;
    if (rule == kLastFrame && outlier && (*outlier)[i]) continue;
    s.hasMP[i] = 1;
    const auto X = p->GetWorldPos();
    for (int k = 0; k < 3; ++k) s.mpPos[3 * i + k] = X(k);
    MapPointAccess<MP>::distances(p, s.mpMax[i], s.mpMin[i]);
    const auto d = p->GetDescriptor();
    std::memcpy(&s.mpDesc[(size_t)i * 32], d.template ptr<uint8_t>(0), 32);
  }
  s.v.hasTrackedMapPoint = s.tracked.data(); s.v.hasMapPoint = s.hasMP.data(); s.v.mpWorldPos = s.mpPos.data();
  s.v.mpMaxDistance = s.mpMax.data(); s.v.mpMinDistance = s.mpMin.data(); s.v.mpDescriptor = s.mpDesc.data();
  s.v.mpHasObservations = s.mpObs.data();
}
// keypoints of a frame: mvKeysUn, or on a fisheye rig mvKeys | mvKeysRight in one row (ORBmatcher.cc:109-112)
template <class View, class F>
inline void fill_keys(Store<View>& s, const F& f, int N, int nLeft) {
  s.kps.resize(N);
  if (nLeft == -1) { for (int i = 0; i < N && i < (int)f.mvKeysUn.size(); ++i) s.kps[i] = to_kp(f.mvKeysUn[i]); }
  else for (int i = 0; i < N; ++i) s.kps[i] = to_kp(i < nLeft ? f.mvKeys[i] : f.mvKeysRight[i - nLeft]);
  s.desc.assign((size_t)N * 32, 0);
  for (int i = 0; i < N; ++i) std::memcpy(&s.desc[(size_t)i * 32], f.mDescriptors.template ptr<uint8_t>(i), 32);
  s.v.N = N; s.v.mvKeysUn = s.kps.data(); s.v.mDescriptors = s.desc.data();
  s.ur.assign(f.mvuRight.begin(), f.mvuRight.end());
  s.v.mvuRight = (int)s.ur.size() >= N && N > 0 ? s.ur.data() : nullptr;
}
template <class View, class F>
inline void frame_view(Store<View>& s, F& f, PointRule rule = kSkipBad) {
  fill_keys(s, f, f.N, f.Nleft);
  fill_params(s.v.params, f, F::mfGridElementWidthInv, F::mfGridElementHeightInv);
  if (f.HasPose()) fill_pose(s.v, f.GetPose());
  node_table(f.mFeatVec, f.N, s.node); s.v.featNode = s.node.data();
  fill_points(s, f.mvpMapPoints, rule, rule == kLastFrame ? &f.mvbOutlier : nullptr);
}
template <class Cam> inline void cam8(Cam* c, float out[8]);
// KeyFrameView::NLeft / rigCam8 of a KannalaBrandt8 rig keyframe (a FrameView has no such members: nothing to do)
template <class View, class KF>
inline auto rig_of(View& v, KF* pKF) -> decltype(v.NLeft, void()) {
  v.NLeft = pKF->NLeft;
  if (pKF->NLeft != -1 && pKF->mpCamera) cam8(pKF->mpCamera, v.rigCam8);
}
inline void rig_of(...) {}
template <class View, class KF>
inline void keyframe_view(Store<View>& s, KF* pKF, PointRule rule = kSkipBad) {
  fill_keys(s, *pKF, pKF->N, pKF->NLeft);
  s.v.nValid = (int)pKF->mvKeysUn.size();
  rig_of(s.v, pKF);
  fill_params(s.v.params, *pKF, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv);
  fill_pose(s.v, pKF->GetPose());
  node_table(pKF->mFeatVec, pKF->N, s.node); s.v.featNode = s.node.data();
  fill_points(s, pKF->GetMapPointMatches(), rule, nullptr);
}
template <class View>
struct PointStore {
  View v;
  std::vector<float> pos, nrm, maxD, minD;
  std::vector<uint8_t> desc, bad, obs, valid;
};
template <class View, class MP>
inline void mappoint_view(PointStore<View>& s, const std::vector<MP*>& mps) {
  const int n = (int)mps.size();
  s.pos.assign((size_t)n * 3, 0.f); s.nrm.assign((size_t)n * 3, 0.f); s.maxD.assign(n, 1.f); s.minD.assign(n, 1.f);
  s.desc.assign((size_t)n * 32, 0); s.bad.assign(n, 1); s.obs.assign(n, 0); s.valid.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    MP* p = mps[i];
    if (!p) continue;
    s.bad[i] = p->isBad() ? 1 : 0; s.obs[i] = p->Observations() > 0 ? 1 : 0; s.valid[i] = s.bad[i] ? 0 : 1;
    const auto X = p->GetWorldPos();
    const auto nv = p->GetNormal();
    for (int k = 0; k < 3; ++k) { s.pos[3 * i + k] = X(k); s.nrm[3 * i + k] = nv(k); }
    MapPointAccess<MP>::distances(p, s.maxD[i], s.minD[i]);
    const auto d = p->GetDescriptor();
    std::memcpy(&s.desc[(size_t)i * 32], d.template ptr<uint8_t>(0), 32);
  }
  s.v.n = n; s.v.worldPos = s.pos.data(); s.v.normal = s.nrm.data(); s.v.maxDistance = s.maxD.data(); s.v.minDistance = s.minD.data();
  s.v.descriptor = s.desc.data(); s.v.isBad = s.bad.data(); s.v.hasObservations = s.obs.data(); s.v.valid = s.valid.data();
}
// Sophus::Sim3f Scw -> Tcw = SE3(Scw.rotationMatrix(), Scw.translation() / Scw.scale()), Ow (ORBmatcher.cc:407-409, :1228-1230), evaluated with
// the reference's own expression; SE3 = the type pKF->GetPose() returns
template <class SE3, class Sim3, class SV>
inline void sim3_view(SV& v, const Sim3& Scw) {
  const SE3 Tcw(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
  pose7(Tcw, v.Tcw);
  const auto Ow = Tcw.inverse().translation();
  for (int k = 0; k < 3; ++k) v.Ow[k] = Ow(k);
}
// Sophus::Sim3f as the 7 floats SearchBySim3 takes: RxSO3 quaternion xyzw (squared norm = scale), then the translation
template <class Sim3>
inline void sim3_raw(const Sim3& S, float out[7]) {
  const auto q = S.rxso3().quaternion();
  const auto t = S.translation();
  out[0] = q.x(); out[1] = q.y(); out[2] = q.z(); out[3] = q.w();
  for (int k = 0; k < 3; ++k) out[4 + k] = t(k);
}
// rotation matrix (row-major) then translation of an SE3: the 12 floats the *_fisheye entry points take per relative pose
template <class SE3>
inline void rt12(const SE3& T, float o[12]) {
  const auto R = T.rotationMatrix();
  const auto t = T.translation();
  for (int r = 0; r < 3; ++r) { o[9 + r] = t(r); for (int c = 0; c < 3; ++c) o[3 * r + c] = R(r, c); }
}
// GeometricCamera (Pinhole / KannalaBrandt8): fx fy cx cy [k0 k1 k2 k3] through getParameter(i) (CameraModels/GeometricCamera.h:95)
template <class Cam>
inline void cam8(Cam* c, float out[8]) {
  for (int i = 0; i < 8; ++i) out[i] = i < (int)c->size() ? c->getParameter(i) : 0.f;
}

}  // namespace morb_glue
}  // namespace ORB_SLAM3
