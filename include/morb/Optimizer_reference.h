// The reference's own Optimizer signatures for the hot path (include/Optimizer.h:67-101) as static member templates of
// ORB_SLAM3::Optimizer — included at the end of Optimizer.h.  Templated on Frame / KeyFrame / Map (and through them MapPoint, Sophus,
// IMU::Preintegrated, ConstraintPoseImu) so this header names none of the reference's or its third parties' types: the call sites
//     Optimizer::PoseOptimization(&mCurrentFrame);                                                    Tracking.cc:2559, :2710, :3020, :3026
//     Optimizer::PoseInertialOptimizationLastFrame(&mCurrentFrame) / ...LastKeyFrame(&mCurrentFrame)  Tracking.cc:3032-3041
//     Optimizer::LocalBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpCurrentKeyFrame->GetMap(), a, b, c, d);              LocalMapping.cc:181
//     Optimizer::LocalInertialBA(mpCurrentKeyFrame, &mbAbortBA, mpCurrentKeyFrame->GetMap(), a, b, c, d, bLarge, !...GetIniertialBA2());  :173
// compile unchanged.  Each member assembles the graph exactly as the reference's method does (same selection loops, same flags written to
// mnBALocalForKF / mnBAFixedForKF, same locks), hands it flattened to the C ABI, and writes the result back as the reference does.
// KannalaBrandt8 rigs (mpCamera2 != NULL) dispatch to the *_fisheye entry points.  See reference_glue.h for what has been compiled.
#pragma once
#include <cmath>
#include <list>
#include <map>
#include <tuple>

#include "reference_glue.h"

namespace ORB_SLAM3 {
namespace morb_glue {

// IMU::Preintegrated (include/ImuTypes.h:154-263) -> the plain record of morb_hip.h
template <class Pre>
inline void fill_pre(morb_imu_preintegrated& o, const Pre* p) {
  auto m3 = [](float* d, const auto& M) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) d[3 * r + c] = M(r, c); };
  auto v3 = [](float* d, const auto& V) { for (int k = 0; k < 3; ++k) d[k] = V(k); };
  o.dT = p->dT;
  m3(o.dR, p->dR); v3(o.dV, p->dV); v3(o.dP, p->dP);
  m3(o.JRg, p->JRg); m3(o.JVg, p->JVg); m3(o.JVa, p->JVa); m3(o.JPg, p->JPg); m3(o.JPa, p->JPa);
  for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) o.C[15 * r + c] = p->C(r, c);
  o.b[0] = p->b.bax; o.b[1] = p->b.bay; o.b[2] = p->b.baz; o.b[3] = p->b.bwx; o.b[4] = p->b.bwy; o.b[5] = p->b.bwz;
  for (int k = 0; k < 6; ++k) { o.nga[k] = p->Nga.diagonal()(k); o.ngaWalk[k] = p->NgaWalk.diagonal()(k); }
  v3(o.avgA, p->avgA); v3(o.avgW, p->avgW);
}
// Rwb (row-major), twb, velocity, gyro bias, acc bias of a Frame / KeyFrame (VertexPose / VertexVelocity / VertexGyroBias / VertexAccBias
// constructors, G2oTypes.h:128-199)
template <class F>
inline void imu_state21(F* f, float s[21]) {
  const auto R = f->GetImuRotation();
  const auto t = f->GetImuPosition();
  const auto v = f->GetVelocity();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) s[3 * r + c] = R(r, c);
  for (int k = 0; k < 3; ++k) { s[9 + k] = t(k); s[12 + k] = v(k); }
}
template <class B> inline void bias6(const B& b, float* s15) { s15[0] = b.bwx; s15[1] = b.bwy; s15[2] = b.bwz; s15[3] = b.bax; s15[4] = b.bay; s15[5] = b.baz; }
template <class SE3> inline void tbc12(const SE3& Tbc, float o[12]) {
  const auto R = Tbc.rotationMatrix(); const auto t = Tbc.translation();
  for (int r = 0; r < 3; ++r) { o[9 + r] = t(r); for (int c = 0; c < 3; ++c) o[3 * r + c] = R(r, c); }
}
// rig28 = left KB8 parameters (8), right ones (8), rotation (9, row-major) and translation (3) of GetRelativePoseTrl()
template <class F> inline void rig28(F* f, float o[28]) {
  cam8(f->mpCamera, o); cam8(f->mpCamera2, o + 8);
  tbc12(f->GetRelativePoseTrl(), o + 16);
}
// ConstraintPoseImu (include/G2oTypes.h:706-729) <-> the 246 doubles of morb_hip.h: Rwb (9, row-major), twb, vwb, bg, ba, H (15 x 15, row-major)
template <class CPI> inline void prior_to_doubles(const CPI* c, double* o) {
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) o[3 * r + k] = c->Rwb(r, k);
  for (int k = 0; k < 3; ++k) { o[9 + k] = c->twb(k); o[12 + k] = c->vwb(k); o[15 + k] = c->bg(k); o[18 + k] = c->ba(k); }
  for (int r = 0; r < 15; ++r) for (int k = 0; k < 15; ++k) o[21 + 15 * r + k] = c->H(r, k);
}
template <class CPI> inline CPI* prior_from_doubles(const double* o) {
  typename std::decay<decltype(CPI::Rwb)>::type R;
  typename std::decay<decltype(CPI::twb)>::type t, v, bg, ba;
  typename std::decay<decltype(CPI::H)>::type H;
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) R(r, k) = o[3 * r + k];
  for (int k = 0; k < 3; ++k) { t(k) = o[9 + k]; v(k) = o[12 + k]; bg(k) = o[15 + k]; ba(k) = o[18 + k]; }
  for (int r = 0; r < 15; ++r) for (int k = 0; k < 15; ++k) H(r, k) = o[21 + 15 * r + k];
  return new CPI(R, t, v, bg, ba, H);   // (the constructor's eigenvalue clamp runs again on the already clamped matrix)
}

// the visual edges of one frame as the pose optimisers build them (Optimizer.cc:806-946, :4444-4528): feature i with a map point ->
// (x, y, uRight or -1), 1 / sigma^2 of its octave, the point's world position; on a rig features [0, Nleft) are left-camera observations
// (mvKeys), the rest right-camera ones (mvKeysRight), all monocular
template <class FrameT>
struct FrameEdges {
  int N = 0, nLeft = -1;
  std::vector<uint8_t> has, close, outlier;
  std::vector<float> obs, inv, Xw;
  explicit FrameEdges(FrameT* pFrame) {
    using MP = typename std::remove_pointer<typename std::decay<decltype(pFrame->mvpMapPoints[0])>::type>::type;
    N = pFrame->N; nLeft = pFrame->Nleft;
    has.assign(N, 0); close.assign(N, 0); obs.assign((size_t)N * 3, 0.f); inv.assign(N, 0.f); Xw.assign((size_t)N * 3, 0.f);
    outlier.assign(pFrame->mvbOutlier.begin(), pFrame->mvbOutlier.end());
    outlier.resize(N, 0);
    std::unique_lock<std::mutex> lock(MP::mGlobalMutex);   // :806, :4445
    for (int i = 0; i < N; ++i) {
      MP* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      has[i] = 1; outlier[i] = 0;                          // :817, :860: pFrame->mvbOutlier[i] = false
      const auto& kp = nLeft == -1 ? pFrame->mvKeysUn[i] : (i < nLeft ? pFrame->mvKeys[i] : pFrame->mvKeysRight[i - nLeft]);
      obs[3 * i] = kp.pt.x; obs[3 * i + 1] = kp.pt.y; obs[3 * i + 2] = nLeft == -1 ? pFrame->mvuRight[i] : -1.f;   // < 0: monocular edge (:811)
      inv[i] = pFrame->mvInvLevelSigma2[kp.octave];
      close[i] = pMP->mTrackDepth < 10.f ? 1 : 0;          // :4576 bClose
      const auto X = pMP->GetWorldPos();
      for (int k = 0; k < 3; ++k) Xw[3 * i + k] = X(k);
    }
  }
  void write_outliers(FrameT* pFrame) const { for (int i = 0; i < N; ++i) pFrame->mvbOutlier[i] = outlier[i] != 0; }
};

}  // namespace morb_glue

// int Optimizer::PoseOptimization(Frame* pFrame)  Optimizer.cc:762-1051
template <class FrameT>
int Optimizer::PoseOptimization(FrameT* pFrame) {
  using SE3 = typename std::decay<decltype(pFrame->GetPose())>::type;
  morb_glue::FrameEdges<FrameT> fe(pFrame);
  PoseOptimizationView f;
  f.N = fe.N; f.hasMapPoint = fe.has.data(); f.obs = fe.obs.data(); f.invSigma2 = fe.inv.data(); f.worldPos = fe.Xw.data();
  f.mvbOutlier = fe.outlier;
  f.fx = pFrame->fx; f.fy = pFrame->fy; f.cx = pFrame->cx; f.cy = pFrame->cy; f.mbf = pFrame->mbf;
  morb_glue::pose7(pFrame->GetPose(), f.pose);                     // :781-783
  float rig[28];
  if (pFrame->mpCamera2) { morb_glue::rig28(pFrame, rig); f.nLeft = pFrame->Nleft; f.rig28 = rig; }   // :880-946
  int nInitialCorrespondences = 0;
  for (int i = 0; i < fe.N; ++i) nInitialCorrespondences += fe.has[i] ? 1 : 0;
  if (fe.N <= 0 || nInitialCorrespondences < 3) {                   // :951: the reference returns without touching the pose; the graph fill
    fe.write_outliers(pFrame);                                      // before it has already cleared mvbOutlier[i] of the features with a map point
    return 0;
  }
  const int nin = PoseOptimization(f);
  pFrame->SetPose(morb_glue::make_se3<SE3>(f.pose));               // :1044-1048
  fe.outlier = f.mvbOutlier;
  fe.write_outliers(pFrame);
  return nin;
}

// int Optimizer::PoseInertialOptimizationLastKeyFrame(Frame* pFrame, bool bRecInit)  Optimizer.cc:4391-4757
template <class FrameT>
int Optimizer::PoseInertialOptimizationLastKeyFrame(FrameT* pFrame, bool bRecInit) {
  using CPI = typename std::remove_pointer<decltype(pFrame->mpcpi)>::type;
  using Bias = typename std::decay<decltype(pFrame->mImuBias)>::type;
  morb_glue::FrameEdges<FrameT> fe(pFrame);
  PoseInertialView v;
  v.N = fe.N; v.nLeft = fe.nLeft; v.hasMapPoint = fe.has.data(); v.obs = fe.obs.data(); v.invSigma2 = fe.inv.data(); v.worldPos = fe.Xw.data();
  v.close = fe.close.data(); v.mvbOutlier = fe.outlier;
  v.fx = pFrame->fx; v.fy = pFrame->fy; v.cx = pFrame->cx; v.cy = pFrame->cy; v.mbf = pFrame->mbf;
  float rig[28];
  if (pFrame->mpCamera2) { morb_glue::rig28(pFrame, rig); v.rig28 = rig; }
  morb_glue::tbc12(pFrame->mImuCalib.mTbc, v.Tbc12);
  morb_glue::imu_state21(pFrame, v.state); morb_glue::bias6(pFrame->mImuBias, v.state + 15);          // VertexPose(pFrame), ... :4411-4426
  auto* pKF = pFrame->mpLastKeyFrame;                                                                  // :4548-4566: fixed vertices
  morb_glue::imu_state21(pKF, v.otherState); morb_glue::bias6(pKF->GetImuBias(), v.otherState + 15);
  morb_glue::fill_pre(v.pre, pFrame->mpImuPreintegrated);
  const int nin = PoseInertialOptimizationLastKeyFrame(v, bRecInit);
  write_back_inertial(pFrame, v);
  pFrame->mImuBias = Bias(v.state[18], v.state[19], v.state[20], v.state[15], v.state[16], v.state[17]);   // :4708-4710
  pFrame->mpcpi = morb_glue::prior_from_doubles<CPI>(v.prior);                                        // :4752-4754
  fe.outlier = v.mvbOutlier; fe.write_outliers(pFrame);
  return nin;
}

// int Optimizer::PoseInertialOptimizationLastFrame(Frame* pFrame, bool bRecInit)  Optimizer.cc:4761-5161
template <class FrameT>
int Optimizer::PoseInertialOptimizationLastFrame(FrameT* pFrame, bool bRecInit) {
  using CPI = typename std::remove_pointer<decltype(pFrame->mpcpi)>::type;
  using Bias = typename std::decay<decltype(pFrame->mImuBias)>::type;
  FrameT* pFp = pFrame->mpPrevFrame;                                                                   // :4920
  if (pFp->mpcpi == nullptr) return 0;                                                                 // :4968-4971 "NO MPCPI"
  morb_glue::FrameEdges<FrameT> fe(pFrame);
  PoseInertialView v;
  v.N = fe.N; v.nLeft = fe.nLeft; v.hasMapPoint = fe.has.data(); v.obs = fe.obs.data(); v.invSigma2 = fe.inv.data(); v.worldPos = fe.Xw.data();
  v.close = fe.close.data(); v.mvbOutlier = fe.outlier;
  v.fx = pFrame->fx; v.fy = pFrame->fy; v.cx = pFrame->cx; v.cy = pFrame->cy; v.mbf = pFrame->mbf;
  float rig[28];
  if (pFrame->mpCamera2) { morb_glue::rig28(pFrame, rig); v.rig28 = rig; }
  morb_glue::tbc12(pFrame->mImuCalib.mTbc, v.Tbc12);
  morb_glue::imu_state21(pFrame, v.state); morb_glue::bias6(pFrame->mImuBias, v.state + 15);
  morb_glue::imu_state21(pFp, v.otherState); morb_glue::bias6(pFp->mImuBias, v.otherState + 15);      // :4922-4937: free vertices of the previous frame
  morb_glue::fill_pre(v.pre, pFrame->mpImuPreintegratedFrame);                                        // :4939 EdgeInertial
  morb_glue::fill_pre(v.preKF, pFrame->mpImuPreintegrated);                                           // :4951-4965 random-walk information
  morb_glue::prior_to_doubles(pFp->mpcpi, v.prevPrior);                                               // :4972 EdgePriorPoseImu
  const int nin = PoseInertialOptimizationLastFrame(v, bRecInit);
  write_back_inertial(pFrame, v);
  pFrame->mImuBias = Bias(v.state[18], v.state[19], v.state[20], v.state[15], v.state[16], v.state[17]);   // :5078-5080
  pFrame->mpcpi = morb_glue::prior_from_doubles<CPI>(v.prior);                                        // :5150-5152
  static CPI* oldMpcpi = nullptr;                                                                     // :4759, :5153-5159: the fork's guard against freeing the same
  if (oldMpcpi != pFp->mpcpi) {                                                                       // address twice ("SAME MPCPI": nothing is deleted, the pointer stays)
    oldMpcpi = pFp->mpcpi;
    delete pFp->mpcpi;
    pFp->mpcpi = NULL;
  }
  fe.outlier = v.mvbOutlier; fe.write_outliers(pFrame);
  return nin;
}
template <class FrameT>
void Optimizer::write_back_inertial(FrameT* pFrame, const PoseInertialView& v) {
  // pFrame->SetImuPoseVelocity(VP->estimate().Rwb.cast<float>(), VP->estimate().twb.cast<float>(), VV->estimate().cast<float>())  :4703-4705
  typename std::decay<decltype(pFrame->GetImuRotation())>::type R;
  typename std::decay<decltype(pFrame->GetImuPosition())>::type t, vel;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = v.state[3 * r + c];
  for (int k = 0; k < 3; ++k) { t(k) = v.state[9 + k]; vel(k) = v.state[12 + k]; }
  pFrame->SetImuPoseVelocity(R, t, vel);
}

// void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges)
// Optimizer.cc:1053-1441
template <class KF, class MapT>
void Optimizer::LocalBundleAdjustment(KF* pKF, bool* pbStopFlag, MapT* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges) {
  using SE3 = typename std::decay<decltype(pKF->GetPose())>::type;
  using MP = typename std::remove_pointer<typename std::decay<decltype(pKF->GetMapPointMatches()[0])>::type>::type;
  // :1058-1122 — the reference's graph selection, unchanged
  std::list<KF*> lLocalKeyFrames;
  lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  MapT* pCurrentMap = pKF->GetMap();
  const std::vector<KF*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (KF* pKFi : vNeighKFs) {
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(pKFi);
  }
  num_fixedKF = 0;
  std::list<MP*> lLocalMapPoints;
  for (KF* pKFi : lLocalKeyFrames) {
    if (pKFi->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
    for (MP* pMP : pKFi->GetMapPointMatches())
      if (pMP && !pMP->isBad() && pMP->GetMap() == pCurrentMap && pMP->mnBALocalForKF != pKF->mnId) {
        lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->mnId;
      }
  }
  std::list<KF*> lFixedCameras;
  for (MP* pMP : lLocalMapPoints)
    for (const auto& ob : pMP->GetObservations()) {
      KF* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lFixedCameras.push_back(pKFi);
      }
    }
  num_fixedKF = (int)lFixedCameras.size() + num_fixedKF;
  if (num_fixedKF == 0) return;   // :1118-1122 "LBA aborted"
  // :1150-1351 — vertices and edges, flattened
  const bool rig = pKF->mpCamera2 != nullptr;
  std::map<KF*, int> kfIndex;
  std::vector<KF*> kfs;
  std::vector<float> kfPose;
  std::vector<uint8_t> kfFixed;
  auto add_kf = [&](KF* k, bool fixed) {
    kfIndex[k] = (int)kfs.size(); kfs.push_back(k);
    float p[7];
    morb_glue::pose7(k->GetPose(), p);
    kfPose.insert(kfPose.end(), p, p + 7);
    kfFixed.push_back(fixed ? 1 : 0);
  };
  for (KF* k : lLocalKeyFrames) add_kf(k, k->mnId == pMap->GetInitKFid());   // :1160
  num_OptKF = (int)lLocalKeyFrames.size();
  for (KF* k : lFixedCameras) add_kf(k, true);
  std::vector<MP*> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());
  std::vector<float> mpPos, eObs, eInv;
  std::vector<int> eKF, eMP;
  std::vector<uint8_t> eRight;
  std::vector<std::pair<KF*, MP*>> eOwner;
  for (int j = 0; j < (int)mps.size(); ++j) {
    MP* pMP = mps[j];
    const auto X = pMP->GetWorldPos();
    mpPos.push_back(X(0)); mpPos.push_back(X(1)); mpPos.push_back(X(2));
    for (const auto& ob : pMP->GetObservations()) {
      KF* pKFi = ob.first;
      if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
      const auto it = kfIndex.find(pKFi);
      if (it == kfIndex.end()) continue;   // (a keyframe g2o has no vertex for: optimizer.vertex(id) would be NULL)
      const int leftIndex = std::get<0>(ob.second);
      if (leftIndex != -1) {               // :1236-1290: monocular (mvuRight < 0) or stereo observation of the left camera
        const auto& kp = pKFi->NLeft == -1 ? pKFi->mvKeysUn[leftIndex] : pKFi->mvKeys[leftIndex];
        eKF.push_back(it->second); eMP.push_back(j); eRight.push_back(0);
        eObs.push_back(kp.pt.x); eObs.push_back(kp.pt.y); eObs.push_back(pKFi->mvuRight[leftIndex]);
        eInv.push_back(pKFi->mvInvLevelSigma2[kp.octave]);
        eOwner.emplace_back(pKFi, pMP);
      }
      if (pKFi->mpCamera2) {               // :1292-1330: EdgeSE3ProjectXYZToBody on the right camera
        const int rightIndex = std::get<1>(ob.second);
        if (rightIndex != -1) {
          const auto& kp = pKFi->mvKeysRight[rightIndex - pKFi->NLeft];
          eKF.push_back(it->second); eMP.push_back(j); eRight.push_back(1);
          eObs.push_back(kp.pt.x); eObs.push_back(kp.pt.y); eObs.push_back(-1.f);
          eInv.push_back(pKFi->mvInvLevelSigma2[kp.octave]);
          eOwner.emplace_back(pKFi, pMP);
        }
      }
    }
  }
  num_MPs = (int)mps.size();
  num_edges = (int)eKF.size();
  if (pbStopFlag && *pbStopFlag) return;   // :1353-1354
  LocalBAView g;
  g.nKF = (int)kfs.size(); g.nMP = (int)mps.size(); g.nE = (int)eKF.size();
  g.kfPose = kfPose.data(); g.kfFixed = kfFixed.data(); g.mpPos = mpPos.data(); g.eKF = eKF.data(); g.eMP = eMP.data();
  g.eObs = eObs.data(); g.eInvSigma2 = eInv.data();
  g.fx = pKF->fx; g.fy = pKF->fy; g.cx = pKF->cx; g.cy = pKF->cy; g.mbf = pKF->mbf;
  g.inertialMap = pMap->IsInertial();   // :1137
  float rigp[28];
  if (rig) { morb_glue::rig28(pKF, rigp); g.rig28 = rigp; g.eRight = eRight.data(); }
  LocalBundleAdjustment(g, pbStopFlag);
  // :1366-1401: the library marks the observations the reference erases (chi2 > 5.991 / 7.815 or non-positive depth); bad points are skipped here
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);   // :1404
  for (size_t e = 0; e < eOwner.size(); ++e)
    if (g.eraseFlag[e] && !eOwner[e].second->isBad()) {
      eOwner[e].first->EraseMapPointMatch(eOwner[e].second);
      eOwner[e].second->EraseObservation(eOwner[e].first);
    }
  int i = 0;
  for (KF* k : lLocalKeyFrames) k->SetPose(morb_glue::make_se3<SE3>(&kfPose[(size_t)7 * i++]));   // :1416-1425
  for (int j = 0; j < (int)mps.size(); ++j) {   // :1428-1436
    typename std::decay<decltype(mps[j]->GetWorldPos())>::type X;
    for (int k = 0; k < 3; ++k) X(k) = mpPos[3 * j + k];
    mps[j]->SetWorldPos(X);
    mps[j]->UpdateNormalAndDepth();
  }
  pMap->IncreaseChangeIndex();
}

// void Optimizer::LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int&, int&, int&, int&, bool bLarge, bool bRecInit)  Optimizer.cc:2324-2897
template <class KF, class MapT>
void Optimizer::LocalInertialBA(KF* pKF, bool* pbStopFlag, MapT* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges, bool bLarge,
                                bool bRecInit) {
  using SE3 = typename std::decay<decltype(pKF->GetPose())>::type;
  using MP = typename std::remove_pointer<typename std::decay<decltype(pKF->GetMapPointMatches()[0])>::type>::type;
  using Bias = typename std::decay<decltype(pKF->GetImuBias())>::type;
  MapT* pCurrentMap = pKF->GetMap();
  const int maxOpt = bLarge ? 25 : 10;                                                // :2330-2335
  const int Nd = std::min((int)pCurrentMap->KeyFramesInMap() - 2, maxOpt);
  // :2339-2352: the temporal window
  std::vector<KF*> vpOptimizableKFs;
  vpOptimizableKFs.reserve(Nd > 0 ? Nd : 1);
  vpOptimizableKFs.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  for (int i = 1; i < Nd; i++) {
    if (vpOptimizableKFs.back()->mPrevKF) {
      vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
      vpOptimizableKFs.back()->mnBALocalForKF = pKF->mnId;
    } else break;
  }
  int N = (int)vpOptimizableKFs.size();
  std::list<MP*> lLocalMapPoints;                                                       // :2357-2371
  for (int i = 0; i < N; i++)
    for (MP* pMP : vpOptimizableKFs[i]->GetMapPointMatches())
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
  std::list<KF*> lFixedKeyFrames;                                                       // :2374-2383
  if (vpOptimizableKFs.back()->mPrevKF) {
    lFixedKeyFrames.push_back(vpOptimizableKFs.back()->mPrevKF);
    vpOptimizableKFs.back()->mPrevKF->mnBAFixedForKF = pKF->mnId;
  } else {
    vpOptimizableKFs.back()->mnBALocalForKF = 0;
    vpOptimizableKFs.back()->mnBAFixedForKF = pKF->mnId;
    lFixedKeyFrames.push_back(vpOptimizableKFs.back());
    vpOptimizableKFs.pop_back();
  }
  // (:2386-2410: maxCovKF = 0 — no optimizable visual keyframes)
  const int maxFixKF = 200;                                                             // :2413-2435
  for (MP* pMP : lLocalMapPoints) {
    for (const auto& ob : pMP->GetObservations()) {
      KF* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad()) { lFixedKeyFrames.push_back(pKFi); break; }
      }
    }
    if ((int)lFixedKeyFrames.size() >= maxFixKF) break;
  }
  // vertices (:2461-2521): temporal keyframes (kind 0), then the fixed ones: the first is the keyframe before the window (kind 1), the rest kind 2
  N = (int)vpOptimizableKFs.size();
  std::map<KF*, int> kfIndex;
  std::vector<KF*> kfs;
  std::vector<float> kfState;
  std::vector<uint8_t> kfKind;
  auto add_kf = [&](KF* k, int kind) {
    kfIndex[k] = (int)kfs.size(); kfs.push_back(k);
    float s[21];
    morb_glue::imu_state21(k, s); morb_glue::bias6(k->GetImuBias(), s + 15);
    kfState.insert(kfState.end(), s, s + 21);
    kfKind.push_back((uint8_t)kind);
  };
  for (KF* k : vpOptimizableKFs) add_kf(k, 0);
  { bool first = true; for (KF* k : lFixedKeyFrames) { add_kf(k, first && k->bImu ? 1 : 2); first = false; } }
  // inertial links (:2524-2601)
  std::vector<int> iKF1, iKF2;
  std::vector<morb_imu_preintegrated> iPre;
  std::vector<uint8_t> iRobust;
  std::vector<float> iInfoScale;
  for (int i = 0; i < N; i++) {
    KF* pKFi = vpOptimizableKFs[i];
    if (!pKFi->mPrevKF) continue;
    if (!(pKFi->bImu && pKFi->mPrevKF->bImu && pKFi->mpImuPreintegrated)) continue;
    pKFi->mpImuPreintegrated->SetNewBias(pKFi->mPrevKF->GetImuBias());                // :2533 (before the vertices are looked up, as the reference does)
    const auto a = kfIndex.find(pKFi->mPrevKF);
    if (a == kfIndex.end()) continue;                                                 // (:2547-2552: a vertex is missing)
    iKF1.push_back(a->second); iKF2.push_back(kfIndex[pKFi]);
    iPre.emplace_back(); morb_glue::fill_pre(iPre.back(), pKFi->mpImuPreintegrated);
    iRobust.push_back((i == N - 1 || bRecInit) ? 1 : 0);                              // :2563-2573
    iInfoScale.push_back(i == N - 1 ? 1e-2f : 1.f);
  }
  // points and visual edges (:2643-2762)
  const bool rig = pKF->mpCamera2 != nullptr;
  std::vector<MP*> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());
  std::vector<float> mpPos, eObs, eInv;
  std::vector<uint8_t> mpClose, eRight;
  std::vector<int> eKF, eMP;
  std::vector<std::pair<KF*, MP*>> eOwner;
  for (int j = 0; j < (int)mps.size(); ++j) {
    MP* pMP = mps[j];
    const auto X = pMP->GetWorldPos();
    mpPos.push_back(X(0)); mpPos.push_back(X(1)); mpPos.push_back(X(2));
    mpClose.push_back(pMP->mTrackDepth < 10.f ? 1 : 0);                               // :2781
    for (const auto& ob : pMP->GetObservations()) {
      KF* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) continue;
      if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
      const auto it = kfIndex.find(pKFi);
      if (it == kfIndex.end()) continue;
      const int leftIndex = std::get<0>(ob.second);
      if (leftIndex != -1) {                                                          // :2666-2722: EdgeMono(0) / EdgeStereo(0)
        const auto& kp = pKFi->mvKeysUn[leftIndex];
        eKF.push_back(it->second); eMP.push_back(j); eRight.push_back(0);
        eObs.push_back(kp.pt.x); eObs.push_back(kp.pt.y); eObs.push_back(pKFi->mvuRight[leftIndex]);
        eInv.push_back(pKFi->mvInvLevelSigma2[kp.octave]);
        eOwner.emplace_back(pKFi, pMP);
      }
      if (pKFi->mpCamera2) {                                                          // :2725-2758: EdgeMono(1) on the right camera
        const int rightIndex = std::get<1>(ob.second);
        if (rightIndex != -1) {
          const auto& kp = pKFi->mvKeysRight[rightIndex - pKFi->NLeft];
          eKF.push_back(it->second); eMP.push_back(j); eRight.push_back(1);
          eObs.push_back(kp.pt.x); eObs.push_back(kp.pt.y); eObs.push_back(-1.f);
          eInv.push_back(pKFi->mvInvLevelSigma2[kp.octave]);
          eOwner.emplace_back(pKFi, pMP);
        }
      }
    }
  }
  num_fixedKF = (int)lFixedKeyFrames.size(); num_OptKF = N; num_MPs = (int)mps.size(); num_edges = (int)eKF.size();
  LocalInertialBAView g;
  g.nKF = (int)kfs.size(); g.kfState = kfState.data(); g.kfKind = kfKind.data();
  g.nMP = (int)mps.size(); g.mpPos = mpPos.data(); g.mpClose = mpClose.data();
  g.nE = (int)eKF.size(); g.eKF = eKF.data(); g.eMP = eMP.data(); g.eObs = eObs.data(); g.eInvSigma2 = eInv.data();
  g.nI = (int)iKF1.size(); g.iKF1 = iKF1.data(); g.iKF2 = iKF2.data(); g.iPre = iPre.data(); g.iRobust = iRobust.data(); g.iInfoScale = iInfoScale.data();
  g.fx = pKF->fx; g.fy = pKF->fy; g.cx = pKF->cx; g.cy = pKF->cy; g.mbf = pKF->mbf;
  morb_glue::tbc12(pKF->mImuCalib.mTbc, g.Tbc12);
  float rigp[28];
  if (rig) { morb_glue::rig28(pKF, rigp); g.rig28 = rigp; g.eRight = eRight.data(); }
  g.bLarge = bLarge;
  LocalInertialBA(g);
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                             // :2808
  if (!g.ok) return;                                                                    // :2811-2815 "FAIL LOCAL-INERTIAL BA"
  for (size_t e = 0; e < eOwner.size(); ++e)                                            // :2773-2826
    if (g.eraseFlag[e] && !eOwner[e].second->isBad()) {
      eOwner[e].first->EraseMapPointMatch(eOwner[e].second);
      eOwner[e].second->EraseObservation(eOwner[e].first);
    }
  for (KF* k : lFixedKeyFrames) k->mnBAFixedForKF = 0;                                  // :2828-2831
  // :2836-2858: Tcw = (Rcb Rwb^T, tcb - Rcb Rwb^T twb) (ImuCamPose, G2oTypes.cc:74-113), velocity, bias
  float Tbc[12];
  morb_glue::tbc12(pKF->mImuCalib.mTbc, Tbc);
  for (int i = 0; i < N; i++) {
    KF* pKFi = vpOptimizableKFs[i];
    const float* s = &kfState[(size_t)21 * i];
    double Rcw[9], tcw[3];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) { double a = 0; for (int k = 0; k < 3; ++k) a += (double)Tbc[3 * k + r] * s[3 * c + k]; Rcw[3 * r + c] = a; }   // Rcb Rwb^T, Rcb = Rbc^T
    }
    for (int r = 0; r < 3; ++r) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a -= (double)Tbc[3 * k + r] * Tbc[9 + k];            // tcb = -Rbc^T tbc
      for (int c = 0; c < 3; ++c) a -= Rcw[3 * r + c] * s[9 + c];
      tcw[r] = a;
    }
    pKFi->SetPose(se3_from_matrix<SE3>(Rcw, tcw));
    pKFi->mnBALocalForKF = 0;
    if (pKFi->bImu) {
      typename std::decay<decltype(pKFi->GetVelocity())>::type vel;
      for (int k = 0; k < 3; ++k) vel(k) = s[12 + k];
      pKFi->SetVelocity(vel);
      pKFi->SetNewBias(Bias(s[18], s[19], s[20], s[15], s[16], s[17]));
    }
  }
  for (int j = 0; j < (int)mps.size(); ++j) {                                           // :2873-2881
    typename std::decay<decltype(mps[j]->GetWorldPos())>::type X;
    for (int k = 0; k < 3; ++k) X(k) = mpPos[3 * j + k];
    mps[j]->SetWorldPos(X);
    mps[j]->UpdateNormalAndDepth();
  }
  pMap->IncreaseChangeIndex();
}
// Sophus::SE3f(Matrix3f Rcw, Vector3f tcw) (:2840-2842)
template <class SE3>
SE3 Optimizer::se3_from_matrix(const double R[9], const double t[3]) {
  typename std::decay<decltype(std::declval<SE3>().rotationMatrix())>::type Rm;
  typename SE3::Point tv;
  for (int r = 0; r < 3; ++r) { tv(r) = (float)t[r]; for (int c = 0; c < 3; ++c) Rm(r, c) = (float)R[3 * r + c]; }
  return SE3(Rm, tv);
}

}  // namespace ORB_SLAM3
