// Reference-typed entry points: the glue between the reference's Frame / KeyFrame / MapPoint / Map objects and the view-taking
// adapters of ORBmatcher.h / Optimizer.h, so that the call sites of src/Tracking.cc and src/LocalMapping.cc keep their shape:
//
//   ORBmatcher matcher(0.8);                                         // Tracking.cc:3222-3285 SearchLocalPoints
//   int n = morb_glue::SearchByProjection(matcher, mCurrentFrame, vpMapPoints, th, mpLocalMapper->mbFarPoints, mpLocalMapper->mThFarPoints);
//   int n = morb_glue::SearchByProjection(matcher, mCurrentFrame, mLastFrame, th, mSensor == System::MONOCULAR);   // :2677-2690
//   int n = morb_glue::SearchByBoW(matcher, mpReferenceKF, mCurrentFrame, vpMapPointMatches);                      // :2541
//   int n = morb_glue::SearchForTriangulation(matcher, mpCurrentKeyFrame, pKF2, vMatchedIndices, false, bCoarse);  // LocalMapping.cc:424-473
//   int nin = morb_glue::PoseOptimization(&mCurrentFrame);                                                         // Tracking.cc:2559, :2710, :2764
//   morb_glue::LocalBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpCurrentKeyFrame->GetMap(), a, b, c, d);      // LocalMapping.cc:181
//
// Compiled only inside the reference tree (it needs Frame.h, KeyFrame.h, MapPoint.h, Map.h and with them OpenCV / Eigen / Sophus /
// DBoW2).  THIS FILE HAS NEVER BEEN COMPILED AGAINST THE REFERENCE: the build container has none of those headers.  What is checked
// here (tests/test_oracle_cpu.py::test_reference_glue_parses) is that it parses and type-checks against mock declarations of the
// members it touches (tests/native/mock_ref: names and types read off the reference headers, no behaviour).  It is written against the member
// names of /root/reference/include/{Frame,KeyFrame,MapPoint,Map}.h and follows, statement for statement, the gathering and
// write-back code of the methods it replaces (cited per function); the views it fills are the ones tests/native/adapters_check.cc
// drives.  Pinhole cameras only: on a KannalaBrandt8 rig (Nleft != -1 / mpCamera2) call the *_fisheye entry points of morb_hip.h with
// the left + right features in one row.
// Two members the reference keeps private or protected are read through their public accessors: mfMaxDistance / mfMinDistance as
// GetMaxDistanceInvariance() / 1.2f and GetMinDistanceInvariance() / 0.8f (the library multiplies the factors back in; the
// round trip can move a value by one ulp — adding two plain getters to MapPoint removes even that).
#pragma once
#if __has_include("Frame.h") && __has_include("KeyFrame.h") && __has_include("MapPoint.h") && __has_include("Map.h")
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "ORBmatcher.h"   // include/morb/ORBmatcher.h (this directory first on the include path)
#include "Optimizer.h"    // include/morb/Optimizer.h

namespace ORB_SLAM3 {
namespace morb_glue {

// ---- views ---------------------------------------------------------------------------------------------------------------
struct FrameStore {           // owns the arrays a FrameView points at
  KeyFrameView v;
  std::vector<uint8_t> tracked, hasMP, mpDesc, mpObs;
  std::vector<float> mpPos, mpMax, mpMin;
  std::vector<int> node;
};
inline void fill_pose(FrameView& v, const Sophus::SE3f& Tcw) {
  const Eigen::Matrix3f R = Tcw.rotationMatrix();
  const Eigen::Vector3f t = Tcw.translation(), Ow = Tcw.inverse().translation();
  const Eigen::Quaternionf q = Tcw.unit_quaternion();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) v.mRcw[3 * r + c] = R(r, c);
  for (int k = 0; k < 3; ++k) { v.mtcw[k] = t(k); v.mOw[k] = Ow(k); v.Tcw[4 + k] = t(k); }
  v.Tcw[0] = q.x(); v.Tcw[1] = q.y(); v.Tcw[2] = q.z(); v.Tcw[3] = q.w();
}
template <typename F>   // Frame or KeyFrame: the members have the same names
inline void fill_params(morb_frame_params& P, const F& f, float gridInvW, float gridInvH) {
  P.minX = (float)f.mnMinX; P.minY = (float)f.mnMinY; P.maxX = (float)f.mnMaxX; P.maxY = (float)f.mnMaxY;
  P.gridInvW = gridInvW; P.gridInvH = gridInvH;
  P.fx = f.fx; P.fy = f.fy; P.cx = f.cx; P.cy = f.cy; P.mbf = f.mbf; P.mb = f.mb; P.logScaleFactor = f.mfLogScaleFactor;
  P.nlevels = f.mnScaleLevels;
  for (int l = 0; l < f.mnScaleLevels && l < 16; ++l) { P.scaleFactors[l] = f.mvScaleFactors[l]; P.levelSigma2[l] = f.mvLevelSigma2[l]; }
}
inline void node_table(const DBoW2::FeatureVector& fv, int N, std::vector<int>& node) {   // mFeatVec: node -> feature indices
  node.assign(N, -1);
  for (const auto& kv : fv) for (unsigned int i : kv.second) if ((int)i < N) node[i] = (int)kv.first;
}
// the map point of every feature, read under the locks the getters take
inline void fill_points(FrameStore& s, const std::vector<MapPoint*>& mps, const std::vector<bool>* outlier) {
  const int N = s.v.N;
  s.hasMP.assign(N, 0); s.tracked.assign(N, 0); s.mpObs.assign(N, 0); s.mpDesc.assign((size_t)N * 32, 0);
  s.mpPos.assign((size_t)N * 3, 0.f); s.mpMax.assign(N, 1.f); s.mpMin.assign(N, 1.f);
  for (int i = 0; i < N && i < (int)mps.size(); ++i) {
    MapPoint* p = mps[i];
    if (!p) continue;
    const bool obs = p->Observations() > 0;
    s.tracked[i] = obs ? 1 : 0; s.mpObs[i] = obs ? 1 : 0;
    if (p->isBad() || (outlier && (*outlier)[i])) continue;
    s.hasMP[i] = 1;
    const Eigen::Vector3f X = p->GetWorldPos();
    for (int k = 0; k < 3; ++k) s.mpPos[3 * i + k] = X(k);
    s.mpMax[i] = p->GetMaxDistanceInvariance() / 1.2f; s.mpMin[i] = p->GetMinDistanceInvariance() / 0.8f;
    const cv::Mat d = p->GetDescriptor();
    std::memcpy(&s.mpDesc[(size_t)i * 32], d.ptr<uint8_t>(0), 32);
  }
  s.v.hasTrackedMapPoint = s.tracked.data(); s.v.hasMapPoint = s.hasMP.data(); s.v.mpWorldPos = s.mpPos.data();
  s.v.mpMaxDistance = s.mpMax.data(); s.v.mpMinDistance = s.mpMin.data(); s.v.mpDescriptor = s.mpDesc.data();
  s.v.mpHasObservations = s.mpObs.data();
}
inline void frame_view(FrameStore& s, Frame& F, bool lastFrameRules = false) {
  static_assert(sizeof(cv::KeyPoint) == sizeof(morb_keypoint), "cv::KeyPoint layout");
  s.v.N = F.N;
  s.v.mvKeysUn = reinterpret_cast<const morb_keypoint*>(F.mvKeysUn.data());
  s.v.mDescriptors = F.mDescriptors.ptr<uint8_t>(0);   // N x 32 CV_8U, continuous (ORBextractor creates it)
  s.v.mvuRight = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
  fill_params(s.v.params, F, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv);
  if (F.HasPose()) fill_pose(s.v, F.GetPose());
  node_table(F.mFeatVec, F.N, s.node); s.v.featNode = s.node.data();
  fill_points(s, F.mvpMapPoints, lastFrameRules ? &F.mvbOutlier : nullptr);
}
inline void keyframe_view(FrameStore& s, KeyFrame* pKF) {
  s.v.N = pKF->N;
  s.v.mvKeysUn = reinterpret_cast<const morb_keypoint*>(pKF->mvKeysUn.data());
  s.v.nValid = (int)pKF->mvKeysUn.size();
  s.v.mDescriptors = pKF->mDescriptors.ptr<uint8_t>(0);
  s.v.mvuRight = pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data();
  fill_params(s.v.params, *pKF, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv);
  fill_pose(s.v, pKF->GetPose());
  node_table(pKF->mFeatVec, pKF->N, s.node); s.v.featNode = s.node.data();
  fill_points(s, pKF->GetMapPointMatches(), nullptr);
}
struct PointStore {
  MapPointView v;
  std::vector<float> pos, nrm, maxD, minD;
  std::vector<uint8_t> desc, bad, obs, valid;
};
inline void mappoint_view(PointStore& s, const std::vector<MapPoint*>& mps) {
  const int n = (int)mps.size();
  s.pos.assign((size_t)n * 3, 0.f); s.nrm.assign((size_t)n * 3, 0.f); s.maxD.assign(n, 1.f); s.minD.assign(n, 1.f);
  s.desc.assign((size_t)n * 32, 0); s.bad.assign(n, 1); s.obs.assign(n, 0); s.valid.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    MapPoint* p = mps[i];
    if (!p) continue;
    s.bad[i] = p->isBad() ? 1 : 0; s.obs[i] = p->Observations() > 0 ? 1 : 0; s.valid[i] = s.bad[i] ? 0 : 1;
    const Eigen::Vector3f X = p->GetWorldPos(), nv = p->GetNormal();
    for (int k = 0; k < 3; ++k) { s.pos[3 * i + k] = X(k); s.nrm[3 * i + k] = nv(k); }
    s.maxD[i] = p->GetMaxDistanceInvariance() / 1.2f; s.minD[i] = p->GetMinDistanceInvariance() / 0.8f;
    std::memcpy(&s.desc[(size_t)i * 32], p->GetDescriptor().ptr<uint8_t>(0), 32);
  }
  s.v.n = n; s.v.worldPos = s.pos.data(); s.v.normal = s.nrm.data(); s.v.maxDistance = s.maxD.data(); s.v.minDistance = s.minD.data();
  s.v.descriptor = s.desc.data(); s.v.isBad = s.bad.data(); s.v.hasObservations = s.obs.data(); s.v.valid = s.valid.data();
}

// ---- ORBmatcher ----------------------------------------------------------------------------------------------------------
// int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)  ORBmatcher.cc:42-209,
// preceded by the isInFrustum loop of Tracking::SearchLocalPoints (Tracking.cc:3258-3275): points already tracked in this frame
// (pMP->mnLastFrameSeen == F.mnId) or bad are passed as isBad so that they are skipped exactly as the reference skips them.
inline int SearchByProjection(ORBmatcher& m, Frame& F, const std::vector<MapPoint*>& vpMapPoints, float th = 3, bool bFarPoints = false,
                              float thFarPoints = 50.0f) {
  FrameStore fs; frame_view(fs, F);
  PointStore ps; mappoint_view(ps, vpMapPoints);
  for (size_t i = 0; i < vpMapPoints.size(); ++i)
    if (vpMapPoints[i] && vpMapPoints[i]->mnLastFrameSeen == F.mnId) ps.bad[i] = 1;
  std::vector<int> matchF(F.N, -1);
  const int n = m.SearchByProjection(fs.v, ps.v, matchF, th, bFarPoints, thFarPoints);
  for (int i = 0; i < F.N; ++i)
    if (matchF[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[matchF[i]];   // :132, :197
  return n;
}
// int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)  ORBmatcher.cc:1521-1733
inline int SearchByProjection(ORBmatcher& m, Frame& CurrentFrame, Frame& LastFrame, float th, bool bMono) {
  FrameStore cur, last; frame_view(cur, CurrentFrame); frame_view(last, LastFrame, true);
  std::vector<int> matchCur(CurrentFrame.N, -1);
  const int n = m.SearchByProjection(cur.v, last.v, matchCur, th, bMono);
  for (int i = 0; i < CurrentFrame.N; ++i)
    if (matchCur[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[matchCur[i]];   // :1627
  return n;
}
// int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  ORBmatcher.cc:218-395
inline int SearchByBoW(ORBmatcher& m, KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
  FrameStore kf, fr; keyframe_view(kf, pKF); frame_view(fr, F);
  const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
  std::vector<int> idx;
  const int n = m.SearchByBoW(kf.v, static_cast<const FrameView&>(fr.v), idx);
  vpMapPointMatches.assign(F.N, static_cast<MapPoint*>(NULL));   // :222
  for (int j = 0; j < F.N; ++j) if (idx[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[idx[j]];
  return n;
}
// int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vMatchedPairs, bOnlyStereo, bCoarse)  ORBmatcher.cc:821-1042
inline int SearchForTriangulation(ORBmatcher& m, KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                                  bool bOnlyStereo, bool bCoarse = false) {
  FrameStore a, b; keyframe_view(a, pKF1); keyframe_view(b, pKF2);
  // :829-838
  const Sophus::SE3f T1w = pKF1->GetPose(), T2w = pKF2->GetPose(), Tw2 = pKF2->GetPoseInverse();
  const Eigen::Vector3f Cw = pKF1->GetCameraCenter(), C2 = T2w * Cw;
  const Eigen::Vector2f ep = pKF2->mpCamera->project(C2);
  const Sophus::SE3f T12 = T1w * Tw2;
  const Eigen::Matrix3f R12 = T12.rotationMatrix();
  const Eigen::Vector3f t12 = T12.translation();
  float R[9], t[3], e[2] = {ep(0), ep(1)};
  for (int r = 0; r < 3; ++r) { t[r] = t12(r); for (int c = 0; c < 3; ++c) R[3 * r + c] = R12(r, c); }
  // SearchForTriangulation looks at features WITHOUT a map point: GetMapPoint(idx) != NULL, bad or not (:876, :903)
  const std::vector<MapPoint*> mp1 = pKF1->GetMapPointMatches(), mp2 = pKF2->GetMapPointMatches();
  for (int i = 0; i < a.v.N; ++i) a.hasMP[i] = (i < (int)mp1.size() && mp1[i]) ? 1 : 0;
  for (int i = 0; i < b.v.N; ++i) b.hasMP[i] = (i < (int)mp2.size() && mp2[i]) ? 1 : 0;
  return m.SearchForTriangulation(a.v, b.v, R, t, e, vMatchedPairs, bOnlyStereo, bCoarse);
}

// ---- Optimizer -----------------------------------------------------------------------------------------------------------
// int Optimizer::PoseOptimization(Frame* pFrame)  Optimizer.cc:762-1051 (pinhole: monocular and stereo edges)
inline int PoseOptimization(Frame* pFrame, int device = 0) {
  const int N = pFrame->N;
  std::vector<uint8_t> has(N, 0);
  std::vector<float> obs((size_t)N * 3, 0.f), inv(N, 0.f), Xw((size_t)N * 3, 0.f);
  PoseOptimizationView f;
  f.mvbOutlier.assign(pFrame->mvbOutlier.begin(), pFrame->mvbOutlier.end());
  {
    std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);   // :806
    for (int i = 0; i < N; ++i) {
      MapPoint* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      has[i] = 1; f.mvbOutlier[i] = 0;                          // :817, :860: pFrame->mvbOutlier[i] = false
      const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
      obs[3 * i] = kpUn.pt.x; obs[3 * i + 1] = kpUn.pt.y; obs[3 * i + 2] = pFrame->mvuRight[i];   // < 0: monocular edge (:811)
      inv[i] = pFrame->mvInvLevelSigma2[kpUn.octave];
      const Eigen::Vector3f X = pMP->GetWorldPos();
      for (int k = 0; k < 3; ++k) Xw[3 * i + k] = X(k);
    }
  }
  f.N = N; f.hasMapPoint = has.data(); f.obs = obs.data(); f.invSigma2 = inv.data(); f.worldPos = Xw.data();
  f.fx = pFrame->fx; f.fy = pFrame->fy; f.cx = pFrame->cx; f.cy = pFrame->cy; f.mbf = pFrame->mbf;
  const Sophus::SE3f Tcw = pFrame->GetPose();                    // :781-783
  const Eigen::Quaternionf q = Tcw.unit_quaternion();
  f.pose[0] = q.x(); f.pose[1] = q.y(); f.pose[2] = q.z(); f.pose[3] = q.w();
  for (int k = 0; k < 3; ++k) f.pose[4 + k] = Tcw.translation()(k);
  const int nin = Optimizer::PoseOptimization(f, device);
  pFrame->SetPose(Sophus::SE3f(Eigen::Quaternionf(f.pose[3], f.pose[0], f.pose[1], f.pose[2]), Eigen::Vector3f(f.pose[4], f.pose[5], f.pose[6])));   // :1044-1048
  for (int i = 0; i < N; ++i) pFrame->mvbOutlier[i] = f.mvbOutlier[i] != 0;
  return nin;
}

// void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges)
// Optimizer.cc:1053-1441 (pinhole keyframes: monocular and stereo edges)
inline void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges,
                                  int device = 0) {
  // :1058-1122 — the reference's graph selection, unchanged
  std::list<KeyFrame*> lLocalKeyFrames;
  lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  Map* pCurrentMap = pKF->GetMap();
  const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (KeyFrame* pKFi : vNeighKFs) {
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(pKFi);
  }
  num_fixedKF = 0;
  std::list<MapPoint*> lLocalMapPoints;
  for (KeyFrame* pKFi : lLocalKeyFrames) {
    if (pKFi->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
    for (MapPoint* pMP : pKFi->GetMapPointMatches())
      if (pMP && !pMP->isBad() && pMP->GetMap() == pCurrentMap && pMP->mnBALocalForKF != pKF->mnId) {
        lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->mnId;
      }
  }
  std::list<KeyFrame*> lFixedCameras;
  for (MapPoint* pMP : lLocalMapPoints)
    for (const auto& ob : pMP->GetObservations()) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lFixedCameras.push_back(pKFi);
      }
    }
  num_fixedKF = (int)lFixedCameras.size() + num_fixedKF;
  if (num_fixedKF == 0) return;   // :1118-1122 "LBA aborted"
  // :1150-1351 — vertices and edges, flattened
  std::map<KeyFrame*, int> kfIndex;
  std::vector<KeyFrame*> kfs;
  std::vector<float> kfPose;
  std::vector<uint8_t> kfFixed;
  auto add_kf = [&](KeyFrame* k, bool fixed) {
    kfIndex[k] = (int)kfs.size(); kfs.push_back(k);
    const Sophus::SE3f Tcw = k->GetPose();
    const Eigen::Quaternionf q = Tcw.unit_quaternion();
    const float p[7] = {q.x(), q.y(), q.z(), q.w(), Tcw.translation()(0), Tcw.translation()(1), Tcw.translation()(2)};
    kfPose.insert(kfPose.end(), p, p + 7);
    kfFixed.push_back(fixed ? 1 : 0);
  };
  for (KeyFrame* k : lLocalKeyFrames) add_kf(k, k->mnId == pMap->GetInitKFid());   // :1160
  num_OptKF = (int)lLocalKeyFrames.size();
  for (KeyFrame* k : lFixedCameras) add_kf(k, true);
  std::vector<MapPoint*> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());
  std::vector<float> mpPos;
  std::vector<int> eKF, eMP;
  std::vector<float> eObs, eInv;
  std::vector<std::pair<KeyFrame*, MapPoint*>> eOwner;
  for (int j = 0; j < (int)mps.size(); ++j) {
    MapPoint* pMP = mps[j];
    const Eigen::Vector3f X = pMP->GetWorldPos();
    mpPos.push_back(X(0)); mpPos.push_back(X(1)); mpPos.push_back(X(2));
    for (const auto& ob : pMP->GetObservations()) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
      const int leftIndex = std::get<0>(ob.second);
      if (leftIndex == -1) continue;
      const auto it = kfIndex.find(pKFi);
      if (it == kfIndex.end()) continue;   // (a keyframe g2o has no vertex for: optimizer.vertex(id) would be NULL)
      const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
      eKF.push_back(it->second); eMP.push_back(j);
      eObs.push_back(kpUn.pt.x); eObs.push_back(kpUn.pt.y); eObs.push_back(pKFi->mvuRight[leftIndex]);   // < 0: EdgeSE3ProjectXYZ, else stereo
      eInv.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
      eOwner.emplace_back(pKFi, pMP);
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
  Optimizer::LocalBundleAdjustment(g, pbStopFlag, device);
  // :1366-1401: the library marks the observations the reference erases (chi2 > 5.991 / 7.815 or non-positive depth); bad points are skipped here
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);   // :1404
  for (size_t e = 0; e < eOwner.size(); ++e)
    if (g.eraseFlag[e] && !eOwner[e].second->isBad()) {
      eOwner[e].first->EraseMapPointMatch(eOwner[e].second);
      eOwner[e].second->EraseObservation(eOwner[e].first);
    }
  int i = 0;
  for (KeyFrame* k : lLocalKeyFrames) {   // :1416-1425
    const float* p = &kfPose[(size_t)7 * i++];
    k->SetPose(Sophus::SE3f(Eigen::Quaternionf(p[3], p[0], p[1], p[2]), Eigen::Vector3f(p[4], p[5], p[6])));
  }
  for (int j = 0; j < (int)mps.size(); ++j) {   // :1428-1436
    mps[j]->SetWorldPos(Eigen::Vector3f(mpPos[3 * j], mpPos[3 * j + 1], mpPos[3 * j + 2]));
    mps[j]->UpdateNormalAndDepth();
  }
  pMap->IncreaseChangeIndex();
}

}  // namespace morb_glue
}  // namespace ORB_SLAM3
#endif  // reference headers present
