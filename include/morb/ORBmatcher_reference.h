// The reference's own ORBmatcher signatures (include/ORBmatcher.h:41-114) as member templates of ORB_SLAM3::ORBmatcher — included at the
// end of ORBmatcher.h.  Templated on the Frame / KeyFrame / MapPoint / Sophus types so that this header names none of them: a call site of
// src/Tracking.cc, src/LocalMapping.cc or src/LoopClosing.cc (`matcher.SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches)`)
// compiles UNCHANGED against it and instantiates the member with the reference's types.  Each member fills the views of ORBmatcher.h
// from the objects (reference_glue.h), calls the view-taking overload / the C ABI, and writes the result back the way the reference's
// method does (cited per member).  See reference_glue.h for what has and has not been compiled.
//
// KannalaBrandt8 rigs (Frame::Nleft != -1 / KeyFrame::NLeft != -1): every member in which the reference has a rig branch dispatches to the rig form
// behind the C ABI — SearchByProjection(F, vpMapPoints), SearchByProjection(Cur, Last), SearchByBoW(pKF, F), SearchByBoW(pKF1, pKF2) (the
// mvKeysUn.size() bound, :734), SearchForTriangulation, Fuse(pKF, vpMapPoints, th, bRight).  Loop closing's members have NO rig branch in the reference
// and run there with the left camera's features (GetFeaturesInArea's bRight = false, mvKeysUn = mvKeys), projecting with mpCamera->project (the left
// KannalaBrandt8 camera: SearchByProjection(pKF, Scw, ...), Fuse(pKF, Scw, ...)) or with the pinhole formula on pKF->fx ... (that search's twin with
// vpPointsKFs, SearchBySim3): the keyframe views carry NLeft / the camera and the rig entry points do exactly that.  Two members throw
// std::runtime_error on a rig instead of running the pinhole form silently: relocalisation's SearchByProjection(Frame, KeyFrame), where the reference
// reads mvKeysUn[i] past its NLeft entries (:1807 with i < N), and SearchForInitialization (monocular initialisation only).
#pragma once
#include <set>

#include "reference_glue.h"

namespace ORB_SLAM3 {


// static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b)  ORBmatcher.cc:1880-1894
template <class Mat, class>
int ORBmatcher::DescriptorDistance(const Mat& a, const Mat& b) {
  return DescriptorDistance(a.template ptr<uint8_t>(0), b.template ptr<uint8_t>(0));
}

// int SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)  ORBmatcher.cc:42-209.
// Called by Tracking::SearchLocalPoints (Tracking.cc:3117-3183) AFTER its own isInFrustum loop, which stays as it is (it also does
// IncreaseVisible / mnLastFrameSeen): this member reads the tracking fields that loop left in the map points (mbTrackInView, mTrackProjX, ...).
template <class FrameT, class MP>
int ORBmatcher::SearchByProjection(FrameT& F, const std::vector<MP*>& vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints) {
  const int N = F.N, M = (int)vpMapPoints.size();
  if (N <= 0 || M <= 0) return 0;
  const bool rig = F.Nleft != -1;
  morb_glue::Store<FrameView> fs;
  morb_glue::fill_keys(fs, F, N, F.Nleft);
  morb_glue::fill_params(fs.v.params, F, FrameT::mfGridElementWidthInv, FrameT::mfGridElementHeightInv);
  std::vector<uint8_t> blocked(N, 0), inL(M, 0), inR(M, 0), bad(M, 1), hasObs(M, 0), desc((size_t)M * 32, 0);
  std::vector<float> depth(M, 0.f), pxL(M, 0.f), pyL(M, 0.f), pxrL(M, 0.f), cosL(M, 0.f), pxR(M, 0.f), pyR(M, 0.f), cosR(M, 0.f);
  std::vector<int> lvL(M, 0), lvR(M, 0);
  for (int i = 0; i < N; ++i) blocked[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;   // :93-94
  for (int j = 0; j < M; ++j) {
    MP* p = vpMapPoints[j];
    if (!p) continue;
    inL[j] = p->mbTrackInView ? 1 : 0; inR[j] = p->mbTrackInViewR ? 1 : 0;
    if (!inL[j] && !inR[j]) continue;                 // :52
    bad[j] = p->isBad() ? 1 : 0;                      // :56
    hasObs[j] = p->Observations() > 0 ? 1 : 0;
    depth[j] = p->mTrackDepth;
    pxL[j] = p->mTrackProjX; pyL[j] = p->mTrackProjY; pxrL[j] = p->mTrackProjXR; lvL[j] = p->mnTrackScaleLevel; cosL[j] = p->mTrackViewCos;
    pxR[j] = p->mTrackProjXR; pyR[j] = p->mTrackProjYR; lvR[j] = p->mnTrackScaleLevelR; cosR[j] = p->mTrackViewCosR;
    const auto d = p->GetDescriptor();
    std::memcpy(&desc[(size_t)j * 32], d.template ptr<uint8_t>(0), 32);
  }
  morb_adapter::StreamScope scope_(morb_matcher_stream(h_));
  Staging& s = staging();
  const int zero = 0;
  s.kp[0].assign(fs.kps.data(), N); s.u8[0].assign(fs.desc.data(), (size_t)N * 32); s.u8[1].assign(blocked.data(), N);
  s.i32[0].assign(&zero, 1); s.i32[1].assign(&N, 1); s.i32[2].assign(&M, 1); s.i32[3].resize(1); s.i32[3].fill_bytes(0);
  s.u8[2].assign(inL.data(), M); s.u8[3].assign(bad.data(), M); s.u8[4].assign(hasObs.data(), M); s.u8[5].assign(desc.data(), desc.size());
  s.f32[11].assign(depth.data(), M); s.f32[8].assign(pxL.data(), M); s.f32[9].assign(pyL.data(), M); s.f32[12].assign(cosL.data(), M);
  s.i32[4].assign(lvL.data(), M); s.i32[5].resize(N); s.i32[5].fill_bytes(0xFF);
  if (!rig) {
    if (fs.v.mvuRight) s.f32[0].assign(fs.v.mvuRight, N);
    s.f32[10].assign(pxrL.data(), M);
    check(morb_search_by_projection_mps_batch(h_, &fs.v.params, 1, s.i32[0].get(), N, s.i32[1].get(), s.kp[0].get(), s.u8[0].get(),
                                              fs.v.mvuRight ? s.f32[0].get() : nullptr, s.u8[1].get(), M, s.i32[2].get(), s.u8[2].get(), s.u8[3].get(),
                                              s.f32[11].get(), s.f32[8].get(), s.f32[9].get(), s.f32[10].get(), s.i32[4].get(), s.f32[12].get(),
                                              s.u8[5].get(), s.u8[4].get(), th, bFarPoints ? 1 : 0, thFarPoints, mfNNratio, s.i32[5].get(),
                                              s.i32[3].get(), nullptr));
  } else {
    // :97-133, :142-206: left pass + right pass, stereo partners through mvLeftToRightMatch / mvRightToLeftMatch
    std::vector<int> l2r(N, -1), r2l(N, -1);
    for (int i = 0; i < (int)F.mvLeftToRightMatch.size() && i < N; ++i) l2r[i] = F.mvLeftToRightMatch[i];
    for (int i = 0; i < (int)F.mvRightToLeftMatch.size() && i < N; ++i) r2l[i] = F.mvRightToLeftMatch[i];
    const int nl = F.Nleft;
    s.i32[6].assign(&nl, 1); s.i32[7].assign(l2r.data(), N);
    morb_adapter::DeviceBuffer<int>& dR2L = rigI32(0); dR2L.assign(r2l.data(), N);
    morb_adapter::DeviceBuffer<int>& dLvR = rigI32(1); dLvR.assign(lvR.data(), M);
    s.u8[6].assign(inR.data(), M); s.f32[10].assign(pxR.data(), M); s.f32[7].assign(pyR.data(), M); s.f32[6].assign(cosR.data(), M);
    check(morb_search_by_projection_mps_fisheye_batch(h_, &fs.v.params, 1, s.i32[0].get(), N, s.i32[1].get(), s.i32[6].get(), s.kp[0].get(), s.u8[0].get(),
                                                      s.i32[7].get(), dR2L.get(), s.u8[1].get(), M, s.i32[2].get(), s.u8[2].get(), s.u8[6].get(), s.u8[3].get(),
                                                      s.f32[11].get(), s.f32[8].get(), s.f32[9].get(), s.i32[4].get(), s.f32[12].get(), s.f32[10].get(),
                                                      s.f32[7].get(), dLvR.get(), s.f32[6].get(), s.u8[5].get(), s.u8[4].get(), th, bFarPoints ? 1 : 0,
                                                      thFarPoints, mfNNratio, s.i32[5].get(), s.i32[3].get(), nullptr));
  }
  sync();
  const std::vector<int> matchF = s.i32[5].to_host();
  for (int i = 0; i < N; ++i)
    if (matchF[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[matchF[i]];   // :132, :197
  return s.i32[3].to_host()[0];
}

// int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)  ORBmatcher.cc:1521-1733 (Tracking::TrackWithMotionModel)
template <class FrameT>
int ORBmatcher::SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
  FrameT& Last = const_cast<FrameT&>(LastFrame);   // (GetPose() and the map points' getters are not const in the reference)
  morb_glue::Store<FrameView> cur, last;
  morb_glue::frame_view(cur, CurrentFrame, morb_glue::kAny);
  morb_glue::frame_view(last, Last, morb_glue::kLastFrame);   // :1540-1542: pMP && !LastFrame.mvbOutlier[i], nothing else
  std::vector<int> matchCur(CurrentFrame.N, -1);
  int n;
  if (CurrentFrame.Nleft == -1) {
    n = SearchByProjection(static_cast<const FrameView&>(cur.v), static_cast<const FrameView&>(last.v), matchCur, th, bMono);
  } else {
    // fisheye current frame: left pass + right pass through GetRelativePoseTrl() with the LEFT camera model (:1577-1710)
    const int N = cur.v.N, NL = last.v.N;
    if (N <= 0 || NL <= 0) return 0;
    float c8[8], trl[7];
    morb_glue::cam8(CurrentFrame.mpCamera, c8);
    morb_glue::pose7(CurrentFrame.GetRelativePoseTrl(), trl);
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));
    Staging& s = staging();
    const int cap = load_pool(s, {&cur.v, &last.v});
    float tlc2 = last.v.mtcw[2];
    for (int k = 0; k < 3; ++k) tlc2 += last.v.mRcw[6 + k] * cur.v.mOw[k];   // :1536-1539
    const uint8_t fwd = (tlc2 > cur.v.params.mb && !bMono) ? 1 : 0, bwd = (-tlc2 > cur.v.params.mb && !bMono) ? 1 : 0;
    const int ci = 0, li = 1, nl = CurrentFrame.Nleft;
    s.i32[2].assign(&ci, 1); s.i32[3].assign(&li, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[6].assign(&nl, 1);
    s.u8[2].assign(&fwd, 1); s.u8[3].assign(&bwd, 1); s.f32[1].assign(cur.v.Tcw, 7);
    up_row(s.u8[4], cur.v.hasTrackedMapPoint, N, cap, 1); up_row(s.u8[5], last.v.hasMapPoint, NL, cap, 1);
    up_row(s.f32[2], last.v.mpWorldPos, NL, cap, 3); up_row(s.u8[6], last.v.mpDescriptor, NL, cap, 32); up_row(s.u8[7], last.v.mpHasObservations, NL, cap, 1);
    init_match(s.i32[5], matchCur, N, cap);
    check(morb_search_by_projection_last_fisheye_batch(h_, &cur.v.params, c8, trl, 1, s.i32[2].get(), s.i32[3].get(), s.i32[6].get(), cap, s.i32[0].get(),
                                                       s.kp[0].get(), s.u8[0].get(), s.u8[4].get(), s.f32[1].get(), s.u8[5].get(), s.f32[2].get(), s.u8[6].get(),
                                                       s.u8[7].get(), th, s.u8[2].get(), s.u8[3].get(), mbCheckOrientation ? 1 : 0, s.i32[5].get(),
                                                       s.i32[4].get(), nullptr));
    sync();
    matchCur = s.i32[5].to_host(); matchCur.resize(N);
    n = s.i32[4].to_host()[0];
  }
  for (int i = 0; i < CurrentFrame.N; ++i)
    if (matchCur[i] >= 0) CurrentFrame.mvpMapPoints[i] = Last.mvpMapPoints[matchCur[i]];   // :1627, :1692
  return n;
}

// int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)  ORBmatcher.cc:1735-1842 (relocalisation)
template <class FrameT, class KF, class MP>
int ORBmatcher::SearchByProjection(FrameT& CurrentFrame, KF* pKF, const std::set<MP*>& sAlreadyFound, const float th, const int ORBdist) {
  morb_glue::Store<FrameView> cur;
  morb_glue::Store<KeyFrameView> kf;
  morb_glue::frame_view(cur, CurrentFrame, morb_glue::kAny);        // :1790: CurrentFrame.mvpMapPoints[i2] non-NULL blocks the feature
  morb_glue::keyframe_view(kf, pKF, morb_glue::kSkipBad);           // :1753-1754
  const std::vector<MP*> vpMPs = pKF->GetMapPointMatches();
  std::vector<uint8_t> found(kf.v.N, 0);
  for (int i = 0; i < kf.v.N && i < (int)vpMPs.size(); ++i) found[i] = (vpMPs[i] && sAlreadyFound.count(vpMPs[i])) ? 1 : 0;
  std::vector<int> matchCur(CurrentFrame.N, -1);
  // a KannalaBrandt8 rig frame: the reference (no rig branch here) projects with mpCamera and searches the left grid (include/morb_hip.h:
  // morb_search_by_projection_kf_rig_batch, which also says what replaces the reference's out-of-bounds mvKeysUn[i] of a right keyframe feature)
  float camL[8];
  const bool rig = CurrentFrame.Nleft != -1;
  if (rig) morb_glue::cam8(CurrentFrame.mpCamera, camL);
  const int n = SearchByProjection(static_cast<const FrameView&>(cur.v), kf.v, found, matchCur, th, ORBdist, rig ? camL : nullptr,
                                   rig ? (int)CurrentFrame.Nleft : -1);
  for (int i = 0; i < CurrentFrame.N; ++i)
    if (matchCur[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[matchCur[i]];   // :1809
  return n;
}

// int SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, th, ratioHamming)
// ORBmatcher.cc:397-494 (loop detection)
template <class KF, class Sim3, class MP>
int ORBmatcher::SearchByProjection(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, std::vector<MP*>& vpMatched, int th, float ratioHamming) {
  return sim3_projection_ref(pKF, Scw, vpPoints, static_cast<const std::vector<KF*>*>(nullptr), vpMatched, static_cast<std::vector<KF*>*>(nullptr), th, ratioHamming);
}
// ... with vpPointsKFs / vpMatchedKF  ORBmatcher.cc:496-601 (place recognition)
template <class KF, class Sim3, class MP>
int ORBmatcher::SearchByProjection(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, const std::vector<KF*>& vpPointsKFs, std::vector<MP*>& vpMatched,
                                   std::vector<KF*>& vpMatchedKF, int th, float ratioHamming) {
  return sim3_projection_ref(pKF, Scw, vpPoints, &vpPointsKFs, vpMatched, &vpMatchedKF, th, ratioHamming);
}
template <class KF, class Sim3, class MP>
int ORBmatcher::sim3_projection_ref(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, const std::vector<KF*>* vpPointsKFs, std::vector<MP*>& vpMatched,
                                    std::vector<KF*>* vpMatchedKF, int th, float ratioHamming) {
  using SE3 = typename std::decay<decltype(pKF->GetPose())>::type;
  morb_glue::Store<KeyFrameView> kf;
  morb_glue::keyframe_view(kf, pKF);
  Sim3View sv;
  morb_glue::sim3_view<SE3>(sv, Scw);
  // :411-413: spAlreadyFound = the points already in vpMatched; a candidate that is bad or already found is skipped (:421)
  std::set<MP*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
  spAlreadyFound.erase(static_cast<MP*>(NULL));
  morb_glue::PointStore<MapPointView> ps;
  morb_glue::mappoint_view(ps, vpPoints);
  for (size_t i = 0; i < vpPoints.size(); ++i)
    if (vpPoints[i] && spAlreadyFound.count(vpPoints[i])) ps.valid[i] = 0;
  const int N = kf.v.N, M = (int)vpPoints.size();
  vpMatched.resize(N, static_cast<MP*>(NULL));
  // :471 `if (vpMatched[idx]) continue;`: a feature that already holds a match is blocked — marked with an index no candidate has
  const int kTaken = 0x7FFFFFFF;
  std::vector<int> idx(N, -1);
  for (int i = 0; i < N; ++i) if (vpMatched[i]) idx[i] = kTaken;
  const int n = sim3_projection(kf.v, sv, ps.v, idx, th, ratioHamming, vpPointsKFs ? 1 : 0);
  if (vpMatchedKF) vpMatchedKF->resize(N, static_cast<KF*>(NULL));
  for (int i = 0; i < N; ++i)
    if (idx[i] >= 0 && idx[i] < M) {
      vpMatched[i] = vpPoints[idx[i]];                                          // :483
      if (vpMatchedKF && vpPointsKFs) (*vpMatchedKF)[i] = (*vpPointsKFs)[idx[i]];   // :594
    }
  return n;
}

// int SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  ORBmatcher.cc:218-395 (TrackReferenceKeyFrame, relocalisation)
template <class KF, class FrameT, class MP, class>
int ORBmatcher::SearchByBoW(KF* pKF, FrameT& F, std::vector<MP*>& vpMapPointMatches) {
  morb_glue::Store<KeyFrameView> kf;
  morb_glue::Store<FrameView> fr;
  morb_glue::keyframe_view(kf, pKF);                 // :243-246: the keyframe feature's map point, skipped when NULL or bad
  morb_glue::frame_view(fr, F, morb_glue::kAny);
  const std::vector<MP*> vpMapPointsKF = pKF->GetMapPointMatches();
  std::vector<int> idx;
  int n;
  if (F.Nleft == -1) {
    n = SearchByBoW(kf.v, static_cast<const FrameView&>(fr.v), idx);
  } else {
    // :262-299, :333-365: left and right candidates of a node ranked separately
    if (kf.v.N <= 0 || fr.v.N <= 0) { vpMapPointMatches.assign(F.N > 0 ? F.N : 0, static_cast<MP*>(NULL)); return 0; }
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));
    Staging& s = staging();
    const int cap = load_pool(s, {&kf.v, &fr.v});
    const int ki = 0, fi = 1, nl = F.Nleft;
    s.i32[2].assign(&ki, 1); s.i32[3].assign(&fi, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap); s.i32[6].assign(&nl, 1);
    check(morb_search_by_bow_fisheye_batch(h_, 1, s.i32[2].get(), s.i32[3].get(), s.i32[6].get(), 2, s.kp[0].get(), s.u8[0].get(), s.i32[1].get(), s.i32[0].get(),
                                           s.u8[1].get(), cap, mfNNratio, mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    idx = s.i32[5].to_host(); idx.resize(F.N);
    n = s.i32[4].to_host()[0];
  }
  vpMapPointMatches.assign(F.N, static_cast<MP*>(NULL));   // :222
  for (int j = 0; j < F.N; ++j)
    if (idx[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[idx[j]];   // :316, :352
  return n;
}

// int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)  ORBmatcher.cc:702-819 (loop closing / merging)
template <class KF, class MP>
int ORBmatcher::SearchByBoW(KF* pKF1, KF* pKF2, std::vector<MP*>& vpMatches12) {
  morb_glue::Store<KeyFrameView> a, b;
  morb_glue::keyframe_view(a, pKF1); morb_glue::keyframe_view(b, pKF2);   // :738-742, :756-760: NULL or bad map points are skipped on both sides
  const std::vector<MP*> vpMapPoints2 = pKF2->GetMapPointMatches();
  std::vector<int> idx;
  const int n = SearchByBoW(a.v, b.v, idx);
  vpMatches12.assign(pKF1->GetMapPointMatches().size(), static_cast<MP*>(NULL));   // :710
  for (int i = 0; i < (int)idx.size() && i < (int)vpMatches12.size(); ++i)
    if (idx[i] >= 0) vpMatches12[i] = vpMapPoints2[idx[i]];   // :783
  return n;
}

// int SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, windowSize)  ORBmatcher.cc:603-700
template <class FrameT, class Pt, class>
int ORBmatcher::SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<Pt>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize) {
  // (rig frames: the member has no rig branch — it walks F1.mvKeysUn (the left keypoints), F2's left grid and descriptor rows [0, Nleft): the
  // left-camera search, which is what the views below hold)
  const int n1 = F1.Nleft != -1 ? (int)F1.Nleft : (int)F1.N, n2 = F2.Nleft != -1 ? (int)F2.Nleft : (int)F2.N;
  morb_glue::Store<FrameView> a, b;
  morb_glue::fill_keys(a, F1, n1, F1.Nleft); morb_glue::fill_keys(b, F2, n2, F2.Nleft);
  morb_glue::fill_params(a.v.params, F1, FrameT::mfGridElementWidthInv, FrameT::mfGridElementHeightInv);
  b.v.params = a.v.params;
  std::vector<float> prev((size_t)n1 * 2, 0.f);
  for (int i = 0; i < n1 && i < (int)vbPrevMatched.size(); ++i) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
  const int n = SearchForInitialization(static_cast<const FrameView&>(a.v), static_cast<const FrameView&>(b.v), prev, vnMatches12, windowSize);
  for (int i = 0; i < n1 && i < (int)vbPrevMatched.size(); ++i) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }   // :696-698
  return n;
}

// int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<pair<size_t, size_t>>& vMatchedPairs, bOnlyStereo, bCoarse)  ORBmatcher.cc:821-1042
template <class KF>
int ORBmatcher::SearchForTriangulation(KF* pKF1, KF* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse) {
  morb_glue::Store<KeyFrameView> a, b;
  morb_glue::keyframe_view(a, pKF1, morb_glue::kAny); morb_glue::keyframe_view(b, pKF2, morb_glue::kAny);   // :876, :903: GetMapPoint(idx) != NULL, bad or not
  if (pKF1->mpCamera2 || pKF2->mpCamera2) {
    // KannalaBrandt8 rig (:845-852, :884, :925, :934-975): left features then right ones in one row, the four relative poses of the sides
    vMatchedPairs.clear();
    if (a.v.N <= 0 || b.v.N <= 0) return 0;
    const auto T1w = pKF1->GetPose();
    const auto Tw2 = pKF2->GetPoseInverse();
    const auto Tr1w = pKF1->GetRightPose();
    const auto Twr2 = pKF2->GetRightPoseInverse();
    float T4[4][12], camL[8], camR[8];
    morb_glue::rt12(T1w * Tw2, T4[0]); morb_glue::rt12(T1w * Twr2, T4[1]); morb_glue::rt12(Tr1w * Tw2, T4[2]); morb_glue::rt12(Tr1w * Twr2, T4[3]);
    morb_glue::cam8(pKF1->mpCamera, camL); morb_glue::cam8(pKF1->mpCamera2, camR);
    morb_adapter::StreamScope scope_(morb_matcher_stream(h_));
    Staging& s = staging();
    const int cap = load_pool(s, {&a.v, &b.v});
    const int i1 = 0, i2 = 1, nl1 = pKF1->NLeft, nl2 = pKF2->NLeft;
    s.i32[2].assign(&i1, 1); s.i32[3].assign(&i2, 1); s.i32[4].resize(1); s.i32[4].fill_bytes(0); s.i32[5].resize(cap); s.i32[6].assign(&nl1, 1); s.i32[7].assign(&nl2, 1);
    check(morb_search_for_triangulation_fisheye_batch(h_, &a.v.params, 1, s.i32[2].get(), s.i32[3].get(), s.i32[6].get(), s.i32[7].get(), 2, cap, s.i32[0].get(),
                                                      s.kp[0].get(), s.u8[0].get(), s.i32[1].get(), s.u8[1].get(), camL, camR, &T4[0][0], bOnlyStereo ? 1 : 0,
                                                      bCoarse ? 1 : 0, mbCheckOrientation ? 1 : 0, s.i32[5].get(), s.i32[4].get(), nullptr));
    sync();
    const std::vector<int> m12 = s.i32[5].to_host();
    for (int i = 0; i < a.v.N; ++i)
      if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);   // :1030-1036
    return s.i32[4].to_host()[0];
  }
  // :829-838
  const auto T1w = pKF1->GetPose();
  const auto T2w = pKF2->GetPose();
  const auto Tw2 = pKF2->GetPoseInverse();
  const auto Cw = pKF1->GetCameraCenter();
  const auto C2 = T2w * Cw;
  const auto ep = pKF2->mpCamera->project(C2);
  const auto T12 = T1w * Tw2;
  const auto R12 = T12.rotationMatrix();
  const auto t12 = T12.translation();
  float R[9], t[3], e[2] = {ep(0), ep(1)};
  for (int r = 0; r < 3; ++r) { t[r] = t12(r); for (int c = 0; c < 3; ++c) R[3 * r + c] = R12(r, c); }
  return SearchForTriangulation(a.v, b.v, R, t, e, vMatchedPairs, bOnlyStereo, bCoarse);
}

// int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, th)  ORBmatcher.cc:1323-1519
template <class KF, class MP, class Sim3>
int ORBmatcher::SearchBySim3(KF* pKF1, KF* pKF2, std::vector<MP*>& vpMatches12, const Sim3& S12, const float th) {
  morb_glue::Store<KeyFrameView> a, b;
  morb_glue::keyframe_view(a, pKF1); morb_glue::keyframe_view(b, pKF2);
  const std::vector<MP*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
  // :1356-1369: vbAlreadyMatched1[i] = vpMatches12[i] != NULL, vbAlreadyMatched2[idx2] through pMP->GetIndexInKeyFrame(pKF2)
  std::vector<int> idx(N1, -1);
  for (int i = 0; i < N1 && i < (int)vpMatches12.size(); ++i)
    if (vpMatches12[i]) {
      const int idx2 = std::get<0>(vpMatches12[i]->GetIndexInKeyFrame(pKF2));
      if (idx2 >= 0 && idx2 < N2) idx[i] = idx2;
      else a.hasMP[i] = 0;   // already matched, partner not in pKF2: blocked on side 1 only
    }
  float s12[7], s21[7];
  morb_glue::sim3_raw(S12, s12); morb_glue::sim3_raw(S12.inverse(), s21);
  const std::vector<int> before = idx;
  const int n = SearchBySim3(a.v, b.v, idx, s12, s21, th);
  vpMatches12.resize(N1, static_cast<MP*>(NULL));
  for (int i = 0; i < N1; ++i)
    if (idx[i] >= 0 && before[i] < 0) vpMatches12[i] = vpMapPoints2[idx[i]];   // :1506-1515
  return n;
}

// int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th, bRight)  ORBmatcher.cc:1044-1213 (LocalMapping::SearchInNeighbors)
template <class KF, class MP>
int ORBmatcher::Fuse(KF* pKF, const std::vector<MP*>& vpMapPoints, const float th, const bool bRight) {
  morb_glue::Store<KeyFrameView> kf;
  morb_glue::keyframe_view(kf, pKF);
  morb_glue::PointStore<MapPointView> ps;
  morb_glue::mappoint_view(ps, vpMapPoints);
  for (size_t i = 0; i < vpMapPoints.size(); ++i)
    if (vpMapPoints[i] && vpMapPoints[i]->IsInKeyFrame(pKF)) ps.valid[i] = 0;   // :1076-1089
  std::vector<int> bestIdx, bestDist;
  if (pKF->NLeft != -1) {   // KannalaBrandt8 rig: the side's pose, centre, camera and features (:1050-1058, :1130-1147, :1177)
    RigSide side;
    const auto O = bRight ? pKF->GetRightCameraCenter() : pKF->GetCameraCenter();
    morb_glue::pose7(bRight ? pKF->GetRightPose() : pKF->GetPose(), side.Tcw);
    for (int k = 0; k < 3; ++k) side.Ow[k] = O(k);
    morb_glue::cam8(bRight ? pKF->mpCamera2 : pKF->mpCamera, side.cam8);
    side.jLo = bRight ? pKF->NLeft : 0; side.jHi = bRight ? pKF->N : pKF->NLeft;
    Fuse(kf.v, ps.v, bestIdx, bestDist, th, &side);
  } else {
    if (bRight) throw std::runtime_error("Fuse: bRight on a keyframe without a right camera (NLeft == -1)");   // (the reference would dereference mpCamera2 == NULL)
    Fuse(kf.v, ps.v, bestIdx, bestDist, th);
  }
  int nFused = 0;
  for (size_t i = 0; i < vpMapPoints.size(); ++i) {
    if (bestIdx[i] < 0) continue;
    MP* pMP = vpMapPoints[i];
    // the search ran on the state before the loop; an earlier iteration's Replace / AddMapPoint may have changed this point's
    if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
    MP* pMPinKF = pKF->GetMapPoint(bestIdx[i]);   // :1196-1208
    if (pMPinKF) {
      if (!pMPinKF->isBad()) {
        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
        else pMPinKF->Replace(pMP);
      }
    } else {
      pMP->AddObservation(pKF, bestIdx[i]);
      pKF->AddMapPoint(pMP, bestIdx[i]);
    }
    ++nFused;
  }
  return nFused;
}

// int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, th, vector<MapPoint*>& vpReplacePoint)  ORBmatcher.cc:1215-1321
template <class KF, class Sim3, class MP>
int ORBmatcher::Fuse(KF* pKF, Sim3& Scw, const std::vector<MP*>& vpPoints, float th, std::vector<MP*>& vpReplacePoint) {
  using SE3 = typename std::decay<decltype(pKF->GetPose())>::type;
  morb_glue::Store<KeyFrameView> kf;
  morb_glue::keyframe_view(kf, pKF);
  Sim3View sv;
  morb_glue::sim3_view<SE3>(sv, Scw);
  const auto spAlreadyFound = pKF->GetMapPoints();   // :1237
  morb_glue::PointStore<MapPointView> ps;
  morb_glue::mappoint_view(ps, vpPoints);
  for (size_t i = 0; i < vpPoints.size(); ++i)
    if (vpPoints[i] && spAlreadyFound.count(vpPoints[i])) ps.valid[i] = 0;   // :1248
  std::vector<int> bestIdx, bestDist;
  Fuse(kf.v, sv, ps.v, th, bestIdx, bestDist);
  int nFused = 0;
  for (size_t i = 0; i < vpPoints.size(); ++i) {
    if (bestIdx[i] < 0) continue;
    MP* pMP = vpPoints[i];
    MP* pMPinKF = pKF->GetMapPoint(bestIdx[i]);   // :1307-1316
    if (pMPinKF) {
      if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
    } else {
      pMP->AddObservation(pKF, bestIdx[i]);
      pKF->AddMapPoint(pMP, bestIdx[i]);
    }
    ++nFused;
  }
  return nFused;
}

}  // namespace ORB_SLAM3
