// Drives the REFERENCE-TYPED members of include/morb/ORBmatcher.h / Optimizer.h (the exact signatures of the reference's
// include/ORBmatcher.h:41-114 and Optimizer.h:86) with mock Frame / KeyFrame / MapPoint objects (tests/native/mock_ref) filled from the
// SAME input files matcher_adapters_check.cc / adapters_check.cc read, and dumps what they leave in the objects as index tables
// (out_ref_*).  tests/test_adapter_matcher_gpu.py / test_adapter_gpu.py require out_ref_* == out_* (the view-taking adapters' results, which they
// compare with the oracle): the gather and write-back code of ORBmatcher_reference.h / Optimizer_reference.h is thereby checked end to end.
//   reference_members_check <dir> matcher|tracking|rig|inertial|iba
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "Frame.h"      // tests/native/mock_ref
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "ORBmatcher.h"   // include/morb
#include "Optimizer.h"

using namespace ORB_SLAM3;
namespace ORB_SLAM3 {
float Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv,
    Frame::mfGridElementHeightInv;
std::mutex MapPoint::mGlobalMutex;
}

static std::string g_dir;
template <typename T>
static std::vector<T> load(const std::string& name, bool optional = false) {
  std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
  if (!f) { if (optional) return {}; std::fprintf(stderr, "missing %s\n", name.c_str()); std::exit(3); }
  const size_t bytes = (size_t)f.tellg();
  std::vector<T> v(bytes / sizeof(T));
  f.seekg(0); f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)bytes);
  return v;
}
template <typename T>
static void dump(const std::string& name, const T* p, size_t n) {
  std::ofstream f(g_dir + "/out_ref_" + name + ".bin", std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

static std::vector<std::unique_ptr<MapPoint>> g_points;   // owns every mock map point
static MapPoint* new_point(const float* pos, const uint8_t* desc, float maxD, float minD, int nObs) {
  g_points.emplace_back(new MapPoint());
  MapPoint* p = g_points.back().get();
  if (pos) p->mWorldPos = Eigen::Vector3f(pos[0], pos[1], pos[2]);
  p->mDescriptor.data.assign(32, 0);
  if (desc) p->mDescriptor.data.assign(desc, desc + 32);
  p->mock_set_distances(maxD, minD);
  p->nObs = nObs;
  return p;
}
static Sophus::SE3f canned_pose(const std::vector<float>& pose22) {
  Sophus::SE3f T;
  if (pose22.size() == 22) {
    for (int i = 0; i < 9; ++i) T.R[i] = pose22[i];
    for (int i = 0; i < 3; ++i) { T.t[i] = pose22[9 + i]; T.Ow[i] = pose22[12 + i]; }
    for (int i = 0; i < 4; ++i) T.q[i] = pose22[15 + i];
  }
  return T;
}
// the per-feature arrays of a test frame (files written by the Python side)
struct Arrays {
  std::vector<morb_keypoint> kps; std::vector<uint8_t> desc, tracked, hasmp, mpdesc, mpobs; std::vector<float> uright, mppos, mpmax, mpmin, pose;
  std::vector<int> node, nvalid; std::vector<morb_frame_params> prm;
  explicit Arrays(const std::string& p) {
    kps = load<morb_keypoint>(p + "_kps"); desc = load<uint8_t>(p + "_desc"); uright = load<float>(p + "_uright", true);
    tracked = load<uint8_t>(p + "_tracked", true); hasmp = load<uint8_t>(p + "_hasmp", true); mpdesc = load<uint8_t>(p + "_mpdesc", true);
    mpobs = load<uint8_t>(p + "_mpobs", true); mppos = load<float>(p + "_mppos", true); mpmax = load<float>(p + "_mpmax", true);
    mpmin = load<float>(p + "_mpmin", true); pose = load<float>(p + "_pose", true); node = load<int>(p + "_node", true);
    nvalid = load<int>(p + "_nvalid", true); prm = load<morb_frame_params>(p + "_params");
  }
  // feature i's map point as the reference would hold it: NULL unless the test marks one; Observations() from tracked / mpobs
  std::vector<MapPoint*> points() const {
    const int N = (int)kps.size();
    std::vector<MapPoint*> v(N, nullptr);
    for (int i = 0; i < N; ++i) {
      const bool has = !hasmp.empty() && hasmp[i], trk = !tracked.empty() && tracked[i];
      if (!has && !trk) continue;
      const int nObs = trk ? 1 : (mpobs.empty() ? 1 : (int)mpobs[i]);
      v[i] = new_point(mppos.empty() ? nullptr : &mppos[3 * i], mpdesc.empty() ? nullptr : &mpdesc[(size_t)32 * i], mpmax.empty() ? 1.f : mpmax[i],
                       mpmin.empty() ? 1.f : mpmin[i], nObs);
    }
    return v;
  }
  template <class F> void common(F& f) const {
    const int N = (int)kps.size();
    f.N = N;
    f.mvKeysUn.resize(N);
    for (int i = 0; i < N; ++i) { cv::KeyPoint& k = f.mvKeysUn[i]; k.pt.x = kps[i].x; k.pt.y = kps[i].y; k.size = kps[i].size; k.angle = kps[i].angle; k.response = kps[i].response; k.octave = kps[i].octave; k.class_id = kps[i].class_id; }
    f.mvKeys = f.mvKeysUn;
    f.mDescriptors.data = desc; f.mDescriptors.rows = N;
    f.mvuRight = uright.empty() ? std::vector<float>(N, -1.f) : uright;
    for (int i = 0; i < (int)node.size(); ++i) if (node[i] >= 0) f.mFeatVec[(unsigned)node[i]].push_back((unsigned)i);
    const morb_frame_params& P = prm[0];
    f.mbf = P.mbf; f.mb = P.mb; f.mfLogScaleFactor = P.logScaleFactor; f.mnScaleLevels = P.nlevels;
    f.mvScaleFactors.assign(P.scaleFactors, P.scaleFactors + P.nlevels); f.mvLevelSigma2.assign(P.levelSigma2, P.levelSigma2 + P.nlevels);
    f.mvInvLevelSigma2.resize(P.nlevels);
    for (int l = 0; l < P.nlevels; ++l) f.mvInvLevelSigma2[l] = 1.0f / P.levelSigma2[l];
  }
  void fill(Frame& f) const {
    common(f);
    const morb_frame_params& P = prm[0];
    Frame::fx = P.fx; Frame::fy = P.fy; Frame::cx = P.cx; Frame::cy = P.cy; Frame::mnMinX = P.minX; Frame::mnMaxX = P.maxX; Frame::mnMinY = P.minY;
    Frame::mnMaxY = P.maxY; Frame::mfGridElementWidthInv = P.gridInvW; Frame::mfGridElementHeightInv = P.gridInvH;
    f.mvpMapPoints = points(); f.mvbOutlier.assign(f.N, false);
    if (pose.size() == 22) f.SetPose(canned_pose(pose));
  }
  void fill(KeyFrame& k) const {
    common(k);
    const morb_frame_params& P = prm[0];
    k.fx = P.fx; k.fy = P.fy; k.cx = P.cx; k.cy = P.cy; k.mnMinX = (int)P.minX; k.mnMaxX = (int)P.maxX; k.mnMinY = (int)P.minY; k.mnMaxY = (int)P.maxY;
    k.mfGridElementWidthInv = P.gridInvW; k.mfGridElementHeightInv = P.gridInvH;
    k.mvpMapPoints = points();
    if (!nvalid.empty() && nvalid[0] < k.N) k.mvKeysUn.resize(nvalid[0]);   // (a fisheye keyframe's shorter mvKeysUn, ORBmatcher.cc:734)
    k.mTcw = canned_pose(pose);
  }
};
template <class V> static std::vector<int> index_of(const std::vector<MapPoint*>& got, const V& pool) {
  std::vector<int> out(got.size(), -1);
  for (size_t i = 0; i < got.size(); ++i)
    if (got[i]) for (size_t j = 0; j < pool.size(); ++j) if (pool[j] == got[i]) { out[i] = (int)j; break; }
  return out;
}

static int run_matcher() {
  {   // int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
    Arrays ac("last_cur"), al("last_last");
    Frame cur, last; ac.fill(cur); al.fill(last);
    const auto cfg = load<float>("last_cfg");
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    const std::vector<MapPoint*> before = cur.mvpMapPoints;
    const int n = matcher.SearchByProjection(cur, last, cfg[2], cfg[3] != 0);
    std::vector<int> match = index_of(cur.mvpMapPoints, last.mvpMapPoints);
    dump("last_n", &n, 1); dump("last_match", match.data(), match.size());
  }
  {   // int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist)
    Arrays ac("kfp_cur"), ak("kfp_kf");
    Frame cur; KeyFrame kf; ac.fill(cur); ak.fill(kf);
    const auto cfg = load<float>("kfp_cfg");
    const auto found = load<uint8_t>("kfp_found");
    std::set<MapPoint*> sFound;
    for (int i = 0; i < kf.N; ++i) if (found[i] && kf.mvpMapPoints[i]) sFound.insert(kf.mvpMapPoints[i]);
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    const int n = matcher.SearchByProjection(cur, &kf, sFound, cfg[2], (int)cfg[3]);
    std::vector<int> match = index_of(cur.mvpMapPoints, kf.mvpMapPoints);
    dump("kfp_n", &n, 1); dump("kfp_match", match.data(), match.size());
  }
  {   // int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) and SearchByBoW(pKF1, pKF2, vpMatches12)
    Arrays ak("bow_kf"), af("bow_f");
    KeyFrame kf; Frame fr; ak.fill(kf); af.fill(fr);
    const auto cfg = load<float>("bow_cfg");
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    std::vector<MapPoint*> vpMapPointMatches;
    int n = matcher.SearchByBoW(&kf, fr, vpMapPointMatches);
    std::vector<int> match = index_of(vpMapPointMatches, kf.mvpMapPoints);
    dump("bow_n", &n, 1); dump("bow_match", match.data(), match.size());
    Arrays a1("bowkk_1"), a2("bowkk_2");
    KeyFrame k1, k2; a1.fill(k1); a2.fill(k2);
    std::vector<MapPoint*> vpMatches12;
    n = matcher.SearchByBoW(&k1, &k2, vpMatches12);
    match = index_of(vpMatches12, k2.mvpMapPoints);
    dump("bowkk_n", &n, 1); dump("bowkk_match", match.data(), match.size());
  }
  {   // int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize)
    Arrays a1("ini_1"), a2("ini_2");
    Frame f1, f2; a1.fill(f1); a2.fill(f2);
    const auto cfg = load<float>("ini_cfg");
    const auto prev = load<float>("ini_prev");
    std::vector<cv::Point2f> vbPrevMatched(prev.size() / 2);
    for (size_t i = 0; i < vbPrevMatched.size(); ++i) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    std::vector<int> vnMatches12;
    const int n = matcher.SearchForInitialization(f1, f2, vbPrevMatched, vnMatches12, (int)cfg[2]);
    std::vector<float> flat;
    for (auto& p : vbPrevMatched) { flat.push_back(p.x); flat.push_back(p.y); }
    dump("ini_n", &n, 1); dump("ini_match", vnMatches12.data(), vnMatches12.size()); dump("ini_prev", flat.data(), flat.size());
  }
  {   // int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vMatchedPairs, const bool bOnlyStereo, const bool bCoarse)
    Arrays a1("tri_1"), a2("tri_2");
    KeyFrame k1, k2; a1.fill(k1); a2.fill(k2);
    const auto cfg = load<float>("tri_cfg");   // nnratio, checkOri, bOnlyStereo, bCoarse, R12 (9), t12 (3), ep (2)
    for (int i = 0; i < 9; ++i) k1.mTcw.R[i] = cfg[4 + i];     // canned: T1w = (R12, t12), Tw2 = identity -> T1w * Tw2 = T12
    for (int i = 0; i < 3; ++i) k1.mTcw.t[i] = cfg[13 + i];
    GeometricCamera cam; cam.ep.v[0] = cfg[16]; cam.ep.v[1] = cfg[17];
    k1.mpCamera = k2.mpCamera = &cam;
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    std::vector<std::pair<size_t, size_t>> pairs;
    const int n = matcher.SearchForTriangulation(&k1, &k2, pairs, cfg[2] != 0, cfg[3] != 0);
    std::vector<int> flat;
    for (auto& p : pairs) { flat.push_back((int)p.first); flat.push_back((int)p.second); }
    dump("tri_n", &n, 1); dump("tri_pairs", flat.data(), flat.size());
  }
  {   // Fuse x2 and SearchByProjection(pKF, Scw, ...) x2
    Arrays ak("lc_kf");
    const auto pos = load<float>("lc_pts_pos"), nrm = load<float>("lc_pts_normal"), maxd = load<float>("lc_pts_maxd"), mind = load<float>("lc_pts_mind");
    const auto desc = load<uint8_t>("lc_pts_desc"), valid = load<uint8_t>("lc_pts_valid");
    const auto cfg = load<float>("lc_cfg");   // nnratio, checkOri, thFuse, thFuseSim3, thProj, ratioProj
    const auto sim = load<float>("lc_sim3");  // Tcw 7, Ow 3
    const int M = (int)maxd.size();
    auto make_points = [&]() {
      std::vector<MapPoint*> v(M);
      for (int i = 0; i < M; ++i) {
        v[i] = new_point(&pos[3 * i], &desc[(size_t)32 * i], maxd[i], mind[i], 1);
        v[i]->mNormalVector = Eigen::Vector3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
        v[i]->mbBad = valid[i] == 0;   // valid = non-NULL && !isBad() && !IsInKeyFrame(pKF)
      }
      return v;
    };
    Sophus::Sim3f Scw;
    for (int i = 0; i < 4; ++i) Scw.T.q[i] = sim[i];
    for (int i = 0; i < 3; ++i) { Scw.T.t[i] = sim[4 + i]; Scw.T.Ow[i] = sim[7 + i]; }
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    auto slots = [&](KeyFrame& kf, const std::vector<MapPoint*>& pts, const char* tag, int n) {   // feature -> the point the call put there
      std::vector<int> s = index_of(kf.mvpMapPoints, pts);
      dump(std::string(tag) + "_n", &n, 1); dump(std::string(tag) + "_slots", s.data(), s.size());
    };
    {
      KeyFrame kf; ak.fill(kf);
      const auto kfpose = load<float>("lc_kf_pose");
      kf.mTcw = canned_pose(kfpose);
      std::vector<MapPoint*> pts = make_points();
      const int n = matcher.Fuse(&kf, pts, cfg[2]);
      slots(kf, pts, "fuse", n);
    }
    {
      KeyFrame kf; ak.fill(kf);
      std::vector<MapPoint*> pts = make_points();
      std::vector<MapPoint*> vpReplacePoint(M, static_cast<MapPoint*>(NULL));
      const int n = matcher.Fuse(&kf, Scw, pts, cfg[3], vpReplacePoint);
      slots(kf, pts, "fuse3", n);
    }
    const auto matched = load<int>("lc_matched");
    for (int variant = 0; variant < 2; ++variant) {
      KeyFrame kf; ak.fill(kf);
      std::vector<MapPoint*> pts = make_points();
      MapPoint* taken = new_point(nullptr, nullptr, 1.f, 1.f, 1);   // what already sits in vpMatched on entry
      std::vector<MapPoint*> vpMatched(kf.N, static_cast<MapPoint*>(NULL));
      for (int i = 0; i < kf.N; ++i) if (matched[i] >= 0) vpMatched[i] = taken;
      std::vector<KeyFrame*> vpPointsKFs(M, &kf), vpMatchedKF;
      const int n = variant == 0 ? matcher.SearchByProjection(&kf, Scw, pts, vpMatched, (int)cfg[4], cfg[5])
                                 : matcher.SearchByProjection(&kf, Scw, pts, vpPointsKFs, vpMatched, vpMatchedKF, (int)cfg[4], cfg[5]);
      std::vector<int> idx = index_of(vpMatched, pts);
      for (int i = 0; i < kf.N; ++i) if (vpMatched[i] == taken) idx[i] = matched[i];
      const char* tag = variant == 0 ? "sim3p" : "sim3k";
      dump(std::string(tag) + "_n", &n, 1); dump(std::string(tag) + "_match", idx.data(), idx.size());
    }
  }
  {   // int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th)
    Arrays a1("s3_1"), a2("s3_2");
    KeyFrame k1, k2; a1.fill(k1); a2.fill(k2);
    const auto cfg = load<float>("s3_cfg");   // nnratio, checkOri, th, S12 (7), S21 (7)
    Sophus::Sim3f S12, S21;
    for (int i = 0; i < 7; ++i) { S12.raw[i] = cfg[3 + i]; S21.raw[i] = cfg[10 + i]; }
    S12.inv = &S21;
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    std::vector<MapPoint*> vpMatches12(k1.N, static_cast<MapPoint*>(NULL));
    const int n = matcher.SearchBySim3(&k1, &k2, vpMatches12, S12, cfg[2]);
    std::vector<int> m12 = index_of(vpMatches12, k2.mvpMapPoints);
    dump("s3_n", &n, 1); dump("s3_match", m12.data(), m12.size());
  }
  std::printf("reference members (matcher) ok\n");
  return 0;
}

// SearchForTriangulation between two keyframes of a KannalaBrandt8 rig (KeyFrame::NLeft != -1, mpCamera2 set): ORBmatcher.cc:821-1040
static int run_rig() {
  Arrays a1("rig_1"), a2("rig_2");
  const auto cfg = load<float>("rig_cfg");   // nnratio, checkOri, bOnlyStereo, bCoarse, NLeft1, NLeft2
  const auto cams = load<float>("rig_cams");   // left 8, right 8
  const auto poses = load<float>("rig_poses");   // T1w, Tr1w, Tw2, Twr2: R (9 row-major) + t (3) each
  GeometricCamera camL, camR;
  camL.mvParameters.assign(cams.begin(), cams.begin() + 8); camR.mvParameters.assign(cams.begin() + 8, cams.begin() + 16);
  auto rt = [&](int i) { Sophus::SE3f T; for (int k = 0; k < 9; ++k) T.R[k] = poses[12 * i + k]; for (int k = 0; k < 3; ++k) T.t[k] = poses[12 * i + 9 + k]; return T; };
  auto split = [&](KeyFrame& k, int nl) {   // mvKeys | mvKeysRight as Frame's fisheye constructor leaves them (Frame.cc:1160-1162)
    k.NLeft = nl; k.NRight = k.N - nl;
    k.mvKeysRight.assign(k.mvKeys.begin() + nl, k.mvKeys.end()); k.mvKeys.resize(nl); k.mvKeysUn = k.mvKeys;
    k.mpCamera = &camL; k.mpCamera2 = &camR;
  };
  for (int variant = 0; variant < 2; ++variant) {
    KeyFrame k1, k2; a1.fill(k1); a2.fill(k2);
    split(k1, (int)cfg[4]); split(k2, (int)cfg[5]);
    k1.mTcw = rt(0); k1.mTrw = rt(1);
    const Sophus::SE3f Tw2 = rt(2);
    k2.mTcw = Sophus::SE3f(); for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) k2.mTcw.R[3 * r + c] = Tw2.R[3 * c + r];
    for (int k = 0; k < 3; ++k) k2.mTcw.Ow[k] = Tw2.t[k];   // (GetPoseInverse() of the mock: R transposed, Ow carried)
    k2.mTwr = rt(3);
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    std::vector<std::pair<size_t, size_t>> pairs;
    const int n = matcher.SearchForTriangulation(&k1, &k2, pairs, variant == 1, cfg[3] != 0);
    std::vector<int> flat;
    for (auto& p : pairs) { flat.push_back((int)p.first); flat.push_back((int)p.second); }
    const std::string tag = variant == 0 ? "rig" : "rig_stereo";
    dump(tag + "_n", &n, 1); dump(tag + "_pairs", flat.data(), flat.size());
  }
  for (int right = 0; right < 2; ++right) {   // int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th, const bool bRight) on the rig keyframe
    const std::string t = right ? "rigf_r" : "rigf_l";
    const auto pos = load<float>(t + "_pos"), nrm = load<float>(t + "_normal"), maxd = load<float>(t + "_maxd"), mind = load<float>(t + "_mind");
    const auto desc = load<uint8_t>(t + "_desc"), valid = load<uint8_t>(t + "_valid");
    const auto side = load<float>(t + "_pose");   // Tcw (quaternion xyzw, t), Ow of the camera searched
    const int M = (int)maxd.size();
    KeyFrame kf; a2.fill(kf);
    split(kf, (int)cfg[5]);
    Sophus::SE3f& T = right ? kf.mTrw : kf.mTcw;
    for (int i = 0; i < 4; ++i) T.q[i] = side[i];
    for (int i = 0; i < 3; ++i) { T.t[i] = side[4 + i]; T.Ow[i] = side[7 + i]; }
    std::vector<MapPoint*> pts(M);
    for (int i = 0; i < M; ++i) {
      pts[i] = new_point(&pos[3 * i], &desc[(size_t)32 * i], maxd[i], mind[i], 1);
      pts[i]->mNormalVector = Eigen::Vector3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
      pts[i]->mbBad = valid[i] == 0;
    }
    ORBmatcher matcher(cfg[0], cfg[1] != 0);
    const int n = matcher.Fuse(&kf, pts, 3.0f, right != 0);
    const std::vector<int> slots = index_of(kf.mvpMapPoints, pts);
    dump(t + "_n", &n, 1); dump(t + "_slots", slots.data(), slots.size());
  }
  {   // loop closing on the rig keyframe: Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) and SearchByProjection(pKF, Scw, ...) x 2 — no rig branch in the
      // reference: the left features, mpCamera->project (the twin: the pinhole formula on pKF->fx ...)
    const auto pos = load<float>("rlc_pos"), nrm = load<float>("rlc_normal"), maxd = load<float>("rlc_maxd"), mind = load<float>("rlc_mind");
    const auto desc = load<uint8_t>("rlc_desc"), valid = load<uint8_t>("rlc_valid");
    const auto sim = load<float>("rlc_sim3");   // Tcw 7, Ow 3
    const auto matched = load<int>("rlc_matched");
    const int M = (int)maxd.size();
    auto make_points = [&]() {
      std::vector<MapPoint*> v(M);
      for (int i = 0; i < M; ++i) {
        v[i] = new_point(&pos[3 * i], &desc[(size_t)32 * i], maxd[i], mind[i], 1);
        v[i]->mNormalVector = Eigen::Vector3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
        v[i]->mbBad = valid[i] == 0;
      }
      return v;
    };
    Sophus::Sim3f Scw;
    for (int i = 0; i < 4; ++i) Scw.T.q[i] = sim[i];
    for (int i = 0; i < 3; ++i) { Scw.T.t[i] = sim[4 + i]; Scw.T.Ow[i] = sim[7 + i]; }
    ORBmatcher matcher(0.8f, true);
    {
      KeyFrame kf; a2.fill(kf); split(kf, (int)cfg[5]);
      for (auto& p : kf.mvpMapPoints) p = nullptr;   // (Fuse skips points the keyframe holds: the search's own selection is what is compared)
      std::vector<MapPoint*> pts = make_points();
      std::vector<MapPoint*> vpReplacePoint(M, static_cast<MapPoint*>(NULL));
      const int n = matcher.Fuse(&kf, Scw, pts, 6.0f, vpReplacePoint);
      const std::vector<int> slots = index_of(kf.mvpMapPoints, pts);
      dump("rlc_fuse_n", &n, 1); dump("rlc_fuse_slots", slots.data(), slots.size());
    }
    for (int variant = 0; variant < 2; ++variant) {
      KeyFrame kf; a2.fill(kf); split(kf, (int)cfg[5]);
      std::vector<MapPoint*> pts = make_points();
      MapPoint* taken = new_point(nullptr, nullptr, 1.f, 1.f, 1);
      std::vector<MapPoint*> vpMatched(kf.N, static_cast<MapPoint*>(NULL));
      for (int i = 0; i < kf.N; ++i) if (matched[i] >= 0) vpMatched[i] = taken;
      std::vector<KeyFrame*> vpPointsKFs(M, &kf), vpMatchedKF;
      const int n = variant == 0 ? matcher.SearchByProjection(&kf, Scw, pts, vpMatched, 8, 0.9f)
                                 : matcher.SearchByProjection(&kf, Scw, pts, vpPointsKFs, vpMatched, vpMatchedKF, 8, 0.9f);
      std::vector<int> idx = index_of(vpMatched, pts);
      for (int i = 0; i < kf.N; ++i) if (vpMatched[i] == taken) idx[i] = -2;
      const std::string tag = variant == 0 ? "rlc_proj" : "rlc_projk";
      dump(tag + "_n", &n, 1); dump(tag + "_match", idx.data(), idx.size());
    }
  }
  {   // SearchBySim3 between the two rig keyframes
    Arrays b1("rs3_1"), b2("rs3_2");
    KeyFrame k1, k2; b1.fill(k1); b2.fill(k2);
    split(k1, (int)cfg[4]); split(k2, (int)cfg[5]);
    const auto c3 = load<float>("rs3_cfg");   // th, S12 (7), S21 (7)
    Sophus::Sim3f S12, S21;
    for (int i = 0; i < 7; ++i) { S12.raw[i] = c3[1 + i]; S21.raw[i] = c3[8 + i]; }
    S12.inv = &S21;
    ORBmatcher matcher(0.8f, true);
    std::vector<MapPoint*> vpMatches12(k1.N, static_cast<MapPoint*>(NULL));
    const int n = matcher.SearchBySim3(&k1, &k2, vpMatches12, S12, c3[0]);
    const std::vector<int> m12 = index_of(vpMatches12, k2.mvpMapPoints);
    dump("rs3_n", &n, 1); dump("rs3_match", m12.data(), m12.size());
  }
  std::printf("reference members (rig) ok\n");
  return 0;
}

// Optimizer::PoseInertialOptimizationLastKeyFrame(Frame*) then PoseInertialOptimizationLastFrame(Frame*) on the next frame (Tracking.cc:3032-3041):
// keyframe -> frame A (its mpcpi comes out of the first call) -> frame B (mpPrevFrame = A).  Dumps, per frame: the IMU state written back
// (SetImuPoseVelocity + mImuBias, 21 floats), mvbOutlier, the return value, mpcpi as 246 doubles.
static void fill_pre(IMU::Preintegrated& P, const float* r) {   // the record of morb_hip.h (310 floats) -> the mock IMU::Preintegrated
  auto m3 = [&](Eigen::Matrix3f& M) { for (int k = 0; k < 9; ++k) M.m[k] = *r++; };
  auto v3 = [&](Eigen::Vector3f& V) { for (int k = 0; k < 3; ++k) V.v[k] = *r++; };
  P.dT = *r++;
  m3(P.dR); v3(P.dV); v3(P.dP); m3(P.JRg); m3(P.JVg); m3(P.JVa); m3(P.JPg); m3(P.JPa);
  for (int k = 0; k < 225; ++k) P.C.m[k] = *r++;
  P.b = IMU::Bias(r[0], r[1], r[2], r[3], r[4], r[5]); r += 6;
  for (int k = 0; k < 6; ++k) P.Nga.d.v[k] = *r++;
  for (int k = 0; k < 6; ++k) P.NgaWalk.d.v[k] = *r++;
  v3(P.avgA); v3(P.avgW);
}
template <class F> static void set_state(F& f, const float* s) {   // Rwb (9), twb, velocity; the biases: gyro (3), acc (3)
  for (int k = 0; k < 9; ++k) f.mRwb.m[k] = s[k];
  for (int k = 0; k < 3; ++k) { f.mOwb.v[k] = s[9 + k]; f.mVw.v[k] = s[12 + k]; }
  f.mImuBias = IMU::Bias(s[18], s[19], s[20], s[15], s[16], s[17]);
}
static void fill_visual(Frame& F, const std::string& t) {
  const auto has = load<uint8_t>(t + "_has"), close = load<uint8_t>(t + "_close"); const auto obs = load<float>(t + "_obs"), inv = load<float>(t + "_inv"), Xw = load<float>(t + "_xw");
  const int N = (int)has.size();
  F.N = N; F.mvKeysUn.resize(N); F.mvuRight.resize(N); F.mvInvLevelSigma2.resize(N); F.mvpMapPoints.assign(N, static_cast<MapPoint*>(NULL)); F.mvbOutlier.assign(N, true);
  for (int i = 0; i < N; ++i) {
    F.mvKeysUn[i].pt.x = obs[3 * i]; F.mvKeysUn[i].pt.y = obs[3 * i + 1]; F.mvuRight[i] = obs[3 * i + 2];
    F.mvKeysUn[i].octave = i; F.mvInvLevelSigma2[i] = inv[i];   // (mock: one "octave" per feature so that any 1 / sigma^2 can be carried)
    if (has[i]) { F.mvpMapPoints[i] = new_point(&Xw[3 * i], nullptr, 1.f, 1.f, 1); F.mvpMapPoints[i]->mTrackDepth = close[i] ? 5.f : 20.f; }
  }
}
static void dump_frame(Frame& F, const std::string& t, int nin) {
  float st[21];
  for (int k = 0; k < 9; ++k) st[k] = F.mRwb.m[k];
  for (int k = 0; k < 3; ++k) { st[9 + k] = F.mOwb.v[k]; st[12 + k] = F.mVw.v[k]; }
  st[15] = F.mImuBias.bwx; st[16] = F.mImuBias.bwy; st[17] = F.mImuBias.bwz; st[18] = F.mImuBias.bax; st[19] = F.mImuBias.bay; st[20] = F.mImuBias.baz;
  std::vector<uint8_t> outl(F.N);
  for (int i = 0; i < F.N; ++i) outl[i] = (F.mvpMapPoints[i] && F.mvbOutlier[i]) ? 1 : 0;
  double pr[246] = {0};
  if (F.mpcpi) {
    const ConstraintPoseImu& c = *F.mpcpi;
    for (int k = 0; k < 9; ++k) pr[k] = c.Rwb.m[k];
    for (int k = 0; k < 3; ++k) { pr[9 + k] = c.twb.v[k]; pr[12 + k] = c.vwb.v[k]; pr[15 + k] = c.bg.v[k]; pr[18 + k] = c.ba.v[k]; }
    for (int k = 0; k < 225; ++k) pr[21 + k] = c.H.m[k];
  }
  dump(t + "_state", st, 21); dump(t + "_outlier", outl.data(), outl.size()); dump(t + "_n", &nin, 1); dump(t + "_prior", pr, 246);
}
static int run_inertial() {
  const auto cam = load<float>("vi_cam");   // fx fy cx cy bf
  const auto tbc = load<float>("vi_tbc");   // rotation (9) + translation (3)
  Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3];
  IMU::Preintegrated preA, preBF, preBK;
  fill_pre(preA, load<float>("vi_a_pre").data()); fill_pre(preBF, load<float>("vi_b_pref").data()); fill_pre(preBK, load<float>("vi_b_prek").data());
  KeyFrame kf;
  { const auto s = load<float>("vi_kf_state"); set_state(kf, s.data()); }
  auto calib = [&](Frame& F) { F.mbf = cam[4]; for (int k = 0; k < 9; ++k) F.mImuCalib.mTbc.R[k] = tbc[k]; for (int k = 0; k < 3; ++k) F.mImuCalib.mTbc.t[k] = tbc[9 + k]; };
  Frame A; fill_visual(A, "vi_a"); calib(A);
  { const auto s = load<float>("vi_a_state0"); set_state(A, s.data()); }
  A.mpLastKeyFrame = &kf; A.mpImuPreintegrated = &preA;
  const int nA = Optimizer::PoseInertialOptimizationLastKeyFrame(&A);
  dump_frame(A, "vi_a", nA);
  Frame B; fill_visual(B, "vi_b"); calib(B);
  { const auto s = load<float>("vi_b_state0"); set_state(B, s.data()); }
  B.mpLastKeyFrame = &kf; B.mpPrevFrame = &A; B.mpImuPreintegratedFrame = &preBF; B.mpImuPreintegrated = &preBK;
  const int nB = Optimizer::PoseInertialOptimizationLastFrame(&B);
  dump_frame(B, "vi_b", nB);
  const int freed = A.mpcpi == nullptr ? 1 : 0;   // :5157-5158: the previous frame's prior is deleted
  dump("vi_a_freed", &freed, 1);
  dump_frame(A, "vi_a_after", nA);   // LastFrame does not write the previous frame back (its vertices are only read, :5144-5149)
  std::printf("reference members (inertial) ok\n");
  return 0;
}

// void Optimizer::LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int&, int&, int&, int&, bool bLarge, bool bRecInit): the flattened graph of
// the test rebuilt as mock objects — the temporal chain through mPrevKF (keyframe 0 = the fixed one before the window), IMU::Preintegrated per link,
// a keyframe's features = its edges — so that the member's own selection (Optimizer.cc:2337-2435) finds the same graph in its own order.
static int run_iba() {
  const auto st = load<float>("iba_kf_state"); const auto kind = load<uint8_t>("iba_kf_kind"); auto mpp = load<float>("iba_mp_pos"); const auto mpc = load<uint8_t>("iba_mp_close");
  const auto eKF = load<int>("iba_e_kf"), eMP = load<int>("iba_e_mp"); const auto eObs = load<float>("iba_e_obs"), eInv = load<float>("iba_e_inv");
  const auto iKF2 = load<int>("iba_i_kf2"); const auto iPre = load<float>("iba_i_pre"); const auto cam = load<float>("iba_cam"), tbc = load<float>("iba_tbc");
  const auto cfg = load<int>("iba_cfg");   // bLarge, bRecInit
  const int nKF = (int)kind.size(), nMP = (int)mpc.size(), nE = (int)eKF.size(), nI = (int)iKF2.size();
  int nOpt = 0; for (int k = 0; k < nKF; ++k) nOpt += kind[k] == 0 ? 1 : 0;
  Map map; map.initKF = 1u << 30; map.nKFs = (unsigned long)nOpt + 2; map.inertial = true;
  std::vector<std::unique_ptr<KeyFrame>> kfs(nKF);
  std::vector<std::unique_ptr<IMU::Preintegrated>> pres(nI);
  std::vector<MapPoint*> mps(nMP);
  for (int j = 0; j < nMP; ++j) { mps[j] = new_point(&mpp[3 * j], nullptr, 1.f, 1.f, 0); mps[j]->mpMap = &map; mps[j]->mnId = (unsigned long)j + 1; mps[j]->mTrackDepth = mpc[j] ? 5.f : 20.f; }
  std::vector<int> featOfEdge(nE, -1);
  for (int k = 0; k < nKF; ++k) {
    kfs[k].reset(new KeyFrame());
    KeyFrame& K = *kfs[k];
    K.mnId = (unsigned long)k + 1; K.mpMap = &map; K.fx = cam[0]; K.fy = cam[1]; K.cx = cam[2]; K.cy = cam[3]; K.mbf = cam[4];
    set_state(K, &st[(size_t)21 * k]); K.bImu = kind[k] != 2;
    for (int i = 0; i < 9; ++i) K.mImuCalib.mTbc.R[i] = tbc[i];
    for (int i = 0; i < 3; ++i) K.mImuCalib.mTbc.t[i] = tbc[9 + i];
    for (int e = 0; e < nE; ++e) {
      if (eKF[e] != k) continue;
      const int f = (int)K.mvKeysUn.size();
      featOfEdge[e] = f;
      cv::KeyPoint kp; kp.pt.x = eObs[3 * e]; kp.pt.y = eObs[3 * e + 1]; kp.octave = f;
      K.mvKeysUn.push_back(kp); K.mvuRight.push_back(eObs[3 * e + 2]); K.mvInvLevelSigma2.push_back(eInv[e]);
      K.mvpMapPoints.push_back(mps[eMP[e]]);
      mps[eMP[e]]->mObservations[&K] = std::make_tuple(f, -1); mps[eMP[e]]->nObs++;
    }
    K.N = (int)K.mvKeysUn.size();
  }
  for (int i = 0; i < nI; ++i) {   // link i: keyframe iKF2[i] owns the preintegration since iKF2[i] - 1 (the chain is 0 -> 1 -> ... -> nOpt in time)
    pres[i].reset(new IMU::Preintegrated()); fill_pre(*pres[i], &iPre[(size_t)310 * i]);
    kfs[iKF2[i]]->mpImuPreintegrated = pres[i].get(); kfs[iKF2[i]]->mPrevKF = kfs[iKF2[i] - 1].get();
  }
  bool stop = false;
  int num_fixedKF = 0, num_OptKF = 0, num_MPs = 0, num_edges = 0;
  Optimizer::LocalInertialBA(kfs[nOpt].get(), &stop, &map, num_fixedKF, num_OptKF, num_MPs, num_edges, cfg[0] != 0, cfg[1] != 0);
  std::vector<float> kfo((size_t)nKF * 21), mpo((size_t)nMP * 3);
  for (int k = 0; k < nKF; ++k) {   // what the member wrote: SetPose(Tcw), SetVelocity, SetNewBias
    const KeyFrame& K = *kfs[k];
    float* o = &kfo[(size_t)21 * k];
    for (int i = 0; i < 9; ++i) o[i] = K.mTcw.R[i];
    for (int i = 0; i < 3; ++i) { o[9 + i] = K.mTcw.t[i]; o[12 + i] = K.mVw.v[i]; }
    o[15] = K.mImuBias.bwx; o[16] = K.mImuBias.bwy; o[17] = K.mImuBias.bwz; o[18] = K.mImuBias.bax; o[19] = K.mImuBias.bay; o[20] = K.mImuBias.baz;
  }
  for (int j = 0; j < nMP; ++j) for (int i = 0; i < 3; ++i) mpo[3 * j + i] = mps[j]->mWorldPos(i);
  std::vector<uint8_t> erased(nE);
  for (int e = 0; e < nE; ++e) erased[e] = kfs[eKF[e]]->mvpMapPoints[featOfEdge[e]] == nullptr ? 1 : 0;
  int flagsLeft = 0; for (int k = 0; k < nKF; ++k) flagsLeft += (kfs[k]->mnBALocalForKF != 0 || kfs[k]->mnBAFixedForKF != 0) ? 1 : 0;   // :2828-2831, :2845: reset on the way out
  const int counts[7] = {num_fixedKF, num_OptKF, num_MPs, num_edges, map.changes, mps[0]->nUpdates, flagsLeft};
  dump("iba_kf", kfo.data(), kfo.size()); dump("iba_mp", mpo.data(), mpo.size()); dump("iba_erase", erased.data(), erased.size()); dump("iba_counts", counts, 7);
  std::printf("reference members (iba) ok\n");
  return 0;
}

static int run_tracking() {
  {   // int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints)
    // after the host's isInFrustum loop (Tracking.cc:3117-3183): here that loop's results come from morb_is_in_frustum_batch
    using morb_adapter::DeviceBuffer;
    const auto kps = load<morb_keypoint>("f_kps"); const auto desc = load<uint8_t>("f_desc"); const auto ur = load<float>("f_uright");
    const auto blocked = load<uint8_t>("f_blocked"); const auto prm = load<morb_frame_params>("f_params"); const auto pose = load<float>("f_pose");
    const auto Xw = load<float>("mp_xw"); const auto nrm = load<float>("mp_normal"); const auto maxD = load<float>("mp_maxd"); const auto minD = load<float>("mp_mind");
    const auto mdesc = load<uint8_t>("mp_desc"); const auto bad = load<uint8_t>("mp_bad"); const auto obs = load<uint8_t>("mp_hasobs");
    const auto cfg = load<float>("sbp_cfg");   // nnratio, th, bFar, thFar
    const int N = (int)kps.size(), M = (int)maxD.size();
    ORBmatcher matcher(cfg[0], true);
    DeviceBuffer<float> dR(pose.data(), 9), dt(pose.data() + 9, 3), dO(pose.data() + 12, 3), dX(Xw), dn(nrm), dmax(maxD), dmin(minD), px(M), py(M), pxr(M), dep(M), vc(M);
    DeviceBuffer<uint8_t> inv(M); DeviceBuffer<int> lvl(M), nmp(&M, 1);
    if (morb_is_in_frustum_batch(matcher.handle(), &prm[0], 1, dR.get(), dt.get(), dO.get(), M, nmp.get(), dX.get(), dn.get(), dmax.get(), dmin.get(), 0.5f, inv.get(),
                                 px.get(), py.get(), pxr.get(), dep.get(), lvl.get(), vc.get(), nullptr) != MORB_OK) return 4;
    morb_matcher_sync(matcher.handle());
    const auto hin = inv.to_host(); const auto hpx = px.to_host(), hpy = py.to_host(), hpxr = pxr.to_host(), hdep = dep.to_host(), hvc = vc.to_host(); const auto hl = lvl.to_host();
    Frame F;
    F.N = N; F.mvKeysUn.resize(N);
    for (int i = 0; i < N; ++i) { cv::KeyPoint& k = F.mvKeysUn[i]; k.pt.x = kps[i].x; k.pt.y = kps[i].y; k.size = kps[i].size; k.angle = kps[i].angle; k.response = kps[i].response; k.octave = kps[i].octave; k.class_id = kps[i].class_id; }
    F.mDescriptors.data = desc; F.mDescriptors.rows = N; F.mvuRight = ur;
    const morb_frame_params& P = prm[0];
    Frame::fx = P.fx; Frame::fy = P.fy; Frame::cx = P.cx; Frame::cy = P.cy; Frame::mnMinX = P.minX; Frame::mnMaxX = P.maxX; Frame::mnMinY = P.minY;
    Frame::mnMaxY = P.maxY; Frame::mfGridElementWidthInv = P.gridInvW; Frame::mfGridElementHeightInv = P.gridInvH;
    F.mbf = P.mbf; F.mb = P.mb; F.mfLogScaleFactor = P.logScaleFactor; F.mnScaleLevels = P.nlevels;
    F.mvScaleFactors.assign(P.scaleFactors, P.scaleFactors + P.nlevels); F.mvLevelSigma2.assign(P.levelSigma2, P.levelSigma2 + P.nlevels);
    F.mvpMapPoints.assign(N, static_cast<MapPoint*>(NULL));
    for (int i = 0; i < N; ++i) if (blocked[i]) F.mvpMapPoints[i] = new_point(nullptr, nullptr, 1.f, 1.f, 1);
    std::vector<MapPoint*> vpMapPoints(M);
    for (int j = 0; j < M; ++j) {
      MapPoint* p = new_point(&Xw[3 * j], &mdesc[(size_t)32 * j], maxD[j], minD[j], obs[j] ? 1 : 0);
      p->mbBad = bad[j] != 0;
      p->mbTrackInView = hin[j] != 0 && !p->mbBad;     // (the host loop skips bad points before isInFrustum, Tracking.cc:3152)
      p->mTrackProjX = hpx[j]; p->mTrackProjY = hpy[j]; p->mTrackProjXR = hpxr[j]; p->mTrackDepth = hdep[j]; p->mnTrackScaleLevel = hl[j]; p->mTrackViewCos = hvc[j];
      vpMapPoints[j] = p;
    }
    const std::vector<MapPoint*> before = F.mvpMapPoints;
    const int n = matcher.SearchByProjection(F, vpMapPoints, cfg[1], cfg[2] != 0, cfg[3]);
    std::vector<int> match = index_of(F.mvpMapPoints, vpMapPoints);
    dump("sbp_n", &n, 1); dump("sbp_match", match.data(), match.size());
    cv::Mat a, b; a.data.assign(desc.begin(), desc.begin() + 32); b.data.assign(mdesc.begin(), mdesc.begin() + 32);
    const int dd = ORBmatcher::DescriptorDistance(a, b);
    dump("dist", &dd, 1);
  }
  {   // int Optimizer::PoseOptimization(Frame* pFrame)
    const auto has = load<uint8_t>("po_has"); const auto obs = load<float>("po_obs"); const auto inv = load<float>("po_inv"); const auto Xw = load<float>("po_xw");
    const auto cam = load<float>("po_cam"); const auto pose = load<float>("po_pose");
    const int N = (int)has.size();
    Frame F;
    F.N = N; F.mvKeysUn.resize(N); F.mvuRight.resize(N); F.mvInvLevelSigma2.resize(N); F.mvpMapPoints.assign(N, static_cast<MapPoint*>(NULL)); F.mvbOutlier.assign(N, true);
    for (int i = 0; i < N; ++i) {
      F.mvKeysUn[i].pt.x = obs[3 * i]; F.mvKeysUn[i].pt.y = obs[3 * i + 1]; F.mvuRight[i] = obs[3 * i + 2];
      F.mvKeysUn[i].octave = i; F.mvInvLevelSigma2[i] = inv[i];   // (mock: one "octave" per feature so that any 1 / sigma^2 can be carried)
      if (has[i]) F.mvpMapPoints[i] = new_point(&Xw[3 * i], nullptr, 1.f, 1.f, 1);
    }
    Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3]; F.mbf = cam[4];
    Sophus::SE3f T; for (int i = 0; i < 4; ++i) T.q[i] = pose[i]; for (int i = 0; i < 3; ++i) T.t[i] = pose[4 + i];
    F.SetPose(T);
    const int nin = Optimizer::PoseOptimization(&F);
    const Sophus::SE3f To = F.GetPose();
    const float po[7] = {To.q[0], To.q[1], To.q[2], To.q[3], To.t[0], To.t[1], To.t[2]};
    std::vector<uint8_t> outl(N);
    for (int i = 0; i < N; ++i) outl[i] = (has[i] && F.mvbOutlier[i]) ? 1 : 0;
    dump("po_nin", &nin, 1); dump("po_pose", po, 7); dump("po_outlier", outl.data(), outl.size());
  }
  {   // void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges)
    // the flattened graph of the test (keyframes, fixed flags, points, edges) rebuilt as mock objects: a keyframe's features are its edges
    auto kfp = load<float>("ba_kf"); auto mpp = load<float>("ba_mp"); const auto fixed = load<uint8_t>("ba_fixed"); const auto eKF = load<int>("ba_ekf");
    const auto eMP = load<int>("ba_emp"); const auto eObs = load<float>("ba_eobs"); const auto eInv = load<float>("ba_einv"); const auto cam = load<float>("po_cam");
    const int nKF = (int)fixed.size(), nMP = (int)mpp.size() / 3, nE = (int)eKF.size();
    Map map; map.initKF = 1u << 30; map.nKFs = (unsigned long)nKF;
    std::vector<std::unique_ptr<KeyFrame>> kfs(nKF);
    std::vector<MapPoint*> mps(nMP);
    for (int j = 0; j < nMP; ++j) { mps[j] = new_point(&mpp[3 * j], nullptr, 1.f, 1.f, 0); mps[j]->mpMap = &map; mps[j]->mnId = (unsigned long)j + 1; }
    std::vector<int> featOfEdge(nE, -1);
    for (int k = 0; k < nKF; ++k) {
      kfs[k].reset(new KeyFrame());
      KeyFrame& K = *kfs[k];
      // (ids from 1: mnBALocalForKF / mnBAFixedForKF start at 0, as in the reference, where keyframe 0 never runs a local BA)
      K.mnId = (unsigned long)k + 1; K.mpMap = &map; K.fx = cam[0]; K.fy = cam[1]; K.cx = cam[2]; K.cy = cam[3]; K.mbf = cam[4];
      for (int i = 0; i < 4; ++i) K.mTcw.q[i] = kfp[7 * k + i];
      for (int i = 0; i < 3; ++i) K.mTcw.t[i] = kfp[7 * k + 4 + i];
      for (int e = 0; e < nE; ++e) {
        if (eKF[e] != k) continue;
        const int f = (int)K.mvKeysUn.size();
        featOfEdge[e] = f;
        cv::KeyPoint kp; kp.pt.x = eObs[3 * e]; kp.pt.y = eObs[3 * e + 1]; kp.octave = f;   // (one "octave" per feature carries the edge's 1 / sigma^2)
        K.mvKeysUn.push_back(kp); K.mvuRight.push_back(eObs[3 * e + 2]); K.mvInvLevelSigma2.push_back(eInv[e]);
        K.mvpMapPoints.push_back(mps[eMP[e]]);
        mps[eMP[e]]->mObservations[&K] = std::make_tuple(f, -1); mps[eMP[e]]->nObs++;
      }
      K.N = (int)K.mvKeysUn.size();
    }
    KeyFrame* pKF = nullptr;
    for (int k = 0; k < nKF; ++k) if (!fixed[k]) { if (!pKF) pKF = kfs[k].get(); else pKF->mvpCov.push_back(kfs[k].get()); }
    bool stop = false;
    int num_fixedKF = 0, num_OptKF = 0, num_MPs = 0, num_edges = 0;
    Optimizer::LocalBundleAdjustment(pKF, &stop, &map, num_fixedKF, num_OptKF, num_MPs, num_edges);
    std::vector<float> kfo((size_t)nKF * 7), mpo((size_t)nMP * 3);
    for (int k = 0; k < nKF; ++k) { for (int i = 0; i < 4; ++i) kfo[7 * k + i] = kfs[k]->mTcw.q[i]; for (int i = 0; i < 3; ++i) kfo[7 * k + 4 + i] = kfs[k]->mTcw.t[i]; }
    for (int j = 0; j < nMP; ++j) for (int i = 0; i < 3; ++i) mpo[3 * j + i] = mps[j]->mWorldPos(i);
    std::vector<uint8_t> erased(nE);
    for (int e = 0; e < nE; ++e) erased[e] = kfs[eKF[e]]->mvpMapPoints[featOfEdge[e]] == nullptr ? 1 : 0;   // EraseMapPointMatch + EraseObservation (:1404-1414)
    const int counts[6] = {num_fixedKF, num_OptKF, num_MPs, num_edges, map.changes, mps[0]->nUpdates};
    dump("ba_kf", kfo.data(), kfo.size()); dump("ba_mp", mpo.data(), mpo.size()); dump("ba_erase", erased.data(), erased.size()); dump("ba_counts", counts, 6);
  }
  std::printf("reference members (tracking) ok\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  g_dir = argv[1];
  const std::string mode = argv[2];
  return mode == "matcher" ? run_matcher() : mode == "rig" ? run_rig() : mode == "inertial" ? run_inertial() : mode == "iba" ? run_iba() : run_tracking();
}
