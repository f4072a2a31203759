// Drives the three C++ adapters (include/morb/ORBextractor.h, ORBmatcher.h, Optimizer.h) on inputs written by
// tests/test_adapter_gpu.py and dumps what they return; the Python side compares the dumps with the CPU oracle.
//   adapters_check <dir>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/morb/ORBextractor.h"
#include "../../include/morb/ORBmatcher.h"
#include "../../include/morb/Optimizer.h"

static std::string g_dir;
template <typename T>
static std::vector<T> load(const char* name) {
  std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
  if (!f) { std::fprintf(stderr, "missing %s\n", name); std::exit(3); }
  const size_t bytes = (size_t)f.tellg();
  std::vector<T> v(bytes / sizeof(T));
  f.seekg(0); f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)bytes);
  return v;
}
template <typename T>
static void dump(const char* name, const T* p, size_t n) {
  std::ofstream f(g_dir + "/out_" + name + ".bin", std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
  using namespace ORB_SLAM3;
  if (argc < 2) return 2;
  g_dir = argv[1];
  // ---- ORBextractor::operator()
  {
    const auto dims = load<int>("img_dims");   // w, h, nfeatures, nlevels, lap0, lap1
    podcv::Mat8u img; img.cols = dims[0]; img.rows = dims[1]; img.step = dims[0]; img.data = load<uint8_t>("img");
    ORBextractor ext(dims[2], 1.2f, dims[3], 20, 7);
    std::vector<podcv::KeyPoint> k; std::vector<uint8_t> d;
    const int mono = ext(img, k, d, {dims[4], dims[5]});
    const int head[2] = {mono, (int)k.size()};
    dump("ext_head", head, 2); dump("ext_kps", k.data(), k.size()); dump("ext_desc", d.data(), d.size());
    // mvImagePyramid (read by Frame::ComputeStereoMatches, Frame.cc:895): nothing is downloaded by operator() itself, the first access fetches
    const int lazy0 = ext.mvImagePyramid.downloaded() ? 1 : 0;
    const auto& lvl2 = ext.mvImagePyramid[2];
    const int pyr[5] = {lazy0, ext.mvImagePyramid.downloaded() ? 1 : 0, (int)ext.mvImagePyramid.size(), lvl2.cols, lvl2.rows};
    dump("ext_pyr_head", pyr, 5); dump("ext_pyr_l2", lvl2.data.data(), lvl2.data.size());
    const int mono2 = ext(img, k, d, {dims[4], dims[5]});   // a second extraction invalidates the host copies
    const int again[2] = {mono2 == mono ? 1 : 0, ext.mvImagePyramid.downloaded() ? 1 : 0};
    dump("ext_pyr_again", again, 2);
  }
  // ---- ORBmatcher::SearchByProjection(Frame, MapPoints)
  {
    const auto kps = load<morb_keypoint>("f_kps"); const auto desc = load<uint8_t>("f_desc"); const auto ur = load<float>("f_uright");
    const auto blocked = load<uint8_t>("f_blocked"); const auto prm = load<morb_frame_params>("f_params"); const auto pose = load<float>("f_pose");   // R 9, t 3, Ow 3
    const auto Xw = load<float>("mp_xw"); const auto nrm = load<float>("mp_normal"); const auto maxD = load<float>("mp_maxd"); const auto minD = load<float>("mp_mind");
    const auto mdesc = load<uint8_t>("mp_desc"); const auto bad = load<uint8_t>("mp_bad"); const auto obs = load<uint8_t>("mp_hasobs");
    const auto cfg = load<float>("sbp_cfg");   // nnratio, th, bFar, thFar
    FrameView F; F.N = (int)kps.size(); F.mvKeysUn = kps.data(); F.mDescriptors = desc.data(); F.mvuRight = ur.data(); F.hasTrackedMapPoint = blocked.data();
    F.params = prm[0];
    for (int i = 0; i < 9; ++i) F.mRcw[i] = pose[i];
    for (int i = 0; i < 3; ++i) { F.mtcw[i] = pose[9 + i]; F.mOw[i] = pose[12 + i]; }
    MapPointView M; M.n = (int)maxD.size(); M.worldPos = Xw.data(); M.normal = nrm.data(); M.maxDistance = maxD.data(); M.minDistance = minD.data();
    M.descriptor = mdesc.data(); M.isBad = bad.data(); M.hasObservations = obs.data();
    ORBmatcher matcher(cfg[0], true);
    std::vector<int> match;
    const int n = matcher.SearchByProjection(F, M, match, cfg[1], cfg[2] != 0, cfg[3]);
    dump("sbp_n", &n, 1); dump("sbp_match", match.data(), match.size());
    const int dd = ORBmatcher::DescriptorDistance(desc.data(), mdesc.data());
    dump("dist", &dd, 1);
  }
  // ---- Optimizer::PoseOptimization
  {
    PoseOptimizationView f;
    const auto has = load<uint8_t>("po_has"); const auto obs = load<float>("po_obs"); const auto inv = load<float>("po_inv"); const auto Xw = load<float>("po_xw");
    const auto cam = load<float>("po_cam"); const auto pose = load<float>("po_pose");
    f.N = (int)has.size(); f.hasMapPoint = has.data(); f.obs = obs.data(); f.invSigma2 = inv.data(); f.worldPos = Xw.data();
    f.fx = cam[0]; f.fy = cam[1]; f.cx = cam[2]; f.cy = cam[3]; f.mbf = cam[4];
    for (int i = 0; i < 7; ++i) f.pose[i] = pose[i];
    const int nin = Optimizer::PoseOptimization(f);
    dump("po_nin", &nin, 1); dump("po_pose", f.pose, 7); dump("po_outlier", f.mvbOutlier.data(), f.mvbOutlier.size());
  }
  // ---- Optimizer::LocalBundleAdjustment
  {
    LocalBAView g;
    auto kf = load<float>("ba_kf"); auto mp = load<float>("ba_mp"); const auto fixed = load<uint8_t>("ba_fixed"); const auto eKF = load<int>("ba_ekf");
    const auto eMP = load<int>("ba_emp"); const auto eObs = load<float>("ba_eobs"); const auto eInv = load<float>("ba_einv"); const auto cam = load<float>("po_cam");
    g.nKF = (int)fixed.size(); g.nMP = (int)mp.size() / 3; g.nE = (int)eKF.size(); g.kfPose = kf.data(); g.kfFixed = fixed.data(); g.mpPos = mp.data();
    g.eKF = eKF.data(); g.eMP = eMP.data(); g.eObs = eObs.data(); g.eInvSigma2 = eInv.data(); g.fx = cam[0]; g.fy = cam[1]; g.cx = cam[2]; g.cy = cam[3]; g.mbf = cam[4];
    bool stop = false;
    Optimizer::LocalBundleAdjustment(g, &stop);
    const int st[2] = {g.outerIterations, g.lmTrials};
    dump("ba_kf", kf.data(), kf.size()); dump("ba_mp", mp.data(), mp.size()); dump("ba_erase", g.eraseFlag.data(), g.eraseFlag.size()); dump("ba_stats", st, 2);
  }
  std::printf("adapters ok\n");
  return 0;
}
