// Drives the view-taking methods of include/morb/ORBmatcher.h (all of the reference's ORBmatcher surface except
// SearchByProjection(Frame, MapPoints), which adapters_check.cc covers) on inputs written by tests/test_adapter_matcher_gpu.py and
// dumps what they return; the Python side compares the dumps with the CPU oracle.
//   matcher_adapters_check <dir>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/morb/ORBmatcher.h"

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
  std::ofstream f(g_dir + "/out_" + name + ".bin", std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}
template <typename T> static const T* ptr(const std::vector<T>& v) { return v.empty() ? nullptr : v.data(); }

using namespace ORB_SLAM3;
struct FrameFiles {   // keeps the arrays a view points at
  std::vector<morb_keypoint> kps; std::vector<uint8_t> desc, tracked, hasmp, mpdesc, mpobs; std::vector<float> uright, mppos, mpmax, mpmin, pose;
  std::vector<int> node, nvalid; std::vector<morb_frame_params> prm;
  KeyFrameView v;
  explicit FrameFiles(const std::string& p) {
    kps = load<morb_keypoint>(p + "_kps"); desc = load<uint8_t>(p + "_desc"); uright = load<float>(p + "_uright", true);
    tracked = load<uint8_t>(p + "_tracked", true); hasmp = load<uint8_t>(p + "_hasmp", true); mpdesc = load<uint8_t>(p + "_mpdesc", true);
    mpobs = load<uint8_t>(p + "_mpobs", true); mppos = load<float>(p + "_mppos", true); mpmax = load<float>(p + "_mpmax", true);
    mpmin = load<float>(p + "_mpmin", true); pose = load<float>(p + "_pose", true); node = load<int>(p + "_node", true);
    nvalid = load<int>(p + "_nvalid", true); prm = load<morb_frame_params>(p + "_params");
    v.N = (int)kps.size(); v.mvKeysUn = kps.data(); v.mDescriptors = desc.data(); v.mvuRight = ptr(uright); v.hasTrackedMapPoint = ptr(tracked);
    v.params = prm[0]; v.featNode = ptr(node); v.nValid = nvalid.empty() ? -1 : nvalid[0]; v.hasMapPoint = ptr(hasmp); v.mpWorldPos = ptr(mppos);
    v.mpMaxDistance = ptr(mpmax); v.mpMinDistance = ptr(mpmin); v.mpDescriptor = ptr(mpdesc); v.mpHasObservations = ptr(mpobs);
    if (pose.size() == 22) {   // R 9, t 3, Ow 3, Tcw 7
      for (int i = 0; i < 9; ++i) v.mRcw[i] = pose[i];
      for (int i = 0; i < 3; ++i) { v.mtcw[i] = pose[9 + i]; v.mOw[i] = pose[12 + i]; }
      for (int i = 0; i < 7; ++i) v.Tcw[i] = pose[15 + i];
    }
  }
};
struct PointFiles {
  std::vector<float> pos, nrm, maxd, mind; std::vector<uint8_t> desc, valid;
  MapPointView v;
  explicit PointFiles(const std::string& p) {
    pos = load<float>(p + "_pos"); nrm = load<float>(p + "_normal"); maxd = load<float>(p + "_maxd"); mind = load<float>(p + "_mind");
    desc = load<uint8_t>(p + "_desc"); valid = load<uint8_t>(p + "_valid");
    v.n = (int)maxd.size(); v.worldPos = pos.data(); v.normal = nrm.data(); v.maxDistance = maxd.data(); v.minDistance = mind.data();
    v.descriptor = desc.data(); v.valid = valid.data();
  }
};

// MORB_ADAPTER_TIMING=1: every member is called 20 more times and its mean host-to-host latency printed (what a drop-in caller pays per call:
// the views' uploads, the kernels, the downloads and the synchronisations of the handle's stream)
#include <chrono>
#include <cstdlib>
static bool g_time = false;
template <class Fn> static void timeit(const char* name, Fn&& fn) {
  if (!g_time) return;
  fn();
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20; ++i) fn();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 20;
  std::printf("TIMING %-52s %.3f ms per call\n", name, ms);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  g_dir = argv[1];
  g_time = std::getenv("MORB_ADAPTER_TIMING") != nullptr;
  {   // SearchByProjection(CurrentFrame, LastFrame, th, bMono)
    FrameFiles cur("last_cur"), last("last_last");
    const auto cfg = load<float>("last_cfg");   // nnratio, checkOri, th, bMono
    ORBmatcher m(cfg[0], cfg[1] != 0);
    std::vector<int> match;
    const int n = m.SearchByProjection(static_cast<const FrameView&>(cur.v), static_cast<const FrameView&>(last.v), match, cfg[2], cfg[3] != 0);
    timeit("SearchByProjection(CurrentFrame, LastFrame)", [&] { std::vector<int> mm; m.SearchByProjection(static_cast<const FrameView&>(cur.v), static_cast<const FrameView&>(last.v), mm, cfg[2], cfg[3] != 0); });
    dump("last_n", &n, 1); dump("last_match", match.data(), match.size());
  }
  {   // SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)
    FrameFiles cur("kfp_cur"), kf("kfp_kf");
    const auto cfg = load<float>("kfp_cfg");   // nnratio, checkOri, th, ORBdist
    const auto found = load<uint8_t>("kfp_found");
    ORBmatcher m(cfg[0], cfg[1] != 0);
    std::vector<int> match;
    const int n = m.SearchByProjection(static_cast<const FrameView&>(cur.v), kf.v, found, match, cfg[2], (int)cfg[3]);
    timeit("SearchByProjection(CurrentFrame, KeyFrame)", [&] { std::vector<int> mm; m.SearchByProjection(static_cast<const FrameView&>(cur.v), kf.v, found, mm, cfg[2], (int)cfg[3]); });
    dump("kfp_n", &n, 1); dump("kfp_match", match.data(), match.size());
  }
  {   // SearchByBoW(pKF, F) and SearchByBoW(pKF1, pKF2)
    FrameFiles kf("bow_kf"), fr("bow_f");
    const auto cfg = load<float>("bow_cfg");   // nnratio, checkOri
    ORBmatcher m(cfg[0], cfg[1] != 0);
    std::vector<int> match;
    int n = m.SearchByBoW(kf.v, static_cast<const FrameView&>(fr.v), match);
    timeit("SearchByBoW(KeyFrame, Frame)", [&] { std::vector<int> mm; m.SearchByBoW(kf.v, static_cast<const FrameView&>(fr.v), mm); });
    dump("bow_n", &n, 1); dump("bow_match", match.data(), match.size());
    FrameFiles k1("bowkk_1"), k2("bowkk_2");
    n = m.SearchByBoW(k1.v, k2.v, match);
    dump("bowkk_n", &n, 1); dump("bowkk_match", match.data(), match.size());
  }
  {   // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
    FrameFiles f1("ini_1"), f2("ini_2");
    const auto cfg = load<float>("ini_cfg");   // nnratio, checkOri, windowSize
    auto prev = load<float>("ini_prev");
    ORBmatcher m(cfg[0], cfg[1] != 0);
    std::vector<int> m12;
    const int n = m.SearchForInitialization(f1.v, f2.v, prev, m12, (int)cfg[2]);
    dump("ini_n", &n, 1); dump("ini_match", m12.data(), m12.size()); dump("ini_prev", prev.data(), prev.size());
  }
  {   // SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse)
    FrameFiles k1("tri_1"), k2("tri_2");
    const auto cfg = load<float>("tri_cfg");   // nnratio, checkOri, bOnlyStereo, bCoarse, R12 (9), t12 (3), ep (2)
    ORBmatcher m(cfg[0], cfg[1] != 0);
    std::vector<std::pair<size_t, size_t>> pairs;
    const int n = m.SearchForTriangulation(k1.v, k2.v, &cfg[4], &cfg[13], &cfg[16], pairs, cfg[2] != 0, cfg[3] != 0);
    timeit("SearchForTriangulation(KeyFrame, KeyFrame)", [&] { decltype(pairs) pp; m.SearchForTriangulation(k1.v, k2.v, &cfg[4], &cfg[13], &cfg[16], pp, cfg[2] != 0, cfg[3] != 0); });
    std::vector<int> flat;
    for (auto& p : pairs) { flat.push_back((int)p.first); flat.push_back((int)p.second); }
    dump("tri_n", &n, 1); dump("tri_pairs", flat.data(), flat.size());
  }
  {   // Fuse x2 and SearchByProjection(pKF, Scw, ...) x2
    FrameFiles kf("lc_kf");
    PointFiles pts("lc_pts");
    const auto cfg = load<float>("lc_cfg");   // nnratio, checkOri, thFuse, thFuseSim3, thProj, ratioProj
    const auto sim = load<float>("lc_sim3");  // Tcw 7, Ow 3
    Sim3View S; for (int i = 0; i < 7; ++i) S.Tcw[i] = sim[i]; for (int i = 0; i < 3; ++i) S.Ow[i] = sim[7 + i];
    ORBmatcher m(cfg[0], cfg[1] != 0);
    std::vector<int> bi, bd;
    int n = m.Fuse(kf.v, pts.v, bi, bd, cfg[2]);
    timeit("Fuse(KeyFrame, MapPoints)", [&] { auto b1 = bi; auto b2 = bd; m.Fuse(kf.v, pts.v, b1, b2, cfg[2]); });
    dump("fuse_n", &n, 1); dump("fuse_idx", bi.data(), bi.size()); dump("fuse_dist", bd.data(), bd.size());
    n = m.Fuse(kf.v, S, pts.v, cfg[3], bi, bd);
    dump("fuse3_n", &n, 1); dump("fuse3_idx", bi.data(), bi.size()); dump("fuse3_dist", bd.data(), bd.size());
    auto matched = load<int>("lc_matched");   // vpMatched on entry (index or -1)
    std::vector<int> vm = matched;
    n = m.SearchByProjection(kf.v, S, pts.v, vm, (int)cfg[4], cfg[5]);
    dump("sim3p_n", &n, 1); dump("sim3p_match", vm.data(), vm.size());
    vm = matched;
    std::vector<int> vkf;
    n = m.SearchByProjection(kf.v, S, pts.v, vm, vkf, (int)cfg[4], cfg[5]);
    dump("sim3k_n", &n, 1); dump("sim3k_match", vm.data(), vm.size());
  }
  {   // SearchBySim3(pKF1, pKF2, vpMatches12, S12, th)
    FrameFiles k1("s3_1"), k2("s3_2");
    const auto cfg = load<float>("s3_cfg");   // nnratio, checkOri, th, S12 (7), S21 (7)
    ORBmatcher m(cfg[0], cfg[1] != 0);
    std::vector<int> m12;
    const int n = m.SearchBySim3(k1.v, k2.v, m12, &cfg[3], &cfg[10], cfg[2]);
    dump("s3_n", &n, 1); dump("s3_match", m12.data(), m12.size());
  }
  std::printf("matcher adapters ok\n");
  return 0;
}
