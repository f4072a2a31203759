// Drop-in adapter: ORB_SLAM3::Optimizer (reference include/Optimizer.h:46-139) over libmorb_hip.so — the two static methods of the
// hot path, PoseOptimization and LocalBundleAdjustment.  As in ORBmatcher.h the arguments are plain views of what the reference
// methods read from Frame / KeyFrame / MapPoint / Map and write back to them; INTEGRATION.md shows the glue inside the reference tree.
#pragma once
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../morb_hip.h"
#include "device_buffer.h"

namespace ORB_SLAM3 {

// What Optimizer::PoseOptimization(Frame* pFrame) reads and writes (Optimizer.cc:762-1051): per feature i "mvpMapPoints[i] != NULL",
// (mvKeysUn[i].pt.x, .pt.y, mvuRight[i]), mvInvLevelSigma2[octave], the map point's world position; the camera; in / out the pose
// (Sophus::SE3f as unit quaternion xyzw + translation, :781-783) and mvbOutlier.
struct PoseOptimizationView {
  int N = 0;
  const uint8_t* hasMapPoint = nullptr;   // [N]
  const float* obs = nullptr;             // [N][3]
  const float* invSigma2 = nullptr;       // [N]
  const float* worldPos = nullptr;        // [N][3]
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  float pose[7] = {0, 0, 0, 1, 0, 0, 0};  // in / out
  std::vector<uint8_t> mvbOutlier;        // in / out [N] (empty on input = all false)
};
// The graph Optimizer::LocalBundleAdjustment assembles (Optimizer.cc:1058-1351), flattened.
struct LocalBAView {
  int nKF = 0, nMP = 0, nE = 0;
  float* kfPose = nullptr;          // [nKF][7] in / out
  const uint8_t* kfFixed = nullptr; // [nKF]
  float* mpPos = nullptr;           // [nMP][3] in / out
  const int* eKF = nullptr;         // [nE]
  const int* eMP = nullptr;         // [nE]
  const float* eObs = nullptr;      // [nE][3] (x, y, uRight or < 0)
  const float* eInvSigma2 = nullptr;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  bool inertialMap = false;         // pMap->IsInertial() (:1137)
  std::vector<uint8_t> eraseFlag;   // out [nE]: observations the reference erases (:1366-1401)
  int outerIterations = 0, lmTrials = 0;
};

class Optimizer {
 public:
  // static int PoseOptimization(Frame* pFrame)  Optimizer.h:86 -> number of inliers
  static int PoseOptimization(PoseOptimizationView& f, int device = 0) {
    using morb_adapter::DeviceBuffer;
    if (f.N <= 0) return 0;
    const int N = f.N;
    Slot& o = slot(device, kTracking);
    std::lock_guard<std::mutex> lock(o.mu);   // one caller per handle at a time (Tracking's handle is not LocalMapping's: see slot())
    morb_adapter::hip_check(hipSetDevice(device), "hipSetDevice");
    // per-thread, per-device staging buffers that only grow: Tracking calls this once per frame, and eight hipMalloc / hipFree pairs
    // per call cost more than the optimisation itself
    struct Staging { DeviceBuffer<uint8_t> has, outl; DeviceBuffer<float> obs, inv, Xw, pose; DeviceBuffer<int> nin, cnt; };
    static thread_local Staging per_device[kMaxDevices];
    Staging& s = per_device[device];
    s.has.assign(f.hasMapPoint, N); s.outl.resize(N);
    // mvbOutlier is only written for features that hold a map point (Optimizer.cc:817, :860): the others keep what they had
    if ((int)f.mvbOutlier.size() == N) s.outl.upload(f.mvbOutlier.data(), N); else s.outl.fill_bytes(0);
    s.obs.assign(f.obs, (size_t)N * 3); s.inv.assign(f.invSigma2, N); s.Xw.assign(f.worldPos, (size_t)N * 3); s.pose.assign(f.pose, 7);
    s.nin.resize(1); s.cnt.assign(&N, 1);
    check(morb_pose_optimization_batch(o.h, 1, N, s.cnt.get(), s.has.get(), s.obs.get(), s.inv.get(), s.Xw.get(), f.fx, f.fy, f.cx, f.cy, f.mbf,
                                       s.pose.get(), s.outl.get(), s.nin.get(), nullptr, nullptr));
    morb_adapter::hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    s.pose.download(f.pose, 7);
    f.mvbOutlier = s.outl.to_host();
    return s.nin.to_host()[0];
  }

  // static void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int&, int&, int&, int&)  Optimizer.h:67-69.  pbStopFlag is
  // the reference's own bool (LocalMapping::mbAbortBA): it is polled at every LM iteration and trial.
  static void LocalBundleAdjustment(LocalBAView& g, bool* pbStopFlag, int device = 0) {
    static_assert(sizeof(bool) == 1, "pbStopFlag is read as one byte");
    g.eraseFlag.assign(g.nE, 0);
    int stats[2] = {0, 0};
    Slot& o = slot(device, kMapping);
    std::lock_guard<std::mutex> lock(o.mu);   // the one-shot entry point works in the handle's grow-only workspace
    morb_adapter::hip_check(hipSetDevice(device), "hipSetDevice");
    check(morb_local_bundle_adjustment(o.h, g.nKF, g.kfPose, g.kfFixed, g.nMP, g.mpPos, g.nE, g.eKF, g.eMP, g.eObs, g.eInvSigma2, g.fx,
                                       g.fy, g.cx, g.cy, g.mbf, g.inertialMap ? 1 : 0, reinterpret_cast<const unsigned char*>(pbStopFlag),
                                       g.eraseFlag.data(), stats));
    g.outerIterations = stats[0]; g.lmTrials = stats[1];
  }

  // The reference's Optimizer is a stateless static class entered concurrently from Tracking (PoseOptimization, every frame) and from
  // LocalMapping (LocalBundleAdjustment, hundreds of milliseconds of LM trials): each role has its own handle — own stream, own
  // workspace — per device, created once (std::call_once), so a tracked frame never queues behind local-mapping trials; a mutex per
  // handle serialises callers of the same role.
  enum Role { kTracking = 0, kMapping = 1 };
  static morb_optimizer* optimizer(int device = 0, Role role = kTracking) { return slot(device, role).h; }

 private:
  static constexpr int kMaxDevices = 16;
  struct Slot { std::once_flag once; std::mutex mu; morb_optimizer* h = nullptr; };
  static Slot& slot(int device, Role role) {
    if (device < 0 || device >= kMaxDevices) throw std::runtime_error("bad device");
    static Slot slots[kMaxDevices][2];
    Slot& o = slots[device][role];
    std::call_once(o.once, [&] {
      if (morb_optimizer_create(&o.h, device) != MORB_OK) { o.h = nullptr; throw std::runtime_error(std::string("morb_optimizer_create: ") + morb_last_error()); }
    });
    if (!o.h) throw std::runtime_error("morb_optimizer_create failed earlier");
    return o;
  }
  static void check(int rc) { if (rc < 0) throw std::runtime_error(morb_last_error()); }
};

}  // namespace ORB_SLAM3
