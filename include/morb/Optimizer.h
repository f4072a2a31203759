// Drop-in adapter: ORB_SLAM3::Optimizer (reference include/Optimizer.h:46-139) over libmorb_hip.so — the two static methods of the
// hot path, PoseOptimization and LocalBundleAdjustment.  As in ORBmatcher.h the arguments are plain views of what the reference
// methods read from Frame / KeyFrame / MapPoint / Map and write back to them; INTEGRATION.md shows the glue inside the reference tree.
#pragma once
#include <cmath>
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
  // KannalaBrandt8 rig (pFrame->mpCamera2 != NULL, Optimizer.cc:880-946): features [0, nLeft) on the left camera, the rest on the right one;
  // rig28 = left KB8 parameters (8), right ones (8), rotation (9, row-major) + translation (3) of GetRelativePoseTrl().  NULL = pinhole.
  int nLeft = -1;
  const float* rig28 = nullptr;
};
// What PoseInertialOptimizationLastKeyFrame / LastFrame read and write (Optimizer.cc:4391-5161): the frame's visual edges as above plus
// close[i] (mTrackDepth < 10), the IMU states as 21 floats (Rwb row-major, twb, velocity, gyro bias, acc bias), the preintegrations,
// mImuCalib.mTbc, and the ConstraintPoseImu priors as 246 doubles (state in FP64, then the 15 x 15 H row-major).
struct PoseInertialView {
  int N = 0, nLeft = -1;
  const uint8_t* hasMapPoint = nullptr;
  const float* obs = nullptr;
  const float* invSigma2 = nullptr;
  const float* worldPos = nullptr;
  const uint8_t* close = nullptr;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  const float* rig28 = nullptr;
  float Tbc12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  float state[21] = {0};         // in / out: the frame
  float otherState[21] = {0};    // LastKeyFrame: pFrame->mpLastKeyFrame (fixed); LastFrame: pFrame->mpPrevFrame (free)
  morb_imu_preintegrated pre{};  // LastKeyFrame: mpImuPreintegrated; LastFrame: mpImuPreintegratedFrame
  morb_imu_preintegrated preKF{};   // LastFrame only: mpImuPreintegrated (the random-walk information)
  double prevPrior[246] = {0};   // LastFrame only: pFp->mpcpi
  double prior[246] = {0};       // out: the frame's new mpcpi
  std::vector<uint8_t> mvbOutlier;
};
// The graph Optimizer::LocalInertialBA assembles (Optimizer.cc:2337-2768), flattened as morb_local_inertial_ba takes it.
struct LocalInertialBAView {
  int nKF = 0, nMP = 0, nE = 0, nI = 0;
  float* kfState = nullptr;            // [nKF][21] in / out
  const uint8_t* kfKind = nullptr;     // [nKF] 0 temporal optimizable, 1 the fixed keyframe before the window, 2 fixed observer
  float* mpPos = nullptr;              // [nMP][3] in / out
  const uint8_t* mpClose = nullptr;
  const int *eKF = nullptr, *eMP = nullptr;
  const float *eObs = nullptr, *eInvSigma2 = nullptr;
  const uint8_t* eRight = nullptr;     // rig only
  const int *iKF1 = nullptr, *iKF2 = nullptr;
  const morb_imu_preintegrated* iPre = nullptr;
  const uint8_t* iRobust = nullptr;
  const float* iInfoScale = nullptr;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  const float* rig28 = nullptr;
  float Tbc12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  bool bLarge = false;
  std::vector<uint8_t> eraseFlag;      // out [nE]
  int outerIterations = 0, lmTrials = 0;
  bool ok = false;                     // false = "FAIL LOCAL-INERTIAL BA" (nothing to write back)
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
  const float* rig28 = nullptr;     // KannalaBrandt8 rig (as PoseOptimizationView::rig28): edges with eRight[e] != 0 are right-camera observations
  const uint8_t* eRight = nullptr;
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
    morb_adapter::StreamScope scope_(morb_optimizer_stream(o.h));   // uploads, the kernel, downloads: this handle's stream, never the null stream
    // per-thread, per-device staging buffers that only grow: Tracking calls this once per frame, and eight hipMalloc / hipFree pairs
    // per call cost more than the optimisation itself
    struct Staging { DeviceBuffer<uint8_t> has, outl; DeviceBuffer<float> obs, inv, Xw, pose; DeviceBuffer<int> nin, cnt, nl; };
    static thread_local Staging per_device[kMaxDevices];
    Staging& s = per_device[device];
    s.has.assign(f.hasMapPoint, N); s.outl.resize(N);
    // mvbOutlier is only written for features that hold a map point (Optimizer.cc:817, :860): the others keep what they had
    if ((int)f.mvbOutlier.size() == N) s.outl.upload(f.mvbOutlier.data(), N); else s.outl.fill_bytes(0);
    s.obs.assign(f.obs, (size_t)N * 3); s.inv.assign(f.invSigma2, N); s.Xw.assign(f.worldPos, (size_t)N * 3); s.pose.assign(f.pose, 7);
    s.nin.resize(1); s.cnt.assign(&N, 1);
    if (f.rig28) {
      float trl7[7];
      rot_to_pose7(f.rig28 + 16, trl7);
      s.nl.assign(&f.nLeft, 1);
      check(morb_pose_optimization_fisheye_batch(o.h, 1, N, s.cnt.get(), s.nl.get(), s.has.get(), s.obs.get(), s.inv.get(), s.Xw.get(), f.rig28, f.rig28 + 8,
                                                 trl7, s.pose.get(), s.outl.get(), s.nin.get(), nullptr, nullptr));
    } else
    check(morb_pose_optimization_batch(o.h, 1, N, s.cnt.get(), s.has.get(), s.obs.get(), s.inv.get(), s.Xw.get(), f.fx, f.fy, f.cx, f.cy, f.mbf,
                                       s.pose.get(), s.outl.get(), s.nin.get(), nullptr, nullptr));
    morb_adapter::sync_current_stream();   // (the tracking handle's stream: a LocalBundleAdjustment on the mapping handle keeps running)
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
    if (g.rig28) {
      float trl7[7];
      rot_to_pose7(g.rig28 + 16, trl7);
      std::vector<float> obs2((size_t)g.nE * 2);
      for (int e = 0; e < g.nE; ++e) { obs2[2 * e] = g.eObs[3 * e]; obs2[2 * e + 1] = g.eObs[3 * e + 1]; }
      check(morb_local_bundle_adjustment_fisheye(o.h, g.nKF, g.kfPose, g.kfFixed, g.nMP, g.mpPos, g.nE, g.eKF, g.eMP, obs2.data(), g.eRight, g.eInvSigma2,
                                                 g.rig28, g.rig28 + 8, trl7, g.inertialMap ? 1 : 0, reinterpret_cast<const unsigned char*>(pbStopFlag),
                                                 g.eraseFlag.data(), stats));
    } else
    check(morb_local_bundle_adjustment(o.h, g.nKF, g.kfPose, g.kfFixed, g.nMP, g.mpPos, g.nE, g.eKF, g.eMP, g.eObs, g.eInvSigma2, g.fx,
                                       g.fy, g.cx, g.cy, g.mbf, g.inertialMap ? 1 : 0, reinterpret_cast<const unsigned char*>(pbStopFlag),
                                       g.eraseFlag.data(), stats));
    g.outerIterations = stats[0]; g.lmTrials = stats[1];
  }

  // static int PoseInertialOptimizationLastKeyFrame(Frame* pFrame, bool bRecInit)  Optimizer.h:87 / ...LastFrame  Optimizer.h:88: one frame per call
  static int PoseInertialOptimizationLastKeyFrame(PoseInertialView& v, bool bRecInit = false, int device = 0) { return pose_inertial(v, bRecInit, false, device); }
  static int PoseInertialOptimizationLastFrame(PoseInertialView& v, bool bRecInit = false, int device = 0) { return pose_inertial(v, bRecInit, true, device); }

  // static void LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int&, int&, int&, int&, bool bLarge, bool bRecInit)  Optimizer.h:97-101
  // (bRecInit is folded into iRobust by the caller, Optimizer.cc:2563)
  static void LocalInertialBA(LocalInertialBAView& g, int device = 0) {
    g.eraseFlag.assign(g.nE, 0);
    int stats[3] = {0, 0, 0};
    Slot& o = slot(device, kMapping);
    std::lock_guard<std::mutex> lock(o.mu);
    morb_adapter::hip_check(hipSetDevice(device), "hipSetDevice");
    if (g.rig28)
      check(morb_local_inertial_ba_fisheye(o.h, g.nKF, g.kfState, g.kfKind, g.nMP, g.mpPos, g.mpClose, g.nE, g.eKF, g.eMP, g.eObs, g.eRight, g.eInvSigma2, g.nI,
                                           g.iKF1, g.iKF2, g.iPre, g.iRobust, g.iInfoScale, g.rig28, g.Tbc12, g.bLarge ? 1 : 0, g.eraseFlag.data(), stats));
    else
      check(morb_local_inertial_ba(o.h, g.nKF, g.kfState, g.kfKind, g.nMP, g.mpPos, g.mpClose, g.nE, g.eKF, g.eMP, g.eObs, g.eInvSigma2, g.nI, g.iKF1, g.iKF2,
                                   g.iPre, g.iRobust, g.iInfoScale, g.fx, g.fy, g.cx, g.cy, g.mbf, g.Tbc12, g.bLarge ? 1 : 0, g.eraseFlag.data(), stats));
    g.outerIterations = stats[0]; g.lmTrials = stats[1]; g.ok = stats[2] != 0;
  }

  // ---- the reference's own signatures (include/Optimizer.h:67-101) as static member templates: the call sites of src/Tracking.cc and
  // src/LocalMapping.cc compile unchanged (definitions: Optimizer_reference.h, included below) ----
  template <class FrameT> static int PoseOptimization(FrameT* pFrame);
  template <class FrameT> static int PoseInertialOptimizationLastKeyFrame(FrameT* pFrame, bool bRecInit = false);
  template <class FrameT> static int PoseInertialOptimizationLastFrame(FrameT* pFrame, bool bRecInit = false);
  template <class KF, class MapT>
  static void LocalBundleAdjustment(KF* pKF, bool* pbStopFlag, MapT* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges);
  template <class KF, class MapT>
  static void LocalInertialBA(KF* pKF, bool* pbStopFlag, MapT* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges, bool bLarge = false,
                              bool bRecInit = false);

  // The reference's Optimizer is a stateless static class entered concurrently from Tracking (PoseOptimization, every frame) and from
  // LocalMapping (LocalBundleAdjustment, hundreds of milliseconds of LM trials): each role has its own handle — own stream, own
  // workspace — per device, created once (std::call_once), so a tracked frame never queues behind local-mapping trials; a mutex per
  // handle serialises callers of the same role.
  enum Role { kTracking = 0, kMapping = 1 };
  static morb_optimizer* optimizer(int device = 0, Role role = kTracking) { return slot(device, role).h; }

 private:
  template <class FrameT> static void write_back_inertial(FrameT* pFrame, const PoseInertialView& v);
  template <class SE3> static SE3 se3_from_matrix(const double R[9], const double t[3]);
  // rotation (9, row-major) + translation (3) -> unit quaternion xyzw + translation (the *_fisheye pose entries take Trl that way)
  static void rot_to_pose7(const float* Rt12, float out[7]) {
    const float* R = Rt12;
    const float tr = R[0] + R[4] + R[8];
    float q[4];   // x y z w
    if (tr > 0.f) { const float s = std::sqrt(tr + 1.f) * 2.f; q[3] = 0.25f * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { const float s = std::sqrt(1.f + R[0] - R[4] - R[8]) * 2.f; q[3] = (R[7] - R[5]) / s; q[0] = 0.25f * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { const float s = std::sqrt(1.f + R[4] - R[0] - R[8]) * 2.f; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25f * s; q[2] = (R[5] + R[7]) / s; }
    else { const float s = std::sqrt(1.f + R[8] - R[0] - R[4]) * 2.f; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25f * s; }
    for (int k = 0; k < 4; ++k) out[k] = q[k];
    for (int k = 0; k < 3; ++k) out[4 + k] = Rt12[9 + k];
  }
  // one frame through morb_pose_inertial_optimization_last_{keyframe,frame}[_fisheye]_batch
  static int pose_inertial(PoseInertialView& v, bool bRecInit, bool lastFrame, int device) {
    using morb_adapter::DeviceBuffer;
    if (v.N <= 0) return 0;
    const int N = v.N;
    Slot& o = slot(device, kTracking);
    std::lock_guard<std::mutex> lock(o.mu);
    morb_adapter::hip_check(hipSetDevice(device), "hipSetDevice");
    morb_adapter::StreamScope scope_(morb_optimizer_stream(o.h));
    struct Staging { DeviceBuffer<uint8_t> has, close, outl; DeviceBuffer<float> obs, inv, Xw, st, other; DeviceBuffer<int> nin, cnt, nl;
                     DeviceBuffer<morb_imu_preintegrated> pre, preKF; DeviceBuffer<double> prior, prevPrior; };
    static thread_local Staging per_device[kMaxDevices];
    Staging& s = per_device[device];
    s.has.assign(v.hasMapPoint, N); s.close.assign(v.close, N); s.outl.resize(N);
    if ((int)v.mvbOutlier.size() == N) s.outl.upload(v.mvbOutlier.data(), N); else s.outl.fill_bytes(0);
    s.obs.assign(v.obs, (size_t)N * 3); s.inv.assign(v.invSigma2, N); s.Xw.assign(v.worldPos, (size_t)N * 3);
    s.st.assign(v.state, 21); s.other.assign(v.otherState, 21); s.pre.assign(&v.pre, 1); s.nin.resize(1); s.cnt.assign(&N, 1); s.prior.resize(246);
    if (v.rig28) s.nl.assign(&v.nLeft, 1);
    if (!lastFrame) {
      if (v.rig28)
        check(morb_pose_inertial_optimization_last_keyframe_fisheye_batch(o.h, 1, N, s.cnt.get(), s.nl.get(), s.has.get(), s.obs.get(), s.inv.get(), s.Xw.get(),
                                                                          s.close.get(), v.rig28, v.Tbc12, s.other.get(), s.pre.get(), bRecInit ? 1 : 0,
                                                                          s.st.get(), s.outl.get(), s.nin.get(), s.prior.get(), nullptr));
      else
        check(morb_pose_inertial_optimization_last_keyframe_batch(o.h, 1, N, s.cnt.get(), s.has.get(), s.obs.get(), s.inv.get(), s.Xw.get(), s.close.get(), v.fx,
                                                                  v.fy, v.cx, v.cy, v.mbf, v.Tbc12, s.other.get(), s.pre.get(), bRecInit ? 1 : 0, s.st.get(),
                                                                  s.outl.get(), s.nin.get(), s.prior.get(), nullptr));
    } else {
      s.preKF.assign(&v.preKF, 1); s.prevPrior.assign(v.prevPrior, 246);
      if (v.rig28)
        check(morb_pose_inertial_optimization_last_frame_fisheye_batch(o.h, 1, N, s.cnt.get(), s.nl.get(), s.has.get(), s.obs.get(), s.inv.get(), s.Xw.get(),
                                                                       s.close.get(), v.rig28, v.Tbc12, s.other.get(), s.pre.get(), s.preKF.get(),
                                                                       s.prevPrior.get(), bRecInit ? 1 : 0, s.st.get(), s.outl.get(), s.nin.get(),
                                                                       s.prior.get(), nullptr));
      else
        check(morb_pose_inertial_optimization_last_frame_batch(o.h, 1, N, s.cnt.get(), s.has.get(), s.obs.get(), s.inv.get(), s.Xw.get(), s.close.get(), v.fx, v.fy,
                                                               v.cx, v.cy, v.mbf, v.Tbc12, s.other.get(), s.pre.get(), s.preKF.get(), s.prevPrior.get(),
                                                               bRecInit ? 1 : 0, s.st.get(), s.outl.get(), s.nin.get(), s.prior.get(), nullptr));
    }
    morb_adapter::sync_current_stream();
    s.st.download(v.state, 21); s.prior.download(v.prior, 246);
    v.mvbOutlier = s.outl.to_host();
    return s.nin.to_host()[0];
  }
  static constexpr int kMaxDevices = 16;
  struct Slot { std::once_flag once; std::mutex mu; morb_optimizer* h = nullptr; };
  static Slot& slot(int device, Role role) {
    if (device < 0 || device >= kMaxDevices) throw std::runtime_error("bad device");
    static Slot slots[kMaxDevices][2];
    Slot& o = slots[device][role];
    std::call_once(o.once, [&] {
      if (morb_optimizer_create(&o.h, device) != MORB_OK) { o.h = nullptr; throw std::runtime_error(std::string("morb_optimizer_create: ") + morb_last_error()); }
      morb_optimizer_set_exact_order(o.h, 1);   // the drop-in follows g2o's LM path decision for decision (edge-order sums); a frame at a time the cost is latency, not throughput
    });
    if (!o.h) throw std::runtime_error("morb_optimizer_create failed earlier");
    return o;
  }
  static void check(int rc) { if (rc < 0) throw std::runtime_error(morb_last_error()); }
};

}  // namespace ORB_SLAM3

#include "Optimizer_reference.h"
