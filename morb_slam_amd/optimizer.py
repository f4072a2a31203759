"""Optimizer — host-side mirror of the reference's static Optimizer functions on the hot path
(include/Optimizer.h:46-139): PoseOptimization, LocalBundleAdjustment and the inertial functions of SURVEY 8(f) N1
(IMU preintegration, PoseInertialOptimizationLastKeyFrame / LastFrame, LocalInertialBA), over the device-resident kernels."""
import ctypes as C

import numpy as np

from .capi import check, lib, ptr, stream_arg


# morb_imu_preintegrated as a float32 record: field -> (offset, length)
PREINT_FIELDS = {"dT": (0, 1), "dR": (1, 9), "dV": (10, 3), "dP": (13, 3), "JRg": (16, 9), "JVg": (25, 9), "JVa": (34, 9),
                 "JPg": (43, 9), "JPa": (52, 9), "C": (61, 225), "b": (286, 6), "nga": (292, 6), "ngaWalk": (298, 6),
                 "avgA": (304, 3), "avgW": (307, 3)}
PREINT_FLOATS = 310


class Optimizer:
    def __init__(self, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.morb_optimizer_create(C.byref(self._h), device))
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.morb_optimizer_destroy(self._h)
            self._h = None

    __del__ = close

    def set_exact_order(self, on=True):
        """PoseOptimization sums in edge order (the default: g2o's LM path decision for decision) or as a tree (~3 % faster, the trial count may differ by one); morb_optimizer_set_exact_order."""
        check(self._L.morb_optimizer_set_exact_order(self._h, 1 if on else 0))

    def info(self):
        """morb_optimizer_info: {"exact_order": 0|1, "mfma_chain": 0|1, "mfma_selftest": 1 passed | 0 device rejected | -1 not run} — which
        PoseOptimization path this handle runs."""
        a, b, c = C.c_int(-9), C.c_int(-9), C.c_int(-9)
        check(self._L.morb_optimizer_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"mfma_chain": a.value, "exact_order": b.value, "mfma_selftest": c.value}

    def PoseOptimization(self, hasMP, obs, invSigma2, Xw, pose, cam, count=None, out=None, stream=None):
        """Batched PoseOptimization.  Device tensors: hasMP u8 [F, cap], obs f32 [F, cap, 3] (x, y, uRight),
        invSigma2 f32 [F, cap], Xw f32 [F, cap, 3], pose f32 [F, 7] (in/out), count i32 [F] or None.
        Returns (nInliers i32 [F], outlier u8 [F, cap], stats i32 [F, 2]); pose is updated in place."""
        import torch
        F, cap = hasMP.shape
        if out is None:
            out = (torch.empty((F,), dtype=torch.int32, device=hasMP.device),
                   torch.zeros((F, cap), dtype=torch.uint8, device=hasMP.device),
                   torch.empty((F, 2), dtype=torch.int32, device=hasMP.device))
        st = stream_arg(stream)
        check(self._L.morb_pose_optimization_batch(self._h, F, cap, ptr(count), ptr(hasMP), ptr(obs), ptr(invSigma2), ptr(Xw),
                                                   cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"], ptr(pose),
                                                   ptr(out[1]), ptr(out[0]), ptr(out[2]), st))
        return out

    def PoseOptimizationFisheye(self, hasMP, obs, invSigma2, Xw, pose, nLeft, count, camL, camR, Trl, out=None, stream=None):
        """PoseOptimization for a fisheye rig: features [0, nLeft[f]) left camera, the rest right camera ("ToBody").
        camL/camR: 8 floats; Trl: 7 floats (quaternion xyzw + translation, left-camera -> right-camera frame)."""
        import torch
        F, cap = hasMP.shape
        if out is None:
            out = (torch.empty((F,), dtype=torch.int32, device=hasMP.device),
                   torch.zeros((F, cap), dtype=torch.uint8, device=hasMP.device),
                   torch.empty((F, 2), dtype=torch.int32, device=hasMP.device))
        a = [np.ascontiguousarray(x, np.float32) for x in (camL, camR, Trl)]
        st = stream_arg(stream)
        check(self._L.morb_pose_optimization_fisheye_batch(self._h, F, cap, ptr(count), ptr(nLeft), ptr(hasMP), ptr(obs), ptr(invSigma2),
                                                           ptr(Xw), ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(pose), ptr(out[1]), ptr(out[0]),
                                                           ptr(out[2]), st))
        return out

    # ---- visual-inertial tracking and mapping ---------------------------------------------------
    def PreintegrateIMU(self, start, acc, gyro, dt, bias, nga, walk, out=None, stream=None):
        """IMU::Preintegrated::Initialize + IntegrateNewMeasurement for many measurement sequences at once.
        Device tensors: start i32 [S+1] (sequence s owns measurements start[s]:start[s+1]), acc / gyro f32 [M, 3],
        dt f32 [M], bias f32 [S, 6] (bax bay baz bwx bwy bwz).  nga / walk: 6 floats each (diagonals of Calib::Cov /
        CovWalk).  Returns f32 [S, PREINT_FLOATS] (morb_imu_preintegrated records, see PREINT_FIELDS)."""
        import torch
        S = start.shape[0] - 1
        if out is None:
            out = torch.empty((S, PREINT_FLOATS), dtype=torch.float32, device=acc.device)
        a = [np.ascontiguousarray(x, np.float32) for x in (nga, walk)]
        st = stream_arg(stream)
        check(self._L.morb_imu_preintegrate_batch(self._h, S, ptr(start), ptr(acc), ptr(gyro), ptr(dt), ptr(bias), ptr(a[0]), ptr(a[1]),
                                                  ptr(out), st))
        return out

    def PoseInertialOptimizationLastKeyFrame(self, hasMP, obs, invSigma2, Xw, close, cam, Tbc, kfState, pre, state, bRecInit=False,
                                             count=None, want_prior=True, out=None, stream=None, rig=None, nLeft=None):
        """Batched Optimizer::PoseInertialOptimizationLastKeyFrame.  Device tensors as PoseOptimization plus close u8
        [F, cap] (mTrackDepth < 10), kfState f32 [F, 21] (fixed), pre f32 [F, PREINT_FLOATS], state f32 [F, 21] in/out
        (Rwb row-major, twb, velocity, gyro bias, acc bias).  Tbc: 12 floats (rotation row-major + translation).
        Returns (nInliers i32 [F], outlier u8 [F, cap], prior f64 [F, 246] or None)."""
        import torch
        F, cap = hasMP.shape
        if out is None:
            out = (torch.empty((F,), dtype=torch.int32, device=hasMP.device),
                   torch.zeros((F, cap), dtype=torch.uint8, device=hasMP.device),
                   torch.empty((F, 246), dtype=torch.float64, device=hasMP.device) if want_prior else None)
        t = np.ascontiguousarray(Tbc, np.float32)
        assert t.size == 12 and pre.shape[1] == PREINT_FLOATS and state.shape[1] == 21 and kfState.shape[1] == 21
        st = stream_arg(stream)
        if rig is not None:   # fisheye rig: rig = 28 floats (see morb_hip.h), nLeft i32 [F] on the device; cam is ignored
            r28 = np.ascontiguousarray(rig, np.float32)
            assert r28.size == 28 and nLeft is not None
            check(self._L.morb_pose_inertial_optimization_last_keyframe_fisheye_batch(
                self._h, F, cap, ptr(count), ptr(nLeft), ptr(hasMP), ptr(obs), ptr(invSigma2), ptr(Xw), ptr(close), ptr(r28), ptr(t),
                ptr(kfState), ptr(pre), int(bool(bRecInit)), ptr(state), ptr(out[1]), ptr(out[0]), ptr(out[2]), st))
            return out
        check(self._L.morb_pose_inertial_optimization_last_keyframe_batch(
            self._h, F, cap, ptr(count), ptr(hasMP), ptr(obs), ptr(invSigma2), ptr(Xw), ptr(close), cam["fx"], cam["fy"], cam["cx"],
            cam["cy"], cam["bf"], ptr(t), ptr(kfState), ptr(pre), int(bool(bRecInit)), ptr(state), ptr(out[1]), ptr(out[0]),
            ptr(out[2]), st))
        return out

    def PoseInertialOptimizationLastFrame(self, hasMP, obs, invSigma2, Xw, close, cam, Tbc, prevState, preFrame, preKF, prevPrior, state,
                                          bRecInit=False, count=None, want_prior=True, out=None, stream=None, rig=None, nLeft=None):
        """Batched Optimizer::PoseInertialOptimizationLastFrame: as PoseInertialOptimizationLastKeyFrame, but the previous
        frame's state (prevState f32 [F, 21]) is free and carries the prior prevPrior f64 [F, 246] (a previous call's third
        result); preFrame = preintegration since the previous frame, preKF = since the last keyframe."""
        import torch
        F, cap = hasMP.shape
        if out is None:
            out = (torch.empty((F,), dtype=torch.int32, device=hasMP.device),
                   torch.zeros((F, cap), dtype=torch.uint8, device=hasMP.device),
                   torch.empty((F, 246), dtype=torch.float64, device=hasMP.device) if want_prior else None)
        t = np.ascontiguousarray(Tbc, np.float32)
        assert t.size == 12 and preFrame.shape[1] == PREINT_FLOATS and preKF.shape[1] == PREINT_FLOATS
        assert state.shape[1] == 21 and prevState.shape[1] == 21 and prevPrior.shape[1] == 246 and prevPrior.dtype == torch.float64
        st = stream_arg(stream)
        if rig is not None:
            r28 = np.ascontiguousarray(rig, np.float32)
            assert r28.size == 28 and nLeft is not None
            check(self._L.morb_pose_inertial_optimization_last_frame_fisheye_batch(
                self._h, F, cap, ptr(count), ptr(nLeft), ptr(hasMP), ptr(obs), ptr(invSigma2), ptr(Xw), ptr(close), ptr(r28), ptr(t),
                ptr(prevState), ptr(preFrame), ptr(preKF), ptr(prevPrior), int(bool(bRecInit)), ptr(state), ptr(out[1]), ptr(out[0]),
                ptr(out[2]), st))
            return out
        check(self._L.morb_pose_inertial_optimization_last_frame_batch(
            self._h, F, cap, ptr(count), ptr(hasMP), ptr(obs), ptr(invSigma2), ptr(Xw), ptr(close), cam["fx"], cam["fy"], cam["cx"],
            cam["cy"], cam["bf"], ptr(t), ptr(prevState), ptr(preFrame), ptr(preKF), ptr(prevPrior), int(bool(bRecInit)), ptr(state),
            ptr(out[1]), ptr(out[0]), ptr(out[2]), st))
        return out

    def LocalInertialBA(self, kfState, kfKind, mpPos, mpClose, eKF, eMP, eObs, eInvSigma2, iKF1, iKF2, iPre, iRobust, iInfoScale, cam, Tbc,
                        bLarge=False, rig=None, eRight=None):
        """Optimizer::LocalInertialBA on host numpy arrays (see morb_local_inertial_ba).  iPre: float32 [nI, PREINT_FLOATS].
        Returns (kfState, mpPos, eraseFlag, stats) with stats = (outer LM iterations, LM trials, ok)."""
        a = [np.ascontiguousarray(kfState, np.float32).copy(), np.ascontiguousarray(kfKind, np.uint8),
             np.ascontiguousarray(mpPos, np.float32).copy(), np.ascontiguousarray(mpClose, np.uint8),
             np.ascontiguousarray(eKF, np.int32), np.ascontiguousarray(eMP, np.int32), np.ascontiguousarray(eObs, np.float32),
             np.ascontiguousarray(eInvSigma2, np.float32), np.ascontiguousarray(iKF1, np.int32), np.ascontiguousarray(iKF2, np.int32),
             np.ascontiguousarray(iPre, np.float32), np.ascontiguousarray(iRobust, np.uint8), np.ascontiguousarray(iInfoScale, np.float32),
             np.ascontiguousarray(Tbc, np.float32)]
        nI = len(a[8])
        assert a[10].shape == (nI, PREINT_FLOATS) and a[0].shape[1] == 21 and a[6].shape[1] == 3
        erase = np.zeros(len(a[4]), np.uint8); stats = np.zeros(3, np.int32)
        if rig is not None:   # fisheye rig: rig = 28 floats, eRight uint8 [nE]; cam is ignored
            r28 = np.ascontiguousarray(rig, np.float32); er = np.ascontiguousarray(eRight, np.uint8)
            assert r28.size == 28 and len(er) == len(a[4])
            check(self._L.morb_local_inertial_ba_fisheye(self._h, len(a[0]), ptr(a[0]), ptr(a[1]), len(a[2]), ptr(a[2]), ptr(a[3]), len(a[4]),
                                                         ptr(a[4]), ptr(a[5]), ptr(a[6]), ptr(er), ptr(a[7]), nI, ptr(a[8]), ptr(a[9]),
                                                         ptr(a[10]), ptr(a[11]), ptr(a[12]), ptr(r28), ptr(a[13]), int(bool(bLarge)),
                                                         ptr(erase), ptr(stats)))
            return a[0], a[2], erase, stats
        check(self._L.morb_local_inertial_ba(self._h, len(a[0]), ptr(a[0]), ptr(a[1]), len(a[2]), ptr(a[2]), ptr(a[3]), len(a[4]), ptr(a[4]),
                                             ptr(a[5]), ptr(a[6]), ptr(a[7]), nI, ptr(a[8]), ptr(a[9]), ptr(a[10]), ptr(a[11]), ptr(a[12]),
                                             cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"], ptr(a[13]), int(bool(bLarge)),
                                             ptr(erase), ptr(stats)))
        return a[0], a[2], erase, stats

    def LocalBundleAdjustment(self, kfPose, kfFixed, mpPos, eKF, eMP, eObs, eInvSigma2, cam, inertial=False, stop=False, mode=0,
                              rig=None):
        """One-shot LocalBundleAdjustment on host numpy arrays; returns (kfPose, mpPos, eraseFlag, stats)."""
        p = BAProblem(self, kfPose, kfFixed, mpPos, eKF, eMP, eObs, eInvSigma2, cam, inertial, rig=rig)
        p.set_mode(mode)
        if stop:
            p.set_stop(True)
        p.solve()
        return p.results()


def local_bundle_adjustment_oneshot(opt, kfPose, kfFixed, mpPos, eKF, eMP, eObs, eInvSigma2, cam, inertial=False, stop_flag=None):
    """The one-shot ABI entry (`morb_local_bundle_adjustment`), as the reference binding calls it: `stop_flag` is a one-byte
    numpy array standing for `bool* pbStopFlag` (another thread may set it while the call runs)."""
    L = lib()
    a = [np.ascontiguousarray(kfPose, np.float32).copy(), np.ascontiguousarray(kfFixed, np.uint8),
         np.ascontiguousarray(mpPos, np.float32).copy(), np.ascontiguousarray(eKF, np.int32), np.ascontiguousarray(eMP, np.int32),
         np.ascontiguousarray(eObs, np.float32), np.ascontiguousarray(eInvSigma2, np.float32)]
    erase = np.zeros(len(a[3]), np.uint8); stats = np.zeros(2, np.int32)
    L.morb_local_bundle_adjustment.restype = C.c_int
    L.morb_local_bundle_adjustment.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                               C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.morb_local_bundle_adjustment(opt._h, len(a[0]), ptr(a[0]), ptr(a[1]), len(a[2]), ptr(a[2]), len(a[3]), ptr(a[3]), ptr(a[4]),
                                         ptr(a[5]), ptr(a[6]), cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"],
                                         1 if inertial else 0, ptr(stop_flag) if stop_flag is not None else None, ptr(erase), ptr(stats)))
    return a[0], a[2], erase, stats


def local_bundle_adjustment_fisheye_oneshot(opt, kfPose, kfFixed, mpPos, eKF, eMP, eObs2, eRight, eInvSigma2, camL, camR, Trl, inertial=False,
                                            stop_flag=None):
    """The one-shot ABI entry on the KannalaBrandt8 rig (`morb_local_bundle_adjustment_fisheye`)."""
    L = lib()
    a = [np.ascontiguousarray(kfPose, np.float32).copy(), np.ascontiguousarray(kfFixed, np.uint8),
         np.ascontiguousarray(mpPos, np.float32).copy(), np.ascontiguousarray(eKF, np.int32), np.ascontiguousarray(eMP, np.int32),
         np.ascontiguousarray(eObs2, np.float32), np.ascontiguousarray(eRight, np.uint8), np.ascontiguousarray(eInvSigma2, np.float32),
         np.ascontiguousarray(camL, np.float32), np.ascontiguousarray(camR, np.float32), np.ascontiguousarray(Trl, np.float32)]
    erase = np.zeros(len(a[3]), np.uint8); stats = np.zeros(2, np.int32)
    L.morb_local_bundle_adjustment_fisheye.restype = C.c_int
    L.morb_local_bundle_adjustment_fisheye.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + \
        [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    check(L.morb_local_bundle_adjustment_fisheye(opt._h, len(a[0]), ptr(a[0]), ptr(a[1]), len(a[2]), ptr(a[2]), len(a[3]), ptr(a[3]), ptr(a[4]),
                                                 ptr(a[5]), ptr(a[6]), ptr(a[7]), ptr(a[8]), ptr(a[9]), ptr(a[10]), 1 if inertial else 0,
                                                 ptr(stop_flag) if stop_flag is not None else None, ptr(erase), ptr(stats)))
    return a[0], a[2], erase, stats


class BAProblem:
    """A LocalBundleAdjustment graph resident in HBM (create once, solve repeatedly).
    rig = dict(eRight uint8 [nE], camL, camR (8 floats each), Trl (7 floats)) selects the KannalaBrandt8 stereo rig
    (eObs is then [nE, 2]; cam is ignored)."""

    def __init__(self, opt, kfPose, kfFixed, mpPos, eKF, eMP, eObs, eInvSigma2, cam, inertial=False, rig=None):
        self._L = lib()
        self._opt = opt
        self._h = C.c_void_p()
        a = [np.ascontiguousarray(kfPose, np.float32), np.ascontiguousarray(kfFixed, np.uint8),
             np.ascontiguousarray(mpPos, np.float32), np.ascontiguousarray(eKF, np.int32), np.ascontiguousarray(eMP, np.int32),
             np.ascontiguousarray(eObs, np.float32), np.ascontiguousarray(eInvSigma2, np.float32)]
        self.nKF, self.nMP, self.nE = len(a[0]), len(a[2]), len(a[3])
        self._init = (a[0], a[2])
        if rig is not None:
            r = [np.ascontiguousarray(rig["eRight"], np.uint8), np.ascontiguousarray(rig["camL"], np.float32),
                 np.ascontiguousarray(rig["camR"], np.float32), np.ascontiguousarray(rig["Trl"], np.float32)]
            assert a[5].shape == (self.nE, 2)
            check(self._L.morb_ba_problem_create_fisheye(opt._h, C.byref(self._h), self.nKF, ptr(a[0]), ptr(a[1]), self.nMP, ptr(a[2]),
                                                         self.nE, ptr(a[3]), ptr(a[4]), ptr(a[5]), ptr(r[0]), ptr(a[6]), ptr(r[1]),
                                                         ptr(r[2]), ptr(r[3]), 1 if inertial else 0))
            return
        check(self._L.morb_ba_problem_create(opt._h, C.byref(self._h), self.nKF, ptr(a[0]), ptr(a[1]), self.nMP, ptr(a[2]),
                                             self.nE, ptr(a[3]), ptr(a[4]), ptr(a[5]), ptr(a[6]), cam["fx"], cam["fy"],
                                             cam["cx"], cam["cy"], cam["bf"], 1 if inertial else 0))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.morb_ba_problem_destroy(self._h)
            self._h = None

    __del__ = close

    def set_mode(self, mode):
        """0 = grid (phase kernels over the whole GPU, default), 1 = one persistent workgroup."""
        check(self._L.morb_ba_set_mode(self._h, int(mode)))

    def schur_profile(self, iters=50):
        """(ms per launch, MFMA flops issued per launch, flops of the sparse block-pair form) of the Schur product."""
        ms = C.c_float(); fl = C.c_double(); uf = C.c_double()
        self._L.morb_ba_schur_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        check(self._L.morb_ba_schur_profile(self._h, int(iters), C.byref(ms), C.byref(fl), C.byref(uf)))
        return ms.value, fl.value, uf.value

    def set_stop(self, on):
        check(self._L.morb_ba_set_stop(self._h, 1 if on else 0))

    def solve(self, stream=None):
        check(self._L.morb_ba_solve(self._h, stream_arg(stream)))

    def results(self):
        kf = self._init[0].copy(); mp = self._init[1].copy()
        erase = np.zeros(self.nE, np.uint8); stats = np.zeros(2, np.int32)
        check(self._L.morb_ba_results(self._h, ptr(kf), ptr(mp), ptr(erase), ptr(stats)))
        return kf, mp, erase, stats
