#!/usr/bin/env python3
"""Timing of the visual-inertial tracking kernels on MI355X: PreintegrateIMU and PoseInertialOptimizationLastKeyFrame."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morb_slam_amd import Optimizer
from morb_slam_amd.synth import imu_calib_diagonals, make_inertial_problem

dev = torch.device("cuda", 0)
opt = Optimizer(0)
nga, walk = imu_calib_diagonals()
for F in (1, 64, 256):
    probs = [make_inertial_problem(600, seed=s % 16, n_imu=20) for s in range(F)]
    st = lambda k: torch.from_numpy(np.stack([p[k] for p in probs])).to(dev)
    start = torch.from_numpy(np.cumsum([0] + [len(p["dt"]) for p in probs]).astype(np.int32)).to(dev)
    cat = lambda k: torch.from_numpy(np.concatenate([p[k] for p in probs])).to(dev)
    acc, gyro, dt, bias = cat("acc"), cat("gyro"), cat("dt"), st("bias")
    a = [st(k) for k in ("hasMP", "obs", "invSigma2", "Xw", "close", "kfState")]
    s0 = st("state0")
    pre = opt.PreintegrateIMU(start, acc, gyro, dt, bias, nga, walk)
    out = None
    def run():
        global out
        state = s0.clone()
        out = opt.PoseInertialOptimizationLastKeyFrame(a[0], a[1], a[2], a[3], a[4], probs[0]["cam"], probs[0]["Tbc12"], a[5], pre, state, out=out)
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): opt.PreintegrateIMU(start, acc, gyro, dt, bias, nga, walk, out=pre)
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 20
    print(f"F={F}: PoseInertialOptimizationLastKeyFrame {t1*1e3:.3f} ms/batch ({F/t1:.0f} frames/s), PreintegrateIMU(20 samples) {t2*1e3:.3f} ms/batch")
# CPU oracle for scale
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
p = make_inertial_problem(600, seed=0, n_imu=20)
t0 = time.perf_counter()
for _ in range(20):
    pr = orc.imu_preintegrate(p["bias"], nga, walk, p["acc"], p["gyro"], p["dt"])
    orc.pose_inertial_optimization_last_keyframe(p, pr)
print(f"CPU oracle: {(time.perf_counter()-t0)/20*1e3:.3f} ms/frame (1 core)")
