#!/usr/bin/env python3
"""Timing of LocalInertialBA on MI355X vs the CPU oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from morb_slam_amd import Optimizer
from morb_slam_amd.synth import imu_calib_diagonals, make_inertial_ba_problem
import oracle_lib as orc
opt = Optimizer(0)
nga, walk = imu_calib_diagonals()
for large, n_opt, npts in ((False, 10, 1500), (False, 10, 3000), (True, 25, 3000)):
    p = make_inertial_ba_problem(n_opt=n_opt, seed=1, n_points=npts)
    pre = np.stack([orc.imu_preintegrate(p["bias"], nga, walk, p["acc"][a:b], p["gyro"][a:b], p["dt"][a:b]) for a, b in zip(p["imuStart"][:-1], p["imuStart"][1:])])
    args = (p["kfState"], p["kfKind"], p["mpPos"], p["mpClose"], p["eKF"], p["eMP"], p["eObs"], p["eInvSigma2"], p["iKF1"], p["iKF2"], pre, p["iRobust"], p["iInfoScale"], p["cam"], p["Tbc12"])
    opt.LocalInertialBA(*args, bLarge=large)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); _, _, _, st = opt.LocalInertialBA(*args, bLarge=large); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]   # median: an occasional host hiccup (tens of ms) would dominate a mean of a few calls
    t0 = time.perf_counter(); r = orc.local_inertial_ba(p, pre, bLarge=large); dc = time.perf_counter() - t0
    print(f"LocalInertialBA N={n_opt} points={npts} edges={len(p['eKF'])} bLarge={large}: {dt*1e3:.2f} ms/solve ({st[0]} its, {st[1]} trials) -> {st[0]/dt:.0f} LM it/s; CPU oracle {dc*1e3:.1f} ms -> {r[4][0]/dc:.0f} LM it/s")
