"""LocalBundleAdjustment at window sizes beyond the LDS-resident LDL^T (developer tool, GPU only): parity vs the oracle and time per solve
for 20 / 30 / 45 / 60 free keyframes (the reference takes every covisible keyframe, Optimizer.cc:1058-1070).  Usage: python tools/lba_sizes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from morb_slam_amd import Optimizer, BAProblem
from morb_slam_amd.synth import make_ba_problem
opt = Optimizer()
for nf in (20, 30, 45, 60):
    b = make_ba_problem(seed=5, n_free=nf, n_fixed=6, n_points=3000)
    p = BAProblem(opt, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"])
    p.solve(); kf, mp, er, st = p.results()
    t0 = time.perf_counter()
    for _ in range(5): p.solve()
    kf, mp, er, st = p.results()
    dt = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter(); its, kfe, mpe, ee, se = O.local_ba(b); dc = time.perf_counter() - t0
    print(f"free KFs {nf:3d} edges {len(b['eKF']):6d}: {dt * 1e3:7.2f} ms/solve  its {st[0]} trials {st[1]} (oracle {se[0]} {se[1]}, {dc:.1f} s)  "
          f"max|dpose| {np.abs(kf - kfe).max():.2e} max|dpoint| {np.abs(mp - mpe).max():.2e} erase flags differ {int((er != ee).sum())}", flush=True)
