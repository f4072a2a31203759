#!/usr/bin/env python3
"""Micro-benchmark of the optimisers on MI355X (BASELINE config 5: LocalBA 20 KF x 3000 MP; config 3: PoseOptimization)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from morb_slam_amd import Optimizer, BAProblem
from morb_slam_amd.synth import make_ba_problem, make_pose_problem


def main():
    opt = Optimizer()
    b = make_ba_problem(seed=1)
    p = BAProblem(opt, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"])
    p.solve(); kf, mp, er, st = p.results()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        p.solve()
    kf, mp, er, st = p.results()
    dt = (time.perf_counter() - t0) / n
    print(f"LBA: {len(b['eKF'])} edges, {st[0]} outer its, {st[1]} trials: {dt*1e3:.3f} ms/solve -> {st[0]/dt:.0f} LM iters/s")
    import oracle_lib as O
    t0 = time.perf_counter(); its, *_ = O.local_ba(b); dc = time.perf_counter() - t0
    print(f"oracle CPU: {dc*1e3:.1f} ms -> {its/dc:.0f} LM iters/s")
    F = 256
    probs = [make_pose_problem(600, seed=s % 8) for s in range(F)]
    cap = 600
    has = np.stack([q["hasMP"] for q in probs]); obs = np.stack([q["obs"] for q in probs])
    inv = np.stack([q["invSigma2"] for q in probs]); Xw = np.stack([q["Xw"] for q in probs]); pose = np.stack([q["pose0"] for q in probs])
    t = [torch.from_numpy(a).cuda() for a in (has, obs, inv, Xw)]
    pose0 = torch.from_numpy(pose).cuda()
    out = None
    for _ in range(3):
        ps = pose0.clone(); out = opt.PoseOptimization(t[0], t[1], t[2], t[3], ps, probs[0]["cam"], out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ps = pose0.clone(); out = opt.PoseOptimization(t[0], t[1], t[2], t[3], ps, probs[0]["cam"], out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"PoseOptimization: {F} frames x 600 pts: {dt*1e3:.3f} ms/batch -> {F/dt:.0f} frames/s; stats {out[2][0].tolist()}")
    t0 = time.perf_counter(); O.pose_optimization(probs[0]); dc = time.perf_counter() - t0
    print(f"oracle CPU PoseOptimization: {dc*1e3:.2f} ms/frame")


if __name__ == "__main__":
    main()
