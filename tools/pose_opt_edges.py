import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import Optimizer
from morb_slam_amd.synth import make_pose_problem
opt = Optimizer()
for n in (60, 150, 256, 300, 512, 600, 768, 1200):
    F = 64
    probs = [make_pose_problem(n, seed=s % 8) for s in range(8)]
    st = lambda k: torch.from_numpy(np.stack([probs[s % 8][k] for s in range(F)])).cuda()
    t = [st(k) for k in ("hasMP", "obs", "invSigma2", "Xw")]; pose0 = st("pose0")
    for _ in range(2): o = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"], want_stats=True) if False else opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): o = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"])
    torch.cuda.synchronize()
    print(n, "edges:", f"{(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
