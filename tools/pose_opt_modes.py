import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import Optimizer
from morb_slam_amd.synth import make_pose_problem
F = 256
probs = [make_pose_problem(600, seed=s % 8) for s in range(F)]
t = [torch.from_numpy(np.stack([q[k] for q in probs])).cuda() for k in ("hasMP", "obs", "invSigma2", "Xw")]
pose0 = torch.from_numpy(np.stack([q["pose0"] for q in probs])).cuda()
for mode in (True, False):
    opt = Optimizer(); opt.set_exact_order(mode)
    out = None
    for _ in range(2): out = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"], out=out)
    torch.cuda.synchronize()
    poses = [pose0.clone() for _ in range(10)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(10): out = opt.PoseOptimization(t[0], t[1], t[2], t[3], poses[k], probs[0]["cam"], out=out)
    torch.cuda.synchronize()
    print("exact order" if mode else "tree sums", f"{(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per 256 frames")
