"""PoseOptimization (tree-sum mode) per launch at 64 / 256 / 1024 frames, pinhole (600 edges) and KannalaBrandt8 rig (700 edges): developer tool for
the threads-per-frame choice (MORB_PO_NT builds through tools/ab_build.py, MORB_HIP_LIB selects the library)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import Optimizer
from morb_slam_amd.synth import make_pose_problem, make_pose_problem_fisheye


def run(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


opt = Optimizer()
for F in (64, 256, 1024):
    probs = [make_pose_problem(600, seed=s % 8) for s in range(min(F, 8))]
    st = lambda k: torch.from_numpy(np.stack([probs[s % len(probs)][k] for s in range(F)])).cuda()
    t = [st(k) for k in ("hasMP", "obs", "invSigma2", "Xw")]; pose0 = st("pose0")
    ms = run(lambda: opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"]))
    fp = [make_pose_problem_fisheye(seed=s) for s in range(4)]
    cap = max(len(p["hasMP"]) for p in fp)
    def padded(k, shape, dt, fill=0):
        a = np.full((F,) + shape, fill, dt)
        for f in range(F):
            p = fp[f % len(fp)]; a[f, :len(p[k])] = p[k]
        return torch.from_numpy(a).cuda()
    has, obs, inv, Xw = padded("hasMP", (cap,), np.uint8), padded("obs", (cap, 3), np.float32), padded("invSigma2", (cap,), np.float32, 1), padded("Xw", (cap, 3), np.float32)
    pose = torch.from_numpy(np.stack([fp[f % len(fp)]["pose0"] for f in range(F)])).cuda()
    nl = torch.from_numpy(np.array([fp[f % len(fp)]["Nleft"] for f in range(F)], np.int32)).cuda()
    cnt = torch.from_numpy(np.array([len(fp[f % len(fp)]["hasMP"]) for f in range(F)], np.int32)).cuda()
    msf = run(lambda: opt.PoseOptimizationFisheye(has, obs, inv, Xw, pose.clone(), nl, cnt, fp[0]["camL"], fp[0]["camR"], fp[0]["Trl"]))
    print(f"{F:5d} frames: pinhole {ms:.3f} ms ({F / ms:.0f} frames/ms), rig {msf:.3f} ms ({F / msf:.0f} frames/ms)")
