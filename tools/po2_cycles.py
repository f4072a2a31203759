"""Where k_pose_opt2's cycles go (developer build): python tools/ab_build.py po2cyc optimizer.hip -DMORB_PO_CYCLES, then
MORB_HIP_LIB=.../libmorb_hip_po2cyc.so python tools/po2_cycles.py [n_features] [has_frac]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import Optimizer
from morb_slam_amd.capi import lib
from morb_slam_amd.synth import make_pose_problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
probs = [make_pose_problem(n, seed=3)]
t = [torch.from_numpy(np.stack([q[k] for q in probs])).cuda() for k in ("hasMP", "obs", "invSigma2", "Xw")]
pose0 = torch.from_numpy(np.stack([q["pose0"] for q in probs])).cuda()
L = lib()
for mode in (True, False):
    opt = Optimizer(); opt.set_exact_order(mode)
    out = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"])
    torch.cuda.synchronize()
    z = (C.c_ulonglong * 8)(); L.morb_po2_cycles(z)
    out = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"])
    torch.cuda.synchronize()
    L.morb_po2_cycles(z)
    c = list(z); st = out[2].cpu().numpy()[0]
    names = ["kernel", "solve+bcast", "pass", "pass.compute0", "npass", "compact+load", "classify", "chain"]
    print("exact" if mode else "tree ", f"its {st[0]} trials {st[1]}", " ".join(f"{a}={b}" for a, b in zip(names, c)), f"per pass {c[2] / max(c[4], 1):.0f} per solve {c[1] / max(st[1], 1):.0f}")
