#!/usr/bin/env python3
"""Phase timing of k_pose_inertial (needs a build with -DMORB_INERTIAL_TIMING: HIPCC_EXTRA=-DMORB_INERTIAL_TIMING)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morb_slam_amd import Optimizer, capi
from morb_slam_amd.synth import imu_calib_diagonals, make_inertial_problem
dev = torch.device("cuda", 0); opt = Optimizer(0); L = capi.lib()
nga, walk = imu_calib_diagonals()
probs = [make_inertial_problem(600, seed=0, n_imu=20)]
st = lambda k: torch.from_numpy(np.stack([p[k] for p in probs])).to(dev)
a = [st(k) for k in ("hasMP", "obs", "invSigma2", "Xw", "close", "kfState")]
start = torch.tensor([0, 20], dtype=torch.int32, device=dev)
ins = [st("acc")[0], st("gyro")[0], st("dt")[0], st("bias")]
pre = opt.PreintegrateIMU(start, ins[0], ins[1], ins[2], ins[3], nga, walk)
s0 = st("state0")
for rep in range(2):
    L.morb_inertial_timing(None, 1)
    state = s0.clone()
    opt.PoseInertialOptimizationLastKeyFrame(a[0], a[1], a[2], a[3], a[4], probs[0]["cam"], probs[0]["Tbc12"], a[5], pre, state)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    L.morb_inertial_timing(out, 0)
names = ["setup", "loop-top/classify-in", "visual", "dense-edges", "OmegaJ", "H-assembly", "solve", "update", "classify", "final", "prior"]
tot = sum(out)
for n, v in zip(names, out):
    print(f"{n:22s} {v / 100.0:9.1f} us")
print("total", tot / 100.0, "us")
