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
def show(title, out):
    print("==", title)
    for n, v in zip(names, out):
        print(f"{n:22s} {v / 100.0:9.1f} us")
    print("total", sum(out) / 100.0, "us")
show("PoseInertialOptimizationLastKeyFrame", out)
if len(sys.argv) > 1 and sys.argv[1] == "lf":   # ... and LastFrame: frame B against frame A (its state and prior from the call above)
    from morb_slam_amd.synth import make_inertial_sequence
    pA, pB = make_inertial_sequence(600, seed=0, n_imu=20)
    tb = lambda p, k: torch.from_numpy(p[k][None]).to(dev)
    startA = torch.tensor([0, len(pA["dt"])], dtype=torch.int32, device=dev)
    preA = opt.PreintegrateIMU(startA, tb(pA, "acc")[0], tb(pA, "gyro")[0], tb(pA, "dt")[0], tb(pA, "bias"), nga, walk)
    stA = tb(pA, "state0").clone()
    oA = opt.PoseInertialOptimizationLastKeyFrame(tb(pA, "hasMP"), tb(pA, "obs"), tb(pA, "invSigma2"), tb(pA, "Xw"), tb(pA, "close"), pA["cam"], pA["Tbc12"],
                                                  tb(pA, "kfState"), preA, stA)
    startF = torch.tensor([0, len(pB["dtF"])], dtype=torch.int32, device=dev); startK = torch.tensor([0, len(pB["dt"])], dtype=torch.int32, device=dev)
    preF = opt.PreintegrateIMU(startF, tb(pB, "accF")[0], tb(pB, "gyroF")[0], tb(pB, "dtF")[0], tb(pA, "bias"), nga, walk)
    preK = opt.PreintegrateIMU(startK, tb(pB, "acc")[0], tb(pB, "gyro")[0], tb(pB, "dt")[0], tb(pA, "bias"), nga, walk)
    for rep in range(2):
        L.morb_inertial_timing(None, 1)
        stB = tb(pB, "state0").clone()
        opt.PoseInertialOptimizationLastFrame(tb(pB, "hasMP"), tb(pB, "obs"), tb(pB, "invSigma2"), tb(pB, "Xw"), tb(pB, "close"), pA["cam"], pA["Tbc12"], stA, preF, preK,
                                              oA[2], stB)
        torch.cuda.synchronize()
        out2 = (C.c_ulonglong * 16)()
        L.morb_inertial_timing(out2, 0)
    show("PoseInertialOptimizationLastFrame", out2)
