"""One-shot LocalBundleAdjustment (morb_local_bundle_adjustment: create + solve + results + destroy per call), C5-sized graph (developer tool, GPU only)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from morb_slam_amd import synth
from morb_slam_amd.optimizer import Optimizer, local_bundle_adjustment_oneshot
b = synth.make_ba_problem(20, 6, 3000, seed=1)
opt = Optimizer()
f = local_bundle_adjustment_oneshot
args = (opt, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"])
r = f(*args)
t = time.perf_counter()
for _ in range(10): r = f(*args)
print("one-shot LocalBundleAdjustment: %.2f ms per call" % ((time.perf_counter() - t) / 10 * 1e3), r[-1] if isinstance(r, tuple) else "")
