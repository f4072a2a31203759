#!/usr/bin/env python3
"""The reference's REGISTER_TIMES dumps (Tracking::TrackStats2File / LocalMapStats2File, Tracking.cc:178-255), written from this library's
calls: `TrackingTimeStats.txt` (one row per stereo frame: ORB extraction of both images, ComputeStereoMatches, PoseOptimization as the
"LM track" column — the stages of a frame this library replaces; the others are 0) and `LBA_Stats.txt` (one row per LocalBundleAdjustment:
time, #KF optimised, #KF fixed, #MP, #edges), same header lines and column order, so the reference's plotting scripts read them.
Developer tool, GPU only: one frame per call (the reference's call pattern), synthetic EuRoC-shaped input.
Usage: python tools/time_stats.py [outdir] [frames] [local maps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from morb_slam_amd import BAProblem, Optimizer, ORBextractor, ORBmatcher
from morb_slam_amd.synth import make_ba_problem, make_pose_problem, make_stereo_pair

out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 50
NLBA = int(sys.argv[3]) if len(sys.argv) > 3 else 10
os.makedirs(out_dir, exist_ok=True)
ext = ORBextractor(1200, 1.2, 8, 20, 7)
mt = ORBmatcher(0.7, True)
opt = Optimizer()


def ms(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


pairs = [make_stereo_pair(752, 480, seed=s) for s in range(4)]
frames = [torch.from_numpy(np.stack(p)).cuda() for p in pairs]
probs = [make_pose_problem(600, seed=s) for s in range(4)]
pt = [[torch.from_numpy(q[k][None]).cuda() for k in ("hasMP", "obs", "invSigma2", "Xw", "pose0")] for q in probs]
eo = None
rows = []
for i in range(NF + 3):
    f = frames[i % 4]
    t_ext, eo = ms(lambda: ext.extract_batch(f, out=eo))
    t_st, _ = ms(lambda: mt.ComputeStereoMatches(ext, eo[0], eo[1], eo[2], 458.654 * 0.11, 0.11))
    q = pt[i % 4]
    t_po, _ = ms(lambda: opt.PoseOptimization(q[0], q[1], q[2], q[3], q[4].clone(), probs[i % 4]["cam"]))
    if i >= 3:   # (the first calls allocate)
        rows.append((0.0, 0.0, t_ext, t_st, 0.0, 0.0, t_po, 0.0, t_ext + t_st + t_po))
with open(os.path.join(out_dir, "TrackingTimeStats.txt"), "w") as fh:
    fh.write("#Image Rect[ms], Image Resize[ms], ORB ext[ms], Stereo match[ms], IMU preint[ms], Pose pred[ms], LM track[ms], KF dec[ms], Total[ms]\n")
    for r in rows:
        fh.write(",".join(f"{v:.6f}" for v in r) + "\n")
lba = []
for s in range(NLBA + 1):
    b = make_ba_problem(seed=s)
    p = BAProblem(opt, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"])
    t, _ = ms(p.solve)
    nfix = int(np.asarray(b["kfFixed"]).sum())
    if s >= 1:
        lba.append((t, len(b["kfFixed"]) - nfix, nfix, len(b["mpPos"]), len(b["eKF"])))
with open(os.path.join(out_dir, "LBA_Stats.txt"), "w") as fh:
    fh.write("#LBA time[ms], KF opt[#], KF fixed[#], MP[#], Edges[#]\n")
    for r in lba:
        fh.write(f"{r[0]:.6f},{r[1]},{r[2]},{r[3]},{r[4]}\n")
a = np.array(rows)
print(f"{len(rows)} frames: ORB ext {a[:, 2].mean():.3f} ms, stereo match {a[:, 3].mean():.3f} ms, PoseOptimization {a[:, 6].mean():.3f} ms; "
      f"{len(lba)} local maps: LBA {np.mean([r[0] for r in lba]):.3f} ms")
