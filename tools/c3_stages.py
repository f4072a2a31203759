"""Stage times of the C3 chain (512x512 fisheye stereo, 1500 features): extraction, ComputeStereoFishEyeMatches, PoseOptimization on the
rig — each alone, B frames per call (developer tool, GPU only).  Usage: python tools/c3_stages.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from morb_slam_amd import ORBextractor, ORBmatcher, Optimizer
from morb_slam_amd.synth import TUMVI_CAM_L, TUMVI_CAM_R, TUMVI_T_C1_C2, make_pose_problem_fisheye, make_stereo_pair
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
base = [make_stereo_pair(512, 512, seed=100 + i) for i in range(4)]
imgs = torch.from_numpy(np.stack([base[i % 4][k] for i in range(B) for k in (0, 1)])).to(dev)
ext = ORBextractor(1500, 1.2, 8, 20, 7)
mt = ORBmatcher(0.7, True)
lap = np.tile(np.array([[0, 511], [0, 511]], np.int32), (B, 1))
sigma2 = ext.GetScaleSigmaSquares()
Rlr = TUMVI_T_C1_C2[:3, :3].astype(np.float32); tlr = TUMVI_T_C1_C2[:3, 3].astype(np.float32)
probs = [make_pose_problem_fisheye(seed=s % 4) for s in range(B)]
capP = max(len(q["hasMP"]) for q in probs)
t = {k: torch.from_numpy(np.stack([np.pad(q[k], [(0, capP - len(q[k]))] + [(0, 0)] * (q[k].ndim - 1)) for q in probs])).to(dev)
     for k in ("hasMP", "obs", "invSigma2", "Xw")}
pose0 = torch.from_numpy(np.stack([q["pose0"] for q in probs])).to(dev)
nl = torch.tensor([q["Nleft"] for q in probs], dtype=torch.int32, device=dev)
cn = torch.tensor([len(q["hasMP"]) for q in probs], dtype=torch.int32, device=dev)
opt = Optimizer()
eo = ext.extract_batch(imgs, lap=lap)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a = timed(lambda: ext.extract_batch(imgs, lap=lap, out=eo))
b = timed(lambda: mt.ComputeStereoFishEyeMatches(eo[0], eo[1], eo[2], eo[3], TUMVI_CAM_L, TUMVI_CAM_R, Rlr, tlr, sigma2))
c = timed(lambda: opt.PoseOptimizationFisheye(t["hasMP"], t["obs"], t["invSigma2"], t["Xw"], pose0.clone(), nl, cn, TUMVI_CAM_L, TUMVI_CAM_R, probs[0]["Trl"]))
print(f"B = {B}: extract {a:.3f} ms, fisheye stereo {b:.3f} ms, pose optimisation {c:.3f} ms, sum {a + b + c:.3f} ms")
