import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import Optimizer
from morb_slam_amd.synth import make_pose_problem
F = 256
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 600
MONO = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3   # (pure-stereo / pure-mono frames: what a wave of one edge type would cost)
probs = [make_pose_problem(NF, seed=s % 8, mono_frac=MONO) for s in range(F)]
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
    st = out[2].cpu().numpy()
    print("exact order" if mode else "tree sums", f"{(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per 256 frames", f"(max trials {st[:, 1].max()}, mean {st[:, 1].mean():.1f})")
