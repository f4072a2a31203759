"""What the matchers cost the pipelined step (developer experiment): the bench's step with the stereo matcher, the BoW chain or both left out.
python tools/ablate_matchers.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from morb_slam_amd.frontend import StereoFrontEnd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = torch.from_numpy(bench.make_batch(list(range(B)), B, seed=0)).cuda()
images = frames.view(2 * B, bench.H, bench.W)
fe = StereoFrontEnd(images, 1200, B)
for ab in ("", "stereo", "bow", "all"):
    fe.ablate = ab
    for _ in range(4): fe.step()
    fe.sync()
    t0 = time.perf_counter()
    for _ in range(30): fe.step()
    fe.sync()
    dt = (time.perf_counter() - t0) / 30
    print(f"skipped: {ab or 'nothing':8s} {dt * 1e3:.3f} ms per step  {B / dt / 1e3:.1f} k frames/s")
