"""k_bow_transform alone: ms per 512 stereo frames (1024 images, the right ones with count 0 as in the front end), staged (LDS) and
global-memory descent (MORB_BOW_GLOBAL=1).  python tools/bow_transform_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from morb_slam_amd import ORBextractor, ORBmatcher
from morb_slam_amd.synth import make_vocabulary
B = 512
frames = torch.from_numpy(bench.make_batch(list(range(B)), B, seed=0)).cuda()
images = frames.view(2 * B, bench.H, bench.W)
ext = ORBextractor(1200, 1.2, 8, 20, 7); m = ORBmatcher(0.7, True)
kps, desc, cnt, _ = ext.extract_batch(images)
left = torch.zeros((2 * B,), dtype=torch.int32, device="cuda"); left[0::2] = 1
cl = cnt * left
vd, vf = make_vocabulary(10, 6, seed=0)
vd, vf = torch.from_numpy(vd).cuda(), torch.from_numpy(vf).cuda()
out = None
st = torch.cuda.Stream()
for _ in range(3): out = m.bow_transform(desc, cl, vd, vf, 10, 6, 4, out=out, stream=st.cuda_stream)
st.synchronize(); t0 = time.perf_counter()
for _ in range(20): out = m.bow_transform(desc, cl, vd, vf, 10, 6, 4, out=out, stream=st.cuda_stream)
st.synchronize()
print("%s: %.3f ms per 512 frames" % ("global" if os.environ.get("MORB_BOW_GLOBAL") else "staged", (time.perf_counter() - t0) / 20 * 1e3))
