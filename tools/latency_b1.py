"""Latency of ONE stereo frame through the hot path (developer tool): device-resident batch of 2 images, and the
host-pointer form morb_extract (includes the PCIe copies)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from morb_slam_amd import ORBextractor, ORBmatcher
from morb_slam_amd.synth import make_stereo_pair
W, H, NF = int(os.environ.get("MORB_W", 752)), int(os.environ.get("MORB_H", 480)), int(os.environ.get("MORB_NF", 1200))
l, r = make_stereo_pair(W, H, seed=1)
imgs = torch.from_numpy(np.stack([l, r])).cuda()
ext = ORBextractor(NF, 1.2, 8, 20, 7); m = ORBmatcher(0.7, True)
st = torch.cuda.Stream()
out = None
def one():
    global out
    out = ext.extract_batch(imgs, out=out, stream=st.cuda_stream)
    m.ComputeStereoMatches(ext, out[0], out[1], out[2], 50.4, 0.11, stream=st.cuda_stream)
for _ in range(5): one()
st.synchronize()
ext.set_profiling(True)
t0 = time.perf_counter()
N = 50
for _ in range(N):
    one(); st.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"{W}x{H}/{NF} device-resident stereo frame: extract x2 + stereo match = {dt * 1e3:.3f} ms  stages {ext.stage_ms()}")
ext.set_profiling(False)
t0 = time.perf_counter()
for _ in range(20):
    ext(l)
dth = (time.perf_counter() - t0) / 20
print(f"host-pointer morb_extract (one {W}x{H} image, copies included): {dth * 1e3:.3f} ms")
