"""What the matchers cost the pipelined step (developer tool, GPU only): the bench's StereoFrontEnd timed (a) as benched, (b) with the matchers
left out (extraction only), (c) the matchers of one step alone on the chip.  Usage: python tools/step_split.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from morb_slam_amd import synth
from morb_slam_amd.frontend import StereoFrontEnd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
import bench
host = bench.make_batch(range(B), B, seed=0)   # the bench's synthetic stream
images = torch.from_numpy(np.ascontiguousarray(host).reshape(2 * B, 480, 752)).cuda()
fe = StereoFrontEnd(images, 1200, B)
def run(k):
    for _ in range(3): fe.step()
    fe.sync()
    t0 = time.perf_counter()
    for _ in range(k): fe.step()
    fe.sync()
    return (time.perf_counter() - t0) / k * 1e3
full = run(K)
orig = fe.run_matchers
fe.run_matchers = lambda S, gate=None: None
ext = run(K)
fe.run_matchers = orig
S = fe.step(); fe.sync()
for _ in range(3): fe.run_matchers(S)
fe.sync()
t0 = time.perf_counter()
for _ in range(K): fe.run_matchers(S)
fe.sync()
mat = (time.perf_counter() - t0) / K * 1e3
print(f"B {B}: step {full:.3f} ms, extraction only {ext:.3f} ms, matchers alone {mat:.3f} ms -> the matchers cost the step {full - ext:.3f} ms")
