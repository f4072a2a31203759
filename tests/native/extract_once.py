"""One batched extraction in a process of its own (tests/test_extractor_gpu.py::test_input_ending_on_a_page_boundary): an out-of-bounds read
of the caller's image is a GPU memory fault that kills the process, which the test observes as a non-zero exit code.
Usage: python extract_once.py W H nfeatures B use_lapping_area"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import ORBextractor
from morb_slam_amd.synth import make_stereo_pair
W, H, N, B, uselap = [int(x) for x in sys.argv[1:6]]
base = [make_stereo_pair(W, H, seed=100 + i) for i in range(4)]
imgs = torch.from_numpy(np.stack([base[i % 4][k] for i in range(B) for k in (0, 1)])).cuda()
ext = ORBextractor(N, 1.2, 8, 20, 7)
lap = np.tile(np.array([[0, W - 1], [0, W - 1]], np.int32), (B, 1)) if uselap else None
eo = ext.extract_batch(imgs, lap=lap); torch.cuda.synchronize()
print("OK", sys.argv[1:], int(eo[2].sum()), flush=True)
