import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import synth
from morb_slam_amd.extractor import ORBextractor
from morb_slam_amd.matcher import ORBmatcher
B = 256
ims = [synth.make_stereo_pair(752, 480, seed=i) for i in range(4)]
batch = np.stack([ims[i % 4][k] for i in range(B) for k in (0, 1)])
ex = ORBextractor(1200, 1.2, 8, 20, 7)
dev = torch.from_numpy(batch).cuda()
out = ex.extract_batch(dev); torch.cuda.synchronize()
m = ORBmatcher(0.7, True)
kps, desc, cnt, _ = out
r = m.ComputeStereoMatches(ex, kps, desc, cnt, 458.654 * 0.11, 0.11); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): r = m.ComputeStereoMatches(ex, kps, desc, cnt, 458.654 * 0.11, 0.11)
torch.cuda.synchronize()
print("stereo %.1f us per %d frames, matches %.2f" % ((time.perf_counter() - t) / 20 * 1e6, B, float((r[0] >= 0).sum()) / B))
