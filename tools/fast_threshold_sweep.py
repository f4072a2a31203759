import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd import synth
from morb_slam_amd.extractor import ORBextractor
B = 256
ims = [synth.make_stereo_pair(752, 480, seed=i) for i in range(4)]
batch = np.stack([ims[i % 4][k] for i in range(B) for k in (0, 1)])
dev = torch.from_numpy(batch).cuda()
for th in ((20, 7), (40, 40), (80, 80), (160, 160), (255, 255)):
    ex = ORBextractor(1200, 1.2, 8, th[0], th[1])
    ex.extract_batch(dev); torch.cuda.synchronize()
    ex.set_profiling(True)
    for _ in range(10): ex.extract_batch(dev)
    torch.cuda.synchronize()
    ms = ex.stage_ms()
    print(th, f"fast {ms['fast']*1e3:.1f} us", flush=True)
    ex.close()
