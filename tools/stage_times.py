"""Per-stage time of the C2 extraction, isolated (one extractor alone on the chip; developer tool, GPU only).
Usage: python tools/stage_times.py [B] [calls]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from morb_slam_amd import synth
from morb_slam_amd.extractor import ORBextractor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ims = [synth.make_stereo_pair(752, 480, seed=i) for i in range(4)]
batch = np.stack([ims[i % 4][k] for i in range(B) for k in (0, 1)])
ex = ORBextractor(1200, 1.2, 8, 20, 7)
dev = torch.from_numpy(batch).cuda()
ex.extract_batch(dev); torch.cuda.synchronize()
ex.set_profiling(True)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20): ex.extract_batch(dev)
torch.cuda.synchronize()
ms = ex.stage_ms()
print(" ".join(f"{k} {v * 1e3:.1f}us" for k, v in ms.items()), f"(per {2 * B} images)")
