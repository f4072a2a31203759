"""Per-phase wall-clock split of k_distribute's level-0 wave (developer tool, GPU only; k_fastw's dynamic phase counts:
tools/fastw_stats.py).

Builds a -DMORB_FAST_TIMING variant of the HIP library next to the product one, runs the C2 extract batch through it and
prints the 10-ns ticks each phase of the quadtree takes (mean over images).
Usage: python tools/fast_phases.py [B]
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "morb_slam_amd", "csrc")
out = os.path.join(ROOT, "morb_slam_amd", "libmorb_hip_timing" + os.environ.get("MORB_TIMING_TAG", "") + ".so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                       "-DMORB_FAST_TIMING"] + os.environ.get("MORB_EXTRA_DEFS", "").split() + ["-I", os.path.join(ROOT, "include"), "-o", out] + srcs)
os.environ["MORB_HIP_LIB"] = out
import torch
from morb_slam_amd import capi, synth
from morb_slam_amd.extractor import ORBextractor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H, NF = int(os.environ.get('MORB_W', 752)), int(os.environ.get('MORB_H', 480)), int(os.environ.get('MORB_NF', 1200))
ims = [synth.make_stereo_pair(W, H, seed=i) for i in range(4)]
import numpy as np
batch = np.stack([ims[i % 4][k] for i in range(B) for k in (0, 1)])
ex = ORBextractor(NF, 1.2, 8, 20, 7)
dev = torch.from_numpy(batch).cuda()
lib = capi.lib()
lib.morb_fast_timing.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
ex.extract_batch(dev); torch.cuda.synchronize()
lib.morb_fast_timing(None, 1)
for _ in range(5): ex.extract_batch(dev)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib.morb_fast_timing(buf, 0)
print("k_distribute level 0 (ticks of 10 ns per wave, mean over images):")
nw = 5 * 2 * B
for n, v in zip(["cell prefix", "gather", "quadtree", "candidates T", "selected n", "qt compact", "qt splits", "qt std::sort", "split: child counts", "split: ranks", "split: partition + children", "split: bookkeeping", "ff: histogram", "ff: levels + decision", "ff: radix sort", "ff: prefix", "ff: nodes", "best key per node", "radix: count", "radix: scan", "radix: scatter"], list(buf[8:16]) + list(buf[16:29])): print(f"  {n:28s} {v / nw:10.1f}")
