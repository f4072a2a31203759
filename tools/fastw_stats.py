"""Dynamic phase counts of k_fastw (developer tool, GPU only): builds a -DMORB_FAST_TIMING variant of the HIP library, runs the C2
extraction once and prints, per segment wave, how many reject / strength / NMS rounds, emit-loop trips, survivors, corners and
keypoints it went through.  Usage: python tools/fastw_stats.py [B] [extra -D flags ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "morb_slam_amd", "libmorb_hip_fwstats.so")
os.environ["MORB_HIP_LIB"] = out     # (read by morb_slam_amd.capi at import)
from morb_slam_amd import build as b
import subprocess
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), "fwstats", "extractor.hip", "-DMORB_FAST_TIMING"] + sys.argv[2:], stdout=subprocess.DEVNULL)
import numpy as np, torch
from morb_slam_amd import capi, synth
from morb_slam_amd.extractor import ORBextractor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H, NF = int(os.environ.get('MORB_W', 752)), int(os.environ.get('MORB_H', 480)), int(os.environ.get('MORB_NF', 1200))
ims = [synth.make_stereo_pair(W, H, seed=i) for i in range(4)]
batch = np.stack([ims[i % 4][k] for i in range(B) for k in (0, 1)])
TH = (int(os.environ.get('MORB_INI', 20)), int(os.environ.get('MORB_MIN', 7)))
if os.environ.get('MORB_NOISE'):      # white noise instead of the benchmark images (with MORB_INI=1 MORB_MIN=1: every cell in strip mode)
    batch = np.random.default_rng(12).integers(0, 256, batch.shape, dtype=np.uint8)
ex = ORBextractor(NF, 1.2, 8, TH[0], TH[1])
dev = torch.from_numpy(batch).cuda()
lib = capi.lib()
lib.morb_fw_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
ex.extract_batch(dev); torch.cuda.synchronize()
lib.morb_fw_stats(None, 1)
ex.extract_batch(dev); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.morb_fw_stats(buf, 0)
names = ["waves", "jobs", "reject rounds", "emit loop trips", "survivors", "strength rounds", "corners", "nms rounds", "keypoints",
         "output rank trips", "-", "minThFAST passes", "partial queue takes", "strip-mode passes"]
w = max(buf[0], 1)
for n, v in zip(names, buf): print(f"{n:22s} {v:12d}   {v / w:8.3f} per wave")
