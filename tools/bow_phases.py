"""Per-phase wall-clock split of k_bow_match (developer tool, GPU only); see tools/fast_phases.py."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "morb_slam_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "libmorb_hip_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                       "-DMORB_FAST_TIMING", "-I", os.path.join(ROOT, "include"), "-o", out] + srcs)
os.environ["MORB_HIP_LIB"] = out
import numpy as np, torch
import bench
from morb_slam_amd import capi, ORBextractor, ORBmatcher
from morb_slam_amd.synth import make_vocabulary
B = 64
frames = torch.from_numpy(bench.make_batch(list(range(B)), B, seed=0)).cuda()
images = frames.view(2 * B, bench.H, bench.W)
ext = ORBextractor(1200, 1.2, 8, 20, 7); m = ORBmatcher(0.7, True)
kps, desc, cnt, _ = ext.extract_batch(images)
vd, vf = make_vocabulary(10, 6, seed=0)
w, n = m.bow_transform(desc, cnt, torch.from_numpy(vd).cuda(), torch.from_numpy(vf).cuda(), 10, 6, 4)
kf = torch.tensor([2 * max(f - 1, 0) for f in range(B)], dtype=torch.int32).cuda()
fi = torch.tensor([2 * f for f in range(B)], dtype=torch.int32).cuda()
has = torch.from_numpy((np.random.default_rng(7).random((2 * B, ext.max_keypoints)) < 0.8).astype(np.uint8)).cuda()
lib = capi.lib()
lib.morb_bow_timing.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
m.SearchByBoW(kf, fi, kps, desc, n, cnt, has); torch.cuda.synchronize()
m.SearchByBoW(kf, fi, kps, desc, n, cnt, has); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (16384 * 4))(); lib.morb_bow_timing(buf, 0)
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)[:25 * B * 4]
t0 = t[:, 0].min()
print("kernel span %.1f us" % ((t[:, 3].max() - t0) / 100.0))
print("wave start  (us): mean %.1f max %.1f" % ((t[:, 0] - t0).mean() / 100.0, (t[:, 0] - t0).max() / 100.0))
for k, nm in ((1, "stage"), (2, "compact"), (3, "nodes")):
    d = (t[:, k] - t[:, k - 1]) / 100.0
    print("%-8s (us): mean %.1f  p99 %.1f  max %.1f" % (nm, d.mean(), np.percentile(d, 99), d.max()))
