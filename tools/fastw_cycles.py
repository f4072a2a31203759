"""Per-phase cycle table of k_fastw (GPU box): builds a -DMORB_FAST_CYCLES variant of the HIP library (s_memtime at the phase boundaries,
accumulated in registers, flushed when the wave is done), runs the C2 extraction and prints where a wave's residency goes.
Usage: python tools/fastw_cycles.py [B] [extra -D flags ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "morb_slam_amd", "libmorb_hip_cycles.so")
os.environ["MORB_HIP_LIB"] = out     # (read by morb_slam_amd.capi at import)
from morb_slam_amd import build as b
import subprocess
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), "cycles", "extractor.hip", "-DMORB_FAST_CYCLES"] + sys.argv[2:], stdout=subprocess.DEVNULL)
import numpy as np, torch
from morb_slam_amd import capi, synth
from morb_slam_amd.extractor import ORBextractor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ims = [synth.make_stereo_pair(752, 480, seed=i) for i in range(4)]
batch = np.stack([ims[i % 4][k] for i in range(B) for k in (0, 1)])
ex = ORBextractor(1200, 1.2, 8, 20, 7)
dev = torch.from_numpy(batch).cuda()
lib = capi.lib()
lib.morb_fw_cycles_buffer.argtypes = [ctypes.c_void_p]
ex.set_profiling(True)
for _ in range(3):
    ex.extract_batch(dev)
torch.cuda.synchronize()
ex.stage_ms()
ncell = 4096                                   # >= FAST cells per image over all levels (752 x 480: 700)
buf = torch.zeros((2 * B * ncell, 8), dtype=torch.int64, device="cuda")
lib.morb_fw_cycles_buffer(buf.data_ptr())
ex.extract_batch(dev); torch.cuda.synchronize()
lib.morb_fw_cycles_buffer(None)
ms = ex.stage_ms()["fast"]
h = buf.cpu().numpy()
h = h[h[:, 7] > 0]
cyc = h.sum(0)
life = max(int(cyc[7]), 1)
print(f"# k_fastw, {2 * B} images of 752 x 480 per launch pair: {ms * 1e3:.0f} us in this build (s_memtime + s_waitcnt lgkmcnt(0) at the phase boundaries), {len(h)} waves")
print("# per-phase shader cycles of a wave's residency, summed over all waves (other waves of the SIMD run in between: a phase's share is its")
print("# share of the residency, which is what the kernel's duration follows at a fixed occupancy)")
print(f"# sum of wave lifetimes / (kernel time x 2.4 GHz) = {life / (ms * 1e-3 * 2.4e9):.0f} waves resident on average (of 8192 slots); median wave lifetime {int(np.median(h[:, 7]))} cycles")
for n, v in zip(["load window (+ wait)", "reject", "emit (compaction)", "strength", "nms (zero + scatter + 3x3)", "output + exit", "strip mode", "lifetime"], cyc):
    print(f"{n:28s} {int(v) / len(h):12.0f} cycles per wave   {100.0 * int(v) / life:5.1f} %")
