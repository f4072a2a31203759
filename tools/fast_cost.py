"""Marginal cost of the phases of k_fast (developer tool, GPU only): runs the C2 extraction with MORB_FAST_STOP = 0..5 (the
kernel returns after that phase and reports no candidates) and prints the kernel's stage time for each.
Usage: python tools/fast_cost.py [B]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from morb_slam_amd import synth
    from morb_slam_amd.extractor import ORBextractor
    B = int(sys.argv[2])
    ims = [synth.make_stereo_pair(752, 480, seed=i) for i in range(4)]
    batch = np.stack([ims[i % 4][k] for i in range(B) for k in (0, 1)])
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    dev = torch.from_numpy(batch).cuda()
    ex.extract_batch(dev); torch.cuda.synchronize()
    ex.set_profiling(True)
    for _ in range(10): ex.extract_batch(dev)
    torch.cuda.synchronize()
    ms = ex.stage_ms()
    print("FASTMS", ms[2] if not isinstance(ms, dict) else ms["fast"])
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "64"
names = {-1: "whole kernel", 0: "load", 1: "+ reject", 2: "+ strength", 3: "+ nms", 4: "+ count", 5: "+ prefix"}
prev = 0.0
names[10] = 'whole, no global loads'
for stop in (0, 1, 2, 3, 4, 5, -1, 10):
    env = dict(os.environ, MORB_FAST_STOP=str(stop))
    out = subprocess.run([sys.executable, __file__, "--child", B], env=env, capture_output=True, text=True).stdout
    v = [float(l.split()[1]) for l in out.splitlines() if l.startswith("FASTMS")]
    if not v: print("stop", stop, "failed", out[-300:]); continue
    print(f"{names[stop]:14s} {v[0] * 1e3:8.1f} us   (+{(v[0] - prev) * 1e3:7.1f})")
    prev = v[0]
