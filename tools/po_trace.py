"""Developer tool: the LM trials of ONE PoseOptimization problem of tools/stress_optimizers.py's sweep on both sides — the oracle (orc_set_trace) and the device
(a library built with -DMORB_PO_TRACE: python tools/ab_build.py potrace optimizer.hip -DMORB_PO_TRACE; MORB_HIP_LIB=.../libmorb_hip_potrace.so) — and the first
trial whose (currentChi, tempChi, lambda, rho, scale) differ in a bit, with the first iteration's H and b.  python tools/po_trace.py <case>"""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
from morb_slam_amd import Optimizer
from morb_slam_amd.capi import lib
from morb_slam_amd.synth import make_pose_problem
seed = 11
rng = np.random.default_rng(seed)
case = int(sys.argv[1])
for s in range(case + 1):
    n = int(rng.choice([60, 150, 300, 600, 1200])); of = float(rng.choice([0.05, 0.15, 0.3])); mf = float(rng.choice([0.0, 0.15, 0.5, 1.0]))
p = make_pose_problem(n, seed=500 + 97 * seed + case, outlier_frac=of, mono_frac=mf)
L = O.lib()
buf = np.zeros(6 * 520); L.orc_set_trace.argtypes = [C.c_void_p, C.c_int]; L.orc_set_trace(buf.ctypes.data, 512)
r, pe, oe, se = O.pose_optimization(p)
no = L.orc_trace_count(); L.orc_set_trace(None, 0)
opt = Optimizer(); opt.set_exact_order(True)
t = [torch.from_numpy(p[k][None]).cuda() for k in ("hasMP", "obs", "invSigma2", "Xw")]
pose = torch.from_numpy(p["pose0"][None]).cuda()
out = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose, p["cam"])
torch.cuda.synchronize()
g = np.zeros(6 * 520); ng = C.c_int(0)
lib().morb_po_trace.argtypes = [C.c_void_p, C.c_void_p]; lib().morb_po_trace(g.ctypes.data, C.byref(ng))
print("case", case, n, of, mf, "oracle trials", no, se, "gpu trials", ng.value, out[2].cpu().numpy())
for i in range(max(no, ng.value)):
    a = buf[6 * i:6 * i + 6] if i < no else None; b = g[6 * i:6 * i + 6] if i < ng.value else None
    same = a is not None and b is not None and a.tobytes() == b.tobytes()
    if not same:
        print(i, "ORACLE", None if a is None else [x.hex() for x in a[:5]], "\n   GPU   ", None if b is None else [x.hex() for x in b[:5]])
        print("   values oracle", a, "\n   values gpu   ", b)
        break
else:
    print("identical traces")
print("x oracle", [v.hex() for v in buf[3000:3006]]); print("x gpu   ", [v.hex() for v in g[3000:3006]])
print("T oracle", [v.hex() for v in buf[3006:3013]]); print("T gpu   ", [v.hex() for v in g[3006:3013]])

Ho, Hg = buf[3024:3066], g[3024:3066]
for k in range(42):
    if Ho[k].tobytes() != Hg[k].tobytes(): print("H/b entry", k, Ho[k].hex(), Hg[k].hex())
