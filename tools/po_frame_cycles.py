import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from morb_slam_amd.tracking import build_chains
B = 256
chain, ks, host = build_chains(bench.make_batch(range(256), 256, seed=0), B=B, npairs=2, seq_len=64)
chain.step(); chain.sync(); chain.step(); chain.sync()
for name, po in (("PO1", chain.po1), ("PO2", chain.po2)):
    s = po[2].cpu().numpy()[:B]
    kc = s[:, 0]; tr = s[:, 1] // 100; ce = s[:, 1] % 100
    o = np.argsort(-kc)[:6]
    print(name, "kilocycles: max %d mean %.0f;  slowest frames (kcycles, trials, certain rejections):" % (kc.max(), kc.mean()), [(int(kc[i]), int(tr[i]), int(ce[i])) for i in o])
    print("   fit: kcycles ~ a + b*(trials - certain) + c*certain:", np.linalg.lstsq(np.stack([np.ones(B), tr - ce, ce], 1), kc, rcond=None)[0].round(2))
