import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from morb_slam_amd.tracking import build_chains
from morb_slam_amd.capi import lib
chain, ks, host = build_chains(bench.make_batch(range(256), 256, seed=0), B=1, npairs=2, seq_len=64)
L = lib()
z = (C.c_ulonglong * 8)()
chain.step(); chain.sync(); L.morb_po2_cycles(z)
chain.step(); chain.sync(); L.morb_po2_cycles(z)
names = ["kernel", "solve+bcast", "pass", "pass.compute0", "npass", "compact+load", "classify", "chain"]
print("PO1 + PO2 of frame 0:", " ".join(f"{a}={b}" for a, b in zip(names, list(z))))
print("stats", chain.po1[2].cpu().numpy()[0], chain.po2[2].cpu().numpy()[0])
