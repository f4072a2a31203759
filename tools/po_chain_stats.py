"""LM iterations / trials of the two PoseOptimizations of the tracking chain (and the active edge counts): python tools/po_chain_stats.py [B]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from morb_slam_amd.tracking import build_chains
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chain, ks, host = build_chains(bench.make_batch(range(256), 256, seed=0), B=B, npairs=2, seq_len=64)
chain.step(); chain.sync()
s1 = chain.po1[2].cpu().numpy(); s2 = chain.po2[2].cpu().numpy()
has1 = None
print("PO1 iterations/trials per frame:", s1[:B].tolist())
print("PO2 iterations/trials per frame:", s2[:B].tolist())
print("PO1 inliers", chain.po1[0].cpu().numpy()[:B].tolist(), "PO2 inliers", chain.po2[0].cpu().numpy()[:B].tolist())
