#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (developer tool, GPU box): extraction (keypoint records + descriptors, byte for byte) and
ComputeStereoMatches (float bit patterns) against the CPU oracle over many seeds, image sizes and feature counts — more shapes than the
test suite's fixtures.  Usage: python tools/stress_parity.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
from morb_slam_amd import KP_DTYPE, ORBextractor, ORBmatcher
from morb_slam_amd.synth import make_stereo_pair
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(7)
SHAPES = [(752, 480), (640, 480), (512, 512), (800, 600), (1000, 400), (377, 289), (1280, 720)]
bad = 0
for case in range(N):
    W, H = SHAPES[case % len(SHAPES)]
    nf = int(rng.choice([500, 1000, 1200, 1500, 2000]))
    nfr = int(rng.choice([1, 2, 3, 5, 9]))     # (9 frames = 18 images: beyond 16 the quadtree keeps one wave per level; up to 16 the big levels get teams)
    ini, mn = [(20, 7), (12, 5), (30, 10), (7, 20)][int(rng.integers(0, 4))]
    pairs = [make_stereo_pair(W, H, seed=1000 + 10 * case + (k % 3)) for k in range(nfr)]
    imgs = np.stack([im for p in pairs for im in p])
    if case % 5 == 4:                           # every fifth case: heavy pixel noise on top (many more FAST survivors and corners per cell)
        imgs = np.clip(imgs.astype(np.int16) + rng.integers(-25, 26, imgs.shape), 0, 255).astype(np.uint8)
    ext = ORBextractor(nf, 1.2, 8, ini, mn)
    kps, desc, cnt, mono = ext.extract_batch(torch.from_numpy(imgs).cuda())
    u, d = ORBmatcher().ComputeStereoMatches(ext, kps, desc, cnt, np.float32(458.654 * 0.11), np.float32(0.11))
    torch.cuda.synchronize()
    c = cnt.cpu().numpy(); kn = kps.cpu().numpy(); dn = desc.cpu().numpy(); un = u.cpu().numpy(); dd = d.cpu().numpy()
    ora = []
    ok = True
    for i, im in enumerate(imgs):
        o = O.OracleExtractor(nf, 1.2, 8, ini, mn)
        _, k, de = o(im)
        ora.append((o, k, de))
        if c[i] != len(k) or kn[i, :c[i]].reshape(-1).view(KP_DTYPE).tobytes() != k.tobytes() or dn[i, :c[i]].tobytes() != de.tobytes():
            ok = False
    for f in range(nfr):
        (ol, kl, dl), (orr, kr, dr) = ora[2 * f], ora[2 * f + 1]
        ue, dep = O.stereo_matches(ol, orr, kl, dl, kr, dr, np.float32(458.654 * 0.11), np.float32(0.11))
        n = len(kl)
        if un[f, :n].view(np.uint32).tolist() != ue.view(np.uint32).tolist() or dd[f, :n].view(np.uint32).tolist() != dep.view(np.uint32).tolist():
            ok = False
    print(f"case {case}: {W}x{H} nfeat {nf} frames {nfr} keypoints {int(c.sum())}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    ext.close()
print(f"{N - bad} / {N} cases identical")
sys.exit(1 if bad else 0)
