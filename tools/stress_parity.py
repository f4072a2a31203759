#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (GPU box): extraction (keypoint records + descriptors, byte for byte) and ComputeStereoMatches
(float bit patterns) against the CPU oracle over many seeds, image sizes and feature counts — more shapes than the fixtures of
tests/.  A seeded subset runs in `-m gpu` (tests/test_stress_gpu.py); the full sweep: python tools/stress_parity.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

SHAPES = [(752, 480), (640, 480), (512, 512), (800, 600), (1000, 400), (377, 289), (1280, 720)]
ODD_SHAPES = [(83, 97), (333, 217), (601, 377), (1023, 511), (2048, 1536)]


def special_images(kind, W, H, n, rng):
    """Images that drive the rare paths: white noise (k_fastw's strip mode at low thresholds, the quadtree's global key path), flat (no corner
    anywhere: every cell falls through to minThFAST and stays empty), saturated (0 / 255 blocks: both polarities at the clamp)."""
    if kind == "noise":
        return rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    if kind == "flat":
        return np.full((n, H, W), 117, np.uint8)
    if kind == "saturated":
        img = np.zeros((n, H, W), np.uint8)
        blk = rng.integers(0, 2, (n, (H + 7) // 8, (W + 7) // 8), dtype=np.uint8) * 255
        return np.repeat(np.repeat(blk, 8, axis=1), 8, axis=2)[:, :H, :W].copy()
    raise ValueError(kind)


def levels_for(W, H):
    """8 levels unless the image is so small that the top level would fall below the extractor's 76-px minimum (MORB_ERR_UNSUPPORTED)."""
    n = 8
    while n > 1 and round(min(W, H) / 1.2 ** (n - 1)) < 76:
        n -= 1
    return n


def check_case(imgs, nf, ini, mn, log=None, tag="", scale=1.2, nlev=None):
    """One batch (2 f = left, 2 f + 1 = right) through the HIP extractor + ComputeStereoMatches and through the oracle; True when identical,
    None when the extractor refused the batch (morb_extractor_status: more than 65535 FAST candidates on a level)."""
    import torch
    import oracle_lib as O
    from morb_slam_amd import KP_DTYPE, ORBextractor, ORBmatcher
    mbf, mb = np.float32(458.654 * 0.11), np.float32(0.11)
    nfr = len(imgs) // 2
    nlev = nlev or levels_for(imgs.shape[2], imgs.shape[1])
    ext = ORBextractor(nf, scale, nlev, ini, mn)
    try:
        kps, desc, cnt, mono = ext.extract_batch(torch.from_numpy(imgs).cuda())
        u, d = ORBmatcher().ComputeStereoMatches(ext, kps, desc, cnt, mbf, mb)
        torch.cuda.synchronize()
        ext.check_status()
    except RuntimeError as e:   # (MorbError is a RuntimeError: a configuration the library refuses — e.g. a per-level quota beyond the LDS-resident quadtree)   # the documented limit (a pyramid level with more than 65535 FAST candidates): refused loudly, not a parity case
        if log:
            log(f"{tag}: {imgs.shape[2]}x{imgs.shape[1]} nfeat {nf} th {ini}/{mn} frames {nfr}: refused ({str(e)[:60]}...)")
        ext.close()
        return None
    c = cnt.cpu().numpy(); kn = kps.cpu().numpy(); dn = desc.cpu().numpy(); un = u.cpu().numpy(); dd = d.cpu().numpy()
    ora, ok = [], True
    for i, im in enumerate(imgs):
        o = O.OracleExtractor(nf, scale, nlev, ini, mn)
        _, k, de = o(im)
        ora.append((o, k, de))
        if c[i] != len(k) or kn[i, :c[i]].reshape(-1).view(KP_DTYPE).tobytes() != k.tobytes() or dn[i, :c[i]].tobytes() != de.tobytes():
            ok = False
    for f in range(nfr):
        (ol, kl, dl), (orr, kr, dr) = ora[2 * f], ora[2 * f + 1]
        ue, dep = O.stereo_matches(ol, orr, kl, dl, kr, dr, mbf, mb)
        n = len(kl)
        if un[f, :n].tobytes() != ue.tobytes() or dd[f, :n].tobytes() != dep.tobytes():
            ok = False
    if log:
        log(f"{tag}: {imgs.shape[2]}x{imgs.shape[1]} nfeat {nf} levels {nlev} scale {scale} th {ini}/{mn} frames {nfr} keypoints {int(c.sum())}: {'ok' if ok else 'MISMATCH'}")
    ext.close()
    return ok


def run(cases, seed=7, log=print, specials=True, odd=ODD_SHAPES, max_pixels=None):
    """`cases` random cases + (specials) noise / flat / saturated batches + the odd shapes.  Returns (checked, mismatches)."""
    from morb_slam_amd.synth import make_stereo_pair
    rng = np.random.default_rng(seed)
    bad = n = 0
    for case in range(cases):
        W, H = SHAPES[case % len(SHAPES)]
        nf = int(rng.choice([500, 1000, 1200, 1500, 2000]))
        nfr = int(rng.choice([1, 2, 3, 5, 9]))     # (9 frames = 18 images: beyond 16 the quadtree keeps one wave per level; up to 16 the big levels get teams)
        ini, mn = [(20, 7), (12, 5), (30, 10), (7, 20)][int(rng.integers(0, 4))]
        if max_pixels and W * H * nfr > max_pixels:
            nfr = max(1, max_pixels // (W * H))
        pairs = [make_stereo_pair(W, H, seed=1000 * seed + 10 * case + (k % 3)) for k in range(nfr)]
        imgs = np.stack([im for p in pairs for im in p])
        if case % 5 == 4:                           # every fifth case: heavy pixel noise on top (many more FAST survivors and corners per cell)
            imgs = np.clip(imgs.astype(np.int16) + rng.integers(-25, 26, imgs.shape), 0, 255).astype(np.uint8)
        r = check_case(imgs, nf, ini, mn, log, f"case {case}")
        bad += 1 if r is False else 0
        n += 0 if r is None else 1
    if specials:
        for kind, (W, H), nf, th in (("noise", (320, 240), 800, (20, 7)), ("noise", (377, 289), 1500, (3, 1)), ("flat", (640, 480), 1000, (20, 7)),
                                     ("saturated", (512, 384), 1200, (20, 7)), ("saturated", (333, 217), 500, (60, 3))):
            r = check_case(special_images(kind, W, H, 2, rng), nf, th[0], th[1], log, kind)
            bad += 1 if r is False else 0
            n += 0 if r is None else 1
    for (W, H) in odd:
        l, r = make_stereo_pair(W, H, seed=77 + W)
        rr = check_case(np.stack([l, r]), 700, 20, 7, log, "odd shape")
        bad += 1 if rr is False else 0
        n += 0 if rr is None else 1
    return n, bad


def run_pyramids(cases, seed=9, log=print):
    """other pyramids: scale factors 1.1 - 1.5, 3 - 12 levels, 100 - 5000 features (the reference's yaml files use 1.2 / 8 / 1000 - 2000 throughout)"""
    from morb_slam_amd.synth import make_stereo_pair
    rng = np.random.default_rng(seed)
    bad = n = 0
    for case in range(cases):
        W, H = SHAPES[int(rng.integers(0, len(SHAPES)))]
        scale = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.44, 1.5]))
        nlev = int(rng.choice([3, 5, 6, 8, 10, 12]))
        while nlev > 1 and round(min(W, H) / scale ** (nlev - 1)) < 76:
            nlev -= 1
        nf = int(rng.choice([100, 300, 1000, 3000, 5000]))
        l, r = make_stereo_pair(W, H, seed=5000 + 10 * seed + case)
        res = check_case(np.stack([l, r]), nf, 20, 7, log, f"pyramid {case}", scale=scale, nlev=nlev)
        bad += 1 if res is False else 0
        n += 0 if res is None else 1
    return n, bad


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "pyramids":
        n, bad = run_pyramids(int(sys.argv[2]), log=lambda s: print(s, flush=True))
        print(f"{n - bad} / {n} pyramid cases identical")
        sys.exit(1 if bad else 0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n, bad = run(N, log=lambda s: print(s, flush=True))
    print(f"{n - bad} / {n} cases identical")
    sys.exit(1 if bad else 0)
