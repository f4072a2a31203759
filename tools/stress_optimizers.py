#!/usr/bin/env python3
"""Randomised parity sweep of PoseOptimization and LocalBundleAdjustment against the CPU oracle (GPU box): more seeds and problem shapes
than the fixtures of tests/; the tests' criteria (pose / points within 1e-4, outlier and erase flags identical, the same LM iterations and trials —
PoseOptimization in its deterministic mode; its tree-sum mode is held to the results).
A seeded subset runs in `-m gpu` (tests/test_stress_gpu.py); the full sweep: python tools/stress_optimizers.py [pose cases] [ba cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def run_pose(NP, seed=11, log=print):
    import torch
    import oracle_lib as O
    from morb_slam_amd import Optimizer
    from morb_slam_amd.synth import make_pose_problem
    rng = np.random.default_rng(seed)
    opt = Optimizer()
    bad = 0
    probs = []
    for s in range(NP):
        n = int(rng.choice([60, 150, 300, 600, 1200]))
        probs.append(make_pose_problem(n, seed=500 + 97 * seed + s, outlier_frac=float(rng.choice([0.05, 0.15, 0.3])),
                                       mono_frac=float(rng.choice([0.0, 0.15, 0.5, 1.0]))))
    cap = max(len(p["hasMP"]) for p in probs)
    pad = lambda a: np.pad(a, [(0, cap - len(a))] + [(0, 0)] * (a.ndim - 1))
    t = [torch.from_numpy(np.stack([pad(p[k]) for p in probs])).cuda() for k in ("hasMP", "obs", "invSigma2", "Xw")]
    pose = torch.from_numpy(np.stack([p["pose0"] for p in probs])).cuda()
    cnt = torch.tensor([len(p["hasMP"]) for p in probs], dtype=torch.int32, device="cuda")
    ora = [O.pose_optimization(p) for p in probs]
    # the deterministic mode (edge-order sums) is held to the LM path too: iterations AND trials; the default tree-sum mode to the results
    for exact in (True, False):
        opt.set_exact_order(exact)
        ps = pose.clone()
        out = opt.PoseOptimization(t[0], t[1], t[2], t[3], ps, probs[0]["cam"], count=cnt)
        torch.cuda.synchronize()
        poseg = ps.cpu().numpy(); nin = out[0].cpu().numpy(); outl = out[1].cpu().numpy(); st = out[2].cpu().numpy()
        for f, p in enumerate(probs):
            r, pe, oe, se = ora[f]
            n = len(p["hasMP"])
            ok = np.abs(poseg[f] - pe).max() <= 1e-4 and nin[f] == r and np.array_equal(outl[f, :n], oe)
            if exact:
                ok = ok and int(st[f][0]) == int(se[0]) and int(st[f][1]) == int(se[1])
            if not ok:
                bad += 1
                log(f"PoseOptimization case {f} ({'edge-order' if exact else 'tree'} sums): MISMATCH dpose {np.abs(poseg[f] - pe).max():.2e} inliers {nin[f]} / {r} "
                    f"flags {int((outl[f, :n] != oe).sum())} its {st[f]} / {se}")
    return NP, bad


def run_pose_rig(NP, seed=11, log=print):
    """PoseOptimization on the KannalaBrandt8 rig (left-camera edges + right-camera ToBody edges), both modes."""
    import torch
    import oracle_lib as O
    from morb_slam_amd import Optimizer
    from morb_slam_amd.synth import make_pose_problem_fisheye
    rng = np.random.default_rng(seed + 2)
    probs = [make_pose_problem_fisheye(int(rng.choice([60, 200, 400])), int(rng.choice([40, 150, 300])), seed=900 + 31 * seed + s,
                                       outlier_frac=float(rng.choice([0.05, 0.15, 0.3]))) for s in range(NP)]
    cap = max(len(p["hasMP"]) for p in probs)
    F = len(probs)
    has = np.zeros((F, cap), np.uint8); obs = np.zeros((F, cap, 3), np.float32); inv = np.ones((F, cap), np.float32)
    Xw = np.zeros((F, cap, 3), np.float32); pose = np.zeros((F, 7), np.float32); cnt = np.zeros(F, np.int32); nl = np.zeros(F, np.int32)
    for f, p in enumerate(probs):
        n = len(p["hasMP"]); cnt[f] = n; nl[f] = p["Nleft"]
        has[f, :n] = p["hasMP"]; obs[f, :n] = p["obs"]; inv[f, :n] = p["invSigma2"]; Xw[f, :n] = p["Xw"]; pose[f] = p["pose0"]
    ora = [O.pose_optimization_fisheye(p) for p in probs]
    opt = Optimizer()
    bad = 0
    for exact in (True, False):
        opt.set_exact_order(exact)
        t = [torch.from_numpy(a.copy()).cuda() for a in (has, obs, inv, Xw, pose, nl, cnt)]
        nin, outl, st = opt.PoseOptimizationFisheye(t[0], t[1], t[2], t[3], t[4], t[5], t[6], probs[0]["camL"], probs[0]["camR"], probs[0]["Trl"])
        torch.cuda.synchronize()
        pg = t[4].cpu().numpy(); nin = nin.cpu().numpy(); outl = outl.cpu().numpy(); st = st.cpu().numpy()
        for f, p in enumerate(probs):
            r, pe, oe, se = ora[f]
            n = len(p["hasMP"])
            ok = np.abs(pg[f] - pe).max() <= 1e-4 and int(nin[f]) == r and np.array_equal(outl[f, :n], oe)
            if exact:
                ok = ok and int(st[f][0]) == int(se[0]) and int(st[f][1]) == int(se[1])
            if not ok:
                bad += 1
                log(f"PoseOptimization (rig) case {f} ({'edge-order' if exact else 'tree'} sums): MISMATCH dpose {np.abs(pg[f] - pe).max():.2e} inliers {nin[f]} / {r} "
                    f"flags {int((outl[f, :n] != oe).sum())} its {st[f]} / {se}")
    return NP, bad


def run_ba(NB, seed=11, log=print, max_points=3000):
    import oracle_lib as O
    from morb_slam_amd import Optimizer
    from morb_slam_amd.synth import make_ba_problem
    rng = np.random.default_rng(seed + 1)
    opt = Optimizer()
    bad = 0
    for s in range(NB):
        kw = dict(seed=700 + 97 * seed + s, n_free=int(rng.choice([5, 8, 12, 20])), n_fixed=int(rng.choice([2, 3, 6])),
                  n_points=int(rng.choice([p for p in (200, 500, 1500, 3000) if p <= max_points])), mono_frac=float(rng.choice([0.0, 0.15, 0.5])))
        b = make_ba_problem(**kw)
        kf, mp, erase, stats = opt.LocalBundleAdjustment(b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"])
        its, kfe, mpe, ee, se = O.local_ba(b)
        ok = (int(stats[0]) == int(se[0]) and int(stats[1]) == int(se[1]) and np.abs(kf - kfe).max() <= 1e-4
              and np.abs(mp - mpe).max() <= 1e-4 * max(1.0, np.abs(mpe).max()) and np.array_equal(erase, ee))   # iterations AND trials
        if not ok:
            bad += 1
            log(f"LocalBA case {kw}: MISMATCH dkf {np.abs(kf - kfe).max():.2e} dmp {np.abs(mp - mpe).max():.2e} flags {int((erase != ee).sum())} its {stats} / {se}")
    return NB, bad


def run_ba_rig(NB, seed=11, log=print):
    """LocalBundleAdjustment on the KannalaBrandt8 rig (left-camera and ToBody edges), the oracle's LM path and results."""
    import oracle_lib as O
    from morb_slam_amd import Optimizer
    from morb_slam_amd.synth import make_ba_problem_fisheye
    rng = np.random.default_rng(seed + 3)
    opt = Optimizer()
    bad = 0
    for s in range(NB):
        kw = dict(seed=1300 + 17 * seed + s, n_free=int(rng.choice([4, 6, 10, 16])), n_fixed=int(rng.choice([2, 4])), n_points=int(rng.choice([300, 800, 1500])),
                  right_frac=float(rng.choice([0.3, 0.45, 0.7])))
        b = make_ba_problem_fisheye(**kw)
        rig = dict(eRight=b["eRight"], camL=b["camL"], camR=b["camR"], Trl=b["Trl"])
        inertial = bool(s & 1)
        kf, mp, erase, stats = opt.LocalBundleAdjustment(b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], None, inertial=inertial, mode=0, rig=rig)
        its, kfe, mpe, ee, se = O.local_ba_fisheye(b, lambda100=inertial)
        ok = (int(stats[0]) == int(se[0]) and int(stats[1]) == int(se[1]) and np.abs(kf - kfe).max() <= 1e-4
              and np.abs(mp - mpe).max() <= 1e-4 * max(1.0, np.abs(mpe).max()) and np.array_equal(erase, ee))
        if not ok:
            bad += 1
            log(f"LocalBA (rig) case {kw} inertial {inertial}: MISMATCH dkf {np.abs(kf - kfe).max():.2e} dmp {np.abs(mp - mpe).max():.2e} flags {int((erase != ee).sum())} its {stats} / {se}")
    return NB, bad


if __name__ == "__main__":
    NP = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    NB = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    _, b1 = run_pose(NP)
    nr, b3 = run_pose_rig(max(NP // 2, 1))
    print(f"PoseOptimization on the rig: {nr} cases checked, {b3} mismatches", flush=True)
    b1 += b3
    print(f"PoseOptimization: {NP} cases checked")
    _, b2 = run_ba(NB)
    nr2, b4 = run_ba_rig(max(NB // 2, 1))
    print(f"LocalBundleAdjustment on the rig: {nr2} cases checked, {b4} mismatches", flush=True)
    b2 += b4
    print(f"LocalBundleAdjustment: {NB} cases checked; {b1 + b2} mismatches in total")
    sys.exit(1 if b1 + b2 else 0)
