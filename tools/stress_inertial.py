#!/usr/bin/env python3
"""Randomised sweep of the inertial optimisers against the CPU oracle (GPU box): LocalInertialBA (the same LM path — iterations and trials —, states within
1e-4, erased observations identical) and the tracking pair PoseInertialOptimizationLastKeyFrame -> LastFrame (states within 1e-4, outlier flags identical) over
more seeds than tests/test_inertial_gpu.py.  python tools/stress_inertial.py [ba cases] [tracking cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main(NB, NT):
    import torch
    import oracle_lib as orc
    from morb_slam_amd import Optimizer
    from morb_slam_amd.synth import imu_calib_diagonals, make_inertial_ba_problem, make_inertial_sequence
    opt = Optimizer(0)
    nga, walk = imu_calib_diagonals()
    rng = np.random.default_rng(5)
    bad = 0
    for s in range(NB):
        n_opt = int(rng.choice([4, 10, 10, 25])); large = n_opt > 10
        p = make_inertial_ba_problem(n_opt=n_opt, seed=100 + s, n_points=int(rng.choice([300, 800, 1500])))
        pre = np.stack([orc.imu_preintegrate(p["bias"], nga, walk, p["acc"][a:b], p["gyro"][a:b], p["dt"][a:b]) for a, b in zip(p["imuStart"][:-1], p["imuStart"][1:])])
        r, kf_o, mp_o, er_o, st_o = orc.local_inertial_ba(p, pre, bLarge=large)
        kf, mp, er, st = opt.LocalInertialBA(p["kfState"], p["kfKind"], p["mpPos"], p["mpClose"], p["eKF"], p["eMP"], p["eObs"], p["eInvSigma2"], p["iKF1"], p["iKF2"], pre,
                                             p["iRobust"], p["iInfoScale"], p["cam"], p["Tbc12"], bLarge=large)
        optk = p["kfKind"] == 0
        d = np.abs(mp - mp_o).max(1) / np.maximum(1.0, np.linalg.norm(mp_o, axis=1))
        ok = int(st[2]) == r and (int(st[0]), int(st[1])) == (int(st_o[0]), int(st_o[1])) and np.allclose(kf[optk], kf_o[optk], rtol=0, atol=1e-4) and d.max() < 1e-4 and np.array_equal(er, er_o)
        if not ok:
            bad += 1
            print(f"LocalInertialBA seed {100 + s} n_opt {n_opt}: MISMATCH ok {st[2]} / {r} its {st[:2]} / {st_o[:2]} dkf {np.abs(kf[optk] - kf_o[optk]).max():.2e} dmp {d.max():.2e} flags {int((er != er_o).sum())}", flush=True)
    print(f"LocalInertialBA: {NB} cases checked", flush=True)
    dev = torch.device("cuda", 0)
    one = lambda a: torch.from_numpy(np.ascontiguousarray(a)[None]).to(dev)
    for s in range(NT):
        pA, pB = make_inertial_sequence(int(rng.choice([30, 100, 400])), seed=300 + s, n_imu=int(rng.choice([10, 20])))
        pre = lambda p, a, g, dd: orc.imu_preintegrate(p["bias"], nga, walk, p[a], p[g], p[dd])
        preA, preBF, preBK = pre(pA, "acc", "gyro", "dt"), pre(pB, "accF", "gyroF", "dtF"), pre(pB, "acc", "gyro", "dt")
        rA, sA, oA, prA = orc.pose_inertial_optimization_last_keyframe(pA, preA)
        rB, sB, oB, _ = orc.pose_inertial_optimization_last_frame(pB, sA, preBF, preBK, prA)
        stA = one(pA["state0"]).clone()
        ninA, outA, priA = opt.PoseInertialOptimizationLastKeyFrame(one(pA["hasMP"]), one(pA["obs"]), one(pA["invSigma2"]), one(pA["Xw"]), one(pA["close"]), pA["cam"], pA["Tbc12"],
                                                                    one(pA["kfState"]), one(preA), stA)
        stB = one(pB["state0"]).clone()
        # frame B on both sides from the ORACLE's frame-A results (the two frame-A results agree to 1e-4, not bit for bit)
        ninB, outB, _ = opt.PoseInertialOptimizationLastFrame(one(pB["hasMP"]), one(pB["obs"]), one(pB["invSigma2"]), one(pB["Xw"]), one(pB["close"]), pB["cam"], pB["Tbc12"],
                                                              one(sA), one(preBF), one(preBK), one(prA), stB)
        torch.cuda.synchronize()
        ok = (np.allclose(stA[0].cpu().numpy(), sA, atol=1e-4) and int(ninA[0]) == rA and np.array_equal(outA[0].cpu().numpy(), oA) and
              np.allclose(stB[0].cpu().numpy(), sB, atol=1e-4) and int(ninB[0]) == rB and np.array_equal(outB[0].cpu().numpy(), oB))
        if not ok:
            bad += 1
            print(f"pose-inertial seed {300 + s}: MISMATCH A d {np.abs(stA[0].cpu().numpy() - sA).max():.2e} n {int(ninA[0])} / {rA} flags {int((outA[0].cpu().numpy() != oA).sum())}; "
                  f"B d {np.abs(stB[0].cpu().numpy() - sB).max():.2e} n {int(ninB[0])} / {rB} flags {int((outB[0].cpu().numpy() != oB).sum())}", flush=True)
    print(f"PoseInertialOptimizationLastKeyFrame -> LastFrame: {NT} cases checked; {bad} mismatches in total", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
