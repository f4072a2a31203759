#!/usr/bin/env python3
"""Generate the oracle_* fixtures in this directory.  The reference ships no tests or golden vectors for
this path and cannot be built here (OpenCV/Eigen absent), so these vectors are produced by the CPU oracle
(oracle/): they pin the oracle against regressions and give the GPU tests fixed expected outputs.  They are
oracle_* fixtures, never reference_* (SURVEY.md §8c).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle_lib import OracleExtractor  # noqa: E402
from morb_slam_amd.synth import make_image  # noqa: E402


def main():
    img = make_image(320, 240, seed=1234)
    o = OracleExtractor(300, 1.2, 4, 20, 7)
    mono, k, d = o(img, (100, 200))
    np.savez_compressed(os.path.join(HERE, "oracle_extract_320x240.npz"), image=img, mono=np.int32(mono),
                        kps=k.view(np.uint8).reshape(len(k), 28), desc=d,
                        cand_counts=np.array([len(o.level_candidates(l)) for l in range(4)], np.int32),
                        sel_counts=np.array([len(o.level_keypoints(l)) for l in range(4)], np.int32))
    # full-size case: hashes only (keeps the fixture small)
    img = make_image(752, 480, seed=1)
    o = OracleExtractor(1200, 1.2, 8, 20, 7)
    mono, k, d = o(img)
    with open(os.path.join(HERE, "oracle_extract_752x480_seed1.txt"), "w") as f:
        f.write(f"image_sha256 {hashlib.sha256(img.tobytes()).hexdigest()}\n")
        f.write(f"n {len(k)}\nmono {mono}\n")
        f.write(f"kps_sha256 {hashlib.sha256(k.tobytes()).hexdigest()}\n")
        f.write(f"desc_sha256 {hashlib.sha256(d.tobytes()).hexdigest()}\n")
        for l in range(8):
            f.write(f"level{l}_pyr_sha256 {hashlib.sha256(o.level_image(l).tobytes()).hexdigest()}\n")

    # visual-inertial chain (keyframe -> frame A -> frame B) and a small LocalInertialBA window: inputs are regenerated from the
    # seeds, the fixture holds the oracle's outputs (FP: compared with a tolerance, not by hash)
    import oracle_lib as orc
    from morb_slam_amd.synth import imu_calib_diagonals, make_inertial_ba_problem, make_inertial_sequence
    nga, walk = imu_calib_diagonals()
    pA, pB = make_inertial_sequence(200, seed=5, n_imu=15)
    preA = orc.imu_preintegrate(pA["bias"], nga, walk, pA["acc"], pA["gyro"], pA["dt"])
    rA = orc.pose_inertial_optimization_last_keyframe(pA, preA)
    preF = orc.imu_preintegrate(pB["bias"], nga, walk, pB["accF"], pB["gyroF"], pB["dtF"])
    preK = orc.imu_preintegrate(pB["bias"], nga, walk, pB["acc"], pB["gyro"], pB["dt"])
    rB = orc.pose_inertial_optimization_last_frame(pB, rA[1], preF, preK, rA[3])
    pw = make_inertial_ba_problem(n_opt=4, n_fixed_vis=2, n_points=150, n_imu=20, seed=5)
    prew = np.stack([orc.imu_preintegrate(pw["bias"], nga, walk, pw["acc"][a:b], pw["gyro"][a:b], pw["dt"][a:b])
                     for a, b in zip(pw["imuStart"][:-1], pw["imuStart"][1:])])
    rw = orc.local_inertial_ba(pw, prew)
    np.savez_compressed(os.path.join(HERE, "oracle_inertial.npz"), preA=preA, stateA=rA[1], outlierA=rA[2], priorA=rA[3], nA=np.int32(rA[0]),
                        stateB=rB[1], outlierB=rB[2], priorB=rB[3], nB=np.int32(rB[0]), ba_kf=rw[1], ba_mp=rw[2], ba_erase=rw[3],
                        ba_stats=rw[4], ba_ok=np.int32(rw[0]))


if __name__ == "__main__":
    main()
