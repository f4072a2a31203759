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


if __name__ == "__main__":
    main()
