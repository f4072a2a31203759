#!/usr/bin/env python3
"""Pin the oracle's OpenCV restatements against a REAL OpenCV — for whoever has one (this image has none: the script then says so and exits 0).

The reference's hot path calls five OpenCV primitives whose arithmetic lives outside /root/reference: cv::resize (INTER_LINEAR,
8-bit), cv::GaussianBlur (7 x 7, sigma 2, BORDER_REFLECT_101, 8-bit), cv::FAST (FAST-9/16 with non-maximum suppression, on the 35-px cells
ORBextractor cuts), cv::fastAtan2, and BFMatcher(NORM_HAMMING).knnMatch(k = 2) tie order.  oracle/cvprims.cc restates them from the published
algorithms; every "bit-exact" of this repository is relative to that restatement ("parity unpinned").  With `cv2` importable this script

  1. generates reference_* vectors from cv2 on seeded inputs (written to tests/golden/reference_opencv.npz),
  2. runs the oracle's restatements (oracle/liboracle.so through tests/oracle_lib.py) on the same inputs,
  3. prints, per primitive, identical / first difference — and exits non-zero on any difference.

    python tests/golden/make_opencv_golden.py [--write]       # five minutes with OpenCV >= 4.4 (the reference's CI pins 4.5.2)

A green run turns "parity unpinned" into "pinned against OpenCV <version>" for the five primitives; commit the .npz it writes and let
tests/test_oracle_cpu.py::test_oracle_matches_opencv_vectors (skipped while the file is absent) replay it without cv2."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_opencv.npz")
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def cases():
    """Seeded inputs shared by the generator and the replay test (shapes the extractor really meets)."""
    from morb_slam_amd.synth import make_image
    rng = np.random.default_rng(20260105)
    imgs = [make_image(752, 480, seed=11), make_image(320, 240, seed=12), rng.integers(0, 256, (97, 83), dtype=np.uint8)]
    resize_to = [(627, 400), (267, 200), (69, 81)]                    # (w, h) = round(size / 1.2)
    cells = [np.ascontiguousarray(imgs[0][y:y + 41, x:x + 41]) for (x, y) in ((16, 16), (300, 200), (700, 430))] + \
        [rng.integers(0, 256, (41, 41), dtype=np.uint8)]
    yx = rng.normal(0, 50, (4096, 2)).astype(np.float32)
    yx[:8] = [[0, 0], [0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, -1], [3, -4]]
    q = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t[10] = t[20]; t[30] = t[40]; q[5] = t[10]                        # exact ties: the match order among equal distances is what is pinned
    return dict(imgs=imgs, resize_to=resize_to, cells=cells, yx=yx, q=q, t=t)


def from_opencv(c):
    import cv2
    out = {"opencv_version": np.array(cv2.__version__)}
    for i, (im, (w, h)) in enumerate(zip(c["imgs"], c["resize_to"])):
        out[f"resize{i}"] = cv2.resize(im, (w, h), interpolation=cv2.INTER_LINEAR)
        out[f"blur{i}"] = cv2.GaussianBlur(im, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    for i, cell in enumerate(c["cells"]):
        for th in (20, 7):
            kps = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16).detect(cell)
            out[f"fast{i}_{th}"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
    out["atan2"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in c["yx"]], np.float32)
    m = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(c["q"], c["t"], k=2)
    out["knn_idx"] = np.array([[a.trainIdx, b.trainIdx] for a, b in m], np.int32)
    out["knn_dist"] = np.array([[a.distance, b.distance] for a, b in m], np.int32)
    return out


def from_oracle(c):
    import ctypes as C
    import oracle_lib as O
    L = O.lib()
    out = {}
    for i, (im, (w, h)) in enumerate(zip(c["imgs"], c["resize_to"])):
        dst = np.zeros((h, w), np.uint8)
        L.orc_resize_linear(O._p(im), im.shape[1], im.shape[0], im.shape[1], O._p(dst), w, h, w)
        out[f"resize{i}"] = dst
        bl = np.zeros_like(im)
        L.orc_gaussian7(O._p(im), im.shape[1], im.shape[0], im.shape[1], O._p(bl), im.shape[1])
        out[f"blur{i}"] = bl
    from morb_slam_amd.capi import KP_DTYPE
    for i, cell in enumerate(c["cells"]):
        for th in (20, 7):
            buf = np.zeros(4096, KP_DTYPE)
            n = L.orc_fast(O._p(cell), cell.shape[1], cell.shape[0], cell.shape[1], th, 1, O._p(buf), len(buf))
            out[f"fast{i}_{th}"] = np.stack([buf["x"][:n], buf["y"][:n], buf["response"][:n]], 1).astype(np.float32).reshape(-1, 3)
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    out["atan2"] = np.array([L.orc_fast_atan2(float(y), float(x)) for y, x in c["yx"]], np.float32)
    idx = np.zeros((len(c["q"]), 2), np.int32); dist = np.zeros((len(c["q"]), 2), np.int32)
    L.orc_knn2(O._p(c["q"]), len(c["q"]), O._p(c["t"]), len(c["t"]), O._p(idx), O._p(dist))
    out["knn_idx"], out["knn_dist"] = idx, dist
    return out


def compare(ref, got):
    bad = 0
    for k in sorted(ref):
        if k == "opencv_version":
            continue
        a, b = np.asarray(ref[k]), np.asarray(got[k])
        same = a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all() if a.dtype == np.float32 and a.shape == b.shape else \
            (a.shape == b.shape and np.array_equal(a, b))
        if same:
            print(f"{k:12s} identical ({a.size} values)")
        else:
            bad += 1
            where = "shape %s vs %s" % (a.shape, b.shape) if a.shape != b.shape else "first difference at %s: OpenCV %s, oracle %s" % (
                tuple(np.argwhere(a != b)[0]), a[tuple(np.argwhere(a != b)[0])], b[tuple(np.argwhere(a != b)[0])])
            print(f"{k:12s} DIFFERS: {where}")
    return bad


def main():
    try:
        import cv2  # noqa: F401
    except ImportError:
        print("cv2 is not importable here: nothing generated, nothing compared (the oracle stays 'parity unpinned'; run this where OpenCV >= 4.4 is installed)")
        return 0
    c = cases()
    ref = from_opencv(c)
    bad = compare(ref, from_oracle(c))
    if "--write" in sys.argv:
        np.savez_compressed(OUT, **ref)
        print("wrote", OUT)
    print("OpenCV", str(ref["opencv_version"]), "-", "all five primitives identical" if not bad else f"{bad} arrays differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
