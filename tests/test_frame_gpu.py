"""SURVEY 8(f) N4 (second half): Frame::UndistortKeyPoints, ComputeStereoFromRGBD, ComputeImageBounds — HIP vs the oracle's
restatement of cv::undistortPoints (parity unpinned: OpenCV is not vendored)."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TUM_CAM = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, bf=40.0)      # Examples/RGB-D/TUM1.yaml
TUM_DIST = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)


def _distort(xn, yn, k):
    """Forward Brown-Conrady model on normalised coordinates (independent check of the inverse)."""
    k1, k2, p1, p2, k3 = k
    r2 = xn * xn + yn * yn
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    return xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn), yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn


@pytest.fixture(scope="module")
def matcher():
    from morb_slam_amd import ORBmatcher
    return ORBmatcher(0.7, True, device=0)


def _records(xy, cap):
    """cv::KeyPoint records (28 B) with the given points; other fields arbitrary but fixed."""
    F = len(xy)
    rec = np.zeros((F, cap, 7), np.float32)
    for f in range(F):
        n = len(xy[f])
        rec[f, :n, 0:2] = xy[f]; rec[f, :n, 2] = 31.0; rec[f, :n, 3] = 12.5; rec[f, :n, 4] = 40.0
        rec[f, :n, 5] = np.array([3], np.int32).view(np.float32)[0]; rec[f, :n, 6] = np.array([-1], np.int32).view(np.float32)[0]
    return rec.view(np.uint8).reshape(F, cap, 28)


@pytest.mark.parametrize("dist", [TUM_DIST, (0.2624, -0.9531, -0.0054, 0.0026), (-0.28, 0.07, 0.0002, 0.00002, 0.0), (0.0, 0.1, 0, 0, 0)])
def test_undistort_keypoints(matcher, dist):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    counts = np.array([900, 0, 1, 640], np.int32); cap = 1000
    xy = [np.stack([rng.uniform(0, 640, n), rng.uniform(0, 480, n)], 1).astype(np.float32) for n in counts]
    rec = _records(xy, cap)
    out = matcher.UndistortKeyPoints(torch.from_numpy(rec).to(dev), torch.from_numpy(counts).to(dev), TUM_CAM, dist)
    torch.cuda.synchronize()
    out = out.cpu().numpy().view(np.float32).reshape(len(counts), cap, 7)
    inp = rec.view(np.float32).reshape(len(counts), cap, 7)
    for f, n in enumerate(counts):
        ref = xy[f] if dist[0] == 0.0 else orc.undistort_points(xy[f], TUM_CAM, dist)       # mDistCoef[0] == 0: mvKeysUn = mvKeys
        assert np.array_equal(out[f, :n, :2], ref)                                           # FP64 +,*,/ only: bit-exact
        assert np.array_equal(out[f, :n, 2:].view(np.uint32), inp[f, :n, 2:].view(np.uint32))     # the rest of the record is copied
        if n and dist[0] != 0.0:   # the inverse really inverts the forward model (five iterations: sub-0.05 px inside the image)
            xn = (out[f, :n, 0].astype(np.float64) - TUM_CAM["cx"]) / TUM_CAM["fx"]; yn = (out[f, :n, 1].astype(np.float64) - TUM_CAM["cy"]) / TUM_CAM["fy"]
            d5 = list(dist) + [0.0] * (5 - len(dist))
            xd, yd = _distort(xn, yn, d5)
            err = np.hypot(xd * TUM_CAM["fx"] + TUM_CAM["cx"] - xy[f][:, 0], yd * TUM_CAM["fy"] + TUM_CAM["cy"] - xy[f][:, 1])
            central = np.hypot((xy[f][:, 0] - TUM_CAM["cx"]) / TUM_CAM["fx"], (xy[f][:, 1] - TUM_CAM["cy"]) / TUM_CAM["fy"]) < 0.4
            # (the fixed-point iteration is only contractive where the model is: image corners of strong distortions may not converge)
            assert np.median(err) < 0.05 and (not central.any() or err[central].max() < 0.5), (np.median(err), err[central].max())


def test_stereo_from_rgbd_and_image_bounds(matcher):
    from morb_slam_amd import ORBmatcher
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(6)
    H, W, cap = 480, 640, 700
    counts = np.array([700, 333], np.int32)
    depth = rng.uniform(0.3, 8.0, (2, H, W)).astype(np.float32)
    depth[rng.random((2, H, W)) < 0.2] = 0.0                    # holes
    xy = [np.stack([rng.uniform(0, W - 0.01, n), rng.uniform(0, H - 0.01, n)], 1).astype(np.float32) for n in counts]
    rec = _records(xy, cap)
    kps = torch.from_numpy(rec).to(dev); cnt = torch.from_numpy(counts).to(dev)
    un = matcher.UndistortKeyPoints(kps, cnt, TUM_CAM, TUM_DIST)
    ur, dd = matcher.ComputeStereoFromRGBD(kps, un, cnt, torch.from_numpy(depth).to(dev), TUM_CAM["bf"])
    torch.cuda.synchronize()
    unh = un.cpu().numpy().view(np.float32).reshape(2, cap, 7)
    for f, n in enumerate(counts):
        ur_o, d_o = orc.stereo_from_rgbd(xy[f], unh[f, :n, :2], depth[f], TUM_CAM["bf"])
        assert np.array_equal(ur[f, :n].cpu().numpy(), ur_o) and np.array_equal(dd[f, :n].cpu().numpy(), d_o)
        assert (ur[f, n:] == -1).all() and (dd[f, n:] == -1).all()
        assert 0.7 < (d_o > 0).mean() < 0.9
    b = ORBmatcher.ComputeImageBounds(W, H, TUM_CAM, TUM_DIST)
    c = orc.undistort_points(np.array([[0, 0], [W, 0], [0, H], [W, H]], np.float32), TUM_CAM, TUM_DIST)
    assert b == (float(min(c[0, 0], c[2, 0])), float(max(c[1, 0], c[3, 0])), float(min(c[0, 1], c[1, 1])), float(max(c[2, 1], c[3, 1])))
    assert ORBmatcher.ComputeImageBounds(W, H, TUM_CAM, (0.0, 0, 0, 0)) == (0.0, float(W), 0.0, float(H))


@pytest.mark.parametrize("weighting,scoring", [(0, 0), (1, 1), (0, 5), (1, 5), (2, 0), (3, 5), (0, 3)])
def test_bow_vector(matcher, weighting, scoring):
    """BowVector of TemplatedVocabulary::transform: bit-exact doubles, map order, stopped words, every weighting / norm branch."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(17 + weighting * 7 + scoring)
    n_nodes, cap = 5000, 1500
    weight = rng.uniform(0.5, 9.0, n_nodes); weight[rng.random(n_nodes) < 0.1] = 0.0          # stopped words
    word = rng.permutation(n_nodes).astype(np.int32)                                           # node -> WordId (any injective map)
    counts = np.array([1500, 0, 1, 777], np.int32)
    leaf = np.full((4, cap), -1, np.int32)
    for f, n in enumerate(counts):
        leaf[f, :n] = rng.integers(0, 400 if f == 0 else n_nodes, n)                           # image 0: many repeated words
    leaf[3, 5] = -1                                                                            # a feature outside the vocabulary
    for use_word in (True, False):
        w, v, c = matcher.bow_vector(torch.from_numpy(leaf).to(dev), torch.from_numpy(counts).to(dev), torch.from_numpy(weight).to(dev),
                                     torch.from_numpy(word).to(dev) if use_word else None, weighting, scoring)
        torch.cuda.synchronize()
        w, v, c = w.cpu().numpy(), v.cpu().numpy(), c.cpu().numpy()
        for f, n in enumerate(counts):
            ow, ov = orc.bow_vector(leaf[f, :n], weight, word if use_word else None, weighting, scoring)
            assert int(c[f]) == len(ow)
            assert np.array_equal(w[f, :c[f]], ow) and np.array_equal(v[f, :c[f]], ov)         # bit-exact doubles
            if len(ov) and scoring == 0:
                assert abs(np.abs(ov).sum() - 1.0) < 1e-12
