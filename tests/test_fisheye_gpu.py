"""GPU parity of the fisheye (KannalaBrandt8) pieces: Frame::ComputeStereoFishEyeMatches and PoseOptimization with the
right-camera "ToBody" edges.  The float transcendentals of the KB8 model (atan2f, tanf, cosf, sinf) are glibc's on both sides — the
kernels restate them bit for bit (csrc/libm_f32.h, tools/check_libm_f32.cc) — so match tables, frustum flags, projections and outlier
flags are compared exactly; depths / 3-D points come out of an FP64 Jacobi null vector (different sweep order than Eigen): 1e-4."""
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from morb_slam_amd.synth import make_fisheye_features, make_pose_problem_fisheye

pytestmark = pytest.mark.gpu
_OFF = 1000 * int(__import__("os").environ.get("MORB_TEST_SEED", "0"))   # tools/stress_matchers.sh: the same tests on other scenes


def test_stereo_fisheye_matches():
    import torch
    from morb_slam_amd import KP_DTYPE, ORBmatcher
    sets = [make_fisheye_features(seed=_OFF + s, n_pairs=500 + 100 * s) for s in range(3)]
    cap = max(max(len(f["kL"]), len(f["kR"])) for f in sets) + 7
    nimg = 2 * len(sets)
    kps = np.zeros((nimg, cap), KP_DTYPE); desc = np.zeros((nimg, cap, 32), np.uint8)
    cnt = np.zeros(nimg, np.int32); mono = np.zeros(nimg, np.int32)
    for f, fe in enumerate(sets):
        kps[2 * f, :len(fe["kL"])] = fe["kL"]; desc[2 * f, :len(fe["kL"])] = fe["dL"]; cnt[2 * f] = len(fe["kL"]); mono[2 * f] = fe["monoL"]
        kps[2 * f + 1, :len(fe["kR"])] = fe["kR"]; desc[2 * f + 1, :len(fe["kR"])] = fe["dR"]; cnt[2 * f + 1] = len(fe["kR"]); mono[2 * f + 1] = fe["monoR"]
    sigma2 = (1.2 ** np.arange(8)) ** 2
    m = ORBmatcher()
    dk = torch.from_numpy(kps.view(np.uint8).reshape(nimg, cap, 28)).cuda()
    o = m.ComputeStereoFishEyeMatches(dk, torch.from_numpy(desc).cuda(), torch.from_numpy(cnt).cuda(), torch.from_numpy(mono).cuda(),
                                      sets[0]["camL"], sets[0]["camR"], sets[0]["Rlr"], sets[0]["tlr"], sigma2.astype(np.float32))
    torch.cuda.synchronize()
    tot = 0
    for f, fe in enumerate(sets):
        n, l2r, r2l, dep, p3 = O.stereo_fisheye_matches(fe, sigma2)
        nl, nr = len(fe["kL"]), len(fe["kR"])
        assert int(o["nMatches"][f]) == n
        np.testing.assert_array_equal(o["leftToRight"][f, :nl].cpu().numpy(), l2r)
        np.testing.assert_array_equal(o["rightToLeft"][f, :nr].cpu().numpy(), r2l)
        np.testing.assert_allclose(o["depth"][f, :nl].cpu().numpy(), dep, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(o["p3D"][f, :nl].cpu().numpy(), p3, rtol=1e-4, atol=1e-4)
        assert (l2r[:fe["monoL"]] == -1).all()          # mono-area features are never stereo-matched
        tot += n
    assert tot > 400


def test_pose_optimization_fisheye():
    import torch
    from morb_slam_amd import Optimizer
    probs = [make_pose_problem_fisheye(seed=s) for s in range(4)] + [make_pose_problem_fisheye(200, 500, seed=9, outlier_frac=0.25)]
    cap = max(len(p["hasMP"]) for p in probs)
    F = len(probs)
    has = np.zeros((F, cap), np.uint8); obs = np.zeros((F, cap, 3), np.float32); inv = np.ones((F, cap), np.float32)
    Xw = np.zeros((F, cap, 3), np.float32); pose = np.zeros((F, 7), np.float32); cnt = np.zeros(F, np.int32); nl = np.zeros(F, np.int32)
    for f, p in enumerate(probs):
        n = len(p["hasMP"]); cnt[f] = n; nl[f] = p["Nleft"]
        has[f, :n] = p["hasMP"]; obs[f, :n] = p["obs"]; inv[f, :n] = p["invSigma2"]; Xw[f, :n] = p["Xw"]; pose[f] = p["pose0"]
    t = [torch.from_numpy(a).cuda() for a in (has, obs, inv, Xw, pose, nl, cnt)]
    nin, outl, stats = Optimizer().PoseOptimizationFisheye(t[0], t[1], t[2], t[3], t[4], t[5], t[6], probs[0]["camL"], probs[0]["camR"],
                                                           probs[0]["Trl"])
    torch.cuda.synchronize()
    pg = t[4].cpu().numpy()
    for f, p in enumerate(probs):
        r, pe, oe, se = O.pose_optimization_fisheye(p)
        n = len(p["hasMP"])
        assert np.abs(pg[f] - pe).max() <= 1e-4, (f, pg[f], pe)
        assert int(nin[f]) == r
        np.testing.assert_array_equal(outl[f, :n].cpu().numpy(), oe)
        assert np.abs(pg[f] - p["true"]).max() < 0.03


def _fisheye_params():
    from morb_slam_amd.capi import make_frame_params
    from morb_slam_amd.synth import TUMVI_CAM_L
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    return make_frame_params(512, 512, float(TUMVI_CAM_L[0]), float(TUMVI_CAM_L[1]), float(TUMVI_CAM_L[2]), float(TUMVI_CAM_L[3]),
                             19.0, 0.1, sf, sf * sf), sf


def test_is_in_frustum_checks_kb8():
    """Frame::isInFrustumChecks, both cameras of the TUM-VI rig.  uv goes through atan2f / cosf / sinf, restated from glibc on the
    device: every field bit-exact, flags identical."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.synth import TUMVI_CAM_L, TUMVI_CAM_R, TUMVI_T_C1_C2
    P, _ = _fisheye_params()
    rng = np.random.default_rng(_OFF + 11)
    Fn, M = 3, 900
    Tlr = TUMVI_T_C1_C2.astype(np.float32)
    Trl = np.linalg.inv(TUMVI_T_C1_C2).astype(np.float32)
    m = ORBmatcher()
    tot = 0
    for side, cam in (("L", TUMVI_CAM_L), ("R", TUMVI_CAM_R)):
        R = np.zeros((Fn, 9), np.float32); t = np.zeros((Fn, 3), np.float32); O3 = np.zeros((Fn, 3), np.float32)
        Pw = np.zeros((Fn, M, 3), np.float32); nrm = np.zeros((Fn, M, 3), np.float32)
        maxD = np.zeros((Fn, M), np.float32); minD = np.zeros((Fn, M), np.float32)
        for f in range(Fn):
            a = rng.normal(0, 0.05, 3)
            th = np.linalg.norm(a); k = a / th
            K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            Rcw = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32)
            tcw = rng.normal(0, 0.2, 3).astype(np.float32)
            Rwc = Rcw.T.copy(); Ow = (-Rwc @ tcw).astype(np.float32)
            if side == "R":   # Frame.cc:1283-1288
                mR = (Trl[:3, :3] @ Rcw).astype(np.float32); mt = (Trl[:3, :3] @ tcw + Trl[:3, 3]).astype(np.float32)
                twc = (Rwc @ Tlr[:3, 3] + Ow).astype(np.float32)
            else:
                mR, mt, twc = Rcw, tcw, Ow
            R[f] = mR.reshape(-1); t[f] = mt; O3[f] = twc
            Xc = np.stack([rng.uniform(-6, 6, M), rng.uniform(-5, 5, M), rng.uniform(-1, 9, M)], 1)
            X = (Xc - tcw) @ Rcw            # world points: Rwc (Xc - tcw)
            Pw[f] = X.astype(np.float32)
            d = np.linalg.norm(X - Ow, axis=1)
            v = (X - Ow) / d[:, None] + rng.normal(0, 0.5, (M, 3))
            nrm[f] = (v / np.linalg.norm(v, axis=1)[:, None]).astype(np.float32)
            maxD[f] = (d * rng.uniform(0.6, 3.0, M)).astype(np.float32); minD[f] = (maxD[f] / 4.0).astype(np.float32)
        nMP = np.full(Fn, M, np.int32)
        g = m.isInFrustumChecks(P, cam, *[torch.from_numpy(x).cuda() for x in (R, t, O3, nMP, Pw, nrm, maxD, minD)], 0.5)
        torch.cuda.synchronize()
        g = {k: v.cpu().numpy() for k, v in g.items()}
        kps0 = np.zeros(1, O.KP_DTYPE)
        for f in range(Fn):
            Fo = O.make_frame(P, kps0, np.zeros((1, 32), np.uint8), None)
            e = O.is_in_frustum_kb8(Fo, cam, R[f], t[f], O3[f], Pw[f], nrm[f], maxD[f], minD[f], 0.5)
            same = g["inView"][f] == e["inView"]
            np.testing.assert_array_equal(g["inView"][f], e["inView"])
            both = (g["inView"][f] == 1) & (e["inView"] == 1)
            np.testing.assert_array_equal(g["projX"][f][both], e["projX"][both])
            np.testing.assert_array_equal(g["projY"][f][both], e["projY"][both])
            np.testing.assert_array_equal(g["depth"][f][both], e["depth"][both])
            np.testing.assert_array_equal(g["viewCos"][f][both], e["viewCos"][both])
            np.testing.assert_array_equal(g["level"][f][both], e["level"][both])
            rej = (g["inView"][f] == 0) & same
            assert (g["level"][f][rej] == -1).all() and (g["projX"][f][rej] == -1).all()
            tot += int(both.sum())
    assert tot > 600


def test_search_by_projection_mappoints_fisheye():
    """SearchByProjection(F, MapPoints) with F.Nleft != -1: left pass, right pass, stereo-partner claims, the ratio-test
    `continue` that skips the right pass.  The tracking fields are inputs (same for oracle and GPU): exact tables."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBmatcher
    P, sf = _fisheye_params()
    sigma2 = (sf * sf).astype(np.float32)
    sets = [make_fisheye_features(seed=_OFF + 20 + s, n_pairs=450 + 80 * s) for s in range(3)]
    Fn = len(sets)
    cap = max(len(f["kL"]) + len(f["kR"]) for f in sets) + 5
    M = 700
    rng = np.random.default_rng(_OFF + 77)
    kps = np.zeros((Fn, cap), KP_DTYPE); desc = np.zeros((Fn, cap, 32), np.uint8)
    cnt = np.zeros(Fn, np.int32); nLeft = np.zeros(Fn, np.int32)
    l2r = np.full((Fn, cap), -1, np.int32); r2l = np.full((Fn, cap), -1, np.int32)
    blocked = (rng.random((Fn, cap)) < 0.05).astype(np.uint8)
    keys = ("inView", "projX", "projY", "depth", "level", "viewCos")
    dt = dict(inView=np.uint8, projX=np.float32, projY=np.float32, depth=np.float32, level=np.int32, viewCos=np.float32)
    tL = {k: np.zeros((Fn, M), dt[k]) for k in keys}; tR = {k: np.zeros((Fn, M), dt[k]) for k in keys}
    tL["level"][:] = -1; tR["level"][:] = -1
    isBad = (rng.random((Fn, M)) < 0.03).astype(np.uint8); hasObs = (rng.random((Fn, M)) < 0.7).astype(np.uint8)
    mpDesc = rng.integers(0, 256, (Fn, M, 32), dtype=np.uint8)
    for f, fe in enumerate(sets):
        nl, nr = len(fe["kL"]), len(fe["kR"])
        kps[f, :nl] = fe["kL"]; kps[f, nl:nl + nr] = fe["kR"]; desc[f, :nl] = fe["dL"]; desc[f, nl:nl + nr] = fe["dR"]
        cnt[f] = nl + nr; nLeft[f] = nl
        _, a, b, _, _ = O.stereo_fisheye_matches(fe, sigma2)
        l2r[f, :nl] = a; r2l[f, :nr] = b

        def aim(t, i, kp, d):   # point map point i at feature kp of one camera
            t["inView"][f, i] = 1
            t["projX"][f, i] = kp["x"] + rng.uniform(-1.5, 1.5); t["projY"][f, i] = kp["y"] + rng.uniform(-1.5, 1.5)
            t["level"][f, i] = min(7, int(kp["octave"]) + int(rng.integers(0, 2)))
            t["viewCos"][f, i] = rng.choice([0.9995, 0.9, 0.6]); t["depth"][f, i] = rng.uniform(1, 60)
            flips = rng.choice([3, 10, 25, 70])
            dd = d.copy()
            bits = rng.choice(256, flips, replace=False)
            for bt in bits: dd[bt >> 3] ^= np.uint8(1 << (bt & 7))
            mpDesc[f, i] = dd
        for i in range(M):
            u = rng.random()
            if u < 0.4:
                j = rng.integers(0, nl); aim(tL, i, fe["kL"][j], fe["dL"][j])
            elif u < 0.7:
                j = rng.integers(0, nr); aim(tR, i, fe["kR"][j], fe["dR"][j])
            elif u < 0.9:
                j = rng.integers(0, nl); aim(tL, i, fe["kL"][j], fe["dL"][j])
                jr = a[j] if a[j] >= 0 else rng.integers(0, nr)
                keep = mpDesc[f, i].copy(); aim(tR, i, fe["kR"][jr], fe["dR"][jr]); mpDesc[f, i] = keep
            else:   # in view of the right camera only, nothing nearby in most cases
                tR["inView"][f, i] = 1; tR["projX"][f, i] = rng.uniform(0, 512); tR["projY"][f, i] = rng.uniform(0, 512)
                tR["level"][f, i] = rng.integers(-1, 8); tR["viewCos"][f, i] = 0.8; tL["depth"][f, i] = rng.uniform(1, 60)
    m = ORBmatcher(0.8, True)
    fImg = np.arange(Fn, dtype=np.int32); nMP = np.full(Fn, M, np.int32)
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dk = cu(kps.view(np.uint8).reshape(Fn, cap, 28))
    gL = {k: cu(v) for k, v in tL.items()}; gR = {k: cu(v) for k, v in tR.items()}
    for th, far in ((1.0, False), (3.0, True)):
        mt, nm = m.SearchByProjectionMapPointsFisheye(P, cu(fImg), dk, cu(desc), cu(cnt), cu(nLeft), cu(l2r), cu(r2l), cu(blocked), cu(nMP),
                                                      gL, gR, cu(isBad), cu(mpDesc), cu(hasObs), th, far, 40.0)
        torch.cuda.synchronize()
        mt, nm = mt.cpu().numpy(), nm.cpu().numpy()
        tot = 0
        for f in range(Fn):
            N = int(cnt[f])
            Fo = O.make_frame(P, kps[f, :N], desc[f, :N], None)
            ne, me = O.search_by_projection_mps_fisheye(Fo, int(nLeft[f]), l2r[f], r2l[f], blocked[f, :N],
                                                        {k: v[f] for k, v in tL.items()}, {k: v[f] for k, v in tR.items()},
                                                        isBad[f], mpDesc[f], hasObs[f], th, far, 40.0, 0.8)
            assert int(nm[f]) == ne
            np.testing.assert_array_equal(mt[f, :N], me)
            tot += ne
        assert tot > 600


@pytest.mark.parametrize("k,Lv,lup", [(10, 3, 1), (3, 2, 1), (10, 6, 4)])   # ~10 features per node (register path) / ~400 (general path) / the ORBvoc shape
def test_search_by_bow_fisheye(k, Lv, lup):
    """SearchByBoW(pKF, F, ...) with a fisheye frame: left / right candidates ranked separately, right winner taken
    whenever the left best distance passes TH_LOW (ORBmatcher.cc:262-299, :333-365)."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBmatcher
    from morb_slam_amd.synth import make_vocabulary
    sets = [make_fisheye_features(seed=_OFF + 40 + s, n_pairs=420 + 60 * s) for s in range(3)]
    nimg = len(sets)
    cap = max(len(f["kL"]) + len(f["kR"]) for f in sets) + 3
    kps = np.zeros((nimg, cap), KP_DTYPE); desc = np.zeros((nimg, cap, 32), np.uint8)
    cnt = np.zeros(nimg, np.int32); nl = np.zeros(nimg, np.int32)
    rng = np.random.default_rng(_OFF + 5)
    for f, fe in enumerate(sets):
        a, b = len(fe["kL"]), len(fe["kR"])
        kps[f, :a] = fe["kL"]; kps[f, a:a + b] = fe["kR"]; desc[f, :a] = fe["dL"]; desc[f, a:a + b] = fe["dR"]
        cnt[f] = a + b; nl[f] = a
    # make image 1 and 2 noisy copies of image 0's descriptors so that many keyframe features find close left AND right partners
    for f in (1, 2):
        n0 = min(cnt[0], cnt[f])
        noise = np.packbits(rng.random((n0, 256)) < 0.04, axis=1)
        desc[f, :n0] = desc[0, :n0] ^ noise
        half = min(nl[0], cnt[f] - nl[f])
        desc[f, nl[f]:nl[f] + half] = desc[0, :half] ^ np.packbits(rng.random((half, 256)) < 0.05, axis=1)
    vd, vf = make_vocabulary(k, Lv, seed=3)
    has = (rng.random((nimg, cap)) < 0.8).astype(np.uint8)
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dk, dd, dc = cu(kps.view(np.uint8).reshape(nimg, cap, 28)), cu(desc), cu(cnt)
    kf = np.array([0, 0, 1, 2], np.int32); fr = np.array([1, 2, 2, 0], np.int32)
    for ratio, ori in ((0.7, True), (0.9, False)):
        m = ORBmatcher(ratio, ori)
        _, node = m.bow_transform(dd, dc, cu(vd), cu(vf), k, Lv, lup)
        match, nm = m.SearchByBoW(cu(kf), cu(fr), dk, dd, node, dc, cu(has), nLeft=cu(nl[fr]))
        torch.cuda.synchronize()
        nn_, match, nm = node.cpu().numpy(), match.cpu().numpy(), nm.cpu().numpy()
        tot = right = 0
        for p, (a, b) in enumerate(zip(kf, fr)):
            na, nb = cnt[a], cnt[b]
            ne, me = O.search_by_bow_fisheye(desc[a, :na], kps[a, :na]["angle"], has[a, :na], nn_[a, :na], desc[b, :nb], kps[b, :nb]["angle"],
                                             nn_[b, :nb], int(nl[b]), ratio, ori)
            assert nm[p] == ne
            np.testing.assert_array_equal(match[p, :nb], me)
            tot += ne; right += int((me[nl[b]:] >= 0).sum())
        assert tot > 300 and right > 50


def test_search_by_projection_last_frame_fisheye():
    """SearchByProjection(CurrentFrame, LastFrame, th, bMono) with CurrentFrame.Nleft != -1: left pass, right pass through
    GetRelativePoseTrl() with the LEFT camera model, the empty-left-window `continue`, rotation histogram over both."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBmatcher
    from morb_slam_amd.synth import TUMVI_CAM_L, TUMVI_T_C1_C2, kb8_project, _quat_from_rotvec, _quat_rot, _quat_from_R
    P, sf = _fisheye_params()
    Trl_m = np.linalg.inv(TUMVI_T_C1_C2)
    Trl7 = np.concatenate([_quat_from_R(Trl_m[:3, :3]), Trl_m[:3, 3]]).astype(np.float32)
    Fn, M = 3, 520
    rng = np.random.default_rng(_OFF + 123)
    frames = []
    for f in range(Fn):
        Xw = np.stack([rng.uniform(-3, 3, M), rng.uniform(-2.5, 2.5, M), rng.uniform(1, 8, M)], 1)
        Tcw = np.concatenate([_quat_from_rotvec(rng.normal(0, 0.02, 3)), rng.normal(0, 0.05, 3)]).astype(np.float32)
        Xc = np.array([_quat_rot(Tcw[:4].astype(np.float64), x) + Tcw[4:] for x in Xw])
        Xr = Xc @ Trl_m[:3, :3].T + Trl_m[:3, 3]
        uvL, uvR = kb8_project(TUMVI_CAM_L, Xc), kb8_project(TUMVI_CAM_L, Xr)
        octv = rng.integers(0, 8, M); ang = rng.uniform(0, 360, M)
        mpd = rng.integers(0, 256, (M, 32), dtype=np.uint8)

        def side(uv, frac, nd):
            ok = (uv > 8).all(1) & (uv < 504).all(1) & (rng.random(M) < frac)
            idx = np.nonzero(ok)[0]
            n = len(idx) + nd
            k = np.zeros(n, KP_DTYPE); d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            k["x"] = rng.uniform(5, 507, n); k["y"] = rng.uniform(5, 507, n); k["octave"] = rng.integers(0, 8, n)
            k["angle"] = rng.uniform(0, 360, n); k["size"] = 31; k["class_id"] = -1
            sel = rng.permutation(n)[:len(idx)]
            k["x"][sel] = uv[idx, 0] + rng.normal(0, 1.0, len(idx)); k["y"][sel] = uv[idx, 1] + rng.normal(0, 1.0, len(idx))
            k["octave"][sel] = np.clip(octv[idx] + rng.integers(-1, 2, len(idx)), 0, 7)
            k["angle"][sel] = (ang[idx] + rng.choice([0.0, 0.0, 0.0, 90.0], len(idx)) + rng.normal(0, 3, len(idx))) % 360
            d[sel] = mpd[idx] ^ np.packbits(rng.random((len(idx), 256)) < 0.06, axis=1)
            return k, d
        kL, dL = side(uvL, 0.8, 150); kR, dR = side(uvR, 0.6, 120)
        last = np.zeros(M, KP_DTYPE); last["octave"] = octv; last["angle"] = ang
        frames.append(dict(kL=kL, dL=dL, kR=kR, dR=dR, last=last, Xw=Xw.astype(np.float32), Tcw=Tcw, mpd=mpd,
                           valid=(rng.random(M) < 0.9).astype(np.uint8), hasObs=(rng.random(M) < 0.7).astype(np.uint8)))
    cap = max(max(len(fr["kL"]) + len(fr["kR"]), M) for fr in frames) + 2
    nimg = 2 * Fn
    kps = np.zeros((nimg, cap), KP_DTYPE); desc = np.zeros((nimg, cap, 32), np.uint8); cnt = np.zeros(nimg, np.int32)
    nl = np.zeros(Fn, np.int32); Tcw = np.zeros((Fn, 7), np.float32)
    valid = np.zeros((Fn, cap), np.uint8); Xw = np.zeros((Fn, cap, 3), np.float32); mpd = np.zeros((Fn, cap, 32), np.uint8)
    hasObs = np.zeros((Fn, cap), np.uint8)
    blocked = (rng.random((Fn, cap)) < 0.04).astype(np.uint8)
    for f, fr in enumerate(frames):
        a, b = len(fr["kL"]), len(fr["kR"])
        kps[2 * f, :a] = fr["kL"]; kps[2 * f, a:a + b] = fr["kR"]; desc[2 * f, :a] = fr["dL"]; desc[2 * f, a:a + b] = fr["dR"]
        cnt[2 * f] = a + b; nl[f] = a
        kps[2 * f + 1, :M] = fr["last"]; cnt[2 * f + 1] = M
        Tcw[f] = fr["Tcw"]; valid[f, :M] = fr["valid"]; Xw[f, :M] = fr["Xw"]; mpd[f, :M] = fr["mpd"]; hasObs[f, :M] = fr["hasObs"]
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dk = cu(kps.view(np.uint8).reshape(nimg, cap, 28))
    cur = np.arange(0, nimg, 2, dtype=np.int32); lst = cur + 1
    m = ORBmatcher(0.9, True)
    for th, fwd, bwd in ((7.0, 0, 0), (15.0, 1, 0), (7.0, 0, 1)):
        fw = np.full(Fn, fwd, np.uint8); bw = np.full(Fn, bwd, np.uint8)
        mt, nm = m.SearchByProjectionLastFrameFisheye(P, TUMVI_CAM_L, Trl7, cu(cur), cu(lst), cu(nl), dk, cu(desc), cu(cnt), cu(blocked),
                                                      cu(Tcw), cu(valid), cu(Xw), cu(mpd), cu(hasObs), th, cu(fw), cu(bw))
        torch.cuda.synchronize()
        mt, nm = mt.cpu().numpy(), nm.cpu().numpy()
        tot = right = 0
        for f in range(Fn):
            N = int(cnt[2 * f])
            Fo = O.make_frame(P, kps[2 * f, :N], desc[2 * f, :N], None)
            ne, me = O.search_by_projection_last_fisheye(Fo, int(nl[f]), TUMVI_CAM_L, Trl7, blocked[f, :N], Tcw[f], kps[2 * f + 1, :M],
                                                         valid[f, :M], Xw[f, :M], mpd[f, :M], hasObs[f, :M], th, fwd, bwd, True)
            assert int(nm[f]) == ne
            np.testing.assert_array_equal(mt[f, :N], me)
            tot += ne; right += int((me[nl[f]:] >= 0).sum())
        assert tot > 500 and right > 150


@pytest.mark.parametrize("kw", [dict(seed=1), dict(seed=2, n_free=6, n_fixed=2, n_points=500, right_frac=0.7),
                                dict(seed=3, n_free=45, n_fixed=5, n_points=2500)])   # (45 free keyframes: the global-memory LDL^T)
def test_local_ba_fisheye_matches_oracle(kw):
    """LocalBundleAdjustment on the KB8 rig: EdgeSE3ProjectXYZ (left camera) + EdgeSE3ProjectXYZToBody (right camera
    behind mTrl), both solver modes.  Same tolerances as the pinhole test: poses 1e-4, points 1e-3 relative."""
    from morb_slam_amd import Optimizer
    from morb_slam_amd.synth import make_ba_problem_fisheye
    b = make_ba_problem_fisheye(**kw)
    assert 0.2 < b["eRight"].mean() < 0.8
    rig = dict(eRight=b["eRight"], camL=b["camL"], camR=b["camR"], Trl=b["Trl"])
    opt = Optimizer()
    for inertial, mode in ((False, 0), (True, 0), (False, 1)) if kw.get("n_free", 10) <= 20 else ((False, 0),):
        kf, mp, erase, stats = opt.LocalBundleAdjustment(b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"],
                                                         b["eInvSigma2"], None, inertial=inertial, mode=mode, rig=rig)
        its, kfe, mpe, ee, se = O.local_ba_fisheye(b, lambda100=inertial)
        assert int(stats[0]) == int(se[0]) and int(stats[1]) == int(se[1]), (stats, se)
        assert np.abs(kf - kfe).max() <= 1e-4
        assert np.abs(mp - mpe).max() <= 1e-4 * max(1.0, np.abs(mpe).max())
        np.testing.assert_array_equal(erase, ee)   # (the KB8 projection's libm calls are the same restated code on both sides, libm_f32.h)
        nf = int((b["kfFixed"] == 0).sum())
        assert np.abs(kf[:nf] - b["true_poses"][:nf]).max() < 0.05
        assert 0.01 < ee.mean() < 0.3


def _tri_scene():
    """Three keyframes of the TUM-VI KB8 rig looking at one cloud: per keyframe the left | right feature row, counts, NLeft, poses."""
    from morb_slam_amd import KP_DTYPE
    from morb_slam_amd.synth import (TUMVI_CAM_L, TUMVI_CAM_R, TUMVI_T_C1_C2, kb8_project, make_vocabulary, _quat_from_rotvec,
                                     _quat_rot)
    P, sf = _fisheye_params()
    import os
    rng = np.random.default_rng(314 + int(os.environ.get("MORB_TEST_SEED", "0")))   # (tools/stress_matchers.sh: the rig tests again on other scenes)
    Trl_m = np.linalg.inv(TUMVI_T_C1_C2)

    def se3(rv, t):
        q = _quat_from_rotvec(np.asarray(rv, float))
        R = np.array([_quat_rot(q, e) for e in np.eye(3)]).T
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        return T
    M = 700
    Xw = np.stack([rng.uniform(-3, 3, M), rng.uniform(-2.5, 2.5, M), rng.uniform(1.5, 7, M)], 1)
    base = rng.integers(0, 256, (M, 32), dtype=np.uint8)
    poses = [se3([0, 0, 0], [0, 0, 0]), se3([0.01, -0.03, 0.005], [-0.35, 0.02, 0.05]), se3([-0.02, 0.02, 0.0], [0.3, -0.05, -0.1])]
    feats, vis = [], []
    for Tcw in poses:
        seen = []
        Xl = Xw @ Tcw[:3, :3].T + Tcw[:3, 3]
        Xr = Xl @ Trl_m[:3, :3].T + Trl_m[:3, 3]
        parts = []
        for Xs, cam, frac in ((Xl, TUMVI_CAM_L, 0.75), (Xr, TUMVI_CAM_R, 0.6)):
            uv = kb8_project(cam, Xs)
            ok = (uv > 8).all(1) & (uv < 504).all(1) & (Xs[:, 2] > 0.3) & (rng.random(M) < frac)
            idx = np.nonzero(ok)[0]
            n = len(idx) + 60
            k = np.zeros(n, KP_DTYPE); d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            k["x"] = rng.uniform(5, 507, n); k["y"] = rng.uniform(5, 507, n); k["octave"] = rng.integers(0, 8, n)
            k["angle"] = rng.uniform(0, 360, n); k["size"] = 31; k["class_id"] = -1
            sel = rng.permutation(n)[:len(idx)]
            oc = rng.integers(0, 6, len(idx))
            noise = rng.normal(0, 0.4, (len(idx), 2)) * (1.2 ** oc)[:, None]
            bad = rng.random(len(idx)) < 0.15                      # geometrically inconsistent look-alikes
            noise[bad] += rng.choice([-1, 1], (bad.sum(), 2)) * rng.uniform(8, 25, (bad.sum(), 2))
            k["x"][sel] = uv[idx, 0] + noise[:, 0]; k["y"][sel] = uv[idx, 1] + noise[:, 1]; k["octave"][sel] = oc
            k["angle"][sel] = (37.0 * idx) % 360 + rng.normal(0, 2, len(idx))
            d[sel] = base[idx] ^ np.packbits(rng.random((len(idx), 256)) < 0.04, axis=1)
            parts.append((k, d))
            seen.append((idx, sel, oc))
        feats.append(parts); vis.append(seen)
    nimg = len(feats)
    cap = max(len(p[0][0]) + len(p[1][0]) for p in feats) + 4
    kps = np.zeros((nimg, cap), KP_DTYPE); desc = np.zeros((nimg, cap, 32), np.uint8); cnt = np.zeros(nimg, np.int32); nl = np.zeros(nimg, np.int32)
    for i, ((kl, dl), (kr, dr)) in enumerate(feats):
        a, b = len(kl), len(kr)
        kps[i, :a] = kl; kps[i, a:a + b] = kr; desc[i, :a] = dl; desc[i, a:a + b] = dr; cnt[i] = a + b; nl[i] = a
    has = (rng.random((nimg, cap)) < 0.3).astype(np.uint8)
    return dict(P=P, sf=sf, poses=poses, Trl=Trl_m, kps=kps, desc=desc, cnt=cnt, nl=nl, has=has, cap=cap, nimg=nimg, Xw=Xw, base=base, vis=vis)


def test_search_for_triangulation_fisheye():
    """SearchForTriangulation between two keyframes of the KB8 rig: the four (side, side) camera / relative-pose
    combinations and KannalaBrandt8::epipolarConstrain (TriangulateMatches > 1e-4)."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.synth import TUMVI_CAM_L, TUMVI_CAM_R, make_vocabulary
    S = _tri_scene()
    P, sf, poses, Trl_m, kps, desc, cnt, nl, has, cap, nimg = (S[k] for k in ("P", "sf", "poses", "Trl", "kps", "desc", "cnt", "nl", "has", "cap", "nimg"))
    pairs = [(0, 1), (1, 2), (2, 0)]
    T4 = np.zeros((len(pairs), 4, 12), np.float32)
    for p, (a, b) in enumerate(pairs):
        T1w, T2w = poses[a], poses[b]
        Tr1w, Tr2w = Trl_m @ T1w, Trl_m @ T2w
        Tw2, Twr2 = np.linalg.inv(T2w), np.linalg.inv(Tr2w)
        for c, T in enumerate((T1w @ Tw2, T1w @ Twr2, Tr1w @ Tw2, Tr1w @ Twr2)):
            T4[p, c, :9] = T[:3, :3].reshape(-1); T4[p, c, 9:] = T[:3, 3]
    k, Lv = 5, 3
    vd, vf = make_vocabulary(k, Lv, seed=9)
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dk, dd, dc = cu(kps.view(np.uint8).reshape(nimg, cap, 28)), cu(desc), cu(cnt)
    i1 = np.array([a for a, _ in pairs], np.int32); i2 = np.array([b for _, b in pairs], np.int32)
    sigma2 = (sf * sf).astype(np.float32)
    for coarse, ori in ((False, True), (True, False)):
        m = ORBmatcher(0.6, ori)
        _, node = m.bow_transform(dd, dc, cu(vd), cu(vf), k, Lv, 1)
        m12, nm = m.SearchForTriangulationFisheye(P, cu(i1), cu(i2), cu(nl[i1]), cu(nl[i2]), dk, dd, node, dc, cu(has), TUMVI_CAM_L,
                                                  TUMVI_CAM_R, T4, bOnlyStereo=False, bCoarse=coarse)
        torch.cuda.synchronize()
        nn_, m12, nm = node.cpu().numpy(), m12.cpu().numpy(), nm.cpu().numpy()
        tot = cross = 0
        for p, (a, b) in enumerate(pairs):
            na, nb = cnt[a], cnt[b]
            ne, me = O.search_for_triangulation_fisheye(kps[a, :na], nl[a], desc[a, :na], nn_[a, :na], has[a, :na], kps[b, :nb], nl[b],
                                                        desc[b, :nb], nn_[b, :nb], has[b, :nb], sigma2, TUMVI_CAM_L, TUMVI_CAM_R, T4[p],
                                                        False, coarse, ori)
            np.testing.assert_array_equal(m12[p, :na], me)   # TriangulateMatches: tanf / atan2f restated from glibc on the device
            assert int(nm[p]) == ne
            tot += ne
            ok = me >= 0
            cross += int(((np.arange(na)[ok] < nl[a]) != (me[ok] < nl[b])).sum())
        assert tot > 300 and cross > 30
    # bOnlyStereo: bStereo1 is false on this rig -> nothing matches (:886-887)
    m12, nm = ORBmatcher(0.6, True).SearchForTriangulationFisheye(P, cu(i1), cu(i2), cu(nl[i1]), cu(nl[i2]), dk, dd, node, dc, cu(has),
                                                                  TUMVI_CAM_L, TUMVI_CAM_R, T4, bOnlyStereo=True)
    torch.cuda.synchronize()
    assert int(nm.sum()) == 0 and int((m12 >= 0).sum()) == 0


def _fuse_rig_problem(S, b, right, seed):
    """Map points of the cloud offered to keyframe b's left / right camera: pose, centre, per-point normal / distance range."""
    from morb_slam_amd.synth import _quat_from_R
    rng = np.random.default_rng(seed)
    T = (S["Trl"] @ S["poses"][b]) if right else S["poses"][b]
    T7 = np.concatenate([_quat_from_R(T[:3, :3]), T[:3, 3]]).astype(np.float32)
    Ow = (-(T[:3, :3].T @ T[:3, 3])).astype(np.float32)
    Xw = S["Xw"].astype(np.float32); M = len(Xw)
    PO = Xw - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = PO / dist[:, None] + rng.normal(0, 0.25, (M, 3))
    normal[rng.random(M) < 0.05] *= -1                                   # seen from behind: rejected by the 60 degree test
    lvl = rng.integers(0, 8, M)
    idx, sel, oc = S["vis"][b][1 if right else 0]
    lvl[idx] = np.clip(oc + rng.integers(0, 2, len(idx)), 0, 7)          # PredictScale lands on the feature's octave or one above
    maxD = (dist * 1.2 ** (lvl - rng.uniform(0.05, 0.95, M))).astype(np.float32)   # ceil(log(maxD / dist) / log 1.2) = lvl
    minD = (maxD / 1.2 ** 7 * 0.8).astype(np.float32)
    far = rng.random(M) < 0.04
    minD[far] = (dist[far] * 2).astype(np.float32)                        # outside the scale pyramid
    mpd = S["base"] ^ np.packbits(rng.random((M, 256)) < 0.03, axis=1)
    valid = (rng.random(M) < 0.93).astype(np.uint8)
    return dict(T7=T7, Ow=Ow, Xw=Xw, normal=normal.astype(np.float32), maxD=maxD, minD=minD, mpd=mpd, valid=valid)


def test_fuse_on_a_rig_matches_oracle():
    """Fuse(pKF, vpMapPoints, th, bRight) on a KannalaBrandt8 rig keyframe, both sides (ORBmatcher.cc:1044-1213: the side's pose, camera,
    grid and keypoints; the returned index counts from the start of the left | right row): morb_fuse_batch with cam8 and a feature range."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.synth import TUMVI_CAM_L, TUMVI_CAM_R
    S = _tri_scene()
    P, sf, kps, desc, cnt, nl = (S[k] for k in ("P", "sf", "kps", "desc", "cnt", "nl"))
    invS = (1.0 / (sf * sf)).astype(np.float32)
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    m = ORBmatcher(0.6, True)
    dk, dd, dc = cu(kps.view(np.uint8).reshape(S["nimg"], S["cap"], 28)), cu(desc), cu(cnt)
    tot = 0
    for b in (1, 2):
        N = int(cnt[b])
        Fo = O.make_frame(P, kps[b, :N], desc[b, :N], None)
        for right in (False, True):
            pr = _fuse_rig_problem(S, b, right, 50 + 2 * b + right)
            M = len(pr["Xw"])
            lo, hi = (int(nl[b]), N) if right else (0, int(nl[b]))
            one = lambda a: cu(np.asarray(a)[None])
            for th in (3.0, 6.0):
                bi, bd = m.Fuse(P, cu(np.array([b], np.int32)), dk, dd, dc, None, one(pr["T7"]), one(pr["Ow"]), cu(np.array([M], np.int32)),
                                one(pr["valid"]), one(pr["Xw"]), one(pr["normal"]), one(pr["maxD"]), one(pr["minD"]), one(pr["mpd"]), th=th,
                                cam8=TUMVI_CAM_R if right else TUMVI_CAM_L, jLo=cu(np.array([lo], np.int32)), jHi=cu(np.array([hi], np.int32)))
                torch.cuda.synchronize()
                ei, ed = O.fuse_search_rig(Fo, int(nl[b]), right, TUMVI_CAM_R if right else TUMVI_CAM_L, invS, pr["T7"], pr["Ow"], pr["valid"], pr["Xw"],
                                           pr["normal"], pr["maxD"], pr["minD"], pr["mpd"], th)
                np.testing.assert_array_equal(bi[0].cpu().numpy(), ei)
                np.testing.assert_array_equal(bd[0].cpu().numpy(), ed)
                hit = ei[ei >= 0]
                assert len(hit) > 60 and ((hit >= nl[b]) == right).all()      # the side's features only
                tot += len(hit)
    assert tot > 800


def _rig_sim3_problem(S, a, b, seed):
    """SearchBySim3 between rig keyframes a and b: the projection is the pinhole formula on fx, fy, cx, cy even on a rig (ORBmatcher.cc:1375-1379), so the
    map points are placed where that formula lands on a left feature of the other keyframe (random depths), owned by left AND right features of their own."""
    from morb_slam_amd.synth import TUMVI_CAM_L, _quat_from_R
    kps, desc, cnt, nl, poses = (S[k] for k in ("kps", "desc", "cnt", "nl", "poses"))
    rng = np.random.default_rng(seed)
    Na, NLa, N, NL = int(cnt[a]), int(nl[a]), int(cnt[b]), int(nl[b])
    T = [poses[a], poses[b]]
    T7 = [np.concatenate([_quat_from_R(t[:3, :3]), t[:3, 3]]).astype(np.float32) for t in T]
    s12 = 1.01
    R12 = T[0][:3, :3] @ T[1][:3, :3].T; t12 = T[0][:3, 3] - R12 @ T[1][:3, 3]
    sim8 = lambda R, t, sc: np.concatenate([_quat_from_R(R) * np.sqrt(sc), t]).astype(np.float32)
    S12 = sim8(R12, t12, s12); S21 = sim8(R12.T, -(R12.T @ t12) / s12, 1.0 / s12)
    fx, fy, cx, cy = (float(TUMVI_CAM_L[i]) for i in range(4))

    def points_for(dst, Ssd_R, Ssd_t, Ssd_s, Tsrc, pairs_src, pairs_dst):
        """map points of the source keyframe's features pairs_src[q] that the search finds at the left feature pairs_dst[q] of keyframe `dst`"""
        valid = np.zeros(S["cap"], np.uint8); Pw = np.zeros((S["cap"], 3), np.float32); maxD = np.ones(S["cap"], np.float32); minD = np.ones(S["cap"], np.float32)
        mpd = rng.integers(0, 256, (S["cap"], 32), dtype=np.uint8)
        for i, j in zip(pairs_src, pairs_dst):
            kp = kps[dst, j]
            z = rng.uniform(1.5, 8.0)
            pd = np.array([(kp["x"] + rng.normal(0, 0.5) - cx) / fx * z, (kp["y"] + rng.normal(0, 0.5) - cy) / fy * z, z])      # in dst's camera
            ps = Ssd_s * (Ssd_R @ pd) + Ssd_t                                                                                       # in src's camera
            Pw[i] = Tsrc[:3, :3].T @ (ps - Tsrc[:3, 3])
            dist = np.linalg.norm(pd)
            maxD[i] = dist * 1.2 ** (int(kp["octave"]) + rng.uniform(0.1, 0.9)); minD[i] = maxD[i] / 1.2 ** 8
            mpd[i] = desc[dst, j] ^ np.packbits(rng.random(256) < 0.03)
            valid[i] = 1
        return valid, Pw, maxD, minD, mpd
    nPair = 150
    i1 = rng.permutation(NLa)[:nPair]; j2 = rng.permutation(NL)[:nPair]                     # mutual pairs among the left features
    x1 = NLa + rng.permutation(Na - NLa)[:60]; y2 = np.setdiff1d(np.arange(NL), j2)[:60]   # points owned by RIGHT features of keyframe 1, found in keyframe 2
    k1 = points_for(b, R12, t12, s12, T[0], np.concatenate([i1, x1]), np.concatenate([j2, y2]))
    k2 = points_for(a, R12.T, -(R12.T @ t12) / s12, 1.0 / s12, T[1], j2, i1)
    return dict(T7=T7, S12=S12, S21=S21, k1=k1, k2=k2, x1=x1)


def test_loop_closing_searches_on_a_rig_match_oracle():
    """The matcher members WITHOUT a rig branch in the reference, on KannalaBrandt8 rig keyframes (what LoopClosing runs on a stereo-fisheye
    sequence): Fuse(pKF, Scw, ...) and SearchByProjection(pKF, Scw, ...) project with pKF->mpCamera (the left KB8 camera), the latter's twin and
    SearchBySim3 with the pinhole formula on pKF->fx ..., and all of them look the candidates up among the LEFT features only
    (GetFeaturesInArea's bRight = false) while every map point of the keyframe is projected: ORBmatcher.cc:397-601, :1215-1519."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.synth import TUMVI_CAM_L, _quat_from_R
    S = _tri_scene()
    P, sf, kps, desc, cnt, nl, poses = (S[k] for k in ("P", "sf", "kps", "desc", "cnt", "nl", "poses"))
    invS = (1.0 / (sf * sf)).astype(np.float32)
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    m = ORBmatcher(0.8, True)
    dk, dd, dc = cu(kps.view(np.uint8).reshape(S["nimg"], S["cap"], 28)), cu(desc), cu(cnt)
    rng = np.random.default_rng(_OFF + 808)
    b = 1
    N = int(cnt[b]); NL = int(nl[b])
    Fo = O.make_frame(P, kps[b, :N], desc[b, :N], None)
    pr = _fuse_rig_problem(S, b, False, 61)
    M = len(pr["Xw"])
    one = lambda a: cu(np.asarray(a)[None])
    kfb = cu(np.array([b], np.int32)); nmp = cu(np.array([M], np.int32)); dnl = cu(np.array([NL], np.int32))
    # ---- Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    bi, bd = m.Fuse(P, kfb, dk, dd, dc, None, one(pr["T7"]), one(pr["Ow"]), nmp, one(pr["valid"]), one(pr["Xw"]), one(pr["normal"]), one(pr["maxD"]),
                    one(pr["minD"]), one(pr["mpd"]), th=6.0, sim3Form=True, cam8=TUMVI_CAM_L, jLo=cu(np.array([0], np.int32)), jHi=dnl)
    torch.cuda.synchronize()
    ei, ed = O.fuse_search_rig_sim3(Fo, NL, TUMVI_CAM_L, invS, pr["T7"], pr["Ow"], pr["valid"], pr["Xw"], pr["normal"], pr["maxD"], pr["minD"], pr["mpd"], 6.0)
    np.testing.assert_array_equal(bi[0].cpu().numpy(), ei); np.testing.assert_array_equal(bd[0].cpu().numpy(), ed)
    assert (ei >= 0).sum() > 100 and ei.max() < NL
    # ---- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) and its twin
    matched = (rng.random(N) < 0.1).astype(np.uint8)
    mpad = np.zeros(S["cap"], np.uint8); mpad[:N] = matched
    for manual in (False, True):
        mf, nm = m.SearchByProjectionSim3(P, kfb, dk, dd, dc, one(pr["T7"]), one(pr["Ow"]), nmp, one(pr["valid"]), one(pr["Xw"]), one(pr["normal"]),
                                          one(pr["maxD"]), one(pr["minD"]), one(pr["mpd"]), one(mpad), 8, 0.9, manual, cam8=TUMVI_CAM_L, nLeft=dnl)
        torch.cuda.synchronize()
        r, me = O.search_by_projection_sim3_rig(Fo, NL, TUMVI_CAM_L, pr["T7"], pr["Ow"], pr["valid"], pr["Xw"], pr["normal"], pr["maxD"], pr["minD"], pr["mpd"],
                                                matched, 8, 0.9, manual)
        assert int(nm[0]) == r
        np.testing.assert_array_equal(mf[0, :N].cpu().numpy(), me)
        assert (np.nonzero(me >= 0)[0] < NL).all()
        if not manual:
            assert r > 100
    a = 0
    Na, NLa = int(cnt[a]), int(nl[a])
    Fa = O.make_frame(P, kps[a, :Na], desc[a, :Na], None)
    q = _rig_sim3_problem(S, a, b, 809)
    T7, S12, S21, x1 = q["T7"], q["S12"], q["S21"], q["x1"]
    (v1, Pw1, mx1, mn1, d1), (v2, Pw2, mx2, mn2, d2) = q["k1"], q["k2"]
    o = m.SearchBySim3(P, cu(np.array([a], np.int32)), kfb, dk, dd, dc, one(T7[0]), one(T7[1]), one(S12), one(S21), one(v1), one(Pw1), one(mx1), one(mn1), one(d1),
                       one(v2), one(Pw2), one(mx2), one(mn2), one(d2), 7.5, nLeft1=cu(np.array([NLa], np.int32)), nLeft2=dnl)
    torch.cuda.synchronize()
    g1, g2, g12, nf = [x.cpu().numpy() for x in o]
    e1 = O.search_by_sim3_dir_rig(Fo, NL, T7[0], S21, v1[:Na], Pw1[:Na], mx1[:Na], mn1[:Na], d1[:Na], 7.5)
    e2 = O.search_by_sim3_dir_rig(Fa, NLa, T7[1], S12, v2[:N], Pw2[:N], mx2[:N], mn2[:N], d2[:N], 7.5)
    np.testing.assert_array_equal(g1[0, :Na], e1); np.testing.assert_array_equal(g2[0, :N], e2)
    e12 = np.array([i2 if (i2 >= 0 and e2[i2] == k1) else -1 for k1, i2 in enumerate(e1)])
    np.testing.assert_array_equal(g12[0, :Na], e12)
    assert int(nf[0]) == int((e12 >= 0).sum()) > 100
    assert (e1[x1] >= 0).sum() > 40 and e1.max() < NL and e2.max() < NLa      # right-owned points are searched too; candidates are left features only


def _mul_f32(A, B):
    """(R, t) of A * B composed in float32 left to right, as tests/native/mock_ref's SE3f does (no FMA: products rounded, then summed)."""
    A = np.asarray(A, np.float32); B = np.asarray(B, np.float32)
    Ra, ta, Rb, tb = A[:9].reshape(3, 3), A[9:], B[:9].reshape(3, 3), B[9:]
    R = np.zeros((3, 3), np.float32); t = np.zeros(3, np.float32)
    for r in range(3):
        for c in range(3):
            R[r, c] = np.float32(np.float32(Ra[r, 0] * Rb[0, c]) + np.float32(Ra[r, 1] * Rb[1, c])) + np.float32(Ra[r, 2] * Rb[2, c])
        t[r] = np.float32(np.float32(np.float32(Ra[r, 0] * tb[0]) + np.float32(Ra[r, 1] * tb[1])) + np.float32(Ra[r, 2] * tb[2])) + ta[r]
    return np.concatenate([R.reshape(-1), t]).astype(np.float32)


def test_search_for_triangulation_rig_through_the_reference_member(tmp_path):
    """ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, ...) of include/morb/ORBmatcher.h with KeyFrame::NLeft != -1 / mpCamera2 set
    (tests/native/reference_members_check.cc `rig`, mock KeyFrames carrying this scene): the pairs it returns are the oracle's, with the
    four relative poses built from GetPose / GetPoseInverse / GetRightPose / GetRightPoseInverse as ORBmatcher.cc:836-852 builds them."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.synth import TUMVI_CAM_L, TUMVI_CAM_R, make_vocabulary
    from test_adapter_gpu import _build
    S = _tri_scene()
    kps, desc, cnt, nl, has, poses, Trl_m, sf = (S[k] for k in ("kps", "desc", "cnt", "nl", "has", "poses", "Trl", "sf"))
    a, b = 0, 1
    na, nb = int(cnt[a]), int(cnt[b])
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    k, Lv = 5, 3
    vd, vf = make_vocabulary(k, Lv, seed=9)
    _, node = ORBmatcher(0.6, True).bow_transform(cu(desc), cu(cnt), cu(vd), cu(vf), k, Lv, 1)
    torch.cuda.synchronize()
    nn_ = node.cpu().numpy()
    d = tmp_path / "io"
    d.mkdir()
    put = lambda name, x: np.ascontiguousarray(x).tofile(str(d / (name + ".bin")))
    Pb = np.frombuffer(bytes(S["P"]), np.uint8)
    for tag, i, n in (("rig_1", a, na), ("rig_2", b, nb)):
        put(tag + "_kps", kps[i, :n]); put(tag + "_desc", desc[i, :n]); put(tag + "_params", Pb); put(tag + "_node", nn_[i, :n].astype(np.int32))
        put(tag + "_hasmp", has[i, :n])
    rt = lambda T: np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]).astype(np.float32)
    T1w, T2w = poses[a], poses[b]
    four = [rt(T1w), rt(Trl_m @ T1w), rt(np.linalg.inv(T2w)), rt(np.linalg.inv(Trl_m @ T2w))]   # T1w, Tr1w, Tw2, Twr2
    put("rig_poses", np.concatenate(four))
    put("rig_cams", np.concatenate([TUMVI_CAM_L, TUMVI_CAM_R]).astype(np.float32))
    ori, coarse = True, False
    put("rig_cfg", np.array([0.6, ori, 0, coarse, nl[a], nl[b]], np.float32))
    fuse = {}
    for right, t in ((False, "rigf_l"), (True, "rigf_r")):   # Fuse(pKF = keyframe b, cloud points, 3.0, bRight)
        pr = fuse[right] = _fuse_rig_problem(S, b, right, 90 + right)
        put(t + "_pos", pr["Xw"]); put(t + "_normal", pr["normal"]); put(t + "_maxd", pr["maxD"]); put(t + "_mind", pr["minD"]); put(t + "_desc", pr["mpd"])
        put(t + "_valid", pr["valid"]); put(t + "_pose", np.concatenate([pr["T7"], pr["Ow"]]).astype(np.float32))
    # loop closing on keyframe b: Fuse(pKF, Scw, ...) / SearchByProjection(pKF, Scw, ...) x 2 with Scw = the left camera's pose, and SearchBySim3(a, b)
    lc = _fuse_rig_problem(S, b, False, 95)
    put("rlc_pos", lc["Xw"]); put("rlc_normal", lc["normal"]); put("rlc_maxd", lc["maxD"]); put("rlc_mind", lc["minD"]); put("rlc_desc", lc["mpd"])
    put("rlc_valid", lc["valid"]); put("rlc_sim3", np.concatenate([lc["T7"], lc["Ow"]]).astype(np.float32))
    lc_matched = np.where(np.random.default_rng(96).random(nb) < 0.1, 0, -1).astype(np.int32)
    put("rlc_matched", lc_matched)
    q3 = _rig_sim3_problem(S, a, b, 97)
    from test_adapter_matcher_gpu import _pose22
    for tag, i, n, (v, Pw, mx, mn, dsc), T7 in (("rs3_1", a, na, q3["k1"], q3["T7"][0]), ("rs3_2", b, nb, q3["k2"], q3["T7"][1])):
        put(tag + "_kps", kps[i, :n]); put(tag + "_desc", desc[i, :n]); put(tag + "_params", Pb); put(tag + "_hasmp", v[:n]); put(tag + "_mppos", Pw[:n])
        put(tag + "_mpmax", mx[:n]); put(tag + "_mpmin", mn[:n]); put(tag + "_mpdesc", dsc[:n]); put(tag + "_pose", _pose22(T7))
    put("rs3_cfg", np.concatenate([[7.5], q3["S12"], q3["S21"]]).astype(np.float32))
    out = subprocess.run([_build(tmp_path, "reference_members_check.cc", mock_ref=True), str(d), "rig"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "reference members (rig) ok" in out.stdout, out.stdout + out.stderr
    T4 = np.stack([_mul_f32(four[0], four[2]), _mul_f32(four[0], four[3]), _mul_f32(four[1], four[2]), _mul_f32(four[1], four[3])])
    sigma2 = (sf * sf).astype(np.float32)
    for tag, only in (("rig", False), ("rig_stereo", True)):
        ne, me = O.search_for_triangulation_fisheye(kps[a, :na], nl[a], desc[a, :na], nn_[a, :na], has[a, :na], kps[b, :nb], nl[b], desc[b, :nb],
                                                    nn_[b, :nb], has[b, :nb], sigma2, TUMVI_CAM_L, TUMVI_CAM_R, T4, only, coarse, ori)
        got = np.fromfile(str(d / ("out_ref_" + tag + "_pairs.bin")), np.int32).reshape(-1, 2)
        assert int(np.fromfile(str(d / ("out_ref_" + tag + "_n.bin")), np.int32)[0]) == ne
        exp = np.stack([np.nonzero(me >= 0)[0], me[me >= 0]], 1)
        np.testing.assert_array_equal(got, exp)
        if not only:
            assert ne > 80 and ((exp[:, 0] < nl[a]) != (exp[:, 1] < nl[b])).sum() > 5      # left-right pairs too
        else:
            assert ne == 0
    Fo = O.make_frame(S["P"], kps[b, :nb], desc[b, :nb], None)
    invS = (1.0 / (sf * sf)).astype(np.float32)
    for right, t in ((False, "rigf_l"), (True, "rigf_r")):
        pr = fuse[right]
        ei, _ = O.fuse_search_rig(Fo, int(nl[b]), right, TUMVI_CAM_R if right else TUMVI_CAM_L, invS, pr["T7"], pr["Ow"], pr["valid"], pr["Xw"], pr["normal"],
                                  pr["maxD"], pr["minD"], pr["mpd"], 3.0)
        slots = np.full(nb, -1, np.int32)    # a free feature ends up holding the first point that chose it; one that had a point keeps it (Replace)
        for i in range(len(ei) - 1, -1, -1):
            if ei[i] >= 0 and not has[b, ei[i]]:
                slots[ei[i]] = i
        np.testing.assert_array_equal(np.fromfile(str(d / ("out_ref_" + t + "_slots.bin")), np.int32), slots)
        assert int(np.fromfile(str(d / ("out_ref_" + t + "_n.bin")), np.int32)[0]) == int((ei >= 0).sum()) > 60
    getr = lambda name: np.fromfile(str(d / ("out_ref_" + name + ".bin")), np.int32)
    ei, _ = O.fuse_search_rig_sim3(Fo, int(nl[b]), TUMVI_CAM_L, invS, lc["T7"], lc["Ow"], lc["valid"], lc["Xw"], lc["normal"], lc["maxD"], lc["minD"], lc["mpd"], 6.0)
    slots = np.full(nb, -1, np.int32)
    for i in range(len(ei) - 1, -1, -1):
        if ei[i] >= 0:
            slots[ei[i]] = i
    np.testing.assert_array_equal(getr("rlc_fuse_slots"), slots)
    assert int(getr("rlc_fuse_n")[0]) == int((ei >= 0).sum()) > 100
    for tag, manual in (("rlc_proj", False), ("rlc_projk", True)):
        r, me = O.search_by_projection_sim3_rig(Fo, int(nl[b]), TUMVI_CAM_L, lc["T7"], lc["Ow"], lc["valid"], lc["Xw"], lc["normal"], lc["maxD"], lc["minD"], lc["mpd"],
                                                (lc_matched >= 0).astype(np.uint8), 8, 0.9, manual)
        assert int(getr(tag + "_n")[0]) == r and (manual or r > 100)
        np.testing.assert_array_equal(getr(tag + "_match"), np.where(lc_matched >= 0, -2, me))
    Fa = O.make_frame(S["P"], kps[a, :na], desc[a, :na], None)
    (v1, Pw1, mx1, mn1, d1), (v2, Pw2, mx2, mn2, d2) = q3["k1"], q3["k2"]
    e1 = O.search_by_sim3_dir_rig(Fo, int(nl[b]), q3["T7"][0], q3["S21"], v1[:na], Pw1[:na], mx1[:na], mn1[:na], d1[:na], 7.5)
    e2 = O.search_by_sim3_dir_rig(Fa, int(nl[a]), q3["T7"][1], q3["S12"], v2[:nb], Pw2[:nb], mx2[:nb], mn2[:nb], d2[:nb], 7.5)
    e12 = np.array([i2 if (i2 >= 0 and e2[i2] == k1) else -1 for k1, i2 in enumerate(e1)])
    np.testing.assert_array_equal(getr("rs3_match"), e12)
    assert int(getr("rs3_n")[0]) == int((e12 >= 0).sum()) > 100


def test_c3_chain_on_extracted_features():
    """BASELINE config 3 as bench.py runs it, on EXTRACTED features: 512x512 / 1500-feature extraction with lapping areas ->
    ComputeStereoFishEyeMatches -> PoseOptimization on the TUM-VI rig, every stage against the oracle on the same inputs."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBextractor, ORBmatcher, Optimizer
    from morb_slam_amd.synth import TUMVI_CAM_L, TUMVI_CAM_R, TUMVI_T_C1_C2, make_stereo_pair
    left, right = make_stereo_pair(512, 512, seed=104, dmin=1.0, dmax=12.0)
    lap = np.array([[0, 511], [0, 511]], np.int32)
    ext = ORBextractor(1500, 1.2, 8, 20, 7)
    kps, desc, cnt, mono = ext.extract_batch(torch.from_numpy(np.stack([left, right])).cuda(), lap=lap)
    sigma2 = np.asarray(ext.GetScaleSigmaSquares(), np.float32)
    Rlr = TUMVI_T_C1_C2[:3, :3].astype(np.float32); tlr = TUMVI_T_C1_C2[:3, 3].astype(np.float32)
    m = ORBmatcher()
    o = m.ComputeStereoFishEyeMatches(kps, desc, cnt, mono, TUMVI_CAM_L, TUMVI_CAM_R, Rlr, tlr, sigma2)
    torch.cuda.synchronize()
    # the oracle's extraction of the same images: byte-identical features, so both matchers see the same input
    oL, oR = O.OracleExtractor(1500), O.OracleExtractor(1500)
    mL, kL, dL = oL(left, (0, 511)); mR, kR, dR = oR(right, (0, 511))
    c = cnt.cpu().numpy()
    assert kps[0, :c[0]].cpu().numpy().reshape(-1).view(KP_DTYPE).tobytes() == kL.tobytes() and int(mono[0]) == mL
    assert kps[1, :c[1]].cpu().numpy().reshape(-1).view(KP_DTYPE).tobytes() == kR.tobytes() and int(mono[1]) == mR
    fe = dict(kL=kL, dL=dL, monoL=mL, kR=kR, dR=dR, monoR=mR, Rlr=Rlr, tlr=tlr, camL=TUMVI_CAM_L, camR=TUMVI_CAM_R)
    n, l2r, r2l, dep, p3 = O.stereo_fisheye_matches(fe, sigma2)
    nl, nr = len(kL), len(kR)
    assert int(o["nMatches"][0]) == n
    np.testing.assert_array_equal(o["leftToRight"][0, :nl].cpu().numpy(), l2r)
    np.testing.assert_array_equal(o["rightToLeft"][0, :nr].cpu().numpy(), r2l)
    np.testing.assert_allclose(o["depth"][0, :nl].cpu().numpy(), dep, rtol=1e-4, atol=1e-5)
    # PoseOptimization on the rig from these very features: map points = points in front of the extracted left / right keypoints (a
    # KB8 ray through the keypoint at a random depth), seen from a slightly wrong initial pose
    rng = np.random.default_rng(_OFF + 9)
    Trl_m = np.linalg.inv(TUMVI_T_C1_C2)

    def rays(cam, k):   # unproject by bisection on theta (test-side helper; the product's unproject is KannalaBrandt8::unproject)
        x = (k["x"] - cam[2]) / cam[0]; y = (k["y"] - cam[3]) / cam[1]
        rd = np.sqrt(x * x + y * y)
        lo, hi = np.zeros_like(rd), np.full_like(rd, np.pi / 2)
        for _ in range(50):
            th = (lo + hi) / 2
            f = th * (1 + cam[4] * th ** 2 + cam[5] * th ** 4 + cam[6] * th ** 6 + cam[7] * th ** 8)
            lo = np.where(f < rd, th, lo); hi = np.where(f < rd, hi, th)
        th = (lo + hi) / 2
        s = np.where(rd > 1e-8, np.tan(th) / np.maximum(rd, 1e-8), 1.0)
        return np.stack([x * s, y * s, np.ones_like(x)], 1)
    selL = rng.choice(nl, 400, replace=False); selR = rng.choice(nr, 300, replace=False)
    XL = rays(TUMVI_CAM_L, kL[selL]) * rng.uniform(1, 8, (400, 1))                                      # left-camera frame
    XRr = rays(TUMVI_CAM_R, kR[selR]) * rng.uniform(1, 8, (300, 1))                                      # right-camera frame
    XR = (XRr - Trl_m[:3, 3]) @ Trl_m[:3, :3]                                                            # -> left-camera frame (= world: true pose is identity)
    Xw = np.concatenate([XL, XR]).astype(np.float32)
    obs = np.concatenate([np.stack([kL[selL]["x"], kL[selL]["y"], np.full(400, -1.0)], 1), np.stack([kR[selR]["x"], kR[selR]["y"], np.full(300, -1.0)], 1)]).astype(np.float32)
    obs[rng.random(700) < 0.1, :2] += rng.uniform(15, 40, 2)                                            # gross outliers
    octv = np.concatenate([kL[selL]["octave"], kR[selR]["octave"]])
    q = np.array([0.004, -0.006, 0.003, 1.0]); q /= np.linalg.norm(q)
    from scipy.spatial.transform import Rotation
    Trl_q = np.concatenate([Rotation.from_matrix(Trl_m[:3, :3]).as_quat(), Trl_m[:3, 3]]).astype(np.float32)
    p = dict(hasMP=(rng.random(700) < 0.95).astype(np.uint8), obs=obs, invSigma2=(1.0 / sigma2[octv]).astype(np.float32), Xw=Xw,
             pose0=np.concatenate([q, [0.02, -0.015, 0.03]]).astype(np.float32), Nleft=400, camL=np.asarray(TUMVI_CAM_L, np.float32),
             camR=np.asarray(TUMVI_CAM_R, np.float32), Trl=Trl_q)
    dev = "cuda"
    t = [torch.from_numpy(p[k][None].copy()).to(dev) for k in ("hasMP", "obs", "invSigma2", "Xw", "pose0")]
    nin, outl, stats = Optimizer().PoseOptimizationFisheye(t[0], t[1], t[2], t[3], t[4], torch.tensor([400], dtype=torch.int32, device=dev),
                                                           torch.tensor([700], dtype=torch.int32, device=dev), p["camL"], p["camR"], p["Trl"])
    torch.cuda.synchronize()
    r, pe, oe, se = O.pose_optimization_fisheye(p)
    assert int(nin[0]) == r and r > 400
    assert np.abs(t[4][0].cpu().numpy() - pe).max() <= 1e-4
    np.testing.assert_array_equal(outl[0, :700].cpu().numpy(), oe)
    assert np.abs(pe[4:]).max() < 0.02 and np.abs(pe[:3]).max() < 0.01                                  # converged back to the identity


def test_local_ba_fisheye_oneshot_equals_the_three_step_form():
    from morb_slam_amd import Optimizer
    from morb_slam_amd.optimizer import local_bundle_adjustment_fisheye_oneshot
    from morb_slam_amd.synth import make_ba_problem_fisheye
    b = make_ba_problem_fisheye(seed=1)
    rig = dict(eRight=b["eRight"], camL=b["camL"], camR=b["camR"], Trl=b["Trl"])
    opt = Optimizer()
    ref = opt.LocalBundleAdjustment(b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], None, rig=rig)
    one = local_bundle_adjustment_fisheye_oneshot(opt, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eRight"],
                                                  b["eInvSigma2"], b["camL"], b["camR"], b["Trl"])
    for x, y in zip(ref, one):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


def test_search_by_projection_keyframe_with_a_rig_current_frame():
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (relocalisation, ORBmatcher.cc:1735-1842) when CurrentFrame is a
    KannalaBrandt8 rig frame: left-camera projection, left features only, a keyframe whose map points sit on left AND right features
    (rotation check with the feature's own keypoint, include/morb_hip.h)."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBmatcher
    from morb_slam_amd.synth import TUMVI_CAM_L, kb8_project, _quat_from_rotvec, _quat_rot
    P, sf = _fisheye_params()
    rng = np.random.default_rng(_OFF + 777)
    Fn, M = 2, 600
    cap = 900
    nimg = 2 * Fn     # image 2 f = current rig frame (left | right), 2 f + 1 = the keyframe's features (left | right in one row)
    kps = np.zeros((nimg, cap), KP_DTYPE); desc = rng.integers(0, 256, (nimg, cap, 32), dtype=np.uint8); cnt = np.zeros(nimg, np.int32)
    nl = np.zeros(Fn, np.int32); Tcw = np.zeros((Fn, 7), np.float32); Ow = np.zeros((Fn, 3), np.float32)
    valid = np.zeros((Fn, cap), np.uint8); Xw = np.zeros((Fn, cap, 3), np.float32); mpd = np.zeros((Fn, cap, 32), np.uint8)
    maxD = np.ones((Fn, cap), np.float32); minD = np.ones((Fn, cap), np.float32); has = np.zeros((Fn, cap), np.uint8)
    for f in range(Fn):
        X = np.stack([rng.uniform(-3, 3, M), rng.uniform(-2.5, 2.5, M), rng.uniform(1, 8, M)], 1)
        T = np.concatenate([_quat_from_rotvec(rng.normal(0, 0.02, 3)), rng.normal(0, 0.05, 3)])
        Xc = np.array([_quat_rot(T[:4], x) + T[4:] for x in X])
        uv = kb8_project(TUMVI_CAM_L, Xc)
        octv = rng.integers(0, 8, M); ang = rng.uniform(0, 360, M)
        ok = (uv > 8).all(1) & (uv < 504).all(1) & (rng.random(M) < 0.8)
        idx = np.nonzero(ok)[0]
        nL, nR = len(idx) + 120, 200
        k = np.zeros(nL + nR, KP_DTYPE)
        k["x"] = rng.uniform(5, 507, nL + nR); k["y"] = rng.uniform(5, 507, nL + nR); k["octave"] = rng.integers(0, 8, nL + nR)
        k["angle"] = rng.uniform(0, 360, nL + nR); k["size"] = 31; k["class_id"] = -1
        sel = rng.permutation(nL)[:len(idx)]            # the observed points land on LEFT features
        k["x"][sel] = uv[idx, 0] + rng.normal(0, 1.0, len(idx)); k["y"][sel] = uv[idx, 1] + rng.normal(0, 1.0, len(idx))
        k["octave"][sel] = np.clip(octv[idx] + rng.integers(-1, 2, len(idx)), 0, 7)
        k["angle"][sel] = (ang[idx] + rng.choice([0.0, 0.0, 0.0, 90.0], len(idx)) + rng.normal(0, 3, len(idx))) % 360
        pd = rng.integers(0, 256, (M, 32), dtype=np.uint8)
        desc[2 * f, sel] = pd[idx] ^ np.packbits(rng.random((len(idx), 256)) < 0.06, axis=1)
        kps[2 * f, :nL + nR] = k; cnt[2 * f] = nL + nR; nl[f] = nL
        kk = np.zeros(M, KP_DTYPE); kk["octave"] = octv; kk["angle"] = ang; kk["size"] = 31     # keyframe feature i holds map point i
        kps[2 * f + 1, :M] = kk; cnt[2 * f + 1] = M
        Tcw[f] = T.astype(np.float32)
        qc = T[:4] * np.array([-1, -1, -1, 1]); Ow[f] = _quat_rot(qc, -T[4:]).astype(np.float32)
        valid[f, :M] = (rng.random(M) < 0.9); Xw[f, :M] = X.astype(np.float32); mpd[f, :M] = pd
        d3 = np.linalg.norm(X - Ow[f], axis=1)
        maxD[f, :M] = (d3 * 1.2 ** octv * rng.uniform(0.95, 1.15, M)).astype(np.float32); minD[f, :M] = (maxD[f, :M] / 1.2 ** 7)
        has[f, :nL + nR] = rng.random(nL + nR) < 0.1
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dk = cu(kps.view(np.uint8).reshape(nimg, cap, 28))
    cur = np.arange(0, nimg, 2, dtype=np.int32); kf = cur + 1
    tot = 0
    for th, orb, ori in ((10.0, 100, True), (3.0, 64, False)):
        m = ORBmatcher(0.9, ori)
        mc, nm = m.SearchByProjectionKeyFrame(P, cu(cur), cu(kf), dk, cu(desc), cu(cnt), cu(has), cu(Tcw), cu(Ow), cu(valid), cu(Xw), cu(maxD),
                                              cu(minD), cu(mpd), th, orb, cam8=TUMVI_CAM_L, nLeftCur=cu(nl))
        torch.cuda.synchronize()
        mc, nm = mc.cpu().numpy(), nm.cpu().numpy()
        for f in range(Fn):
            nL = int(nl[f]); N = int(cnt[2 * f])
            Fo = O.make_frame(P, kps[2 * f, :nL], desc[2 * f, :nL], None)
            r, me = O.search_by_projection_kf_rig(Fo, TUMVI_CAM_L, has[f, :nL], Tcw[f], Ow[f], kps[2 * f + 1, :M], valid[f, :M], Xw[f, :M],
                                                  maxD[f, :M], minD[f, :M], mpd[f, :M], th, orb, ori)
            assert int(nm[f]) == r, (th, f, int(nm[f]), r)
            np.testing.assert_array_equal(mc[f, :nL], me)
            assert (mc[f, nL:N] == -1).all()          # right features are never searched
            tot += r
    assert tot > 400
