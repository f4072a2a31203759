"""GPU parity of the Hamming matchers (through the C ABI) against the CPU oracle.  Integer outputs (distances,
indices, match tables) bit-exact; mvuRight / mvDepth compared as float32 bit patterns."""
import numpy as np
import pytest

import oracle_lib as O
from morb_slam_amd.synth import make_stereo_pair, make_vocabulary, shift_image

pytestmark = pytest.mark.gpu

MBF, MB = np.float32(458.654 * 0.11), np.float32(0.11)   # EuRoC: bf = fx * baseline, b


@pytest.fixture(scope="module")
def batch():
    return make_batch()


def make_batch():
    """4 stereo frames extracted on the GPU (bit-exact with the oracle per test_extractor_gpu) + oracle twins."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBextractor
    import os
    off = 10 * int(os.environ.get("MORB_TEST_SEED", "0"))   # (tools/stress_matchers.sh: the whole file again on other images)
    pairs = [make_stereo_pair(752, 480, seed=60 + off + i) for i in range(2)]
    pairs += [tuple(shift_image(im, 4, 2) for im in pairs[0]), tuple(shift_image(im, 7, -3) for im in pairs[1])]
    imgs = np.stack([im for p in pairs for im in p])
    ext = ORBextractor(1200, 1.2, 8, 20, 7)
    d = torch.from_numpy(imgs).cuda()
    kps, desc, cnt, mono = ext.extract_batch(d)
    torch.cuda.synchronize()
    ora = []
    for im in imgs:
        o = O.OracleExtractor(1200)
        _, k, dd = o(im)
        ora.append((o, k, dd))
    c = cnt.cpu().numpy()
    for i in range(len(imgs)):
        assert kps[i, :c[i]].cpu().numpy().reshape(-1).view(KP_DTYPE).tobytes() == ora[i][1].tobytes()
    return dict(ext=ext, kps=kps, desc=desc, cnt=cnt, ora=ora, imgs=imgs, KP=KP_DTYPE)


def test_descriptor_distance(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    m = ORBmatcher()
    a = batch["desc"][0, :1000].contiguous(); b = batch["desc"][1, :1000].contiguous()
    got = m.DescriptorDistance(a, b).cpu().numpy()
    an, bn = a.cpu().numpy(), b.cpu().numpy()
    L = O.lib()
    exp = [L.orc_descriptor_distance(O._p(an[i]), O._p(bn[i])) for i in range(1000)]
    np.testing.assert_array_equal(got, exp)
    assert m.DescriptorDistance(a, a).cpu().numpy().max() == 0
    ones = torch.full((3, 32), 255, dtype=torch.uint8, device="cuda"); zeros = torch.zeros_like(ones)
    assert m.DescriptorDistance(ones, zeros).cpu().tolist() == [256, 256, 256]


def test_stereo_matches_bit_exact(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    m = ORBmatcher()
    u, d = m.ComputeStereoMatches(batch["ext"], batch["kps"], batch["desc"], batch["cnt"], MBF, MB)
    torch.cuda.synchronize()
    u, d = u.cpu().numpy(), d.cpu().numpy()
    nmatched = 0
    for f in range(4):
        (ol, kl, dl), (orr, kr, dr) = batch["ora"][2 * f], batch["ora"][2 * f + 1]
        ue, de = O.stereo_matches(ol, orr, kl, dl, kr, dr, MBF, MB)
        n = len(kl)
        assert u[f, :n].view(np.uint32).tolist() == ue.view(np.uint32).tolist()
        assert d[f, :n].view(np.uint32).tolist() == de.view(np.uint32).tolist()
        nmatched += int((ue >= 0).sum())
    assert nmatched > 4 * 300   # the synthetic pairs really produce stereo matches


@pytest.mark.parametrize("nframes", [1, 9, 40])
def test_stereo_matches_every_workgroup_shape(batch, nframes):
    # k_stereo_match sizes its workgroups by the batch (4 / 8 / 16 / 32 left keypoints = 1 / 2 / 4 / 8 per wave, sm_lk_for()): the fixture's
    # 4 frames run the 8-keypoint shape; 1, 9 and 40 frames (the fixture's frames, repeated) run the others.  Every copy of a frame must give
    # that frame's oracle result.
    import torch
    from morb_slam_amd import ORBextractor, ORBmatcher
    imgs = np.stack([batch["imgs"][2 * (f % 4) + k] for f in range(nframes) for k in (0, 1)])
    ext = ORBextractor(1200, 1.2, 8, 20, 7)
    kps, desc, cnt, _ = ext.extract_batch(torch.from_numpy(imgs).cuda())
    u, d = ORBmatcher().ComputeStereoMatches(ext, kps, desc, cnt, MBF, MB)
    torch.cuda.synchronize()
    u, d = u.cpu().numpy(), d.cpu().numpy()
    exp = {}
    for f in range(nframes):
        b = f % 4
        if b not in exp:
            (ol, kl, dl), (orr, kr, dr) = batch["ora"][2 * b], batch["ora"][2 * b + 1]
            exp[b] = O.stereo_matches(ol, orr, kl, dl, kr, dr, MBF, MB) + (len(kl),)
        ue, de, n = exp[b]
        assert u[f, :n].view(np.uint32).tolist() == ue.view(np.uint32).tolist(), f"frame {f}"
        assert d[f, :n].view(np.uint32).tolist() == de.view(np.uint32).tolist(), f"frame {f}"


def test_knn2_ratio(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    m = ORBmatcher()
    desc, cnt = batch["desc"], batch["cnt"]
    q = desc[0::2].contiguous(); t = desc[1::2].contiguous()
    nq = cnt[0::2].contiguous(); nt = cnt[1::2].contiguous()
    qoff = torch.tensor([0, 100, 5, 1200], dtype=torch.int32, device="cuda")
    toff = torch.tensor([0, 7, 300, 0], dtype=torch.int32, device="cuda")
    idx, dist = m.knn2(q, nq, t, nt, qoff, toff)
    torch.cuda.synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    qn, tn, nqn, ntn = q.cpu().numpy(), t.cpu().numpy(), nq.cpu().numpy(), nt.cpu().numpy()
    for p in range(4):
        qo, to = int(qoff[p]), int(toff[p])
        ie, de = O.knn2(qn[p, qo:nqn[p]], tn[p, to:ntn[p]])
        n = max(nqn[p] - qo, 0)
        np.testing.assert_array_equal(idx[p, :n], ie)
        np.testing.assert_array_equal(dist[p, :n], de)
    # degenerate train sets: one row (no second neighbour), zero rows
    one = torch.tensor([1, 1, 0, 0], dtype=torch.int32, device="cuda")
    idx, dist = m.knn2(q, nq, t, one)
    idx = idx.cpu().numpy()
    assert (idx[0, :nqn[0], 0] == 0).all() and (idx[0, :nqn[0], 1] == -1).all() and (idx[2, :nqn[2]] == -1).all()


@pytest.mark.parametrize("cap2", [96, 640, 1088, 2304, 4500])
def test_search_by_bow_at_other_frame_capacities(batch, cap2):
    """The per-frame (node, index) sort behind SearchByBoW is a bitonic network over the next power of two of the frame capacity: one thread per
    compare-exchange up to 2048 keys (steps inside a wave's 128 keys skip the workgroup barrier), a loop over the pairs beyond.  The same frames
    cut down / padded to other capacities: 128, 1024, 2048, 4096, 8192 keys."""
    import torch
    from morb_slam_amd import ORBmatcher
    k, Lv, lup = 10, 3, 1
    vd, vf = make_vocabulary(k, Lv, seed=2)
    dvd, dvf = torch.from_numpy(vd).cuda(), torch.from_numpy(vf).cuda()
    nimg, cap = batch["desc"].shape[0], batch["desc"].shape[1]
    keep = min(cap, cap2)
    cnt = torch.clamp(batch["cnt"], max=keep).contiguous()
    desc = torch.zeros((nimg, cap2, 32), dtype=torch.uint8, device="cuda"); desc[:, :keep] = batch["desc"][:, :keep]
    kps = torch.zeros((nimg, cap2) + tuple(batch["kps"].shape[2:]), dtype=batch["kps"].dtype, device="cuda"); kps[:, :keep] = batch["kps"][:, :keep]
    m = ORBmatcher(0.7, True)
    word, node = m.bow_transform(desc, cnt, dvd, dvf, k, Lv, lup)
    rng = np.random.default_rng(7)
    has = (rng.random((nimg, cap2)) < 0.8).astype(np.uint8)
    kf = torch.tensor([0, 2, 4], dtype=torch.int32, device="cuda")
    fr = torch.tensor([4, 6, 0], dtype=torch.int32, device="cuda")
    match, nm = m.SearchByBoW(kf, fr, kps, desc, node, cnt, torch.from_numpy(has).cuda())
    torch.cuda.synchronize()
    match, nm, nn_, cn = match.cpu().numpy(), nm.cpu().numpy(), node.cpu().numpy(), cnt.cpu().numpy()
    tot = 0
    for p, (a, b) in enumerate(zip(kf.cpu().tolist(), fr.cpu().tolist())):
        na, nb = int(cn[a]), int(cn[b])
        ka, da = batch["ora"][a][1][:na], batch["ora"][a][2][:na]
        kb, db = batch["ora"][b][1][:nb], batch["ora"][b][2][:nb]
        ne, me = O.search_by_bow(da, ka["angle"], has[a, :na], nn_[a, :na], db, kb["angle"], nn_[b, :nb], 0.7, True)
        assert nm[p] == ne
        np.testing.assert_array_equal(match[p, :nb], me)
        tot += ne
    assert tot > (10 if cap2 < 200 else 200)


# vocabulary shapes: ~12 features per node (one register slot per lane), ~75 per node (two slots per lane, keyframe
# features in two chunks), ~400 per node (the general path of k_bow_match)
@pytest.mark.parametrize("k,Lv,lup,settings", [(10, 3, 1, ((0.7, True), (0.9, False), (0.6, True))),
                                               (4, 3, 1, ((0.7, True),)), (3, 2, 1, ((0.8, True),)),
                                               # the ORBvoc shape bench.py runs: six levels, node ids four levels above the words
                                               (10, 6, 4, ((0.7, True), (0.9, False)))])
def test_bow_transform_and_search_by_bow(batch, k, Lv, lup, settings):
    import torch
    from morb_slam_amd import ORBmatcher
    vd, vf = make_vocabulary(k, Lv, seed=2)
    dvd, dvf = torch.from_numpy(vd).cuda(), torch.from_numpy(vf).cuda()
    desc, cnt, kps = batch["desc"], batch["cnt"], batch["kps"]
    for ratio, ori in settings:
        m = ORBmatcher(ratio, ori)
        word, node = m.bow_transform(desc, cnt, dvd, dvf, k, Lv, lup)
        torch.cuda.synchronize()
        wn, nn_, cn = word.cpu().numpy(), node.cpu().numpy(), cnt.cpu().numpy()
        for i in range(desc.shape[0]):
            we, ne = O.bow_transform(batch["ora"][i][2], vd, vf, k, Lv, lup)
            np.testing.assert_array_equal(wn[i, :cn[i]], we)
            np.testing.assert_array_equal(nn_[i, :cn[i]], ne)
        rng = np.random.default_rng(5)
        has = (rng.random((desc.shape[0], desc.shape[1])) < 0.8).astype(np.uint8)
        # pairs: frame 2 (shifted copy of frame 0) against keyframe 0, etc.; also a self pair
        kf = torch.tensor([0, 2, 4, 1, 0], dtype=torch.int32, device="cuda")
        fr = torch.tensor([4, 6, 0, 5, 0], dtype=torch.int32, device="cuda")
        match, nm = m.SearchByBoW(kf, fr, kps, desc, node, cnt, torch.from_numpy(has).cuda())
        torch.cuda.synchronize()
        match, nm = match.cpu().numpy(), nm.cpu().numpy()
        tot = 0
        for p, (a, b) in enumerate(zip(kf.cpu().tolist(), fr.cpu().tolist())):
            ka, da = batch["ora"][a][1], batch["ora"][a][2]
            kb, db = batch["ora"][b][1], batch["ora"][b][2]
            ne, me = O.search_by_bow(da, ka["angle"], has[a, :len(ka)], nn_[a, :len(ka)], db, kb["angle"], nn_[b, :len(kb)],
                                     ratio, ori)
            assert nm[p] == ne
            np.testing.assert_array_equal(match[p, :len(kb)], me)
            tot += ne
        assert tot > 300


@pytest.mark.parametrize("k,Lv,lup", [(10, 3, 1), (4, 3, 1), (3, 2, 1), (10, 6, 4)])
def test_search_by_bow_keyframes(batch, k, Lv, lup):
    """SearchByBoW(pKF1, pKF2, vpMatches12) (:702-819): only features with a MapPoint on both sides, strict TH_LOW,
    vbMatched2, table indexed by the pKF1 feature; mvKeysUn.size() limit of fisheye keyframes."""
    import torch
    from morb_slam_amd import ORBmatcher
    vd, vf = make_vocabulary(k, Lv, seed=4)
    desc, cnt, kps = batch["desc"], batch["cnt"], batch["kps"]
    nimg, cap = desc.shape[0], desc.shape[1]
    rng = np.random.default_rng(9)
    has = (rng.random((nimg, cap)) < 0.7).astype(np.uint8)
    cn = cnt.cpu().numpy()
    nvalid = cn.copy(); nvalid[0] = cn[0] * 2 // 3; nvalid[4] = cn[4] // 2          # two "fisheye" keyframes
    kf1 = torch.tensor([0, 2, 4, 1, 0], dtype=torch.int32, device="cuda")
    kf2 = torch.tensor([4, 6, 0, 5, 0], dtype=torch.int32, device="cuda")
    for ratio, ori in ((0.8, True), (0.6, False)):
        m = ORBmatcher(ratio, ori)
        _, node = m.bow_transform(desc, cnt, torch.from_numpy(vd).cuda(), torch.from_numpy(vf).cuda(), k, Lv, lup)
        m12, nm = m.SearchByBoWKeyFrames(kf1, kf2, kps, desc, node, cnt, torch.from_numpy(has).cuda(),
                                         nValid=torch.from_numpy(nvalid.astype(np.int32)).cuda())
        torch.cuda.synchronize()
        nn_, m12, nm = node.cpu().numpy(), m12.cpu().numpy(), nm.cpu().numpy()
        tot = 0
        for p, (a, b) in enumerate(zip(kf1.cpu().tolist(), kf2.cpu().tolist())):
            ka, da = batch["ora"][a][1], batch["ora"][a][2]
            kb, db = batch["ora"][b][1], batch["ora"][b][2]
            ne, me = O.search_by_bow_kfkf(da, ka["angle"], has[a, :len(ka)], nn_[a, :len(ka)], nvalid[a], db, kb["angle"], has[b, :len(kb)],
                                          nn_[b, :len(kb)], nvalid[b], ratio, ori)
            assert nm[p] == ne
            np.testing.assert_array_equal(m12[p, :len(ka)], me)
            tot += ne
        assert tot > 150


# ---- projection-guided searches --------------------------------------------------------------------------
def _scene(batch):
    """Map points = stereo-triangulated features of frame 0 (identity pose), observed again in frame 2 (a shifted copy)."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.capi import make_frame_params
    ext = batch["ext"]
    P = make_frame_params(752, 480, 458.654, 457.296, 367.215, 248.375, float(MBF), float(MB), ext.GetScaleFactors(),
                          ext.GetScaleSigmaSquares())
    m = ORBmatcher(0.8, True)
    u, d = m.ComputeStereoMatches(ext, batch["kps"], batch["desc"], batch["cnt"], MBF, MB)
    torch.cuda.synchronize()
    return P, u, d


def test_is_in_frustum_and_search_by_projection_mappoints(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    P, uR, dep = _scene(batch)
    k0, d0 = batch["ora"][0][1], batch["ora"][0][2]
    z = dep[0, :len(k0)].cpu().numpy()
    valid = z > 0
    Xw = np.stack([(k0["x"] - P.cx) * z / P.fx, (k0["y"] - P.cy) * z / P.fy, z], 1).astype(np.float32)[valid]
    mpDesc = d0[valid]
    n = len(Xw)
    rng = np.random.default_rng(3)
    normal = (Xw / np.linalg.norm(Xw, axis=1, keepdims=True) + rng.normal(0, 0.2, Xw.shape)).astype(np.float32)  # mean viewing direction camera -> point
    dist = np.linalg.norm(Xw, axis=1).astype(np.float32)
    lvl = k0["octave"][valid]
    maxD = (dist * 1.2 ** lvl * rng.uniform(0.9, 1.3, n)).astype(np.float32)   # mfMaxDistance ~ dist * scale^level
    minD = (maxD / 1.2 ** 7).astype(np.float32)
    # two frames looking at the same points: identity, and a small translation + rotation about y
    th_ = 0.01
    R1 = np.array([[np.cos(th_), 0, np.sin(th_)], [0, 1, 0], [-np.sin(th_), 0, np.cos(th_)]], np.float32)
    Rs = np.stack([np.eye(3, dtype=np.float32), R1]); ts = np.array([[0, 0, 0], [0.02, -0.01, 0.05]], np.float32)
    Ows = np.stack([-(Rs[i].T @ ts[i]) for i in range(2)]).astype(np.float32)
    m = ORBmatcher(0.8, True)
    dev = "cuda"
    rep = lambda a: torch.from_numpy(np.stack([a, a])).to(dev)
    trk = m.isInFrustum(P, torch.from_numpy(Rs.reshape(2, 9)).to(dev), torch.from_numpy(ts).to(dev), torch.from_numpy(Ows).to(dev),
                        torch.tensor([n, n], dtype=torch.int32, device=dev), rep(Xw), rep(normal), rep(maxD), rep(minD), 0.5)
    torch.cuda.synchronize()
    isBad = (rng.random(n) < 0.05).astype(np.uint8); hasObs = (rng.random(n) < 0.9).astype(np.uint8)
    fImg = torch.tensor([4, 4], dtype=torch.int32, device=dev)       # current frame = image 4 (frame 2, left)
    k4, d4 = batch["ora"][4][1], batch["ora"][4][2]
    ur4 = uR[2, :len(k4)].cpu().numpy()
    blocked = (rng.random(len(k4)) < 0.1).astype(np.uint8)
    cap = batch["kps"].shape[1]
    pad = lambda a, fill=0: np.concatenate([a, np.full((cap - len(a),) + a.shape[1:], fill, a.dtype)])
    uRt = torch.from_numpy(np.stack([pad(ur4, -1), pad(ur4, -1)])).to(dev)
    blk = torch.from_numpy(np.stack([pad(blocked), pad(blocked)])).to(dev)
    tot = 0
    for th, bFar, thFar in ((1.0, False, 0.0), (3.0, True, 6.0), (5.0, False, 0.0)):
        mt, nm = m.SearchByProjectionMapPoints(P, fImg, batch["kps"], batch["desc"], batch["cnt"], uRt, blk,
                                               torch.tensor([n, n], dtype=torch.int32, device=dev), trk, rep(isBad), rep(mpDesc),
                                               rep(hasObs), th, bFar, thFar)
        torch.cuda.synchronize()
        mt, nm = mt.cpu().numpy(), nm.cpu().numpy()
        Fo = O.make_frame(P, k4, d4, ur4)
        for f in range(2):
            te = O.is_in_frustum(Fo, Rs[f], ts[f], Ows[f], Xw, normal, maxD, minD, 0.5)
            for key in te:
                g = trk[key][f, :n].cpu().numpy()
                if te[key].dtype == np.float32:
                    assert g.view(np.uint32).tolist() == te[key].view(np.uint32).tolist(), key
                else:
                    np.testing.assert_array_equal(g, te[key], err_msg=key)
            r, me = O.search_by_projection_mps(Fo, blocked, te, isBad, mpDesc, hasObs, th, bFar, thFar, 0.8)
            assert nm[f] == r
            np.testing.assert_array_equal(mt[f, :len(k4)], me)
            tot += r
        assert te["inView"].sum() > 0.5 * n
    assert tot > 600


def test_search_by_projection_last_frame(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    P, uR, dep = _scene(batch)
    dev = "cuda"
    cap = batch["kps"].shape[1]
    rng = np.random.default_rng(4)
    pad = lambda a, fill=0: np.concatenate([a, np.full((cap - len(a),) + a.shape[1:], fill, a.dtype)])
    # pairs (last image, current image): (0 -> 4), (2 -> 6), (4 -> 0)
    pairs = [(0, 4), (2, 6), (4, 0)]
    lastValid, lastXw, lastDesc, lastObs, curUR, curBlk, Tcw = [], [], [], [], [], [], []
    for li, ci in pairs:
        kl, dl = batch["ora"][li][1], batch["ora"][li][2]
        z = dep[li // 2, :len(kl)].cpu().numpy()
        v = (z > 0) & (rng.random(len(kl)) < 0.9)
        zz = np.where(z > 0, z, 1.0)
        X = np.stack([(kl["x"] - P.cx) * zz / P.fx, (kl["y"] - P.cy) * zz / P.fy, zz], 1).astype(np.float32)
        lastValid.append(pad(v.astype(np.uint8))); lastXw.append(pad(X)); lastDesc.append(pad(dl))
        lastObs.append(pad((rng.random(len(kl)) < 0.85).astype(np.uint8)))
        kc = batch["ora"][ci][1]
        curUR.append(pad(uR[ci // 2, :len(kc)].cpu().numpy(), -1)); curBlk.append(pad((rng.random(len(kc)) < 0.05).astype(np.uint8)))
        q = np.array([0.0, 0.002, 0.0, 1.0]); q /= np.linalg.norm(q)
        Tcw.append(np.concatenate([q, [0.01, 0.0, 0.02]]).astype(np.float32))
    t = lambda a, dt=None: torch.from_numpy(np.stack(a)).to(dev)
    curImg = torch.tensor([c for _, c in pairs], dtype=torch.int32, device=dev)
    lastImg = torch.tensor([l for l, _ in pairs], dtype=torch.int32, device=dev)
    tot = 0
    for th, fwd, bwd, ori in ((7.0, 0, 0, True), (15.0, 1, 0, True), (7.0, 0, 1, False)):
        m = ORBmatcher(0.9, ori)
        fw = torch.full((3,), fwd, dtype=torch.uint8, device=dev); bw = torch.full((3,), bwd, dtype=torch.uint8, device=dev)
        mc, nm = m.SearchByProjectionLastFrame(P, curImg, lastImg, batch["kps"], batch["desc"], batch["cnt"], t(curUR), t(curBlk), t(Tcw),
                                               t(lastValid), t(lastXw), t(lastDesc), t(lastObs), th, fw, bw)
        torch.cuda.synchronize()
        mc, nm = mc.cpu().numpy(), nm.cpu().numpy()
        for p, (li, ci) in enumerate(pairs):
            kl = batch["ora"][li][1]; kc, dc = batch["ora"][ci][1], batch["ora"][ci][2]
            Fo = O.make_frame(P, kc, dc, curUR[p][:len(kc)])
            r, me = O.search_by_projection_last(Fo, curBlk[p][:len(kc)], Tcw[p], kl, lastValid[p][:len(kl)], lastXw[p][:len(kl)],
                                                lastDesc[p][:len(kl)], lastObs[p][:len(kl)], th, fwd, bwd, ori)
            assert nm[p] == r, (th, p, nm[p], r)
            np.testing.assert_array_equal(mc[p, :len(kc)], me)
            tot += r
    assert tot > 1000


def test_search_for_triangulation(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    P, uR, dep = _scene(batch)
    dev = "cuda"
    k, Lv, lup = 10, 3, 1
    vd, vf = make_vocabulary(k, Lv, seed=2)
    m0 = ORBmatcher(0.6, False)
    word, node = m0.bow_transform(batch["desc"], batch["cnt"], torch.from_numpy(vd).to(dev), torch.from_numpy(vf).to(dev), k, Lv, lup)
    torch.cuda.synchronize()
    nn_ = node.cpu().numpy()
    cap = batch["kps"].shape[1]
    nimg = batch["kps"].shape[0]
    rng = np.random.default_rng(6)
    has = (rng.random((nimg, cap)) < 0.3).astype(np.uint8)
    ur = np.full((nimg, cap), -1, np.float32)
    for i in range(0, nimg, 2):
        n = len(batch["ora"][i][1]); ur[i, :n] = uR[i // 2, :n].cpu().numpy()
    pairs = [(0, 4), (4, 0), (2, 6)]
    R12 = np.stack([np.eye(3, dtype=np.float32)] * 3); t12 = np.array([[0.05, 0.01, 0.0], [-0.05, 0.0, 0.01], [0.0, 0.03, 0.1]], np.float32)
    ep = np.array([[900.0, 250.0], [-150.0, 240.0], [370.0, 250.0]], np.float32)
    img1 = torch.tensor([a for a, _ in pairs], dtype=torch.int32, device=dev); img2 = torch.tensor([b for _, b in pairs], dtype=torch.int32, device=dev)
    tot = 0
    for onlyStereo, coarse, ori in ((False, False, True), (False, True, False), (True, False, True)):
        m = ORBmatcher(0.6, ori)
        m12, nm = m.SearchForTriangulation(P, img1, img2, batch["kps"], batch["desc"], node, batch["cnt"], torch.from_numpy(has).to(dev),
                                           torch.from_numpy(ur).to(dev), R12, t12, ep, onlyStereo, coarse)
        torch.cuda.synchronize()
        m12, nm = m12.cpu().numpy(), nm.cpu().numpy()
        for p, (a, b) in enumerate(pairs):
            ka, da = batch["ora"][a][1], batch["ora"][a][2]; kb, db = batch["ora"][b][1], batch["ora"][b][2]
            r, me = O.search_for_triangulation(ka, da, nn_[a, :len(ka)], has[a, :len(ka)], ur[a, :len(ka)], kb, db, nn_[b, :len(kb)],
                                               has[b, :len(kb)], ur[b, :len(kb)], list(P.levelSigma2)[:8], list(P.scaleFactors)[:8],
                                               [P.fx, P.fy, P.cx, P.cy], R12[p], t12[p], ep[p], onlyStereo, coarse, ori)
            assert nm[p] == r, (p, nm[p], r)
            np.testing.assert_array_equal(m12[p, :len(ka)], me)
            tot += r
    assert tot > 100


def test_search_for_initialization(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    P, uR, dep = _scene(batch)
    dev = "cuda"
    cap = batch["kps"].shape[1]
    pairs = [(0, 4), (2, 6), (4, 0)]
    img1 = torch.tensor([a for a, _ in pairs], dtype=torch.int32, device=dev); img2 = torch.tensor([b for _, b in pairs], dtype=torch.int32, device=dev)
    prev0 = np.zeros((3, cap, 2), np.float32)
    for p, (a, b) in enumerate(pairs):
        ka = batch["ora"][a][1]
        prev0[p, :len(ka), 0] = ka["x"]; prev0[p, :len(ka), 1] = ka["y"]        # Tracking.cc: mvbPrevMatched = F1 keypoints
    tot = 0
    for win, ratio, ori in ((100, 0.9, True), (20, 0.7, False)):
        m = ORBmatcher(ratio, ori)
        prev = torch.from_numpy(prev0.copy()).to(dev)
        m12, nm = m.SearchForInitialization(P, img1, img2, batch["kps"], batch["desc"], batch["cnt"], prev, win)
        torch.cuda.synchronize()
        m12, nm, prevg = m12.cpu().numpy(), nm.cpu().numpy(), prev.cpu().numpy()
        for p, (a, b) in enumerate(pairs):
            ka, da = batch["ora"][a][1], batch["ora"][a][2]; kb, db = batch["ora"][b][1], batch["ora"][b][2]
            F2 = O.make_frame(P, kb, db, None)
            r, me, pe = O.search_for_initialization(ka, da, F2, prev0[p, :len(ka)], win, ratio, ori)
            assert nm[p] == r, (p, nm[p], r)
            np.testing.assert_array_equal(m12[p, :len(ka)], me)
            assert prevg[p, :len(ka)].tobytes() == pe.tobytes()
            tot += r
    assert tot > 300


def test_search_by_projection_keyframe(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    P, uR, dep = _scene(batch)
    dev = "cuda"
    cap = batch["kps"].shape[1]
    rng = np.random.default_rng(9)
    pad = lambda a, fill=0: np.concatenate([a, np.full((cap - len(a),) + a.shape[1:], fill, a.dtype)])
    pairs = [(0, 4), (2, 6)]       # (keyframe image, current image)
    val, Xs, dsc, mx, mn, has, Tcw, Ow = [], [], [], [], [], [], [], []
    for ki, ci in pairs:
        kk, dk = batch["ora"][ki][1], batch["ora"][ki][2]
        z = dep[ki // 2, :len(kk)].cpu().numpy()
        v = (z > 0) & (rng.random(len(kk)) < 0.9)
        zz = np.where(z > 0, z, 1.0)
        X = np.stack([(kk["x"] - P.cx) * zz / P.fx, (kk["y"] - P.cy) * zz / P.fy, zz], 1).astype(np.float32)
        d3 = np.linalg.norm(X, axis=1).astype(np.float32)
        maxD = (d3 * 1.2 ** kk["octave"] * rng.uniform(0.9, 1.2, len(kk))).astype(np.float32)
        val.append(pad(v.astype(np.uint8))); Xs.append(pad(X)); dsc.append(pad(dk)); mx.append(pad(maxD, 1)); mn.append(pad((maxD / 1.2 ** 7).astype(np.float32), 1))
        kc = batch["ora"][ci][1]
        has.append(pad((rng.random(len(kc)) < 0.2).astype(np.uint8)))
        q = np.array([0.001, -0.002, 0.0, 1.0]); q /= np.linalg.norm(q)
        t = np.array([0.01, 0.0, -0.02])
        Tcw.append(np.concatenate([q, t]).astype(np.float32))
        # Ow = -R^T t (float32, like Tcw.inverse().translation())
        qc = q * np.array([-1, -1, -1, 1])
        u = qc[:3]; uv = 2 * np.cross(u, -t); Ow.append((-t + qc[3] * uv + np.cross(u, uv)).astype(np.float32))
    t_ = lambda a: torch.from_numpy(np.stack(a)).to(dev)
    curImg = torch.tensor([c for _, c in pairs], dtype=torch.int32, device=dev); kfImg = torch.tensor([k for k, _ in pairs], dtype=torch.int32, device=dev)
    tot = 0
    for th, orb, ori in ((10.0, 100, True), (3.0, 64, False)):
        m = ORBmatcher(0.9, ori)
        mc, nm = m.SearchByProjectionKeyFrame(P, curImg, kfImg, batch["kps"], batch["desc"], batch["cnt"], t_(has), t_(Tcw), t_(Ow),
                                              t_(val), t_(Xs), t_(mx), t_(mn), t_(dsc), th, orb)
        torch.cuda.synchronize()
        mc, nm = mc.cpu().numpy(), nm.cpu().numpy()
        for p, (ki, ci) in enumerate(pairs):
            kk = batch["ora"][ki][1]; kc, dc = batch["ora"][ci][1], batch["ora"][ci][2]
            Fo = O.make_frame(P, kc, dc, None)
            r, me = O.search_by_projection_kf(Fo, has[p][:len(kc)], Tcw[p], Ow[p], kk, val[p][:len(kk)], Xs[p][:len(kk)], mx[p][:len(kk)],
                                              mn[p][:len(kk)], dsc[p][:len(kk)], th, orb, ori)
            assert nm[p] == r, (p, nm[p], r)
            np.testing.assert_array_equal(mc[p, :len(kc)], me)
            tot += r
    assert tot > 300


# ---- M7: loop-closing / local-mapping searches -----------------------------------------------------------------
def _quat_from_R(R):
    from morb_slam_amd.synth import _quat_from_R as q
    return q(R)


def _lc_scene(batch):
    """Keyframe A = image 0 (world = its camera frame), keyframe B = image 4 (a 4 x 2 px shifted copy) with a nearby pose;
    map points = stereo back-projections of each keyframe's own features."""
    from morb_slam_amd.synth import _quat_from_rotvec, _quat_rot
    P, uR, dep = _scene(batch)
    rng = np.random.default_rng(21)
    out = {}
    q2 = _quat_from_rotvec(np.array([0.002, -0.004, 0.001])); t2 = np.array([-0.03, -0.015, 0.01])
    T = {0: np.array([0, 0, 0, 1, 0, 0, 0], np.float64), 4: np.concatenate([q2, t2])}
    for img, fr in ((0, 0), (4, 2)):
        k, d = batch["ora"][img][1], batch["ora"][img][2]
        z = dep[fr, :len(k)].cpu().numpy()
        Xc = np.stack([(k["x"] - P.cx) * np.abs(z) / P.fx, (k["y"] - P.cy) * np.abs(z) / P.fy, np.abs(z)], 1)
        qinv = T[img][:4] * np.array([-1, -1, -1, 1])
        Xw = np.array([_quat_rot(qinv, x - T[img][4:]) for x in Xc])
        dist = np.linalg.norm(Xc, axis=1)
        maxD = dist * 1.2 ** k["octave"] * rng.uniform(0.9, 1.3, len(k)); minD = maxD / 1.2 ** 7
        Ow = -_quat_rot(qinv, T[img][4:])
        nrm = (Xw - Ow) / np.linalg.norm(Xw - Ow, axis=1, keepdims=True) + rng.normal(0, 0.2, Xw.shape)
        out[img] = dict(k=k, d=d, valid=z > 0, Xw=Xw.astype(np.float32), maxD=maxD.astype(np.float32), minD=minD.astype(np.float32),
                        normal=nrm.astype(np.float32), T=T[img].astype(np.float32), Ow=Ow.astype(np.float32),
                        uR=uR[fr, :len(k)].cpu().numpy())
    return P, out


def test_fuse_and_search_by_projection_sim3(batch):
    """Fuse x2 and SearchByProjection(KF, Sim3) x2: map points of keyframe A searched in keyframe B."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.synth import _quat_from_rotvec, _quat_rot
    P, sc = _lc_scene(batch)
    A = sc[0]
    # the searched keyframe: image 0 again, seen from a pose a few millimetres / a fraction of a milliradian away, so that the
    # projections land within a pixel or two of the features (Fuse's reprojection gate is 5.99 sigma^2)
    qp = _quat_from_rotvec(np.array([0.0004, -0.0006, 0.0003])); tp = np.array([0.002, -0.001, 0.003])
    B = dict(A, T=np.concatenate([qp, tp]).astype(np.float32), Ow=(-_quat_rot(qp * np.array([-1, -1, -1, 1]), tp)).astype(np.float32))
    rng = np.random.default_rng(8)
    n = len(A["Xw"]); nB = len(B["k"])
    cap = batch["kps"].shape[1]
    valid = (A["valid"] & (rng.random(n) < 0.9)).astype(np.uint8)
    invS = (1.0 / np.array(list(P.levelSigma2)[:P.nlevels], np.float32)).astype(np.float32)
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    one = lambda x: cu(x[None])
    padf = lambda a, fill: np.concatenate([a, np.full((cap - len(a),) + a.shape[1:], fill, a.dtype)])
    m = ORBmatcher(0.8, True)
    kf = torch.tensor([0], dtype=torch.int32, device="cuda"); nmp = torch.tensor([n], dtype=torch.int32, device="cuda")
    Fo = O.make_frame(P, B["k"], B["d"], B["uR"])
    args = (one(valid), one(A["Xw"]), one(A["normal"]), one(A["maxD"]), one(A["minD"]), one(A["d"]))
    tot = 0
    for th, sim3 in ((3.0, False), (6.0, True), (10.0, False)):
        bi, bd = m.Fuse(P, kf, batch["kps"], batch["desc"], batch["cnt"], one(padf(B["uR"], -1.0)), one(B["T"]), one(B["Ow"]), nmp, *args,
                        th=th, sim3Form=sim3)
        torch.cuda.synchronize()
        ei, ed = O.fuse_search(Fo, invS, B["T"], B["Ow"], valid, A["Xw"], A["normal"], A["maxD"], A["minD"], A["d"], th, sim3)
        np.testing.assert_array_equal(bi[0, :n].cpu().numpy(), ei)
        np.testing.assert_array_equal(bd[0, :n].cpu().numpy(), ed)
        tot += int((ei >= 0).sum())
    assert tot > 400
    matched = (rng.random(nB) < 0.1).astype(np.uint8)
    tot = 0
    for th, ratio, manual in ((8, 1.0, False), (8, 0.8, True), (4, 1.2, False)):
        mf, nm = m.SearchByProjectionSim3(P, kf, batch["kps"], batch["desc"], batch["cnt"], one(B["T"]), one(B["Ow"]), nmp, *args,
                                          one(padf(matched, 0)), th, ratio, manual)
        torch.cuda.synchronize()
        r, me = O.search_by_projection_sim3(Fo, B["T"], B["Ow"], valid, A["Xw"], A["normal"], A["maxD"], A["minD"], A["d"], matched, th,
                                            ratio, manual)
        g = mf[0, :nB].cpu().numpy()
        assert int(nm[0]) == r
        np.testing.assert_array_equal(g, me)
        tot += r
    assert tot > 300


def test_search_by_sim3(batch):
    """SearchBySim3: both one-way searches through S21 / S12 and the agreement check."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.synth import _quat_rot
    P, sc = _lc_scene(batch)
    A, B = sc[0], sc[4]
    rng = np.random.default_rng(13)
    cap = batch["kps"].shape[1]
    # S12 maps camera-2 coordinates to camera-1 coordinates: p1 = s * R12 p2 + t12 with the true relative pose and s = 1.01
    def Rof(q):
        return np.array([_quat_rot(q.astype(np.float64), e) for e in np.eye(3)]).T
    R1, t1 = Rof(A["T"][:4]), A["T"][4:].astype(np.float64); R2, t2 = Rof(B["T"][:4]), B["T"][4:].astype(np.float64)
    R12 = R1 @ R2.T; t12 = t1 - R12 @ t2
    s = 1.01
    def sim8(R, t, sc_):
        q = _quat_from_R(R) * np.sqrt(sc_)
        return np.concatenate([q, t]).astype(np.float32)
    S12 = sim8(R12, t12, s)
    R21 = R12.T; S21 = sim8(R21, -(R21 @ t12) / s, 1.0 / s)
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    def padded(d, key, fill=0):
        a = d[key]
        return np.concatenate([a, np.full((cap - len(a),) + a.shape[1:], fill, a.dtype)])[None]
    v1 = (A["valid"] & (rng.random(len(A["k"])) < 0.85)).astype(np.uint8); v2 = (B["valid"] & (rng.random(len(B["k"])) < 0.85)).astype(np.uint8)
    A2 = dict(A, valid=v1); B2 = dict(B, valid=v2)
    m = ORBmatcher(0.8, True)
    kf1 = torch.tensor([0], dtype=torch.int32, device="cuda"); kf2 = torch.tensor([4], dtype=torch.int32, device="cuda")
    for th in (7.5, 3.0):
        o = m.SearchBySim3(P, kf1, kf2, batch["kps"], batch["desc"], batch["cnt"], cu(A["T"][None]), cu(B["T"][None]), cu(S12[None]),
                           cu(S21[None]), cu(padded(A2, "valid")), cu(padded(A, "Xw")), cu(padded(A, "maxD")), cu(padded(A, "minD")),
                           cu(padded(A, "d")), cu(padded(B2, "valid")), cu(padded(B, "Xw")), cu(padded(B, "maxD")), cu(padded(B, "minD")),
                           cu(padded(B, "d")), th)
        torch.cuda.synchronize()
        g1, g2, g12, nf = [x.cpu().numpy() for x in o]
        FB = O.make_frame(P, B["k"], B["d"], None); FA = O.make_frame(P, A["k"], A["d"], None)
        e1 = O.search_by_sim3_dir(FB, A["T"], S21, v1, A["Xw"], A["maxD"], A["minD"], A["d"], th)
        e2 = O.search_by_sim3_dir(FA, B["T"], S12, v2, B["Xw"], B["maxD"], B["minD"], B["d"], th)
        np.testing.assert_array_equal(g1[0, :len(e1)], e1)
        np.testing.assert_array_equal(g2[0, :len(e2)], e2)
        e12 = np.array([i2 if (i2 >= 0 and e2[i2] == i1) else -1 for i1, i2 in enumerate(e1)])
        np.testing.assert_array_equal(g12[0, :len(e1)], e12)
        assert int(nf[0]) == int((e12 >= 0).sum()) and int(nf[0]) > (50 if th > 5 else 3)


def test_compute_distinctive_descriptors():
    """MapPoint::ComputeDistinctiveDescriptors: least-median row, first minimum, N from 0 to a few hundred."""
    import torch
    from morb_slam_amd import ORBmatcher
    rng = np.random.default_rng(99)
    sizes = [0, 1, 2, 3, 5, 8, 17, 40, 64, 65, 130, 300] + list(rng.integers(1, 30, 400))
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = np.zeros((start[-1], 32), np.uint8)
    for m, n in enumerate(sizes):
        if n == 0: continue
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        rows = np.tile(base, (n, 1)) ^ np.packbits(rng.random((n, 256)) < rng.uniform(0.02, 0.2), axis=1)
        if m % 5 == 0 and n > 2: rows[1] = rows[0]                 # ties between rows
        desc[start[m]:start[m + 1]] = rows
    g = ORBmatcher().ComputeDistinctiveDescriptors(torch.from_numpy(start).cuda(), torch.from_numpy(desc).cuda())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(g.cpu().numpy(), O.distinctive_descriptors(start, desc))


def test_bow_transform_on_loaded_vocabulary(batch, tmp_path):
    """ComputeBoW on a trained-style tree (nodes with fewer than k children, early leaves) loaded from the DBoW2 text format."""
    import torch
    from morb_slam_amd import ORBmatcher
    from morb_slam_amd.vocabulary import Vocabulary
    from test_oracle_cpu import _irregular_vocabulary
    path, *_ = _irregular_vocabulary(tmp_path, seed=11, k=6, L=4)
    v = Vocabulary.load_text(path)
    desc, cnt = batch["desc"], batch["cnt"]
    cu = lambda x: torch.from_numpy(x).cuda()
    for lup in (1, 2, 4):
        w, nid = ORBmatcher().bow_transform_tree(desc, cnt, cu(v.nodeDesc), cu(v.firstChild), cu(v.childCount), v.L, lup)
        torch.cuda.synchronize()
        w, nid, cn = w.cpu().numpy(), nid.cpu().numpy(), cnt.cpu().numpy()
        for i in range(desc.shape[0]):
            we, ne = O.bow_transform_tree(batch["ora"][i][2], v.nodeDesc, v.firstChild, v.childCount, v.L, lup)
            np.testing.assert_array_equal(w[i, :cn[i]], we)
            np.testing.assert_array_equal(nid[i, :cn[i]], ne)
        assert len(np.unique(w[0, :cn[0]])) > 20


def test_stereo_matches_1080p_4000_bit_exact():
    """ComputeStereoMatches at the size bench.py's C4 workload runs it (1920x1080, 4000 features: LDS tables scale with the
    keypoint capacity), extraction included: mvuRight / mvDepth bit patterns equal the oracle's."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBextractor, ORBmatcher
    left, right = make_stereo_pair(1920, 1080, seed=71)
    ext = ORBextractor(4000, 1.2, 8, 20, 7)
    kps, desc, cnt, _ = ext.extract_batch(torch.from_numpy(np.stack([left, right])).cuda())
    u, d = ORBmatcher().ComputeStereoMatches(ext, kps, desc, cnt, MBF, MB)
    torch.cuda.synchronize()
    ol, orr = O.OracleExtractor(4000), O.OracleExtractor(4000)
    _, kl, dl = ol(left)
    _, kr, dr = orr(right)
    c = cnt.cpu().numpy()
    assert kps[0, :c[0]].cpu().numpy().reshape(-1).view(KP_DTYPE).tobytes() == kl.tobytes()
    assert kps[1, :c[1]].cpu().numpy().reshape(-1).view(KP_DTYPE).tobytes() == kr.tobytes()
    ue, de = O.stereo_matches(ol, orr, kl, dl, kr, dr, MBF, MB)
    n = len(kl)
    assert n > 3900
    assert u[0, :n].cpu().numpy().view(np.uint32).tolist() == ue.view(np.uint32).tolist()
    assert d[0, :n].cpu().numpy().view(np.uint32).tolist() == de.view(np.uint32).tolist()
    assert int((ue >= 0).sum()) > 1500
