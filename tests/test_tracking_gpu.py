"""The tracking chain bench.py times (morb_slam_amd.tracking.TrackingChain: TrackWithMotionModel + TrackLocalMap's searches and
optimisations, device-resident) and LocalMapping's keyframe searches, compared with the oracle stage by stage through
tests/tracking_check.py — at a batch of frames, and one frame at a time (the b = 1 latency shape)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stream(G, seed=0, W=752, H=480, seq_len=16):
    from morb_slam_amd.synth import make_stereo_pair, shift_image
    base = [make_stereo_pair(W, H, seed=seed * 16 + i) for i in range(max(G // seq_len, 1))]
    imgs = np.empty((G, 2, H, W), np.uint8)
    for g in range(G):
        l, r = base[g // seq_len]
        k = g % seq_len
        imgs[g, 0] = shift_image(l, 3 * k, 2 * k); imgs[g, 1] = shift_image(r, 3 * k, 2 * k)
    return imgs


@pytest.fixture(scope="module")
def chains():
    from morb_slam_amd.tracking import build_chains
    imgs = _stream(32, seed=3)
    return build_chains(imgs, B=24, npairs=12, seq_len=16)


def test_tracking_chain_matches_the_oracle_stage_by_stage(chains):
    import tracking_check
    ch, ks, host = chains
    for _ in range(2):            # the second step runs on warm workspaces and must give the same tables
        ch.step(snapshot=True); ch.sync()
        n = tracking_check.verify_tracking(ch, [0, 5, 11, 12, 13, 23], host["kps"], host["cnt"], host["desc"], host["curUR"], host["scene"])
        assert n == 6
    nm = ch.nmLast.cpu().numpy(); nl = ch.nmLocal.cpu().numpy(); ni = ch.nInl.cpu().numpy()
    assert nm.min() > 100 and nl.min() > 20 and ni.min() > 100, (nm.min(), nl.min(), ni.min())


def test_keyframe_searches_match_the_oracle(chains):
    import tracking_check
    ch, ks, host = chains
    ks.step(); ks.sync()
    n = tracking_check.verify_keyframe_searches(ks, [0, 3, 7, 11], host["kps"], host["cnt"], host["desc"], host["node"], host["uR_img"], host["kscene"])
    assert n == 4
    assert int(ks.tri[1].sum()) > 200 and int((ks.fused[0] >= 0).sum()) > 1000


def test_tracking_chain_one_frame_at_a_time(chains):
    """b = 1 (the latency shape): the same frame alone gives the tables it gives inside the batch."""
    import torch
    from morb_slam_amd.tracking import TrackingChain
    ch, ks, host = chains
    ch.step(); ch.sync()
    ref = (ch.frameMP.cpu().numpy(), ch.pose.cpu().numpy(), ch.nInl.cpu().numpy())
    for f in (0, 13):
        one = {k: v[f:f + 1] for k, v in host["scene"].items()}
        c1 = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight[f:f + 1].contiguous(), one)
        c1.step(); c1.sync()
        assert np.array_equal(c1.frameMP.cpu().numpy()[0], ref[0][f]) and c1.nInl.cpu().numpy()[0] == ref[2][f]
        assert c1.pose.cpu().numpy()[0].tobytes() == ref[1][f].tobytes()
        c1.close()


def test_search_paths_beyond_the_register_lists(chains):
    """k_search's slower in-kernel paths, against the oracle: (1) windows so wide that a query's list overflows the 32 stored candidates (the
    query walks the grid again in every pass), (2) more than 2048 queries per frame (no register slot: lists read from L2), both with the
    blocked-feature dependence in play (hasObs set on most points)."""
    import torch
    import oracle_lib as O
    from morb_slam_amd import ORBmatcher
    ch, ks, host = chains
    sc = host["scene"]
    dev = ch.dev
    P = ch.P
    # (1) SearchByProjection(Cur, Last) with th = 45: r = 45 * scale -> windows of 90 .. 320 px
    m = ORBmatcher(0.9, True)
    F = 4
    mt, nm = m.SearchByProjectionLastFrame(P, ch.curImg[:F].contiguous(), ch.lastImg[:F].contiguous(), ch.kps, ch.desc, ch.count, ch.uRight[:F].contiguous(), None,
                                           ch.pose0[:F].contiguous(), ch.lastValid[:F].contiguous(), ch.lastXw[:F].contiguous(), ch.lastDesc[:F].contiguous(),
                                           ch.lastHasObs[:F].contiguous(), 45.0, ch.fwd[:F].contiguous(), ch.bwd[:F].contiguous())
    torch.cuda.synchronize()
    mt, nm = mt.cpu().numpy(), nm.cpu().numpy()
    for f in range(F):
        ci, li = int(sc["curImg"][f]), int(sc["lastImg"][f]); nc, nl = int(host["cnt"][ci]), int(host["cnt"][li])
        Fo = O.make_frame(P, host["kps"][ci, :nc], host["desc"][ci, :nc], host["curUR"][f, :nc])
        lastMP = sc["lastMP"][f, :nl]; lv = (lastMP >= 0).astype(np.uint8); lm = np.maximum(lastMP, 0)
        r, me = O.search_by_projection_last(Fo, np.zeros(nc, np.uint8), sc["pose0"][f], host["kps"][li, :nl], lv, sc["mpXw"][f][lm], sc["mpDesc"][f][lm],
                                            sc["mpHasObs"][f][lm] * lv, 45.0, 0, 0, True)
        assert int(nm[f]) == r, (f, int(nm[f]), r)
        np.testing.assert_array_equal(mt[f, :nc], me)
    # (2) SearchByProjection(F, MapPoints) over 3000 map points (the frame's table, its first 952 rows once more), th = 3
    mp = 3000
    rep = lambda a: torch.cat([a, a[:, :mp - a.shape[1]]], 1).contiguous()
    f0 = slice(0, 2)
    Rcw = torch.eye(3, device=dev).reshape(1, 9).repeat(2, 1).contiguous(); z3 = torch.zeros((2, 3), device=dev)
    Xw, nrm, mx, mn = rep(ch.mpXw[f0]), rep(ch.mpNormal[f0]), rep(ch.mpMaxD[f0]), rep(ch.mpMinD[f0])
    dsc, ho = rep(ch.mpDesc[f0]), rep(ch.mpHasObs[f0])
    nMP = torch.full((2,), mp, dtype=torch.int32, device=dev)
    ml = ORBmatcher(0.8, True)
    trk = ml.isInFrustum(P, Rcw, z3, z3, nMP, Xw, nrm, mx, mn, 0.5)
    bad = torch.zeros((2, mp), dtype=torch.uint8, device=dev)
    blocked = torch.zeros((2, ch.cap), dtype=torch.uint8, device=dev)
    mt2, nm2 = ml.SearchByProjectionMapPoints(P, ch.curImg[f0].contiguous(), ch.kps, ch.desc, ch.count, ch.uRight[f0].contiguous(), blocked, nMP, trk, bad, dsc, ho, 3.0)
    torch.cuda.synchronize()
    mt2, nm2 = mt2.cpu().numpy(), nm2.cpu().numpy()
    cat = lambda a: np.concatenate([a, a[:mp - len(a)]])
    for f in range(2):
        ci = int(sc["curImg"][f]); nc = int(host["cnt"][ci])
        Fo = O.make_frame(P, host["kps"][ci, :nc], host["desc"][ci, :nc], host["curUR"][f, :nc])
        te = O.is_in_frustum(Fo, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32), cat(sc["mpXw"][f]), cat(sc["mpNormal"][f]),
                             cat(sc["mpMaxD"][f]), cat(sc["mpMinD"][f]), 0.5)
        r, me = O.search_by_projection_mps(Fo, np.zeros(nc, np.uint8), te, np.zeros(mp, np.uint8), cat(sc["mpDesc"][f]), cat(sc["mpHasObs"][f]), 3.0, False, 0.0, 0.8)
        assert int(nm2[f]) == r, (f, int(nm2[f]), r)
        np.testing.assert_array_equal(mt2[f, :nc], me)
    assert int(nm.sum()) > 1000 and int(nm2.sum()) > 500


@pytest.mark.parametrize("copies,flip", [(4, 0.03), (8, 0.08)])
def test_search_fixed_point_under_heavy_contention(chains, copies, flip):
    """The order dependence of SearchByProjection(F, MapPoints) — a feature matched to a point with observations is skipped by every LATER point —
    is solved as a fixed point in k_search.  Here every map point exists `copies` times (same position, descriptors a few bits apart), all with
    observations: the copies compete for the same features, the first takes the best one, the next its runner-up, ... long blocking chains that
    need many passes.  Tables and counts must equal the oracle's sequential loop."""
    import torch
    import oracle_lib as O
    from morb_slam_amd import ORBmatcher
    ch, ks, host = chains
    sc = host["scene"]
    dev = ch.dev
    P = ch.P
    rng = np.random.default_rng(77 + copies)
    F, base = 3, 2048 // copies
    mp = base * copies
    Xw = np.zeros((F, mp, 3), np.float32); nrm = np.zeros((F, mp, 3), np.float32); mx = np.ones((F, mp), np.float32); mn = np.ones((F, mp), np.float32)
    dsc = np.zeros((F, mp, 32), np.uint8)
    for f in range(F):
        for c in range(copies):          # interleaved: copies of a point are `base` queries apart AND neighbours differ -> conflicts across the whole range
            sl = slice(c * base, (c + 1) * base)
            Xw[f, sl] = sc["mpXw"][f, :base]; nrm[f, sl] = sc["mpNormal"][f, :base]; mx[f, sl] = sc["mpMaxD"][f, :base]; mn[f, sl] = sc["mpMinD"][f, :base]
            dsc[f, sl] = sc["mpDesc"][f, :base] ^ np.packbits(rng.random((base, 256)) < flip, axis=1)
    ho = np.ones((F, mp), np.uint8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    nMP = torch.full((F,), mp, dtype=torch.int32, device=dev)
    Rcw = torch.eye(3, device=dev).reshape(1, 9).repeat(F, 1).contiguous(); z3 = torch.zeros((F, 3), device=dev)
    m = ORBmatcher(0.8, True)
    trk = m.isInFrustum(P, Rcw, z3, z3, nMP, t(Xw), t(nrm), t(mx), t(mn), 0.5)
    bad = torch.zeros((F, mp), dtype=torch.uint8, device=dev)
    blocked = torch.zeros((F, ch.cap), dtype=torch.uint8, device=dev)
    mt, nm = m.SearchByProjectionMapPoints(P, ch.curImg[:F].contiguous(), ch.kps, ch.desc, ch.count, ch.uRight[:F].contiguous(), blocked, nMP, trk, bad, t(dsc),
                                           t(ho), 5.0)
    torch.cuda.synchronize()
    mt, nm = mt.cpu().numpy(), nm.cpu().numpy()
    for f in range(F):
        ci = int(sc["curImg"][f]); nc = int(host["cnt"][ci])
        Fo = O.make_frame(P, host["kps"][ci, :nc], host["desc"][ci, :nc], host["curUR"][f, :nc])
        te = O.is_in_frustum(Fo, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32), Xw[f], nrm[f], mx[f], mn[f], 0.5)
        r, me = O.search_by_projection_mps(Fo, np.zeros(nc, np.uint8), te, np.zeros(mp, np.uint8), dsc[f], ho[f], 5.0, False, 0.0, 0.8)
        assert int(nm[f]) == r, (f, int(nm[f]), r)
        np.testing.assert_array_equal(mt[f, :nc], me)
        later = (me >= base).sum()
        assert later > 20, "the scene should make later copies win features the first copy's match blocked"
