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
