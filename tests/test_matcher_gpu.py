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
    """4 stereo frames extracted on the GPU (bit-exact with the oracle per test_extractor_gpu) + oracle twins."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBextractor
    pairs = [make_stereo_pair(752, 480, seed=60 + i) for i in range(2)]
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


def test_bow_transform_and_search_by_bow(batch):
    import torch
    from morb_slam_amd import ORBmatcher
    k, Lv, lup = 10, 3, 1
    vd, vf = make_vocabulary(k, Lv, seed=2)
    dvd, dvf = torch.from_numpy(vd).cuda(), torch.from_numpy(vf).cuda()
    desc, cnt, kps = batch["desc"], batch["cnt"], batch["kps"]
    for ratio, ori in ((0.7, True), (0.9, False), (0.6, True)):
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
        assert tot > 500
