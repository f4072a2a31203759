"""GPU parity: HIP ORBextractor (through the C ABI) vs the CPU oracle, stage by stage and end to end.
Bit-exact everywhere: pyramid bytes, blurred bytes, FAST candidates (order included), quadtree selection
(order included), angles (f32 bits), descriptors, final keypoint records and monoIndex."""
import numpy as np
import pytest

from morb_slam_amd.synth import make_image, make_stereo_pair

pytestmark = pytest.mark.gpu


def _extractors(nfeat, **kw):
    from morb_slam_amd import ORBextractor
    from oracle_lib import OracleExtractor
    args = dict(scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)
    args.update(kw)
    return ORBextractor(nfeat, args["scaleFactor"], args["nlevels"], args["iniThFAST"], args["minThFAST"]), \
        OracleExtractor(nfeat, args["scaleFactor"], args["nlevels"], args["iniThFAST"], args["minThFAST"])


def _compare(img, nfeat, lap=(0, 0), stages=True, **kw):
    g, o = _extractors(nfeat, **kw)
    mono_g, kg, dg = g(img, None, lap)
    mono_o, ko, do = o(img, lap)
    if stages:
        for l in range(g.GetLevels()):
            assert g.level_size(l) == o.level_size(l)
            np.testing.assert_array_equal(g.pyramid_level(l), o.level_image(l), err_msg=f"pyramid level {l}")
            co, cg = o.level_candidates(l), g.level_candidates(l)
            assert len(co) == len(cg), f"level {l}: {len(cg)} candidates vs oracle {len(co)}"
            for f in ("x", "y", "response"):
                np.testing.assert_array_equal(cg[f], co[f], err_msg=f"candidates level {l} field {f}")
            so, sg = o.level_keypoints(l), g.level_keypoints(l)
            assert len(so) == len(sg), f"level {l}: {len(sg)} selected vs oracle {len(so)}"
            for f in ("x", "y", "response", "octave", "size"):
                np.testing.assert_array_equal(sg[f], so[f], err_msg=f"selected level {l} field {f}")
            bo = o.level_blurred(l)
            if bo is not None:
                np.testing.assert_array_equal(g.blurred_level(l), bo, err_msg=f"blur level {l}")
    assert mono_g == mono_o
    assert len(kg) == len(ko)
    assert kg.tobytes() == ko.tobytes(), "keypoint records differ"
    np.testing.assert_array_equal(dg, do)
    return len(kg)


def test_vga_1200_bit_exact():
    left, right = make_stereo_pair(752, 480, seed=1)
    n = _compare(left, 1200)
    assert n > 1000
    _compare(right, 1200)


def test_mono_lapping_reverse_fill():
    # Frame.cc:428 passes [0,1000] for mono: every keypoint of a 752-wide image is "stereo" and the output is
    # filled from the back; monoIndex 0
    img = make_image(752, 480, seed=3)
    g, o = _extractors(1000)
    mono_g, kg, dg = g(img, None, (0, 1000))
    mono_o, ko, do = o(img, (0, 1000))
    assert mono_g == mono_o == 0
    assert kg.tobytes() == ko.tobytes()
    np.testing.assert_array_equal(dg, do)


def test_fisheye_512_partial_lap():
    img = make_image(512, 512, seed=5)
    _compare(img, 1500, lap=(100, 400))


def test_low_texture_threshold_fallback_and_short_quota():
    # nearly flat image: most cells need the minThFAST pass and the quadtree ends short of nfeatures
    rng = np.random.default_rng(7)
    img = np.full((480, 752), 120, np.uint8)
    img[100:140, 200:260] = 200
    img[300:310, 500:700] = 30
    img = (img.astype(np.int16) + rng.integers(-3, 4, img.shape)).clip(0, 255).astype(np.uint8)
    n = _compare(img, 1200)
    assert n < 1200


def test_noise_image_many_candidates():
    # white noise: tens of thousands of FAST candidates, exercises the global-memory key path of the quadtree
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (480, 752), dtype=np.uint8)
    _compare(img, 1200)


def test_dense_corners_strip_mode():
    # k_fastw keeps a cell's corners in a 512-entry list and its keypoints in a 64-entry list; white noise at thresholds this low
    # makes about half of the pixels corners and every ~9th a keypoint, so every cell takes the row-by-row strip mode (whole pass
    # redone with a rolling strength buffer), through both triggers: corner-list overflow (T = 1) and keypoint-list overflow (T = 6)
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    _compare(img, 2000, nlevels=3, iniThFAST=1, minThFAST=1)
    _compare(img, 2000, nlevels=3, iniThFAST=6, minThFAST=2)
    # and a window whose every reject round overflows the survivor queue (both polarities flagged): alternating extremes
    chk = ((np.indices((240, 320)).sum(0) & 1) * 255).astype(np.uint8)
    chk[::7, ::5] = 128
    _compare(chk, 1000, nlevels=2, iniThFAST=20, minThFAST=7)


def test_other_sizes_and_params():
    _compare(make_image(640, 480, seed=21), 500, scaleFactor=1.2, nlevels=8)
    _compare(make_image(333, 217, seed=22), 300, nlevels=4)
    _compare(make_image(1280, 720, seed=23), 2000, iniThFAST=15, minThFAST=5)


def test_pyramid_launch_paths():
    # the pyramid stage has several shapes: one level (plain copy kernel), two (levels 0 + 1 from one launch only), other scale
    # factors (row / column tables with other strides; 2.0 leaves the 8-byte source window of four outputs), widths whose padded
    # pitch ends in a 64-, 128- or 192-px partial block column
    _compare(make_image(640, 480, seed=51), 300, nlevels=1)
    _compare(make_image(640, 480, seed=52), 400, nlevels=2)
    _compare(make_image(752, 480, seed=53), 600, scaleFactor=1.5, nlevels=4)
    _compare(make_image(800, 600, seed=54), 600, scaleFactor=2.0, nlevels=3)
    _compare(make_image(601, 377, seed=55), 500, scaleFactor=1.1, nlevels=6)
    for w in (474, 538, 602, 666):   # padded pitch = 512 + 64, 128, 192, 256
        _compare(make_image(w, 240, seed=56 + w), 300, nlevels=3)


def test_batches_of_8_and_16_follow_the_xcd_tile_order():
    # with a multiple of 8 images the pyramid workgroups are dealt to the XCDs image by image (py_tile); other counts use the plain order
    import torch
    from morb_slam_amd import KP_DTYPE
    for nimg in (8, 16, 5):
        g, o = _extractors(700)
        imgs = np.stack([make_image(752, 480, seed=60 + i) for i in range(nimg)])
        kps, desc, cnt, mono = g.extract_batch(torch.from_numpy(imgs).cuda())
        torch.cuda.synchronize()
        cnt = cnt.cpu().numpy(); kps = kps.cpu().numpy(); desc = desc.cpu().numpy()
        for i in range(nimg):
            mo, ko, do = o(imgs[i])
            assert cnt[i] == len(ko)
            assert kps[i, :cnt[i]].reshape(-1).view(KP_DTYPE).tobytes() == ko.tobytes()
            np.testing.assert_array_equal(desc[i, :cnt[i]], do)


def test_threshold_orderings():
    # the reference runs cv::FAST(iniThFAST) and, for an empty cell, cv::FAST(minThFAST), whatever their order: equal and
    # swapped thresholds exercise the second pass after a first pass that left strengths behind
    img = make_image(640, 480, seed=41)
    _compare(img, 800, iniThFAST=12, minThFAST=12)
    _compare(img, 800, iniThFAST=7, minThFAST=20)
    _compare(img, 800, iniThFAST=60, minThFAST=3)     # most cells fall through to the second pass


def test_1080p_4000():
    _compare(make_image(1920, 1080, seed=31), 4000)


def test_empty_image_returns_minus_one():
    g, _ = _extractors(1000)
    mono, k, d = g(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0


def test_batch_matches_single():
    import torch
    from morb_slam_amd import KP_DTYPE
    g, o = _extractors(1200)
    imgs = np.stack([make_image(752, 480, seed=40 + i) for i in range(6)])
    d = torch.from_numpy(imgs).cuda()
    kps, desc, cnt, mono = g.extract_batch(d)
    torch.cuda.synchronize()
    cnt = cnt.cpu().numpy(); mono = mono.cpu().numpy()
    kps = kps.cpu().numpy(); desc = desc.cpu().numpy()
    for i in range(len(imgs)):
        mo, ko, do = o(imgs[i])
        assert cnt[i] == len(ko) and mono[i] == mo
        kg = kps[i, :cnt[i]].reshape(-1).view(KP_DTYPE)
        assert kg.tobytes() == ko.tobytes()
        np.testing.assert_array_equal(desc[i, :cnt[i]], do)


def test_two_extractors_on_two_threads():
    # Frame::Frame runs the left and right extractor on two std::threads (Frame.cc:194-197): two handles, concurrent calls
    import threading
    left, right = make_stereo_pair(752, 480, seed=77)
    from morb_slam_amd import ORBextractor
    from oracle_lib import OracleExtractor
    gl, gr = ORBextractor(1200, 1.2, 8, 20, 7), ORBextractor(1200, 1.2, 8, 20, 7)
    res = {}

    def run(name, g, img):
        for _ in range(5):
            res[name] = g(img, None, (0, 0))

    th = [threading.Thread(target=run, args=("l", gl, left)), threading.Thread(target=run, args=("r", gr, right))]
    [t.start() for t in th]; [t.join() for t in th]
    for name, img in (("l", left), ("r", right)):
        mo, ko, do = OracleExtractor(1200)(img)
        mg, kg, dg = res[name]
        assert mg == mo and kg.tobytes() == ko.tobytes() and np.array_equal(dg, do)


def test_batch_capacity_is_checked():
    # k_layout fills the lapping-area keypoints from the back of [0, count): a caller's cap below the extractor's maximum
    # must be refused, not silently produce a cut block
    import torch
    from morb_slam_amd import ORBextractor
    from morb_slam_amd.capi import MorbError, lib, ptr
    g = ORBextractor(500, 1.2, 8, 20, 7)
    img = torch.from_numpy(make_image(640, 480, seed=2)[None]).cuda()
    cap = g.max_keypoints - 1
    kps = torch.empty((1, cap, 28), dtype=torch.uint8, device="cuda"); desc = torch.empty((1, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda"); mono = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib().morb_extract_batch(g._h, ptr(img), 1, 640, 480, 640, 640 * 480, None, ptr(kps), ptr(desc), cap, ptr(cnt), ptr(mono), None)
    assert rc == -3, rc   # MORB_ERR_CAPACITY


def test_input_ending_on_a_page_boundary():
    # 64 images of 512 x 512 = exactly 16 MiB: the device buffer ends where its last page ends.  The fused level-0 / level-1 kernel
    # read 8-byte windows of the caller's image and the last row's window ran up to 6 bytes past the end of the buffer — silent for
    # most sizes, a memory fault for this one (found with the C3 shape at 32 stereo frames).  Run in a child process: a fault kills it.
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "native", "extract_once.py"), "512", "512", "1500", "32", "1"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_quadtree_team_and_single_wave_packings_agree():
    # k_distribute has two packings of an image's levels into workgroups: calls with <= 64 images give every level of >= 160 k pixels a
    # team of QT_TEAM_WAVES = 16 waves (latency), larger batches keep one wave per (image, level).  Every other test here runs the first; this one
    # runs both on the same images and requires the same bytes — and the oracle's.
    import torch
    from morb_slam_amd import KP_DTYPE
    g, o = _extractors(1200)
    imgs = np.stack([make_image(752, 480, seed=80 + (i % 3)) for i in range(70)])
    exp = [o(imgs[i]) for i in range(3)]
    for n in (70, 20, 3):   # 70 images: single-wave packing; 20 and 3: teams
        kps, desc, cnt, mono = g.extract_batch(torch.from_numpy(imgs[:n]).cuda())
        torch.cuda.synchronize()
        cnt = cnt.cpu().numpy(); kps = kps.cpu().numpy(); desc = desc.cpu().numpy()
        for i in range(n):
            mo, ko, do = exp[i % 3]
            assert cnt[i] == len(ko), (n, i)
            assert kps[i, :cnt[i]].reshape(-1).view(KP_DTYPE).tobytes() == ko.tobytes(), (n, i)
            np.testing.assert_array_equal(desc[i, :cnt[i]], do)


def test_event_after_fast_is_a_live_hip_event():
    """morb_extractor_event_after_fast: the event the last extract_batch recorded behind its FAST stage can be waited on by the caller."""
    import ctypes
    import torch
    ext, _ = _extractors(500)
    img = torch.from_numpy(np.stack([make_image(640, 480, seed=3), make_image(640, 480, seed=4)])).cuda()
    ext.extract_batch(img)
    ev = ext.event_after_fast()
    assert ev
    from morb_slam_amd.capi import lib
    s = torch.cuda.Stream()
    assert lib().morb_stream_wait_event(s.cuda_stream, ev) == 0     # hipStreamWaitEvent through the library's own HIP runtime
    s.synchronize()
    torch.cuda.synchronize()
    assert lib().morb_stream_wait_event(s.cuda_stream, None) == -1  # MORB_ERR_INVALID


def test_event_after_pyramid_is_recorded_once_asked_for():
    """morb_extractor_event_after_pyramid: recorded behind the last pyramid launch by the extractions queued AFTER the first request (callers that never ask
    pay for no event between the pyramid and FAST).  A stream that waits for it and then reads the pyramid's top level sees this call's pyramid."""
    import torch
    from morb_slam_amd.capi import lib
    ext, orc = _extractors(500)
    img = make_image(640, 480, seed=5)
    ev = ext.event_after_pyramid()           # the request; nothing recorded yet
    assert ev
    ext.extract_batch(torch.from_numpy(np.stack([img, img])).cuda())
    assert ext.event_after_pyramid() == ev   # the handle's event, recorded by that call
    s = torch.cuda.Stream()
    assert lib().morb_stream_wait_event(s.cuda_stream, ev) == 0
    s.synchronize()
    orc(img)
    np.testing.assert_array_equal(ext.pyramid_level(7), orc.level_image(7))


def test_more_than_65535_candidates_in_a_level_match_the_oracle():
    """DistributeOctTree has no size limit (ORBextractor.cc:540-738).  White noise at 1080p puts ~200 k FAST candidates into level 0: until round 5
    the quadtree's 16-bit child counts refused such a level (MORB_ERR_UNSUPPORTED); the counts now saturate and nodes of more than 65535 keys are
    counted again when they are split, so the level is distributed exactly — batched (one wave per level) and one image at a time (team of waves)."""
    import torch
    from morb_slam_amd import KP_DTYPE, ORBextractor
    from oracle_lib import OracleExtractor
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (1080, 1920), dtype=np.uint8)
    g = ORBextractor(4000, 1.2, 8, 20, 7)
    o = OracleExtractor(4000, 1.2, 8, 20, 7)
    mono_o, ko, do = o(img)
    assert max(len(o.level_candidates(l)) for l in range(8)) > 65535
    mono, k, d = g(img)                                     # <= 64 images per call: the big levels are worked by a team of 16 waves
    assert mono == mono_o and k.tobytes() == ko.tobytes() and np.array_equal(d, do)
    imgs = np.stack([img] * 65)                             # > 64 images per call: one wave per level
    kps, desc, cnt, _ = g.extract_batch(torch.from_numpy(imgs).cuda())
    torch.cuda.synchronize()
    g.check_status()
    for i in (0, 64):
        n = int(cnt[i])
        assert n == len(ko) and kps[i, :n].cpu().numpy().reshape(-1).view(KP_DTYPE).tobytes() == ko.tobytes()
        assert np.array_equal(desc[i, :n].cpu().numpy(), do)
    mono, k, d = g(make_image(1920, 1080, seed=31))      # and the handle keeps working
    assert len(k) > 3000


def _edge_case_images():
    """Images that steer DistributeOctTree through every way the full sweeps can end (quadtree.h qt_fast_forward): texture in one corner only (a deep,
    lop-sided tree: the histogram's depth is used up and ordinary sweeps go on), a handful of isolated corners (the tree stops because nothing is left
    to divide, far below the quota), flat images (no candidate at all), and the aspect ratios that give 1, 3 and 4 initial nodes."""
    rng = np.random.default_rng(77)
    out = {}
    img = np.full((480, 752), 128, np.uint8)
    img[20:170, 30:230] = rng.integers(0, 256, (150, 200), dtype=np.uint8)
    out["texture in one corner"] = img
    img = np.full((480, 752), 90, np.uint8)
    for k in range(9):
        y, x = 60 + 40 * k, 80 + 70 * k
        img[y:y + 9, x:x + 9] = 230
    out["nine isolated squares"] = img
    img = np.full((480, 752), 128, np.uint8)
    img[200:260, 300:420] = make_image(120, 60, seed=5)
    out["one small textured patch"] = img
    out["square image (one initial node)"] = make_image(480, 480, seed=6)
    out["3 : 1 (three initial nodes)"] = make_image(900, 300, seed=7)
    out["3.2 : 1 (three initial nodes at the bottom of the pyramid, four at the top: the border is a fixed 16 px)"] = make_image(1280, 400, seed=8)
    img = make_image(752, 480, seed=9)
    img[:, 376:] = 128                                    # the right half of the image — one of the two initial nodes — is empty
    out["empty right half"] = img
    return out


@pytest.mark.parametrize("nfeat", [1200, 60])
def test_quadtree_every_ending_of_the_full_sweeps(nfeat):
    """Every image of _edge_case_images, one at a time (the big levels are worked by a team of waves) — selection AND order of every level against the
    oracle's std::list / std::sort restatement — and then all of them in one batch of 72 (one wave per level), which must give the same bytes."""
    import torch
    from morb_slam_amd import ORBextractor
    cases = _edge_case_images()
    for name, img in cases.items():
        n = _compare(img, nfeat)
        assert n >= 0, name
    same = [im for im in cases.values() if im.shape == (480, 752)]
    batch = np.stack([same[i % len(same)] for i in range(72)])
    ext = ORBextractor(nfeat, 1.2, 8, 20, 7)
    kps, desc, cnt, _ = ext.extract_batch(torch.from_numpy(batch).cuda())
    torch.cuda.synchronize()
    for i in range(len(same)):
        _, k1, d1 = ext(same[i])
        c = int(cnt[i])
        assert c == len(k1)
        assert kps[i, :c].cpu().numpy().tobytes() == k1.tobytes() and np.array_equal(desc[i, :c].cpu().numpy(), d1)


def test_quadtree_tiny_quotas_take_the_sweeps():
    """nfeatures so small that a level's node arrays cannot hold the fast-forward's scratch (qt_fast_forward returns -1): the sweeps run one by one."""
    for nfeat in (8, 17):
        _compare(make_image(640, 480, seed=21), nfeat)
        _compare(make_image(480, 480, seed=22), nfeat)
