"""The C++ drop-in adapters (include/morb/ORBextractor.h, ORBmatcher.h, Optimizer.h) compiled with g++ and linked to libmorb_hip.so:
the C++ side of the boundary a reference maintainer would link.  adapters_check.cc runs every adapter on inputs
written here and dumps what comes back; the dumps are compared with the CPU oracle — keypoints and descriptors byte for byte."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from morb_slam_amd.synth import make_ba_problem, make_image, make_pose_problem, make_stereo_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, src, mock_ref=False):
    exe = str(tmp_path / os.path.splitext(src)[0])
    libdir = os.path.join(ROOT, "morb_slam_amd")
    # mock_ref: the reference-typed members are driven with the mock Frame / KeyFrame / MapPoint of tests/native/mock_ref, with include/morb
    # first on the include path as in an integrated reference tree (so that "ORBmatcher.h" is the adapter)
    inc = ["-I" + os.path.join(ROOT, "tests", "native", "mock_ref"), "-I" + os.path.join(ROOT, "include", "morb")] if mock_ref else []
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + inc + ["-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-o", exe,
                           os.path.join(ROOT, "tests", "native", src), "-L" + libdir, "-lmorb_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_adapter_builds_and_runs(tmp_path):
    out = subprocess.run([_build(tmp_path, "adapter_smoke.cc")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter smoke" in out.stdout


def test_cpp_adapters_match_oracle(tmp_path):
    from morb_slam_amd import ORBextractor, ORBmatcher
    from morb_slam_amd.capi import make_frame_params
    d = tmp_path / "io"
    d.mkdir()
    put = lambda name, a: np.ascontiguousarray(a).tofile(str(d / (name + ".bin")))
    get = lambda name, dt: np.fromfile(str(d / ("out_" + name + ".bin")), dtype=dt)
    # ---- extractor input
    img = make_image(640, 480, seed=17)
    nfeat, nlev, lap = 800, 8, (100, 500)
    put("img", img); put("img_dims", np.array([640, 480, nfeat, nlev, lap[0], lap[1]], np.int32))
    # ---- a frame + map points for SearchByProjection: points triangulated from a stereo pair, seen again from a nearby pose
    left, right = make_stereo_pair(752, 480, seed=2)
    ext = ORBextractor(1200, 1.2, 8, 20, 7)
    oe = O.OracleExtractor(1200, 1.2, 8, 20, 7)
    _, k0, d0 = oe(left)
    mbf, mb = np.float32(458.654 * 0.11), np.float32(0.11)
    import torch
    m = ORBmatcher(0.8, True)
    kb = ext.extract_batch(torch.from_numpy(np.stack([left, right])).cuda())
    u_t, z_t = m.ComputeStereoMatches(ext, kb[0], kb[1], kb[2], float(mbf), float(mb))
    torch.cuda.synchronize()
    n0 = int(kb[2][0])
    ur0, z0 = u_t[0, :n0].cpu().numpy(), z_t[0, :n0].cpu().numpy()
    P = make_frame_params(752, 480, 458.654, 457.296, 367.215, 248.375, float(mbf), float(mb), ext.GetScaleFactors(), ext.GetScaleSigmaSquares())
    valid = z0 > 0
    Xw = np.stack([(k0["x"] - P.cx) * z0 / P.fx, (k0["y"] - P.cy) * z0 / P.fy, z0], 1).astype(np.float32)[valid]
    M = len(Xw)
    rng = np.random.default_rng(5)
    normal = (Xw / np.linalg.norm(Xw, axis=1, keepdims=True) + rng.normal(0, 0.2, Xw.shape)).astype(np.float32)
    dist = np.linalg.norm(Xw, axis=1).astype(np.float32)
    maxD = (dist * 1.2 ** k0["octave"][valid] * rng.uniform(0.9, 1.3, M)).astype(np.float32); minD = (maxD / 1.2 ** 7).astype(np.float32)
    th_ = 0.008
    R = np.array([[np.cos(th_), 0, np.sin(th_)], [0, 1, 0], [-np.sin(th_), 0, np.cos(th_)]], np.float32)
    t = np.array([0.015, -0.01, 0.04], np.float32); Ow = (-(R.T @ t)).astype(np.float32)
    isBad = (rng.random(M) < 0.05).astype(np.uint8); hasObs = (rng.random(M) < 0.9).astype(np.uint8)
    blocked = (rng.random(n0) < 0.1).astype(np.uint8)
    put("f_kps", k0); put("f_desc", d0); put("f_uright", ur0.astype(np.float32)); put("f_blocked", blocked)
    put("f_params", np.frombuffer(bytes(P), np.uint8)); put("f_pose", np.concatenate([R.reshape(9), t, Ow]).astype(np.float32))
    put("mp_xw", Xw); put("mp_normal", normal); put("mp_maxd", maxD); put("mp_mind", minD); put("mp_desc", d0[valid]); put("mp_bad", isBad); put("mp_hasobs", hasObs)
    put("sbp_cfg", np.array([0.8, 3.0, 1.0, 6.0], np.float32))
    # ---- PoseOptimization / LocalBundleAdjustment inputs
    pp = make_pose_problem(600, seed=3)
    cam = pp["cam"]
    put("po_has", pp["hasMP"]); put("po_obs", pp["obs"]); put("po_inv", pp["invSigma2"]); put("po_xw", pp["Xw"]); put("po_pose", pp["pose0"])
    put("po_cam", np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"]], np.float32))
    b = make_ba_problem(seed=2, n_free=8, n_fixed=3, n_points=500)
    # the reference only takes the points a LOCAL keyframe sees (Optimizer.cc:1079-1098): keep that sub-graph, so that the view-taking adapter, the
    # oracle and the reference-typed member (which runs that selection itself over mock objects) all solve the same problem
    fr_ = b["kfFixed"] == 0
    keepMP = np.zeros(len(b["mpPos"]), bool); keepMP[b["eMP"][fr_[b["eKF"]]]] = True
    remap = np.cumsum(keepMP) - 1
    keepE = keepMP[b["eMP"]]
    b = dict(b, mpPos=b["mpPos"][keepMP], eKF=b["eKF"][keepE], eMP=remap[b["eMP"][keepE]].astype(b["eMP"].dtype), eObs=b["eObs"][keepE],
             eInvSigma2=b["eInvSigma2"][keepE])
    put("ba_kf", b["kfPose"]); put("ba_mp", b["mpPos"]); put("ba_fixed", b["kfFixed"]); put("ba_ekf", b["eKF"].astype(np.int32)); put("ba_emp", b["eMP"].astype(np.int32))
    put("ba_eobs", b["eObs"]); put("ba_einv", b["eInvSigma2"])
    out = subprocess.run([_build(tmp_path, "adapters_check.cc"), str(d)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "adapters ok" in out.stdout, out.stdout + out.stderr
    # ---- ORBextractor: byte for byte
    mono_o, ko, do = O.OracleExtractor(nfeat, 1.2, nlev, 20, 7)(img, lap)
    head = get("ext_head", np.int32)
    assert head.tolist() == [mono_o, len(ko)]
    assert get("ext_kps", np.uint8).tobytes() == ko.tobytes()
    assert get("ext_desc", np.uint8).tobytes() == do.tobytes()
    # lazy mvImagePyramid: not downloaded by operator(), fetched on first access, invalidated by the next extraction
    oe2 = O.OracleExtractor(nfeat, 1.2, nlev, 20, 7); oe2(img, lap)
    l2 = oe2.level_image(2)[19:-19, 19:-19]      # (the POD build exposes the interior of each level; with OpenCV: ROIs into the padded copies)
    ph = get("ext_pyr_head", np.int32)
    assert ph.tolist() == [0, 1, nlev, l2.shape[1], l2.shape[0]]
    np.testing.assert_array_equal(get("ext_pyr_l2", np.uint8).reshape(l2.shape), l2)
    assert get("ext_pyr_again", np.int32).tolist() == [1, 0]
    # ---- ORBmatcher::SearchByProjection(F, MPs) (+ isInFrustum)
    Fo = O.make_frame(P, k0, d0, ur0)
    te = O.is_in_frustum(Fo, R, t, Ow, Xw, normal, maxD, minD, 0.5)
    r, me = O.search_by_projection_mps(Fo, blocked, te, isBad, d0[valid], hasObs, 3.0, True, 6.0, 0.8)
    assert int(get("sbp_n", np.int32)[0]) == r and r > 100
    np.testing.assert_array_equal(get("sbp_match", np.int32), me)
    assert int(get("dist", np.int32)[0]) == int(np.unpackbits(d0[0] ^ d0[valid][0]).sum())
    # ---- Optimizer::PoseOptimization
    ro, pe, oe_, se = O.pose_optimization(pp)
    # ---- the reference-typed members (ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, ...), Optimizer::PoseOptimization(Frame*)),
    # driven with mock objects filled from the same files: what they leave in the objects must equal what the view-taking adapters returned
    ref = subprocess.run([_build(tmp_path, "reference_members_check.cc", mock_ref=True), str(d), "tracking"], capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0 and "reference members (tracking) ok" in ref.stdout, ref.stdout + ref.stderr
    getr = lambda name, dt: np.fromfile(str(d / ("out_ref_" + name + ".bin")), dtype=dt)
    for name, dt in (("sbp_n", np.int32), ("sbp_match", np.int32), ("dist", np.int32), ("po_nin", np.int32), ("po_outlier", np.uint8)):
        np.testing.assert_array_equal(getr(name, dt), get(name, dt), err_msg=name)
    assert getr("po_pose", np.float32).tobytes() == get("po_pose", np.float32).tobytes()
    # LocalBundleAdjustment(KeyFrame*, bool*, Map*, int& x 4): the reference's graph selection over mock objects (keyframe order, point order and
    # edge order of the flattened graph then differ from the test's arrays: the optimum agrees to the test's tolerance, the erased observations exactly)
    free = b["kfFixed"] == 0
    localMP = np.zeros(len(b["mpPos"]), bool); localMP[b["eMP"][free[b["eKF"]]]] = True           # points seen by a local keyframe (:1079-1098)
    fixedSeen = np.zeros(len(free), bool); fixedSeen[b["eKF"][localMP[b["eMP"]]]] = True; fixedSeen &= ~free   # fixed keyframes that see one of them (:1100-1116)
    assert localMP.all() and fixedSeen.sum() == (~free).sum(), "the synthetic graph is expected to be connected as the reference's selection needs"
    cnts = getr("ba_counts", np.int32)
    assert cnts[0] == int((~free).sum()) and cnts[1] == int(free.sum()) and cnts[2] == len(b["mpPos"]) and cnts[3] == len(b["eKF"]) and cnts[4] == 1 and cnts[5] == 1
    assert np.abs(getr("ba_kf", np.float32) - get("ba_kf", np.float32)).max() <= 1e-4
    mref = get("ba_mp", np.float32)
    assert np.abs(getr("ba_mp", np.float32) - mref).max() <= 1e-4 * max(1.0, np.abs(mref).max())
    np.testing.assert_array_equal(getr("ba_erase", np.uint8), get("ba_erase", np.uint8))
    assert int(get("po_nin", np.int32)[0]) == ro
    assert np.abs(get("po_pose", np.float32) - pe).max() <= 1e-4
    np.testing.assert_array_equal(get("po_outlier", np.uint8), oe_)
    # ---- Optimizer::LocalBundleAdjustment
    its, kfe, mpe, ee, se = O.local_ba(b)
    st = get("ba_stats", np.int32)
    assert int(st[0]) == int(se[0]) and int(st[1]) == int(se[1])
    assert np.abs(get("ba_kf", np.float32).reshape(-1, 7) - kfe).max() <= 1e-4
    assert np.abs(get("ba_mp", np.float32).reshape(-1, 3) - mpe).max() <= 1e-4 * max(1.0, np.abs(mpe).max())
    np.testing.assert_array_equal(get("ba_erase", np.uint8), ee)
