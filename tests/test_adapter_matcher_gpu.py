"""The C++ ORBmatcher adapter (include/morb/ORBmatcher.h): every method of the reference's class surface (include/ORBmatcher.h:36-129)
driven from C++ (tests/native/matcher_adapters_check.cc, g++ + libmorb_hip.so) on views of extracted frames, compared with the CPU
oracle — the match tables exactly.  (SearchByProjection(Frame, MapPoints) and DescriptorDistance: tests/test_adapter_gpu.py.)"""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import test_matcher_gpu as T
from morb_slam_amd.synth import make_vocabulary

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def batch():
    return T.make_batch()


def _R_of(q):
    from morb_slam_amd.synth import _quat_rot
    return np.array([_quat_rot(np.asarray(q, np.float64), e) for e in np.eye(3)]).T


def _pose22(T7):
    """R (row-major), t, Ow = -R^T t, Tcw (quaternion xyzw + t) as the FrameView carries them."""
    T7 = np.asarray(T7, np.float64)
    R = _R_of(T7[:4]); t = T7[4:]
    return np.concatenate([R.reshape(9), t, -(R.T @ t), T7]).astype(np.float32)


def test_cpp_matcher_adapter_matches_oracle(batch, tmp_path):
    import torch
    from morb_slam_amd import ORBmatcher
    from test_adapter_gpu import _build
    d = tmp_path / "io"
    d.mkdir()
    put = lambda name, a: np.ascontiguousarray(a).tofile(str(d / (name + ".bin")))
    get = lambda name, dt: np.fromfile(str(d / ("out_" + name + ".bin")), dtype=dt)
    P, uR, dep = T._scene(batch)
    Pb = np.frombuffer(bytes(P), np.uint8)
    ora = batch["ora"]
    dev = "cuda"

    def put_frame(prefix, img, pose=None, **extra):
        k, dd = ora[img][1], ora[img][2]
        put(prefix + "_kps", k); put(prefix + "_desc", dd); put(prefix + "_params", Pb)
        if pose is not None:
            put(prefix + "_pose", _pose22(pose))
        for key, a in extra.items():
            put(prefix + "_" + key, a)
        return k, dd

    rng = np.random.default_rng(4)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)

    def backproject(img, fr):
        k = ora[img][1]
        z = dep[fr, :len(k)].cpu().numpy()
        zz = np.where(z > 0, z, 1.0)
        X = np.stack([(k["x"] - P.cx) * zz / P.fx, (k["y"] - P.cy) * zz / P.fy, zz], 1).astype(np.float32)
        return z, X

    # ---- SearchByProjection(CurrentFrame, LastFrame): last = image 0 (identity pose), current = image 4
    kl, dl = ora[0][1], ora[0][2]; kc, dc = ora[4][1], ora[4][2]
    z, X = backproject(0, 0)
    lastValid = ((z > 0) & (rng.random(len(kl)) < 0.9)).astype(np.uint8)
    lastObs = (rng.random(len(kl)) < 0.85).astype(np.uint8)
    curUR = uR[2, :len(kc)].cpu().numpy(); curBlk = (rng.random(len(kc)) < 0.05).astype(np.uint8)
    q = np.array([0.0, 0.002, 0.0, 1.0]); q /= np.linalg.norm(q)
    Tcw = np.concatenate([q, [0.01, 0.0, 0.02]])
    put_frame("last_cur", 4, Tcw, uright=curUR, tracked=curBlk)
    put_frame("last_last", 0, ident, hasmp=lastValid, mppos=X, mpdesc=dl, mpobs=lastObs)
    put("last_cfg", np.array([0.9, 1, 7.0, 0], np.float32))
    # ---- SearchByProjection(CurrentFrame, pKF, sAlreadyFound): keyframe = image 0, current = image 4
    d3 = np.linalg.norm(X, axis=1).astype(np.float32)
    maxD = (d3 * 1.2 ** kl["octave"] * rng.uniform(0.9, 1.2, len(kl))).astype(np.float32); minD = (maxD / 1.2 ** 7).astype(np.float32)
    kfHas = ((z > 0) & (rng.random(len(kl)) < 0.9)).astype(np.uint8)
    found = (rng.random(len(kl)) < 0.05).astype(np.uint8)
    curHas = (rng.random(len(kc)) < 0.2).astype(np.uint8)
    q2 = np.array([0.001, -0.002, 0.0, 1.0]); q2 /= np.linalg.norm(q2)
    Tcw2 = np.concatenate([q2, [0.01, 0.0, -0.02]])
    put_frame("kfp_cur", 4, Tcw2, hasmp=curHas)
    put_frame("kfp_kf", 0, ident, hasmp=kfHas, mppos=X, mpmax=maxD, mpmin=minD, mpdesc=dl)
    put("kfp_found", found); put("kfp_cfg", np.array([0.9, 1, 10.0, 100], np.float32))
    # ---- SearchByBoW(pKF, F) / SearchByBoW(pKF1, pKF2)
    kv, Lv, lup = 10, 3, 1
    vd, vf = make_vocabulary(kv, Lv, seed=2)
    m0 = ORBmatcher(0.7, True)
    _, node = m0.bow_transform(batch["desc"], batch["cnt"], torch.from_numpy(vd).to(dev), torch.from_numpy(vf).to(dev), kv, Lv, lup)
    torch.cuda.synchronize()
    nn_ = node.cpu().numpy()
    has0 = (rng.random(len(kl)) < 0.8).astype(np.uint8); has4 = (rng.random(len(kc)) < 0.7).astype(np.uint8)
    put_frame("bow_kf", 0, node=nn_[0, :len(kl)], hasmp=has0); put_frame("bow_f", 4, node=nn_[4, :len(kc)])
    nv0, nv4 = len(kl) * 2 // 3, len(kc)
    put_frame("bowkk_1", 0, node=nn_[0, :len(kl)], hasmp=has0, nvalid=np.array([nv0], np.int32))
    put_frame("bowkk_2", 4, node=nn_[4, :len(kc)], hasmp=has4, nvalid=np.array([nv4], np.int32))
    put("bow_cfg", np.array([0.7, 1], np.float32))
    # ---- SearchForInitialization(F1 = image 0, F2 = image 4)
    prev0 = np.stack([kl["x"], kl["y"]], 1).astype(np.float32)
    put_frame("ini_1", 0); put_frame("ini_2", 4); put("ini_prev", prev0); put("ini_cfg", np.array([0.9, 1, 100], np.float32))
    # ---- SearchForTriangulation(pKF1 = image 0, pKF2 = image 4)
    hasT0 = (rng.random(len(kl)) < 0.3).astype(np.uint8); hasT4 = (rng.random(len(kc)) < 0.3).astype(np.uint8)
    ur0 = uR[0, :len(kl)].cpu().numpy()
    R12 = np.eye(3, dtype=np.float32); t12 = np.array([0.05, 0.01, 0.0], np.float32); ep = np.array([900.0, 250.0], np.float32)
    put_frame("tri_1", 0, node=nn_[0, :len(kl)], hasmp=hasT0, uright=ur0); put_frame("tri_2", 4, node=nn_[4, :len(kc)], hasmp=hasT4, uright=curUR)
    put("tri_cfg", np.concatenate([[0.6, 1, 0, 0], R12.reshape(9), t12, ep]).astype(np.float32))
    # ---- Fuse x2 / SearchByProjection(pKF, Scw) x2: map points of keyframe A (image 0) searched in image 0 seen from a nearby pose
    from morb_slam_amd.synth import _quat_from_rotvec, _quat_rot
    Pl, sc = T._lc_scene(batch)
    A = sc[0]
    qp = _quat_from_rotvec(np.array([0.0004, -0.0006, 0.0003])); tp = np.array([0.002, -0.001, 0.003])
    Tb = np.concatenate([qp, tp]).astype(np.float32); Owb = (-_quat_rot(qp * np.array([-1, -1, -1, 1]), tp)).astype(np.float32)
    nA = len(A["Xw"])
    validA = (A["valid"] & (rng.random(nA) < 0.9)).astype(np.uint8)
    put_frame("lc_kf", 0, None, uright=A["uR"])
    put("lc_kf_pose", np.concatenate([_R_of(Tb[:4]).reshape(9), Tb[4:], Owb, Tb]).astype(np.float32))
    put("lc_pts_pos", A["Xw"]); put("lc_pts_normal", A["normal"]); put("lc_pts_maxd", A["maxD"]); put("lc_pts_mind", A["minD"])
    put("lc_pts_desc", A["d"]); put("lc_pts_valid", validA)
    put("lc_sim3", np.concatenate([Tb, Owb]).astype(np.float32))
    matched = np.where(rng.random(len(kl)) < 0.1, 0, -1).astype(np.int32)
    put("lc_matched", matched); put("lc_cfg", np.array([0.8, 1, 3.0, 6.0, 8, 0.8], np.float32))
    # ---- SearchBySim3(pKF1 = image 0, pKF2 = image 4)
    B = sc[4]
    R1, t1 = _R_of(A["T"][:4]), A["T"][4:].astype(np.float64); R2, t2 = _R_of(B["T"][:4]), B["T"][4:].astype(np.float64)
    R12s = R1 @ R2.T; t12s = t1 - R12s @ t2; sS = 1.01

    def sim8(R, t, s_):
        return np.concatenate([T._quat_from_R(R) * np.sqrt(s_), t]).astype(np.float32)
    S12 = sim8(R12s, t12s, sS); S21 = sim8(R12s.T, -(R12s.T @ t12s) / sS, 1.0 / sS)
    v1 = (A["valid"] & (rng.random(len(A["k"])) < 0.85)).astype(np.uint8); v2 = (B["valid"] & (rng.random(len(B["k"])) < 0.85)).astype(np.uint8)
    put_frame("s3_1", 0, A["T"], hasmp=v1, mppos=A["Xw"], mpmax=A["maxD"], mpmin=A["minD"], mpdesc=A["d"])
    put_frame("s3_2", 4, B["T"], hasmp=v2, mppos=B["Xw"], mpmax=B["maxD"], mpmin=B["minD"], mpdesc=B["d"])
    put("s3_cfg", np.concatenate([[0.8, 1, 7.5], S12, S21]).astype(np.float32))

    out = subprocess.run([_build(tmp_path, "matcher_adapters_check.cc"), str(d)], capture_output=True, text=True, timeout=600)
    if os.environ.get("MORB_ADAPTER_TIMING"):     # developer run: per-call latency of the C++ members (tools/refresh_profiles.sh keeps the lines)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "adapter_call_latency.txt"), "w") as f:
            f.write("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("TIMING")))
    assert out.returncode == 0 and "matcher adapters ok" in out.stdout, out.stdout + out.stderr

    # ---- the reference's own signatures (SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&), ...) driven with mock Frame / KeyFrame / MapPoint
    # objects filled from the same files (tests/native/reference_members_check.cc): the tables they leave in the objects must equal the
    # view-taking adapters' (compared with the oracle below)
    ref = subprocess.run([_build(tmp_path, "reference_members_check.cc", mock_ref=True), str(d), "matcher"], capture_output=True, text=True, timeout=600)
    assert ref.returncode == 0 and "reference members (matcher) ok" in ref.stdout, ref.stdout + ref.stderr
    getr = lambda name, dt: np.fromfile(str(d / ("out_ref_" + name + ".bin")), dtype=dt)
    for name in ("last_n", "last_match", "kfp_n", "kfp_match", "bow_n", "bow_match", "bowkk_n", "bowkk_match", "ini_n", "ini_match", "tri_n", "tri_pairs",
                 "sim3p_n", "sim3p_match", "sim3k_n", "sim3k_match", "s3_n", "s3_match"):
        np.testing.assert_array_equal(getr(name, np.int32), get(name, np.int32), err_msg="reference-typed member: " + name)
    assert getr("ini_prev", np.float32).tobytes() == get("ini_prev", np.float32).tobytes()
    for tag in ("fuse", "fuse3"):   # Fuse writes into the keyframe: feature f ends up holding the first point whose best feature is f
        idx = get(tag + "_idx", np.int32)
        slots = np.full(len(kl), -1, np.int32)
        for i in range(len(idx) - 1, -1, -1):
            if idx[i] >= 0:
                slots[idx[i]] = i
        np.testing.assert_array_equal(getr(tag + "_slots", np.int32), slots, err_msg=tag)
        assert int(getr(tag + "_n", np.int32)[0]) == int(get(tag + "_n", np.int32)[0])

    # ---- expectations from the oracle
    Fo4 = O.make_frame(P, kc, dc, curUR)
    r, me = O.search_by_projection_last(Fo4, curBlk, Tcw.astype(np.float32), kl, lastValid, X, dl, lastObs, 7.0, 0, 0, True)
    assert int(get("last_n", np.int32)[0]) == r and r > 200
    np.testing.assert_array_equal(get("last_match", np.int32), me)
    Fo4n = O.make_frame(P, kc, dc, None)
    R2c = _R_of(q2); Ow2 = (-(R2c.T @ Tcw2[4:])).astype(np.float32)
    r, me = O.search_by_projection_kf(Fo4n, curHas, Tcw2.astype(np.float32), _pose22(Tcw2)[12:15], kl, (kfHas & (1 - found)).astype(np.uint8), X, maxD, minD,
                                      dl, 10.0, 100, True)
    assert int(get("kfp_n", np.int32)[0]) == r and r > 100
    np.testing.assert_array_equal(get("kfp_match", np.int32), me)
    ne, me = O.search_by_bow(dl, kl["angle"], has0, nn_[0, :len(kl)], dc, kc["angle"], nn_[4, :len(kc)], 0.7, True)
    assert int(get("bow_n", np.int32)[0]) == ne and ne > 50
    np.testing.assert_array_equal(get("bow_match", np.int32), me)
    ne, me = O.search_by_bow_kfkf(dl, kl["angle"], has0, nn_[0, :len(kl)], nv0, dc, kc["angle"], has4, nn_[4, :len(kc)], nv4, 0.7, True)
    assert int(get("bowkk_n", np.int32)[0]) == ne and ne > 20
    np.testing.assert_array_equal(get("bowkk_match", np.int32), me)
    r, me, pe = O.search_for_initialization(kl, dl, Fo4n, prev0, 100, 0.9, True)
    assert int(get("ini_n", np.int32)[0]) == r and r > 100
    np.testing.assert_array_equal(get("ini_match", np.int32), me)
    assert get("ini_prev", np.float32).tobytes() == pe.tobytes()
    r, me = O.search_for_triangulation(kl, dl, nn_[0, :len(kl)], hasT0, ur0, kc, dc, nn_[4, :len(kc)], hasT4, curUR, list(P.levelSigma2)[:8],
                                       list(P.scaleFactors)[:8], [P.fx, P.fy, P.cx, P.cy], R12, t12, ep, False, False, True)
    pairs = get("tri_pairs", np.int32).reshape(-1, 2)
    assert int(get("tri_n", np.int32)[0]) == r and r > 10
    exp_pairs = np.array([(i, j) for i, j in enumerate(me) if j >= 0], np.int32).reshape(-1, 2)
    np.testing.assert_array_equal(pairs, exp_pairs)
    invS = (1.0 / np.array(list(P.levelSigma2)[:P.nlevels], np.float32)).astype(np.float32)
    FoA = O.make_frame(P, A["k"], A["d"], A["uR"])
    for tag, th, sim3 in (("fuse", 3.0, False), ("fuse3", 6.0, True)):
        ei, ed = O.fuse_search(FoA, invS, Tb, Owb, validA, A["Xw"], A["normal"], A["maxD"], A["minD"], A["d"], th, sim3)
        np.testing.assert_array_equal(get(tag + "_idx", np.int32), ei)
        np.testing.assert_array_equal(get(tag + "_dist", np.int32), ed)
        assert int(get(tag + "_n", np.int32)[0]) == int((ei >= 0).sum()) > 100
    for tag, manual in (("sim3p", False), ("sim3k", True)):
        r, me = O.search_by_projection_sim3(FoA, Tb, Owb, validA, A["Xw"], A["normal"], A["maxD"], A["minD"], A["d"], (matched >= 0).astype(np.uint8), 8, 0.8,
                                            manual)
        assert int(get(tag + "_n", np.int32)[0]) == r and r > 50
        exp = np.where(me >= 0, me, matched)
        np.testing.assert_array_equal(get(tag + "_match", np.int32), exp)
    FB = O.make_frame(P, B["k"], B["d"], None); FA = O.make_frame(P, A["k"], A["d"], None)
    e1 = O.search_by_sim3_dir(FB, A["T"], S21, v1, A["Xw"], A["maxD"], A["minD"], A["d"], 7.5)
    e2 = O.search_by_sim3_dir(FA, B["T"], S12, v2, B["Xw"], B["maxD"], B["minD"], B["d"], 7.5)
    e12 = np.array([i2 if (i2 >= 0 and e2[i2] == i1) else -1 for i1, i2 in enumerate(e1)])
    np.testing.assert_array_equal(get("s3_match", np.int32), e12)
    assert int(get("s3_n", np.int32)[0]) == int((e12 >= 0).sum()) > 50
