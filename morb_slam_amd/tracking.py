"""TrackingChain — the searches and optimisations Tracking runs on every frame once a map exists, batched over B frames whose
features, map points and poses stay in HBM:

    Tracking::TrackWithMotionModel (Tracking.cc:2655-2740)
        ORBmatcher(0.9, true).SearchByProjection(CurrentFrame, LastFrame, th = 7 stereo / 15 mono)     ORBmatcher.cc:1521-1733
        Optimizer::PoseOptimization(&CurrentFrame)                                                     Optimizer.cc:762-1051
        discard outliers                                                                               Tracking.cc:2716-2740
    Tracking::TrackLocalMap (Tracking.cc:2745-2806) with SearchLocalPoints (:3117-3183)
        Frame::isInFrustum(pMP, 0.5) over the local map points not yet seen in the frame               Frame.cc:611-678
        ORBmatcher(0.8).SearchByProjection(CurrentFrame, mvpLocalMapPoints, th, bFarPoints, thFar)     ORBmatcher.cc:42-209
        Optimizer::PoseOptimization(&CurrentFrame), inlier count                                       Tracking.cc:2779-2806

and LocalMapping's two per-keyframe searches (`KeyframeSearches`): SearchForTriangulation (LocalMapping.cc:473,
ORBmatcher.cc:821-1042) and Fuse (LocalMapping.cc:771-802, ORBmatcher.cc:1044-1183) over keyframe pairs.

One `step()` enqueues the whole chain on one stream through the C ABI (libmorb_hip.so); the marshalling between the stages
(morb_pose_edges_batch, morb_track_discard_outliers_batch, morb_frame_set_pose_batch) runs on the device.  This is the object
`bench.py --extras` times (`extra_metrics.tracking_chain`) and `tests/test_tracking_gpu.py` compares with the oracle stage by stage.
There is no CPU path."""
import ctypes as C

import numpy as np

from .capi import check, lib, ptr
from .matcher import ORBmatcher
from .optimizer import Optimizer


class TrackingChain:
    def __init__(self, params, cam, kps, desc, count, uRight, scene, device=0, th_last=7.0, th_local=1.0, exact_order=True):
        """params: capi.FrameParams; cam: dict fx fy cx cy bf; kps / desc / count: the feature pool [nimg][cap] (device); uRight
        [nframes][cap] of the CURRENT frames (mvuRight) or None; scene: dict of host arrays from `make_tracking_scene`
        (curImg, lastImg [B]; lastMP [B][cap] = LastFrame.mvpMapPoints as rows of the frame's map-point table; the table itself:
        nMP [B], mpXw, mpNormal, mpMaxD, mpMinD, mpDesc, mpHasObs [B][mpCap]; pose0 [B][7] = the motion-model pose)."""
        import torch
        self.torch = torch
        self.dev = dev = torch.device("cuda", device)
        self.P, self.cam = params, cam
        self.kps, self.desc, self.count, self.uRight = kps, desc, count, uRight
        self.B = B = len(scene["curImg"])
        self.cap = cap = kps.shape[1]
        self.mpCap = mpCap = scene["mpXw"].shape[1]
        self.th_last, self.th_local = float(th_last), float(th_local)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.curImg, self.lastImg = t(scene["curImg"].astype(np.int32)), t(scene["lastImg"].astype(np.int32))
        self.lastMP = t(scene["lastMP"].astype(np.int32))
        self.nMP = t(scene["nMP"].astype(np.int32))
        self.mpXw, self.mpNormal = t(scene["mpXw"].astype(np.float32)), t(scene["mpNormal"].astype(np.float32))
        self.mpMaxD, self.mpMinD = t(scene["mpMaxD"].astype(np.float32)), t(scene["mpMinD"].astype(np.float32))
        self.mpDesc, self.mpHasObs = t(scene["mpDesc"].astype(np.uint8)), t(scene["mpHasObs"].astype(np.uint8))
        self.pose0 = t(scene["pose0"].astype(np.float32))
        # LastFrame's per-feature view of its map points (what SearchByProjection(Cur, Last) reads through mvpMapPoints[i])
        lm = self.lastMP.long().clamp(min=0)
        self.lastValid = (self.lastMP >= 0).to(torch.uint8).contiguous()
        self.lastXw = torch.gather(self.mpXw, 1, lm[..., None].expand(-1, -1, 3)).contiguous()
        self.lastDesc = torch.gather(self.mpDesc, 1, lm[..., None].expand(-1, -1, 32)).contiguous()
        self.lastHasObs = (torch.gather(self.mpHasObs, 1, lm) * self.lastValid).contiguous()
        self.fwd = torch.zeros((B,), dtype=torch.uint8, device=dev)   # bForward / bBackward (:1538-1539): |tlc.z| <= mb for these scenes
        self.bwd = torch.zeros((B,), dtype=torch.uint8, device=dev)
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.pose = e((B, 7), torch.float32)
        self.matchLast, self.nmLast = e((B, cap), torch.int32), z((B,), torch.int32)
        self.frameMP = e((B, cap), torch.int32)
        self.hasMP, self.obs, self.invS2, self.Xw = e((B, cap), torch.uint8), e((B, cap, 3), torch.float32), e((B, cap), torch.float32), e((B, cap, 3), torch.float32)
        self.po1 = (e((B,), torch.int32), z((B, cap), torch.uint8), e((B, 2), torch.int32))
        self.po2 = (e((B,), torch.int32), z((B, cap), torch.uint8), e((B, 2), torch.int32))
        self.Rcw, self.tcw, self.Ow = e((B, 9), torch.float32), e((B, 3), torch.float32), e((B, 3), torch.float32)
        self.blocked, self.mpSeen = e((B, cap), torch.uint8), e((B, mpCap), torch.uint8)
        self.nm1, self.nmMap1, self.nmLocal = e((B,), torch.int32), e((B,), torch.int32), z((B,), torch.int32)
        self.nInl, self.nInlMap = e((B,), torch.int32), e((B,), torch.int32)
        self.trk = dict(inView=z((B, mpCap), torch.uint8), projX=e((B, mpCap), torch.float32), projY=e((B, mpCap), torch.float32),
                        projXR=e((B, mpCap), torch.float32), depth=e((B, mpCap), torch.float32), level=e((B, mpCap), torch.int32),
                        viewCos=e((B, mpCap), torch.float32))
        self.m_last = ORBmatcher(0.9, True, device=device)     # TrackWithMotionModel: ORBmatcher(0.9, true)
        self.m_local = ORBmatcher(0.8, True, device=device)    # SearchLocalPoints: ORBmatcher(0.8)
        self.opt = Optimizer(device=device)
        self.opt.set_exact_order(bool(exact_order))   # True (the handle's default): g2o's LM path decision for decision; False: tree sums
        self.stream = torch.cuda.Stream(device=dev)
        self._L = lib()

    def close(self):
        self.m_last.close(); self.m_local.close(); self.opt.close()

    def step(self, stream=None, snapshot=False):
        """Enqueue TrackWithMotionModel + TrackLocalMap for the B frames; asynchronous on `stream` (default: the chain's own).
        snapshot: keep copies of every stage's output in self.snap (what tests/tracking_check.py compares with the oracle)."""
        torch, L, P = self.torch, self._L, self.P
        s = self.stream if stream is None else stream
        st = s.cuda_stream
        self.snap = snap = {}

        def keep(**kw):
            if snapshot:
                with torch.cuda.stream(s):
                    for k, v in kw.items():
                        snap[k] = {kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()
        B, cap, mpCap = self.B, self.cap, self.mpCap
        cam = self.cam
        mh = self.m_last._h
        with torch.cuda.stream(s):
            self.pose.copy_(self.pose0)                      # mCurrentFrame.SetPose(mVelocity * mLastFrame.GetPose())
            self.matchLast.fill_(-1)                         # fill(mvpMapPoints, NULL)
        # ---- TrackWithMotionModel
        self.m_last.SearchByProjectionLastFrame(P, self.curImg, self.lastImg, self.kps, self.desc, self.count, self.uRight, None, self.pose,
                                                self.lastValid, self.lastXw, self.lastDesc, self.lastHasObs, self.th_last, self.fwd,
                                                self.bwd, matchCur=self.matchLast, nm=self.nmLast, stream=st)
        check(L.morb_pose_edges_batch(mh, C.byref(P), B, ptr(self.curImg), cap, ptr(self.count), ptr(self.kps), ptr(self.uRight),
                                      ptr(self.matchLast), ptr(self.lastMP), cap, mpCap, ptr(self.mpXw), ptr(self.frameMP),
                                      ptr(self.hasMP), ptr(self.obs), ptr(self.invS2), ptr(self.Xw), C.c_void_p(st)))
        keep(matchLast=self.matchLast, nmLast=self.nmLast, edges1=dict(hasMP=self.hasMP, obs=self.obs, invS2=self.invS2, Xw=self.Xw))
        self.opt.PoseOptimization(self.hasMP, self.obs, self.invS2, self.Xw, self.pose, cam, out=self.po1, stream=st)
        keep(pose1=self.pose, outlier1=self.po1[1], nin1=self.po1[0], stats1=self.po1[2])
        check(L.morb_track_discard_outliers_batch(mh, B, ptr(self.curImg), cap, ptr(self.count), ptr(self.frameMP), ptr(self.po1[1]),
                                                  mpCap, ptr(self.mpHasObs), ptr(self.blocked), ptr(self.mpSeen), ptr(self.nm1),
                                                  ptr(self.nmMap1), C.c_void_p(st)))
        keep(frameMP1=self.frameMP, blocked=self.blocked, mpSeen=self.mpSeen, nm1=self.nm1, nmMap1=self.nmMap1)
        # ---- TrackLocalMap: SearchLocalPoints + PoseOptimization
        check(L.morb_frame_set_pose_batch(mh, B, ptr(self.pose), ptr(self.Rcw), ptr(self.tcw), ptr(self.Ow), C.c_void_p(st)))
        self.m_local.isInFrustum(P, self.Rcw, self.tcw, self.Ow, self.nMP, self.mpXw, self.mpNormal, self.mpMaxD, self.mpMinD, 0.5,
                                 out=self.trk, stream=st)
        # points already seen in the frame are skipped (mnLastFrameSeen == mnId, :3143): mpSeen takes the isBad slot of the search
        self.m_local.SearchByProjectionMapPoints(P, self.curImg, self.kps, self.desc, self.count, self.uRight, self.blocked, self.nMP,
                                                 self.trk, self.mpSeen, self.mpDesc, self.mpHasObs, self.th_local, False, 0.0,
                                                 matchF=self.frameMP, nm=self.nmLocal, stream=st)
        keep(Rcw=self.Rcw, tcw=self.tcw, Ow=self.Ow, trk=self.trk, frameMP2=self.frameMP)
        check(L.morb_pose_edges_batch(mh, C.byref(P), B, ptr(self.curImg), cap, ptr(self.count), ptr(self.kps), ptr(self.uRight),
                                      None, None, 0, mpCap, ptr(self.mpXw), ptr(self.frameMP), ptr(self.hasMP), ptr(self.obs),
                                      ptr(self.invS2), ptr(self.Xw), C.c_void_p(st)))
        keep(edges2=dict(hasMP=self.hasMP, obs=self.obs, invS2=self.invS2, Xw=self.Xw))
        self.opt.PoseOptimization(self.hasMP, self.obs, self.invS2, self.Xw, self.pose, cam, out=self.po2, stream=st)
        keep(outlier2=self.po2[1])
        check(L.morb_track_discard_outliers_batch(mh, B, ptr(self.curImg), cap, ptr(self.count), ptr(self.frameMP), ptr(self.po2[1]),
                                                  mpCap, ptr(self.mpHasObs), None, None, ptr(self.nInl), ptr(self.nInlMap),
                                                  C.c_void_p(st)))

    def sync(self):
        self.stream.synchronize()


class KeyframeSearches:
    """LocalMapping's two searches per new keyframe, batched over keyframe pairs: CreateNewMapPoints' SearchForTriangulation
    (LocalMapping.cc:473) and SearchInNeighbors' Fuse (:771-802)."""

    def __init__(self, params, kps, desc, node, count, uRight_img, scene, device=0):
        """scene: dict of host arrays from `make_keyframe_scene` (img1, img2 [npairs]; hasMP [nimg][cap]; R12, t12, ep per pair;
        Fuse: kfImg [nprob], Tcw [nprob][7], Ow [nprob][3], nMP [nprob], valid / Pw / normal / maxD / minD / mpDesc [nprob][mpCap])."""
        import torch
        self.torch = torch
        dev = torch.device("cuda", device)
        self.P = params
        self.kps, self.desc, self.node, self.count, self.uRight = kps, desc, node, count, uRight_img
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.img1, self.img2 = t(scene["img1"].astype(np.int32)), t(scene["img2"].astype(np.int32))
        self.hasMP = t(scene["hasMP"].astype(np.uint8))
        self.R12, self.t12, self.ep = (np.ascontiguousarray(scene[k], np.float32) for k in ("R12", "t12", "ep"))
        self.kfImg = t(scene["kfImg"].astype(np.int32))
        self.fuse = {k: t(scene[k]) for k in ("Tcw", "Ow", "nMP", "valid", "Pw", "normal", "maxD", "minD", "mpDesc")}
        self.fuseUR = t(scene["fuseUR"].astype(np.float32))
        self.m = ORBmatcher(0.6, False, device=device)         # CreateNewMapPoints: ORBmatcher(0.6, false)
        self.stream = torch.cuda.Stream(device=dev)
        npairs, cap, mpCap = len(scene["img1"]), kps.shape[1], scene["Pw"].shape[1]
        self.tri = (torch.empty((npairs, cap), dtype=torch.int32, device=dev), torch.zeros((npairs,), dtype=torch.int32, device=dev))
        self.fused = (torch.empty((npairs, mpCap), dtype=torch.int32, device=dev), torch.empty((npairs, mpCap), dtype=torch.int32, device=dev))

    def close(self):
        self.m.close()

    def step(self, stream=None):
        s = self.stream if stream is None else stream
        st = s.cuda_stream
        f = self.fuse
        self.tri = self.m.SearchForTriangulation(self.P, self.img1, self.img2, self.kps, self.desc, self.node, self.count, self.hasMP,
                                                 self.uRight, self.R12, self.t12, self.ep, False, False, out=self.tri, stream=st)
        self.fused = self.m.Fuse(self.P, self.kfImg, self.kps, self.desc, self.count, self.fuseUR, f["Tcw"], f["Ow"], f["nMP"], f["valid"],
                                 f["Pw"], f["normal"], f["maxD"], f["minD"], f["mpDesc"], 3.0, out=self.fused, stream=st)

    def sync(self):
        self.stream.synchronize()


# ---- synthetic scenes (host side, set-up only) --------------------------------------------------------------------------------
def _backproject(P, x, y, z):
    return np.stack([(x - P.cx) * z / P.fx, (y - P.cy) * z / P.fy, z], 1).astype(np.float32)


def make_tracking_scene(P, kps_host, cnt_host, desc_host, depth_host, frames, shift_px=(3, 2), mp_cap=2048, seed=0, extra=(2, 3)):
    """Map + motion model for a batch of tracked frames taken from sequences of globally shifted stereo frames (bench.make_batch).

    frames: list of (cur, last, [older...]) GLOBAL stereo-frame indices of one sequence (left image of frame g = pool row 2 g);
    frame g is the base scene shifted by g * shift_px.  World = the LAST frame's camera (identity pose).  The frame's map-point table:
    first the last frame's stereo points (depth > 0) in feature order — LastFrame.mvpMapPoints — then the stereo points of the older
    frames (the other local keyframes), each moved by its pixel offset to the last frame so that it projects where the scene now is.
    pose0 = identity perturbed by a small rotation / translation (the motion-model prediction)."""
    rng = np.random.default_rng(0x7ac0 + seed)
    B, cap = len(frames), kps_host.shape[1]
    out = dict(curImg=np.zeros(B, np.int32), lastImg=np.zeros(B, np.int32), lastMP=np.full((B, cap), -1, np.int32), nMP=np.zeros(B, np.int32),
               mpXw=np.zeros((B, mp_cap, 3), np.float32), mpNormal=np.zeros((B, mp_cap, 3), np.float32), mpMaxD=np.ones((B, mp_cap), np.float32),
               mpMinD=np.ones((B, mp_cap), np.float32), mpDesc=np.zeros((B, mp_cap, 32), np.uint8), mpHasObs=np.zeros((B, mp_cap), np.uint8),
               pose0=np.zeros((B, 7), np.float32))
    sf = np.array(list(P.scaleFactors)[:P.nlevels], np.float32)
    for f, (cur, last, older) in enumerate(frames):
        out["curImg"][f], out["lastImg"][f] = 2 * cur, 2 * last
        n = 0
        for k, g in enumerate([last] + list(older)):
            img = 2 * g
            N = int(cnt_host[img])
            kk = kps_host[img, :N]
            z = depth_host[g, :N]
            v = z > 0
            dx, dy = (last - g) * shift_px[0], (last - g) * shift_px[1]
            X = _backproject(P, kk["x"][v] + np.float32(dx), kk["y"][v] + np.float32(dy), z[v])
            m = min(len(X), mp_cap - n)
            if m <= 0:
                break
            idx = np.nonzero(v)[0][:m]
            X = X[:m]
            if k == 0:
                out["lastMP"][f, idx] = np.arange(n, n + m)
            dist = np.linalg.norm(X, axis=1).astype(np.float32)
            nrm = X / dist[:, None] + rng.normal(0, 0.1, X.shape)
            out["mpXw"][f, n:n + m] = X
            out["mpNormal"][f, n:n + m] = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
            lvl = kk["octave"][idx]
            maxD = (dist * sf[lvl] * rng.uniform(1.0, 1.25, m)).astype(np.float32)         # mfMaxDistance = dist * scale^level (MapPoint.cc:437-470)
            out["mpMaxD"][f, n:n + m] = maxD
            out["mpMinD"][f, n:n + m] = (maxD / sf[P.nlevels - 1]).astype(np.float32)
            out["mpDesc"][f, n:n + m] = desc_host[img, idx]
            out["mpHasObs"][f, n:n + m] = (rng.random(m) < 0.95).astype(np.uint8)
            n += m
        out["nMP"][f] = n
        rv = rng.normal(0, 0.001, 3)
        q = np.concatenate([rv / 2, [1.0]]); q /= np.linalg.norm(q)
        out["pose0"][f] = np.concatenate([q, rng.normal(0, 0.005, 3)]).astype(np.float32)
    return out


def make_keyframe_scene(P, kps_host, cnt_host, desc_host, depth_host, uRight_host, pairs, shift_px=(3, 2), mp_cap=2048, seed=0):
    """Keyframe pairs (g1, g2) of one sequence for SearchForTriangulation (pure-translation relative poses consistent with a
    lateral baseline) and for Fuse (keyframe g1's stereo points fused into keyframe g2)."""
    rng = np.random.default_rng(0xf05e + seed)
    nimg, cap = kps_host.shape[0], kps_host.shape[1]
    npairs = len(pairs)
    out = dict(img1=np.array([2 * a for a, _ in pairs], np.int32), img2=np.array([2 * b for _, b in pairs], np.int32),
               hasMP=(rng.random((nimg, cap)) < 0.6).astype(np.uint8),           # keyframe features that already hold a map point
               R12=np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (npairs, 1)), t12=np.zeros((npairs, 3), np.float32),
               ep=np.zeros((npairs, 2), np.float32), kfImg=np.array([2 * b for _, b in pairs], np.int32),
               Tcw=np.zeros((npairs, 7), np.float32), Ow=np.zeros((npairs, 3), np.float32), nMP=np.zeros(npairs, np.int32),
               valid=np.zeros((npairs, mp_cap), np.uint8), Pw=np.zeros((npairs, mp_cap, 3), np.float32),
               normal=np.zeros((npairs, mp_cap, 3), np.float32), maxD=np.ones((npairs, mp_cap), np.float32), minD=np.ones((npairs, mp_cap), np.float32),
               mpDesc=np.zeros((npairs, mp_cap, 32), np.uint8), fuseUR=np.full((npairs, cap), -1, np.float32))
    sf = np.array(list(P.scaleFactors)[:P.nlevels], np.float32)
    for p, (a, b) in enumerate(pairs):
        # camera 2 displaced sideways / forwards from camera 1 (t12 = position of camera 2's origin in camera 1's frame, negated: x1 = R12 x2 + t12)
        t = np.array([0.05 * (b - a), 0.01, 0.02], np.float32)
        out["t12"][p] = t
        # epipole of camera 1 in image 2 (LocalMapping.cc:455-470: pKF2->mpCamera->project(R2w * Ow1 + t2w))
        c2 = -t                                                  # camera 1's centre in camera 2's frame (R12 = I)
        out["ep"][p] = (P.fx * c2[0] / c2[2] + P.cx, P.fy * c2[1] / c2[2] + P.cy)
        # Fuse: map points = keyframe a's stereo points, expressed in keyframe b's frame through the pixel offset
        img = 2 * a
        N = int(cnt_host[img]); kk = kps_host[img, :N]; z = depth_host[a, :N]; v = z > 0
        dx, dy = (b - a) * shift_px[0], (b - a) * shift_px[1]
        X = _backproject(P, kk["x"][v] + np.float32(dx), kk["y"][v] + np.float32(dy), z[v])[:mp_cap]
        m = len(X); idx = np.nonzero(v)[0][:m]
        dist = np.linalg.norm(X, axis=1).astype(np.float32)
        out["nMP"][p] = m
        out["valid"][p, :m] = (rng.random(m) < 0.95).astype(np.uint8)
        out["Pw"][p, :m] = X
        out["normal"][p, :m] = (X / dist[:, None]).astype(np.float32)
        maxD = (dist * sf[kk["octave"][idx]] * rng.uniform(1.0, 1.25, m)).astype(np.float32)
        out["maxD"][p, :m] = maxD; out["minD"][p, :m] = (maxD / sf[P.nlevels - 1]).astype(np.float32)
        out["mpDesc"][p, :m] = desc_host[img, idx]
        out["Tcw"][p] = (0, 0, 0, 1, 0, 0, 0)
        out["fuseUR"][p] = uRight_host[b]
    return out


def build_chains(images, B, npairs=20, nfeatures=1200, device=0, seq_len=64, shift_px=(3, 2), mp_cap=2048, vocab=(10, 6, 4), seed=0,
                 exact_order=True, fx=458.654, fy=457.296, cx=367.215, cy=248.375, baseline=0.11):
    """Set-up shared by bench.py, tools/ and tests/: extract the stereo stream `images` (uint8 host array [G, 2, H, W]: sequences of
    `seq_len` frames, frame g = its sequence's base scene shifted by (g % seq_len) * shift_px), match it in stereo, take the BoW node
    ids, then build a TrackingChain of B frames (frame f tracks stream frame cur against cur - 1 with cur - 2, cur - 3 as the other
    local keyframes) and a KeyframeSearches of `npairs` keyframe pairs.  Returns (chain, keyframe_searches, host) where host holds
    the host copies the checkers need."""
    import torch
    from .capi import KP_DTYPE, make_frame_params
    from .extractor import ORBextractor
    from .synth import make_vocabulary
    dev = torch.device("cuda", device)
    G, _, H, W = images.shape
    ext = ORBextractor(nfeatures, 1.2, 8, 20, 7, device=device)
    m = ORBmatcher(0.7, True, device=device)
    mbf, mb = np.float32(fx * baseline), np.float32(baseline)
    d = torch.from_numpy(images.reshape(2 * G, H, W)).to(dev)
    kps, desc, cnt, _ = ext.extract_batch(d)
    uR, dep = m.ComputeStereoMatches(ext, kps, desc, cnt, mbf, mb)
    vd, vf = make_vocabulary(vocab[0], vocab[1], seed=0)
    _, node = m.bow_transform(desc, cnt, torch.from_numpy(vd).to(dev), torch.from_numpy(vf).to(dev), vocab[0], vocab[1], vocab[2])
    torch.cuda.synchronize(dev)
    P = make_frame_params(W, H, fx, fy, cx, cy, float(mbf), float(mb), ext.GetScaleFactors(), ext.GetScaleSigmaSquares())
    cam = dict(fx=fx, fy=fy, cx=cx, cy=cy, bf=float(mbf))
    cap = kps.shape[1]
    host = dict(kps=kps.cpu().numpy().reshape(2 * G, cap, 28).view(KP_DTYPE).reshape(2 * G, cap), desc=desc.cpu().numpy(), cnt=cnt.cpu().numpy(),
                uR=uR.cpu().numpy(), dep=dep.cpu().numpy(), node=node.cpu().numpy(), P=P, cam=cam, voc=(vd, vf))
    nseq = max(G // seq_len, 1)
    span = min(seq_len, G)
    frames = []
    for f in range(B):
        s, k = (f // (span - 3)) % nseq, 3 + f % (span - 3)
        cur = s * seq_len + k
        frames.append((cur, cur - 1, (cur - 2, cur - 3)))
    scene = make_tracking_scene(P, host["kps"], host["cnt"], host["desc"], host["dep"], frames, shift_px, mp_cap, seed)
    curUR = uR[torch.from_numpy(scene["curImg"] // 2).long().to(dev)].contiguous()          # mvuRight of the current frames
    host["curUR"] = host["uR"][scene["curImg"] // 2]
    chain = TrackingChain(P, cam, kps, desc, cnt, curUR, scene, device=device, exact_order=exact_order)
    pairs = []
    for p in range(npairs):
        s, k = (p // (span - 2)) % nseq, 1 + p % (span - 2)
        pairs.append((s * seq_len + k - 1, s * seq_len + k + 1))       # a new keyframe against a covisible neighbour two frames on
    kscene = make_keyframe_scene(P, host["kps"], host["cnt"], host["desc"], host["dep"], host["uR"], pairs, shift_px, mp_cap, seed)
    uR_img = torch.full((2 * G, cap), -1.0, dtype=torch.float32, device=dev)
    uR_img[0::2] = uR
    host["uR_img"] = uR_img.cpu().numpy()
    ks = KeyframeSearches(P, kps, desc, node, cnt, uR_img, kscene, device=device)
    host["scene"], host["kscene"] = scene, kscene
    chain._keep = (ext, m)      # the pool's owners
    return chain, ks, host
