"""ORBmatcher and the Frame-level stereo matchers — host-side mirror of the reference surfaces
(include/ORBmatcher.h:36-129, Frame::ComputeStereoMatches / ComputeStereoFishEyeMatches) over the HIP kernels.
Inputs are torch device tensors in the batched layout morb_extract_batch writes ([nimg, cap, ...])."""
import ctypes as C

import numpy as np

from .capi import check, lib, ptr, stream_arg

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30   # ORBmatcher.cc:35-37


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.morb_matcher_create(C.byref(self._h), device))
        self.mfNNratio, self.mbCheckOrientation, self.device = float(nnratio), bool(checkOri), device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.morb_matcher_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def _st(stream):
        return stream_arg(stream)

    def DescriptorDistance(self, a, b, stream=None):
        """static DescriptorDistance on n pairs: a, b = uint8 device tensors [n, 32] -> int32 [n]."""
        import torch
        n = a.shape[0]
        out = torch.empty((n,), dtype=torch.int32, device=a.device)
        check(self._L.morb_hamming_pairs(self._h, ptr(a), ptr(b), n, ptr(out), self._st(stream)))
        return out

    def knn2(self, query, nq, train, nt, qoff=None, toff=None, stream=None):
        """BFMatcher(NORM_HAMMING).knnMatch(k=2), batched: query/train [nprob, pitch, 32] u8, nq/nt int32 [nprob]."""
        import torch
        nprob, qp = query.shape[0], query.shape[1]
        idx = torch.full((nprob, qp, 2), -1, dtype=torch.int32, device=query.device)
        dist = torch.full((nprob, qp, 2), -1, dtype=torch.int32, device=query.device)
        check(self._L.morb_hamming_knn2_batch(self._h, nprob, ptr(query), ptr(nq), qp, ptr(qoff), ptr(train), ptr(nt),
                                              train.shape[1], ptr(toff), ptr(idx), ptr(dist), self._st(stream)))
        return idx, dist

    def slab_bytes(self, S, cap):
        return int(self._L.morb_feature_slab_bytes(S, cap))

    def pack_slab(self, kps, desc, count, node=None, rows=None, out=None, stream=None):
        """morb_feature_slab_pack: rows `rows` (int32 device tensor, None = all) of the [nimg, cap] feature arrays -> ONE contiguous uint8 slab."""
        import torch
        cap = kps.shape[1]
        S = int(rows.shape[0]) if rows is not None else kps.shape[0]
        nb = self.slab_bytes(S, cap)
        if out is None or out.numel() != nb:
            out = torch.empty((nb,), dtype=torch.uint8, device=kps.device)
        check(self._L.morb_feature_slab_pack(self._h, S, cap, ptr(rows), ptr(kps), ptr(desc), ptr(node), ptr(count), ptr(out), self._st(stream)))
        return out

    def unpack_slab(self, slab, S, kps, desc, count, node=None, rows=None, stream=None):
        """morb_feature_slab_unpack: the slab's S frames -> rows `rows` (None = 0 .. S-1) of the receiver's [nimg, cap] arrays (in place)."""
        cap = kps.shape[1]
        assert slab.numel() == self.slab_bytes(S, cap)
        check(self._L.morb_feature_slab_unpack(self._h, S, cap, ptr(slab), ptr(rows), ptr(kps), ptr(desc), ptr(node), ptr(count), self._st(stream)))

    def ComputeStereoMatches(self, extractor, kps, desc, count, mbf, mb, out=None, stream=None):
        """Frame::ComputeStereoMatches for nframes = nimg/2 frames (left = image 2f, right = 2f+1 of the batch the
        extractor just processed).  Returns (mvuRight, mvDepth) float32 [nframes, cap]."""
        import torch
        nimg, cap = kps.shape[0], kps.shape[1]
        nf = nimg // 2
        if out is None:
            out = (torch.empty((nf, cap), dtype=torch.float32, device=kps.device),
                   torch.empty((nf, cap), dtype=torch.float32, device=kps.device))
        check(self._L.morb_stereo_match_batch(self._h, extractor._h, nf, ptr(kps), ptr(desc), ptr(count), cap,
                                              float(mbf), float(mb), ptr(out[0]), ptr(out[1]), self._st(stream)))
        return out

    def bow_transform(self, desc, count, voc_desc, voc_first, k, L, levelsup=4, out=None, stream=None):
        """DBoW2 transform (Frame::ComputeBoW): returns (wordId, nodeId) int32 [nimg, cap]."""
        import torch
        nimg, cap = desc.shape[0], desc.shape[1]
        if out is None:
            out = (torch.empty((nimg, cap), dtype=torch.int32, device=desc.device),
                   torch.empty((nimg, cap), dtype=torch.int32, device=desc.device))
        check(self._L.morb_bow_transform_batch(self._h, nimg, ptr(desc), ptr(count), cap, ptr(voc_desc), ptr(voc_first),
                                               k, L, levelsup, ptr(out[0]), ptr(out[1]), self._st(stream)))
        return out

    def bow_vector(self, leaf, count, node_weight, node_word=None, weighting=0, scoring=0, out=None, stream=None):
        """The BowVector of TemplatedVocabulary::transform for a batch: leaf i32 [nimg, cap] (bow_transform's first output),
        node_weight f64 [nNodes] (device), node_word i32 [nNodes] or None.  weighting / scoring: DBoW2 enum values (TF_IDF = 0,
        L1_NORM = 0: ORBvoc).  Returns (word i32 [nimg, cap], value f64 [nimg, cap], count i32 [nimg]), words ascending."""
        import torch
        nimg, cap = leaf.shape
        if out is None:
            out = (torch.empty((nimg, cap), dtype=torch.int32, device=leaf.device), torch.empty((nimg, cap), dtype=torch.float64, device=leaf.device),
                   torch.empty((nimg,), dtype=torch.int32, device=leaf.device))
        assert node_weight.dtype == torch.float64
        check(self._L.morb_bow_vector_batch(self._h, nimg, ptr(leaf), ptr(count), cap, ptr(node_word), ptr(node_weight), int(weighting),
                                            int(scoring), ptr(out[0]), ptr(out[1]), ptr(out[2]), self._st(stream)))
        return out

    def bow_transform_tree(self, desc, count, voc_desc, voc_first, voc_nchild, L, levelsup=4, stream=None):
        """DBoW2 transform on a trained (possibly incomplete) tree: children of n = [first[n], first[n] + nchild[n])."""
        import torch
        nimg, cap = desc.shape[0], desc.shape[1]
        out = (torch.empty((nimg, cap), dtype=torch.int32, device=desc.device), torch.empty((nimg, cap), dtype=torch.int32, device=desc.device))
        check(self._L.morb_bow_transform_tree_batch(self._h, nimg, ptr(desc), ptr(count), cap, ptr(voc_desc), ptr(voc_first), ptr(voc_nchild),
                                                    L, levelsup, ptr(out[0]), ptr(out[1]), self._st(stream)))
        return out

    def SearchByBoW(self, kf_img, f_img, kps, desc, node, count, has_mp, out=None, stream=None, nLeft=None):
        """SearchByBoW(pKF, F, vpMapPointMatches) for pairs (kf_img[p], f_img[p]) of images of one pool.
        Returns (matchF int32 [npairs, cap] = keyframe feature index per frame feature or -1, nmatches int32 [npairs]).
        nLeft: int32 [npairs] F.Nleft per pair for fisheye frames (left features first, then right), None = pinhole."""
        import torch
        npairs = kf_img.shape[0]
        nimg, cap = kps.shape[0], kps.shape[1]
        if out is None:
            out = (torch.empty((npairs, cap), dtype=torch.int32, device=kps.device),
                   torch.empty((npairs,), dtype=torch.int32, device=kps.device))
        if nLeft is not None:
            check(self._L.morb_search_by_bow_fisheye_batch(self._h, npairs, ptr(kf_img), ptr(f_img), ptr(nLeft), nimg, ptr(kps),
                                                           ptr(desc), ptr(node), ptr(count), ptr(has_mp), cap, self.mfNNratio,
                                                           1 if self.mbCheckOrientation else 0, ptr(out[0]), ptr(out[1]),
                                                           self._st(stream)))
            return out
        check(self._L.morb_search_by_bow_batch(self._h, npairs, ptr(kf_img), ptr(f_img), nimg, ptr(kps), ptr(desc),
                                               ptr(node), ptr(count), ptr(has_mp), cap, self.mfNNratio,
                                               1 if self.mbCheckOrientation else 0, ptr(out[0]), ptr(out[1]),
                                               self._st(stream)))
        return out

    def SearchByBoWKeyFrames(self, kf1_img, kf2_img, kps, desc, node, count, has_mp, nValid=None, stream=None):
        """SearchByBoW(pKF1, pKF2, vpMatches12): match12 int32 [npairs, cap] (index of the pKF2 feature per pKF1 feature)."""
        import torch
        npairs = kf1_img.shape[0]
        nimg, cap = kps.shape[0], kps.shape[1]
        out = (torch.empty((npairs, cap), dtype=torch.int32, device=kps.device), torch.empty((npairs,), dtype=torch.int32, device=kps.device))
        check(self._L.morb_search_by_bow_kfkf_batch(self._h, npairs, ptr(kf1_img), ptr(kf2_img), ptr(nValid), nimg, ptr(kps), ptr(desc),
                                                    ptr(node), ptr(count), ptr(has_mp), cap, self.mfNNratio,
                                                    1 if self.mbCheckOrientation else 0, ptr(out[0]), ptr(out[1]), self._st(stream)))
        return out

    # ---- Frame-side helpers of the non-rectified / RGB-D input paths ----------------------------
    def UndistortKeyPoints(self, kps, count, cam, dist, out=None, stream=None):
        """Frame::UndistortKeyPoints for a batch: kps u8 [F, cap, 28] (cv::KeyPoint records), count i32 [F] or None, dist =
        (k1, k2, p1, p2[, k3]) as in mDistCoef.  Returns the mvKeysUn records [F, cap, 28]."""
        import torch
        F, cap = kps.shape[0], kps.shape[1]
        if out is None:
            out = torch.zeros_like(kps)
        d5 = np.zeros(5, np.float32); d5[:len(dist)] = np.asarray(dist, np.float32)
        st = stream_arg(stream)
        check(self._L.morb_undistort_keypoints_batch(self._h, F, cap, ptr(count), ptr(kps), cam["fx"], cam["fy"], cam["cx"], cam["cy"], ptr(d5),
                                                     ptr(out), st))
        return out

    def ComputeStereoFromRGBD(self, kps, kpsUn, count, depth, bf, out=None, stream=None):
        """Frame::ComputeStereoFromRGBD: depth f32 [F, H, W] (contiguous).  Returns (mvuRight, mvDepth) f32 [F, cap]."""
        import torch
        F, cap = kps.shape[0], kps.shape[1]
        H, W = depth.shape[1], depth.shape[2]
        if out is None:
            out = (torch.empty((F, cap), dtype=torch.float32, device=kps.device), torch.empty((F, cap), dtype=torch.float32, device=kps.device))
        st = stream_arg(stream)
        check(self._L.morb_stereo_from_rgbd_batch(self._h, F, cap, ptr(count), ptr(kps), ptr(kpsUn), ptr(depth), W, H, W, H * W, float(bf),
                                                  ptr(out[0]), ptr(out[1]), st))
        return out

    @staticmethod
    def ComputeImageBounds(width, height, cam, dist):
        """Frame::ComputeImageBounds -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
        d5 = np.zeros(5, np.float32); d5[:len(dist)] = np.asarray(dist, np.float32)
        b = np.zeros(4, np.float32)
        check(lib().morb_image_bounds(width, height, cam["fx"], cam["fy"], cam["cx"], cam["cy"], ptr(d5), ptr(b)))
        return tuple(float(x) for x in b)

    def ComputeDistinctiveDescriptors(self, start, desc, stream=None):
        """MapPoint::ComputeDistinctiveDescriptors for many map points: start int32 [nMP + 1] (CSR), desc uint8 [total, 32];
        returns int32 [nMP] = row (within each point) of its most representative descriptor."""
        import torch
        nMP = start.shape[0] - 1
        out = torch.empty((nMP,), dtype=torch.int32, device=desc.device)
        check(self._L.morb_distinctive_descriptors_batch(self._h, nMP, ptr(start), ptr(desc), ptr(out), self._st(stream)))
        return out

    # ---- M7: loop-closing / local-mapping searches ---------------------------------------------------------
    def Fuse(self, params, kfImg, kps, desc, count, uRight, Tcw, Ow, nMP, valid, Pw, normal, maxDist, minDist, mpDesc, th=3.0,
             sim3Form=False, cam8=None, jLo=None, jHi=None, out=None, stream=None):
        """The search of ORBmatcher::Fuse (both overloads): (bestIdx, bestDist) int32 [nprob, mpCap]."""
        import torch
        F, cap, mpCap = kfImg.shape[0], kps.shape[1], mpDesc.shape[1]
        if out is not None:
            bi, bd = out
        else:
            bi = torch.empty((F, mpCap), dtype=torch.int32, device=kps.device); bd = torch.empty_like(bi)
        cam = None if cam8 is None else np.ascontiguousarray(cam8, np.float32)
        check(self._L.morb_fuse_batch(self._h, C.byref(params), F, ptr(kfImg), cap, ptr(count), ptr(kps), ptr(desc), ptr(uRight), ptr(Tcw),
                                      ptr(Ow), ptr(cam), ptr(jLo), ptr(jHi), mpCap, ptr(nMP), ptr(valid), ptr(Pw), ptr(normal),
                                      ptr(maxDist), ptr(minDist), ptr(mpDesc), float(th), 1 if sim3Form else 0, ptr(bi), ptr(bd),
                                      self._st(stream)))
        return bi, bd

    def SearchByProjectionSim3(self, params, kfImg, kps, desc, count, Tcw, Ow, nMP, valid, Pw, normal, maxDist, minDist, mpDesc, matched,
                               th, ratioHamming=1.0, manualProjection=False, stream=None, cam8=None, nLeft=None):
        """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) (and the vpPointsKFs twin).  nLeft (i32 [F], device): the keyframes are
        KannalaBrandt8 rig keyframes — left features only, cam8 = the left camera's parameters for the first form."""
        import torch
        F, cap, mpCap = kfImg.shape[0], kps.shape[1], mpDesc.shape[1]
        mf = torch.empty((F, cap), dtype=torch.int32, device=kps.device); nm = torch.zeros((F,), dtype=torch.int32, device=kps.device)
        if nLeft is not None:
            cam = None if cam8 is None else np.ascontiguousarray(cam8, np.float32)
            check(self._L.morb_search_by_projection_sim3_rig_batch(self._h, C.byref(params), F, ptr(kfImg), cap, ptr(count), ptr(kps), ptr(desc),
                                                                   ptr(Tcw), ptr(Ow), mpCap, ptr(nMP), ptr(valid), ptr(Pw), ptr(normal),
                                                                   ptr(maxDist), ptr(minDist), ptr(mpDesc), ptr(matched), int(th),
                                                                   float(ratioHamming), 1 if manualProjection else 0, ptr(cam), ptr(nLeft), ptr(mf),
                                                                   ptr(nm), self._st(stream)))
            return mf, nm
        check(self._L.morb_search_by_projection_sim3_batch(self._h, C.byref(params), F, ptr(kfImg), cap, ptr(count), ptr(kps), ptr(desc),
                                                           ptr(Tcw), ptr(Ow), mpCap, ptr(nMP), ptr(valid), ptr(Pw), ptr(normal),
                                                           ptr(maxDist), ptr(minDist), ptr(mpDesc), ptr(matched), int(th),
                                                           float(ratioHamming), 1 if manualProjection else 0, ptr(mf), ptr(nm),
                                                           self._st(stream)))
        return mf, nm

    def SearchBySim3(self, params, kf1, kf2, kps, desc, count, T1w, T2w, S12, S21, valid1, Pw1, maxD1, minD1, mpDesc1, valid2, Pw2, maxD2,
                     minD2, mpDesc2, th, stream=None, nLeft1=None, nLeft2=None):
        """SearchBySim3(pKF1, pKF2, vpMatches12, S12, th): (vnMatch1, vnMatch2, match12, nFound).  nLeft1 / nLeft2 (i32 [F], device): rig keyframes."""
        import torch
        F, cap = kf1.shape[0], kps.shape[1]
        v1 = torch.empty((F, cap), dtype=torch.int32, device=kps.device); v2 = torch.empty_like(v1); m12 = torch.empty_like(v1)
        nf = torch.zeros((F,), dtype=torch.int32, device=kps.device)
        if nLeft1 is not None:
            check(self._L.morb_search_by_sim3_rig_batch(self._h, C.byref(params), F, ptr(kf1), ptr(kf2), cap, ptr(count), ptr(kps), ptr(desc),
                                                        ptr(T1w), ptr(T2w), ptr(S12), ptr(S21), ptr(valid1), ptr(Pw1), ptr(maxD1), ptr(minD1),
                                                        ptr(mpDesc1), ptr(valid2), ptr(Pw2), ptr(maxD2), ptr(minD2), ptr(mpDesc2), float(th),
                                                        ptr(nLeft1), ptr(nLeft2), ptr(v1), ptr(v2), ptr(m12), ptr(nf), self._st(stream)))
            return v1, v2, m12, nf
        check(self._L.morb_search_by_sim3_batch(self._h, C.byref(params), F, ptr(kf1), ptr(kf2), cap, ptr(count), ptr(kps), ptr(desc),
                                                ptr(T1w), ptr(T2w), ptr(S12), ptr(S21), ptr(valid1), ptr(Pw1), ptr(maxD1), ptr(minD1),
                                                ptr(mpDesc1), ptr(valid2), ptr(Pw2), ptr(maxD2), ptr(minD2), ptr(mpDesc2), float(th),
                                                ptr(v1), ptr(v2), ptr(m12), ptr(nf), self._st(stream)))
        return v1, v2, m12, nf

    # ---- projection-guided searches (projection.hip) ------------------------------------------------------
    def isInFrustum(self, params, Rcw, tcw, Ow, nMP, Pw, normal, maxDist, minDist, viewingCosLimit=0.5, out=None, stream=None):
        """Frame::isInFrustum for [F, mpCap] map points; returns dict of the MapPoint tracking fields (device tensors).
        out: a dict from an earlier call to write into (every field of every row below nMP is written)."""
        import torch
        F, mpCap = Pw.shape[0], Pw.shape[1]
        dev = Pw.device
        o = out if out is not None else dict(inView=torch.zeros((F, mpCap), dtype=torch.uint8, device=dev),
                 projX=torch.full((F, mpCap), -1.0, device=dev), projY=torch.full((F, mpCap), -1.0, device=dev),
                 projXR=torch.full((F, mpCap), -1.0, device=dev), depth=torch.full((F, mpCap), -1.0, device=dev),
                 level=torch.full((F, mpCap), -1, dtype=torch.int32, device=dev), viewCos=torch.full((F, mpCap), -1.0, device=dev))
        check(self._L.morb_is_in_frustum_batch(self._h, C.byref(params), F, ptr(Rcw), ptr(tcw), ptr(Ow), mpCap, ptr(nMP), ptr(Pw),
                                               ptr(normal), ptr(maxDist), ptr(minDist), float(viewingCosLimit), ptr(o["inView"]),
                                               ptr(o["projX"]), ptr(o["projY"]), ptr(o["projXR"]), ptr(o["depth"]), ptr(o["level"]),
                                               ptr(o["viewCos"]), self._st(stream)))
        return o

    def SearchByProjectionMapPoints(self, params, fImg, kps, desc, count, uRight, blocked, nMP, trk, isBad, mpDesc, mpHasObs,
                                    th=1.0, bFarPoints=False, thFarPoints=50.0, matchF=None, nm=None, stream=None):
        """SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints); trk = dict from isInFrustum."""
        import torch
        F, cap = fImg.shape[0], kps.shape[1]
        mpCap = mpDesc.shape[1]
        if matchF is None:
            matchF = torch.full((F, cap), -1, dtype=torch.int32, device=kps.device)
        if nm is None:
            nm = torch.zeros((F,), dtype=torch.int32, device=kps.device)
        check(self._L.morb_search_by_projection_mps_batch(
            self._h, C.byref(params), F, ptr(fImg), cap, ptr(count), ptr(kps), ptr(desc), ptr(uRight), ptr(blocked), mpCap, ptr(nMP),
            ptr(trk["inView"]), ptr(isBad), ptr(trk["depth"]), ptr(trk["projX"]), ptr(trk["projY"]), ptr(trk["projXR"]),
            ptr(trk["level"]), ptr(trk["viewCos"]), ptr(mpDesc), ptr(mpHasObs), float(th), 1 if bFarPoints else 0,
            float(thFarPoints), self.mfNNratio, ptr(matchF), ptr(nm), self._st(stream)))
        return matchF, nm

    def isInFrustumChecks(self, params, cam8, R, t, twc, nMP, Pw, normal, maxDist, minDist, viewingCosLimit=0.5, stream=None):
        """Frame::isInFrustumChecks for one camera of a KannalaBrandt8 rig (cam8 = fx fy cx cy k0..k3; R, t, twc as
        Frame.cc:1283-1293 builds them).  Returns the tracking fields of that camera (device tensors)."""
        import torch
        F, mpCap = Pw.shape[0], Pw.shape[1]
        dev = Pw.device
        o = dict(inView=torch.zeros((F, mpCap), dtype=torch.uint8, device=dev),
                 projX=torch.full((F, mpCap), -1.0, device=dev), projY=torch.full((F, mpCap), -1.0, device=dev),
                 depth=torch.full((F, mpCap), -1.0, device=dev),
                 level=torch.full((F, mpCap), -1, dtype=torch.int32, device=dev), viewCos=torch.full((F, mpCap), -1.0, device=dev))
        cam = np.ascontiguousarray(cam8, np.float32)
        check(self._L.morb_is_in_frustum_kb8_batch(self._h, C.byref(params), ptr(cam), F, ptr(R), ptr(t), ptr(twc), mpCap, ptr(nMP),
                                                   ptr(Pw), ptr(normal), ptr(maxDist), ptr(minDist), float(viewingCosLimit),
                                                   ptr(o["inView"]), ptr(o["projX"]), ptr(o["projY"]), ptr(o["depth"]),
                                                   ptr(o["level"]), ptr(o["viewCos"]), self._st(stream)))
        return o

    def SearchByProjectionMapPointsFisheye(self, params, fImg, kps, desc, count, nLeft, l2r, r2l, blocked, nMP, trkL, trkR, isBad,
                                           mpDesc, mpHasObs, th=1.0, bFarPoints=False, thFarPoints=50.0, matchF=None, stream=None):
        """SearchByProjection(F, vpMapPoints, ...) with F.Nleft != -1; trkL / trkR = dicts from isInFrustumChecks."""
        import torch
        F, cap = fImg.shape[0], kps.shape[1]
        mpCap = mpDesc.shape[1]
        if matchF is None:
            matchF = torch.full((F, cap), -1, dtype=torch.int32, device=kps.device)
        nm = torch.zeros((F,), dtype=torch.int32, device=kps.device)
        check(self._L.morb_search_by_projection_mps_fisheye_batch(
            self._h, C.byref(params), F, ptr(fImg), cap, ptr(count), ptr(nLeft), ptr(kps), ptr(desc), ptr(l2r), ptr(r2l), ptr(blocked),
            mpCap, ptr(nMP), ptr(trkL["inView"]), ptr(trkR["inView"]), ptr(isBad), ptr(trkL["depth"]), ptr(trkL["projX"]),
            ptr(trkL["projY"]), ptr(trkL["level"]), ptr(trkL["viewCos"]), ptr(trkR["projX"]), ptr(trkR["projY"]), ptr(trkR["level"]),
            ptr(trkR["viewCos"]), ptr(mpDesc), ptr(mpHasObs), float(th), 1 if bFarPoints else 0, float(thFarPoints),
            self.mfNNratio, ptr(matchF), ptr(nm), self._st(stream)))
        return matchF, nm

    def SearchByProjectionLastFrame(self, params, curImg, lastImg, kps, desc, count, curURight, curBlocked, Tcw, lastValid,
                                    lastXw, lastMPdesc, lastMPhasObs, th, bForward, bBackward, matchCur=None, nm=None, stream=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono)."""
        import torch
        F, cap = curImg.shape[0], kps.shape[1]
        if matchCur is None:
            matchCur = torch.full((F, cap), -1, dtype=torch.int32, device=kps.device)
        if nm is None:
            nm = torch.zeros((F,), dtype=torch.int32, device=kps.device)
        check(self._L.morb_search_by_projection_last_batch(
            self._h, C.byref(params), F, ptr(curImg), ptr(lastImg), cap, ptr(count), ptr(kps), ptr(desc), ptr(curURight),
            ptr(curBlocked), ptr(Tcw), ptr(lastValid), ptr(lastXw), ptr(lastMPdesc), ptr(lastMPhasObs), float(th), ptr(bForward),
            ptr(bBackward), 1 if self.mbCheckOrientation else 0, ptr(matchCur), ptr(nm), self._st(stream)))
        return matchCur, nm

    def SearchByProjectionLastFrameFisheye(self, params, cam8, Trl7, curImg, lastImg, nLeftCur, kps, desc, count, curBlocked, Tcw,
                                           lastValid, lastXw, lastMPdesc, lastMPhasObs, th, bForward, bBackward, stream=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) with CurrentFrame.Nleft != -1 (left + right pass)."""
        import torch
        F, cap = curImg.shape[0], kps.shape[1]
        matchCur = torch.full((F, cap), -1, dtype=torch.int32, device=kps.device)
        nm = torch.zeros((F,), dtype=torch.int32, device=kps.device)
        cam = np.ascontiguousarray(cam8, np.float32); trl = np.ascontiguousarray(Trl7, np.float32)
        check(self._L.morb_search_by_projection_last_fisheye_batch(
            self._h, C.byref(params), ptr(cam), ptr(trl), F, ptr(curImg), ptr(lastImg), ptr(nLeftCur), cap, ptr(count), ptr(kps),
            ptr(desc), ptr(curBlocked), ptr(Tcw), ptr(lastValid), ptr(lastXw), ptr(lastMPdesc), ptr(lastMPhasObs), float(th),
            ptr(bForward), ptr(bBackward), 1 if self.mbCheckOrientation else 0, ptr(matchCur), ptr(nm), self._st(stream)))
        return matchCur, nm

    def SearchForTriangulationFisheye(self, params, img1, img2, nLeft1, nLeft2, kps, desc, node, count, hasMP, camL8, camR8, T4,
                                      bOnlyStereo=False, bCoarse=False, stream=None):
        """SearchForTriangulation between keyframes of a KB8 rig; T4 = host [npairs, 4, 12] (Tll, Tlr, Trl, Trr as R | t)."""
        import torch
        npairs = img1.shape[0]
        nimg, cap = kps.shape[0], kps.shape[1]
        m12 = torch.full((npairs, cap), -1, dtype=torch.int32, device=kps.device)
        nm = torch.zeros((npairs,), dtype=torch.int32, device=kps.device)
        a = [np.ascontiguousarray(x, np.float32) for x in (camL8, camR8, T4)]
        check(self._L.morb_search_for_triangulation_fisheye_batch(
            self._h, C.byref(params), npairs, ptr(img1), ptr(img2), ptr(nLeft1), ptr(nLeft2), nimg, cap, ptr(count), ptr(kps), ptr(desc),
            ptr(node), ptr(hasMP), ptr(a[0]), ptr(a[1]), ptr(a[2]), 1 if bOnlyStereo else 0, 1 if bCoarse else 0,
            1 if self.mbCheckOrientation else 0, ptr(m12), ptr(nm), self._st(stream)))
        return m12, nm

    def SearchForTriangulation(self, params, img1, img2, kps, desc, node, count, hasMP, uRight, R12, t12, ep,
                               bOnlyStereo=False, bCoarse=False, out=None, stream=None):
        """SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse); R12/t12/ep are host numpy arrays."""
        import torch
        npairs = img1.shape[0]
        nimg, cap = kps.shape[0], kps.shape[1]
        if out is not None:
            m12, nm = out
        else:
            m12 = torch.full((npairs, cap), -1, dtype=torch.int32, device=kps.device)
            nm = torch.zeros((npairs,), dtype=torch.int32, device=kps.device)
        R12 = np.ascontiguousarray(R12, np.float32); t12 = np.ascontiguousarray(t12, np.float32); ep = np.ascontiguousarray(ep, np.float32)
        check(self._L.morb_search_for_triangulation_batch(
            self._h, C.byref(params), npairs, ptr(img1), ptr(img2), nimg, cap, ptr(count), ptr(kps), ptr(desc), ptr(node),
            ptr(hasMP), ptr(uRight), ptr(R12), ptr(t12), ptr(ep), 1 if bOnlyStereo else 0, 1 if bCoarse else 0,
            1 if self.mbCheckOrientation else 0, ptr(m12), ptr(nm), self._st(stream)))
        return m12, nm

    def SearchByProjectionKeyFrame(self, params, curImg, kfImg, kps, desc, count, curHasMP, Tcw, Ow, kfValid, Xw, maxDist, minDist,
                                   mpDesc, th, ORBdist, matchCur=None, stream=None, cam8=None, nLeftCur=None):
        """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (relocalisation).  cam8 + nLeftCur (i32 [F], device): the current
        frames are KannalaBrandt8 rig frames — left-camera projection, left features only."""
        import torch
        F, cap = curImg.shape[0], kps.shape[1]
        if matchCur is None:
            matchCur = torch.full((F, cap), -1, dtype=torch.int32, device=kps.device)
        nm = torch.zeros((F,), dtype=torch.int32, device=kps.device)
        if cam8 is not None:
            cam = np.ascontiguousarray(cam8, np.float32)
            check(self._L.morb_search_by_projection_kf_rig_batch(
                self._h, C.byref(params), ptr(cam), F, ptr(curImg), ptr(kfImg), ptr(nLeftCur), cap, ptr(count), ptr(kps), ptr(desc), ptr(curHasMP),
                ptr(Tcw), ptr(Ow), ptr(kfValid), ptr(Xw), ptr(maxDist), ptr(minDist), ptr(mpDesc), float(th), int(ORBdist),
                1 if self.mbCheckOrientation else 0, ptr(matchCur), ptr(nm), self._st(stream)))
            return matchCur, nm
        check(self._L.morb_search_by_projection_kf_batch(
            self._h, C.byref(params), F, ptr(curImg), ptr(kfImg), cap, ptr(count), ptr(kps), ptr(desc), ptr(curHasMP), ptr(Tcw), ptr(Ow),
            ptr(kfValid), ptr(Xw), ptr(maxDist), ptr(minDist), ptr(mpDesc), float(th), int(ORBdist),
            1 if self.mbCheckOrientation else 0, ptr(matchCur), ptr(nm), self._st(stream)))
        return matchCur, nm

    def SearchForInitialization(self, params, img1, img2, kps, desc, count, prevMatched, windowSize=10, stream=None):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize); prevMatched [P, cap, 2] f32 is updated in place."""
        import torch
        npairs, cap = img1.shape[0], kps.shape[1]
        m12 = torch.full((npairs, cap), -1, dtype=torch.int32, device=kps.device)
        nm = torch.zeros((npairs,), dtype=torch.int32, device=kps.device)
        check(self._L.morb_search_for_initialization_batch(
            self._h, C.byref(params), npairs, ptr(img1), ptr(img2), cap, ptr(count), ptr(kps), ptr(desc), ptr(prevMatched),
            int(windowSize), self.mfNNratio, 1 if self.mbCheckOrientation else 0, ptr(m12), ptr(nm), self._st(stream)))
        return m12, nm

    def ComputeStereoFishEyeMatches(self, kps, desc, count, mono, camL, camR, Rlr, tlr, levelSigma2, stream=None):
        """Frame::ComputeStereoFishEyeMatches (left = image 2f, right = image 2f+1).  camL/camR: 8 floats (fx fy cx cy k0..k3);
        Rlr (3x3), tlr (3): right-to-left rotation / translation.  Returns dict of device tensors."""
        import torch
        nimg, cap = kps.shape[0], kps.shape[1]
        nf = nimg // 2
        dev = kps.device
        o = dict(leftToRight=torch.empty((nf, cap), dtype=torch.int32, device=dev), rightToLeft=torch.empty((nf, cap), dtype=torch.int32, device=dev),
                 depth=torch.empty((nf, cap), dtype=torch.float32, device=dev), p3D=torch.empty((nf, cap, 3), dtype=torch.float32, device=dev),
                 nMatches=torch.empty((nf,), dtype=torch.int32, device=dev))
        a = [np.ascontiguousarray(x, np.float32) for x in (camL, camR, Rlr, tlr, levelSigma2)]
        check(self._L.morb_stereo_fisheye_match_batch(self._h, nf, ptr(kps), ptr(desc), ptr(count), ptr(mono), cap, ptr(a[0]), ptr(a[1]),
                                                      ptr(a[2]), ptr(a[3]), ptr(a[4]), len(a[4]), ptr(o["leftToRight"]), ptr(o["rightToLeft"]),
                                                      ptr(o["depth"]), ptr(o["p3D"]), ptr(o["nMatches"]), self._st(stream)))
        return o
