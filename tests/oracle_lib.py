"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE (the checker), never the product path."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MORB_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")   # (the sanitizer test points this at its own build)
    try:
        _lib = C.CDLL(path)
    except OSError:
        build()
        _lib = C.CDLL(path)
    L = _lib
    u8p, kpp, ip, fp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.orc_extractor_create.restype = C.c_void_p
    L.orc_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.orc_extractor_destroy.argtypes = [C.c_void_p]
    L.orc_extract.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, kpp, u8p, C.c_int, ip]
    L.orc_extractor_tables.argtypes = [C.c_void_p, fp, fp, fp, fp, ip, ip]
    L.orc_level_size.argtypes = [C.c_void_p, C.c_int, ip, ip]
    L.orc_level_image.argtypes = [C.c_void_p, C.c_int, u8p]
    L.orc_level_blurred.argtypes = [C.c_void_p, C.c_int, u8p]
    L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, kpp, C.c_int]
    L.orc_level_keypoints.argtypes = [C.c_void_p, C.c_int, kpp, C.c_int]
    L.orc_resize_linear.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
    L.orc_gaussian7.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
    L.orc_border101.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_fast.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, kpp, C.c_int]
    L.orc_distribute.argtypes = [kpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, kpp, C.c_int]
    L.orc_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, kpp, u8p, C.c_int, kpp, u8p, C.c_float, C.c_float, fp, fp]
    L.orc_descriptor_distance.argtypes = [u8p, u8p]
    L.orc_three_maxima.argtypes = [ip, C.c_int, ip]
    L.orc_knn2.argtypes = [u8p, C.c_int, u8p, C.c_int, ip, ip]
    L.orc_search_by_bow.argtypes = [C.c_int, u8p, fp, u8p, ip, C.c_int, u8p, fp, ip, C.c_float, C.c_int, ip]
    L.orc_bow_transform.argtypes = [u8p, C.c_int, u8p, ip, C.c_int, C.c_int, C.c_int, ip, ip]
    L.orc_fast_atan2.restype = C.c_float
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_cosf.restype = C.c_float
    L.orc_cosf.argtypes = [C.c_float]
    L.orc_sinf.restype = C.c_float
    L.orc_sinf.argtypes = [C.c_float]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """Mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:44-105) over the CPU restatement."""

    def __init__(self, nfeatures=1200, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_extractor_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orc_extractor_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(fpl), _p(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, feat_per_level=fpl, umax=umax)

    def __call__(self, img, lap=(0, 0)):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = self.nfeatures + 16 * self.nlevels
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        mono = self.L.orc_extract(self.h, _p(img), w, h, w, lap[0], lap[1], _p(kps), _p(desc), cap, C.byref(n))
        assert mono != -2
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, lvl):
        w, h = C.c_int(), C.c_int()
        self.L.orc_level_size(self.h, lvl, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_image(self, lvl):
        w, h = self.level_size(lvl)
        out = np.zeros((h + 38, w + 38), np.uint8)
        self.L.orc_level_image(self.h, lvl, _p(out))
        return out

    def level_blurred(self, lvl):
        w, h = self.level_size(lvl)
        out = np.zeros((h, w), np.uint8)
        r = self.L.orc_level_blurred(self.h, lvl, _p(out))
        return out if r == 1 else None

    def level_candidates(self, lvl):
        n = self.L.orc_level_candidates(self.h, lvl, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.orc_level_candidates(self.h, lvl, _p(out), n)
        return out[:n]

    def level_keypoints(self, lvl):
        n = self.L.orc_level_keypoints(self.h, lvl, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.orc_level_keypoints(self.h, lvl, _p(out), n)
        return out[:n]


def stereo_matches(ext_l, ext_r, kl, dl, kr, dr, mbf, mb):
    """Frame::ComputeStereoMatches over two OracleExtractors that just processed the left / right image."""
    L = lib()
    n = len(kl)
    u = np.zeros(max(n, 1), np.float32); d = np.zeros(max(n, 1), np.float32)
    kl = np.ascontiguousarray(kl); kr = np.ascontiguousarray(kr)
    dl = np.ascontiguousarray(dl); dr = np.ascontiguousarray(dr)
    L.orc_stereo_matches(ext_l.h, ext_r.h, n, _p(kl), _p(dl), len(kr), _p(kr), _p(dr), mbf, mb, _p(u), _p(d))
    return u[:n], d[:n]


def knn2(q, t):
    L = lib()
    q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
    idx = np.zeros((max(len(q), 1), 2), np.int32); dist = np.zeros((max(len(q), 1), 2), np.int32)
    L.orc_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx[:len(q)], dist[:len(q)]


def search_by_bow(descKF, angleKF, hasMP, nodeKF, descF, angleF, nodeF, nnratio, checkOri):
    L = lib()
    a = [np.ascontiguousarray(x) for x in (descKF, angleKF.astype(np.float32), hasMP.astype(np.uint8), nodeKF.astype(np.int32),
                                           descF, angleF.astype(np.float32), nodeF.astype(np.int32))]
    m = np.zeros(max(len(descF), 1), np.int32)
    n = L.orc_search_by_bow(len(descKF), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(descF), _p(a[4]), _p(a[5]), _p(a[6]),
                            nnratio, 1 if checkOri else 0, _p(m))
    return n, m[:len(descF)]


def bow_transform_tree(feat, nodeDesc, firstChild, childCount, L, levelsup):
    L_ = lib()
    L_.orc_bow_transform_tree.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    a = [np.ascontiguousarray(feat), np.ascontiguousarray(nodeDesc), np.ascontiguousarray(firstChild, np.int32), np.ascontiguousarray(childCount, np.int32)]
    w = np.zeros(len(feat), np.int32); nid = np.zeros(len(feat), np.int32)
    L_.orc_bow_transform_tree(_p(a[0]), len(feat), _p(a[1]), _p(a[2]), _p(a[3]), L, levelsup, _p(w), _p(nid))
    return w, nid


def distinctive_descriptors(start, desc):
    L = lib()
    L.orc_distinctive_descriptors.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    a = [np.ascontiguousarray(start, np.int32), np.ascontiguousarray(desc)]
    out = np.zeros(len(start) - 1, np.int32)
    L.orc_distinctive_descriptors(len(start) - 1, _p(a[0]), _p(a[1]), _p(out))
    return out


def search_by_bow_kfkf(d1, a1, has1, node1, nv1, d2, a2, has2, node2, nv2, nnratio, checkOri):
    L = lib()
    L.orc_search_by_bow_kfkf.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (d1, a1.astype(np.float32), has1.astype(np.uint8), node1.astype(np.int32),
                                           d2, a2.astype(np.float32), has2.astype(np.uint8), node2.astype(np.int32))]
    m = np.zeros(max(len(d1), 1), np.int32)
    n = L.orc_search_by_bow_kfkf(len(d1), int(nv1), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(d2), int(nv2), _p(a[4]), _p(a[5]), _p(a[6]),
                                 _p(a[7]), nnratio, 1 if checkOri else 0, _p(m))
    return n, m[:len(d1)]


def search_by_bow_fisheye(descKF, angleKF, hasMP, nodeKF, descF, angleF, nodeF, FNleft, nnratio, checkOri):
    L = lib()
    a = [np.ascontiguousarray(x) for x in (descKF, angleKF.astype(np.float32), hasMP.astype(np.uint8), nodeKF.astype(np.int32),
                                           descF, angleF.astype(np.float32), nodeF.astype(np.int32))]
    m = np.zeros(max(len(descF), 1), np.int32)
    L.orc_search_by_bow_fisheye.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_int, C.c_void_p]
    n = L.orc_search_by_bow_fisheye(len(descKF), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(descF), int(FNleft), _p(a[4]), _p(a[5]),
                                    _p(a[6]), nnratio, 1 if checkOri else 0, _p(m))
    return n, m[:len(descF)]


def bow_transform(desc, voc_desc, voc_first, k, Lv, levelsup):
    L = lib()
    desc = np.ascontiguousarray(desc)
    w = np.zeros(max(len(desc), 1), np.int32); nd = np.zeros(max(len(desc), 1), np.int32)
    L.orc_bow_transform(_p(desc), len(desc), _p(np.ascontiguousarray(voc_desc)), _p(np.ascontiguousarray(voc_first)), k, Lv,
                        levelsup, _p(w), _p(nd))
    return w[:len(desc)], nd[:len(desc)]


def pose_optimization(p):
    L = lib()
    L.orc_pose_optimization.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_float] * 5 + [C.c_void_p] * 3
    n = len(p["hasMP"])
    pose = p["pose0"].astype(np.float32).copy()
    outl = np.zeros(n, np.uint8)
    stats = np.zeros(2, np.int32)
    c = p["cam"]
    a = [np.ascontiguousarray(p[k]) for k in ("hasMP", "obs", "invSigma2", "Xw")]
    r = L.orc_pose_optimization(n, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), c["fx"], c["fy"], c["cx"], c["cy"], c["bf"],
                                _p(pose), _p(outl), _p(stats))
    return r, pose, outl, stats


def local_ba(p, lambda100=False, stop=None):
    L = lib()
    L.orc_local_ba.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + \
        [C.c_float] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    kf = p["kfPose"].astype(np.float32).copy(); mp = p["mpPos"].astype(np.float32).copy()
    nE = len(p["eKF"])
    erase = np.zeros(nE, np.uint8); stats = np.zeros(2, np.int32)
    c = p["cam"]
    a = [np.ascontiguousarray(p[k]) for k in ("kfFixed", "eKF", "eMP", "eObs", "eInvSigma2")]
    st = None if stop is None else _p(stop)
    its = L.orc_local_ba(len(kf), _p(kf), _p(a[0]), len(mp), _p(mp), nE, _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]),
                         c["fx"], c["fy"], c["cx"], c["cy"], c["bf"], 1 if lambda100 else 0, st, _p(erase), _p(stats))
    return its, kf, mp, erase, stats


def local_ba_fisheye(p, lambda100=False):
    L = lib()
    L.orc_local_ba_fisheye.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + \
        [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    kf = p["kfPose"].astype(np.float32).copy(); mp = p["mpPos"].astype(np.float32).copy()
    nE = len(p["eKF"])
    erase = np.zeros(nE, np.uint8); stats = np.zeros(2, np.int32)
    a = [np.ascontiguousarray(p[k]) for k in ("kfFixed", "eKF", "eMP", "eObs", "eRight", "eInvSigma2", "camL", "camR", "Trl")]
    its = L.orc_local_ba_fisheye(len(kf), _p(kf), _p(a[0]), len(mp), _p(mp), nE, _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]),
                                 _p(a[6]), _p(a[7]), _p(a[8]), 1 if lambda100 else 0, None, _p(erase), _p(stats))
    return its, kf, mp, erase, stats


class OrcFrame(C.Structure):
    _fields_ = [("N", C.c_int), ("kpsUn", C.c_void_p), ("desc", C.c_void_p), ("uRight", C.c_void_p),
                ("minX", C.c_float), ("minY", C.c_float), ("maxX", C.c_float), ("maxY", C.c_float), ("gridInvW", C.c_float),
                ("gridInvH", C.c_float), ("scaleFactors", C.c_void_p), ("nlevels", C.c_int), ("fx", C.c_float),
                ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float), ("mb", C.c_float),
                ("logScaleFactor", C.c_float)]


def make_frame(P, kps, desc, uRight):
    """OrcFrame from a morb_slam_amd.capi.FrameParams + host arrays (keeps references alive on the struct)."""
    f = OrcFrame()
    f._keep = [np.ascontiguousarray(kps), np.ascontiguousarray(desc), None if uRight is None else np.ascontiguousarray(uRight, np.float32),
               np.array(list(P.scaleFactors)[:P.nlevels], np.float32)]
    f.N = len(kps); f.kpsUn = _p(f._keep[0]); f.desc = _p(f._keep[1]); f.uRight = None if uRight is None else _p(f._keep[2])
    for k in ("minX", "minY", "maxX", "maxY", "gridInvW", "gridInvH", "fx", "fy", "cx", "cy", "mbf", "mb", "logScaleFactor", "nlevels"):
        setattr(f, k, getattr(P, k))
    f.scaleFactors = _p(f._keep[3])
    return f


def is_in_frustum(F, Rcw, tcw, Ow, Pw, normal, maxDist, minDist, cosLimit=0.5):
    L = lib(); n = len(Pw)
    a = [np.ascontiguousarray(x, np.float32) for x in (Rcw, tcw, Ow, Pw, normal, maxDist, minDist)]
    o = dict(inView=np.zeros(n, np.uint8), projX=np.zeros(n, np.float32), projY=np.zeros(n, np.float32), projXR=np.zeros(n, np.float32),
             depth=np.zeros(n, np.float32), level=np.zeros(n, np.int32), viewCos=np.zeros(n, np.float32))
    L.orc_is_in_frustum.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 7
    L.orc_is_in_frustum(C.byref(F), _p(a[0]), _p(a[1]), _p(a[2]), n, _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), cosLimit,
                        _p(o["inView"]), _p(o["projX"]), _p(o["projY"]), _p(o["projXR"]), _p(o["depth"]), _p(o["level"]), _p(o["viewCos"]))
    return o


def search_by_projection_mps(F, blocked, trk, isBad, mpDesc, mpHasObs, th, bFar, thFar, nnratio, match_init=None):
    L = lib(); n = len(mpDesc)
    L.orc_search_by_projection_mps.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10 + [C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
    m = np.full(F.N, -1, np.int32) if match_init is None else match_init.astype(np.int32).copy()
    a = [np.ascontiguousarray(x) for x in (blocked.astype(np.uint8), trk["inView"], isBad.astype(np.uint8), trk["depth"], trk["projX"],
                                           trk["projY"], trk["projXR"], trk["level"], trk["viewCos"], mpDesc, mpHasObs.astype(np.uint8))]
    r = L.orc_search_by_projection_mps(C.byref(F), _p(a[0]), n, _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]),
                                       _p(a[8]), _p(a[9]), _p(a[10]), th, 1 if bFar else 0, thFar, nnratio, _p(m))
    return r, m


def is_in_frustum_kb8(F, cam8, R, t, twc, Pw, normal, maxDist, minDist, cosLimit=0.5):
    L = lib(); n = len(Pw)
    a = [np.ascontiguousarray(x, np.float32) for x in (cam8, R, t, twc, Pw, normal, maxDist, minDist)]
    o = dict(inView=np.zeros(n, np.uint8), projX=np.zeros(n, np.float32), projY=np.zeros(n, np.float32),
             depth=np.zeros(n, np.float32), level=np.zeros(n, np.int32), viewCos=np.zeros(n, np.float32))
    L.orc_is_in_frustum_kb8.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 6
    L.orc_is_in_frustum_kb8(C.byref(F), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), n, _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), cosLimit,
                            _p(o["inView"]), _p(o["projX"]), _p(o["projY"]), _p(o["depth"]), _p(o["level"]), _p(o["viewCos"]))
    return o


def search_by_projection_mps_fisheye(F, Nleft, l2r, r2l, blocked, trkL, trkR, isBad, mpDesc, mpHasObs, th, bFar, thFar, nnratio):
    L = lib(); n = len(mpDesc)
    L.orc_search_by_projection_mps_fisheye.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + \
        [C.c_void_p] * 14 + [C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
    m = np.full(F.N, -1, np.int32)
    a = [np.ascontiguousarray(x) for x in (np.asarray(l2r, np.int32), np.asarray(r2l, np.int32), blocked.astype(np.uint8),
                                           trkL["inView"], trkR["inView"], isBad.astype(np.uint8), trkL["depth"], trkL["projX"],
                                           trkL["projY"], trkL["level"], trkL["viewCos"], trkR["projX"], trkR["projY"], trkR["level"],
                                           trkR["viewCos"], mpDesc, mpHasObs.astype(np.uint8))]
    r = L.orc_search_by_projection_mps_fisheye(C.byref(F), int(Nleft), _p(a[0]), _p(a[1]), _p(a[2]), n, *[_p(x) for x in a[3:]],
                                               th, 1 if bFar else 0, thFar, nnratio, _p(m))
    return r, m


def search_by_projection_last(Cur, blocked, Tcw7, lastKps, lastValid, lastXw, lastMPdesc, lastHasObs, th, fwd, bwd, checkOri):
    L = lib()
    L.orc_search_by_projection_last.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
    m = np.full(Cur.N, -1, np.int32)
    a = [np.ascontiguousarray(x) for x in (blocked.astype(np.uint8), np.asarray(Tcw7, np.float32), lastKps, lastValid.astype(np.uint8),
                                           np.asarray(lastXw, np.float32), lastMPdesc, lastHasObs.astype(np.uint8))]
    r = L.orc_search_by_projection_last(C.byref(Cur), _p(a[0]), _p(a[1]), len(lastKps), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]),
                                        th, int(fwd), int(bwd), int(checkOri), _p(m))
    return r, m


def search_by_projection_last_fisheye(Cur, NleftCur, cam8, Trl7, blocked, Tcw7, lastKps, lastValid, lastXw, lastMPdesc, lastHasObs, th,
                                      fwd, bwd, checkOri):
    L = lib()
    L.orc_search_by_projection_last_fisheye.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 5 + \
        [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
    m = np.full(Cur.N, -1, np.int32)
    a = [np.ascontiguousarray(x) for x in (np.asarray(cam8, np.float32), np.asarray(Trl7, np.float32), blocked.astype(np.uint8),
                                           np.asarray(Tcw7, np.float32), lastKps, lastValid.astype(np.uint8),
                                           np.asarray(lastXw, np.float32), lastMPdesc, lastHasObs.astype(np.uint8))]
    r = L.orc_search_by_projection_last_fisheye(C.byref(Cur), int(NleftCur), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(lastKps),
                                                _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), _p(a[8]), th, int(fwd), int(bwd), int(checkOri), _p(m))
    return r, m


def fuse_search(KF, invSigma2, Tcw7, Ow, valid, Pw, normal, maxD, minD, mpDesc, th, sim3Form):
    L = lib(); n = len(Pw)
    L.orc_fuse_search.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (np.asarray(invSigma2, np.float32), np.asarray(Tcw7, np.float32), np.asarray(Ow, np.float32),
                                           valid.astype(np.uint8), np.asarray(Pw, np.float32), np.asarray(normal, np.float32),
                                           np.asarray(maxD, np.float32), np.asarray(minD, np.float32), mpDesc)]
    bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32)
    L.orc_fuse_search(C.byref(KF), _p(a[0]), _p(a[1]), _p(a[2]), n, _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), _p(a[8]), th,
                      int(sim3Form), _p(bi), _p(bd))
    return bi, bd


def fuse_search_rig(KF, NLeft, bRight, cam8, invSigma2, Tcw7, Ow, valid, Pw, normal, maxD, minD, mpDesc, th):
    """Fuse(pKF, vpMapPoints, th, bRight) on a KannalaBrandt8 rig keyframe (left | right features in one row)."""
    L = lib(); n = len(Pw)
    L.orc_fuse_search_rig.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (np.asarray(cam8, np.float32), np.asarray(invSigma2, np.float32), np.asarray(Tcw7, np.float32),
                                           np.asarray(Ow, np.float32), valid.astype(np.uint8), np.asarray(Pw, np.float32),
                                           np.asarray(normal, np.float32), np.asarray(maxD, np.float32), np.asarray(minD, np.float32), mpDesc)]
    bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32)
    L.orc_fuse_search_rig(C.byref(KF), int(NLeft), int(bRight), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), n, _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]),
                          _p(a[8]), _p(a[9]), th, _p(bi), _p(bd))
    return bi, bd


def search_by_projection_sim3(KF, Tcw7, Ow, valid, Pw, normal, maxD, minD, mpDesc, matched, th, ratio, manual):
    L = lib(); n = len(Pw)
    L.orc_search_by_projection_sim3.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (np.asarray(Tcw7, np.float32), np.asarray(Ow, np.float32), valid.astype(np.uint8),
                                           np.asarray(Pw, np.float32), np.asarray(normal, np.float32), np.asarray(maxD, np.float32),
                                           np.asarray(minD, np.float32), mpDesc, matched.astype(np.uint8))]
    m = np.zeros(KF.N, np.int32)
    r = L.orc_search_by_projection_sim3(C.byref(KF), _p(a[0]), _p(a[1]), n, _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]),
                                        _p(a[8]), int(th), ratio, int(manual), _p(m))
    return r, m


def search_by_sim3_dir(B, TAw7, SBA8, valid, Pw, maxD, minD, mpDesc, th):
    L = lib(); n = len(Pw)
    L.orc_search_by_sim3_dir.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (np.asarray(TAw7, np.float32), np.asarray(SBA8, np.float32), valid.astype(np.uint8),
                                           np.asarray(Pw, np.float32), np.asarray(maxD, np.float32), np.asarray(minD, np.float32), mpDesc)]
    v = np.zeros(n, np.int32)
    L.orc_search_by_sim3_dir(C.byref(B), _p(a[0]), _p(a[1]), n, _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), th, _p(v))
    return v


def search_by_projection_sim3_rig(KF, NLeft, cam8, Tcw7, Ow, valid, Pw, normal, maxD, minD, mpDesc, matched, th, ratio, manual):
    """SearchByProjection(pKF, Scw, ...) (and its twin) on a KannalaBrandt8 rig keyframe: left features, the left KB8 camera (first form only)."""
    L = lib(); n = len(Pw)
    L.orc_search_by_projection_sim3_rig.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (np.asarray(cam8, np.float32), np.asarray(Tcw7, np.float32), np.asarray(Ow, np.float32), valid.astype(np.uint8),
                                           np.asarray(Pw, np.float32), np.asarray(normal, np.float32), np.asarray(maxD, np.float32),
                                           np.asarray(minD, np.float32), mpDesc, matched.astype(np.uint8))]
    m = np.zeros(KF.N, np.int32)
    r = L.orc_search_by_projection_sim3_rig(C.byref(KF), int(NLeft), _p(a[0]), _p(a[1]), _p(a[2]), n, _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]),
                                            _p(a[8]), _p(a[9]), int(th), ratio, int(manual), _p(m))
    return r, m


def search_by_sim3_dir_rig(B, NLeftB, TAw7, SBA8, valid, Pw, maxD, minD, mpDesc, th):
    L = lib(); n = len(Pw)
    L.orc_search_by_sim3_dir_rig.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (np.asarray(TAw7, np.float32), np.asarray(SBA8, np.float32), valid.astype(np.uint8),
                                           np.asarray(Pw, np.float32), np.asarray(maxD, np.float32), np.asarray(minD, np.float32), mpDesc)]
    v = np.zeros(n, np.int32)
    L.orc_search_by_sim3_dir_rig(C.byref(B), int(NLeftB), _p(a[0]), _p(a[1]), n, _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), th, _p(v))
    return v


def fuse_search_rig_sim3(KF, NLeft, cam8, invSigma2, Tcw7, Ow, valid, Pw, normal, maxD, minD, mpDesc, th):
    """Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) on a rig keyframe: the left camera and its features, no reprojection gate."""
    L = lib(); n = len(Pw)
    L.orc_fuse_search_rig_sim3.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p]
    a = [np.ascontiguousarray(x) for x in (np.asarray(cam8, np.float32), np.asarray(invSigma2, np.float32), np.asarray(Tcw7, np.float32),
                                           np.asarray(Ow, np.float32), valid.astype(np.uint8), np.asarray(Pw, np.float32),
                                           np.asarray(normal, np.float32), np.asarray(maxD, np.float32), np.asarray(minD, np.float32), mpDesc)]
    bi = np.zeros(n, np.int32); bd = np.zeros(n, np.int32)
    L.orc_fuse_search_rig_sim3(C.byref(KF), int(NLeft), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), n, _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]),
                               _p(a[8]), _p(a[9]), th, _p(bi), _p(bd))
    return bi, bd


def search_for_triangulation_fisheye(k1, nl1, d1, node1, has1, k2, nl2, d2, node2, has2, sigma2, camL8, camR8, T4, onlyStereo, coarse, checkOri):
    L = lib()
    L.orc_search_for_triangulation_fisheye.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 8 + \
        [C.c_int] * 3 + [C.c_void_p]
    m = np.full(len(k1), -1, np.int32)
    a = [np.ascontiguousarray(x) for x in (k1, d1, node1.astype(np.int32), has1.astype(np.uint8), k2, d2, node2.astype(np.int32),
                                           has2.astype(np.uint8), np.asarray(sigma2, np.float32), np.asarray(camL8, np.float32),
                                           np.asarray(camR8, np.float32), np.asarray(T4, np.float32))]
    n = L.orc_search_for_triangulation_fisheye(len(k1), int(nl1), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(k2), int(nl2), _p(a[4]),
                                               _p(a[5]), _p(a[6]), _p(a[7]), _p(a[8]), _p(a[9]), _p(a[10]), _p(a[11]),
                                               int(onlyStereo), int(coarse), int(checkOri), _p(m))
    return n, m


def search_for_triangulation(k1, d1, node1, has1, ur1, k2, d2, node2, has2, ur2, sigma2, scaleF, K, R12, t12, ep, onlyStereo, coarse, checkOri):
    L = lib()
    F12 = np.zeros(9, np.float32)
    Kf = np.asarray(K, np.float32)
    L.orc_fundamental_f12(_p(Kf), _p(Kf), _p(np.ascontiguousarray(R12, np.float32)), _p(np.ascontiguousarray(t12, np.float32)), _p(F12))
    L.orc_search_for_triangulation.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_void_p]
    m = np.full(len(k1), -1, np.int32)
    a = [np.ascontiguousarray(x) for x in (k1, d1, node1.astype(np.int32), has1.astype(np.uint8), k2, d2, node2.astype(np.int32), has2.astype(np.uint8),
                                           np.asarray(sigma2, np.float32), np.asarray(scaleF, np.float32), np.asarray(ep, np.float32))]
    u1 = None if ur1 is None else np.ascontiguousarray(ur1, np.float32); u2 = None if ur2 is None else np.ascontiguousarray(ur2, np.float32)
    r = L.orc_search_for_triangulation(len(k1), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), None if u1 is None else _p(u1), _p(a[8]),
                                       len(k2), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]), None if u2 is None else _p(u2), _p(a[8]), _p(a[9]),
                                       _p(F12), _p(a[10]), int(onlyStereo), int(coarse), int(checkOri), _p(m))
    return r, m


def search_by_projection_kf(Cur, curHasMP, Tcw7, Ow, kfKps, kfValid, Xw, maxD, minD, mpDesc, th, ORBdist, checkOri):
    L = lib()
    L.orc_search_by_projection_kf.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
    m = np.full(Cur.N, -1, np.int32)
    a = [np.ascontiguousarray(x) for x in (curHasMP.astype(np.uint8), np.asarray(Tcw7, np.float32), np.asarray(Ow, np.float32), kfKps,
                                           kfValid.astype(np.uint8), np.asarray(Xw, np.float32), np.asarray(maxD, np.float32),
                                           np.asarray(minD, np.float32), mpDesc)]
    r = L.orc_search_by_projection_kf(C.byref(Cur), _p(a[0]), _p(a[1]), _p(a[2]), len(kfKps), _p(a[3]), _p(a[4]), _p(a[5]), _p(a[6]),
                                      _p(a[7]), _p(a[8]), th, int(ORBdist), int(checkOri), _p(m))
    return r, m


def search_for_initialization(k1, d1, F2, prev, windowSize, nnratio, checkOri):
    L = lib()
    L.orc_search_for_initialization.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    m = np.full(len(k1), -1, np.int32)
    pv = np.ascontiguousarray(prev, np.float32).copy()
    k1 = np.ascontiguousarray(k1); d1 = np.ascontiguousarray(d1)
    r = L.orc_search_for_initialization(len(k1), _p(k1), _p(d1), C.byref(F2), _p(pv), int(windowSize), nnratio, int(checkOri), _p(m))
    return r, m, pv


def stereo_fisheye_matches(fe, sigma2):
    L = lib()
    L.orc_stereo_fisheye_matches.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 11
    nl, nr = len(fe["kL"]), len(fe["kR"])
    l2r = np.zeros(nl, np.int32); r2l = np.zeros(nr, np.int32); dep = np.zeros(nl, np.float32); p3 = np.zeros((nl, 3), np.float32)
    a = [np.ascontiguousarray(x) for x in (fe["kL"], fe["dL"], fe["kR"], fe["dR"], fe["camL"], fe["camR"], fe["Rlr"], fe["tlr"],
                                           np.asarray(sigma2, np.float32))]
    n = L.orc_stereo_fisheye_matches(nl, fe["monoL"], _p(a[0]), _p(a[1]), nr, fe["monoR"], _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]),
                                     _p(a[6]), _p(a[7]), _p(a[8]), _p(l2r), _p(r2l), _p(dep), _p(p3))
    return n, l2r, r2l, dep, p3


def pose_optimization_fisheye(p):
    L = lib()
    L.orc_pose_optimization_fisheye.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 10
    n = len(p["hasMP"])
    pose = p["pose0"].astype(np.float32).copy(); outl = np.zeros(n, np.uint8); stats = np.zeros(2, np.int32)
    obs2 = np.ascontiguousarray(p["obs"][:, :2], np.float32)
    a = [np.ascontiguousarray(p[k]) for k in ("hasMP", "invSigma2", "Xw", "camL", "camR", "Trl")]
    r = L.orc_pose_optimization_fisheye(n, p["Nleft"], _p(a[0]), _p(obs2), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]), _p(pose),
                                        _p(outl), _p(stats))
    return r, pose, outl, stats


# ---- N1 slice: IMU preintegration + PoseInertialOptimizationLastKeyFrame (oracle/inertial.cc) ----
PREINT_FLOATS = 310   # orc_imu_preintegrated is all floats: same record layout as morb_imu_preintegrated


def imu_preintegrate(bias6, nga6, walk6, acc, gyro, dt):
    L = lib()
    L.orc_imu_preintegrate.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4
    L.orc_imu_preintegrate.restype = None
    out = np.zeros(PREINT_FLOATS, np.float32)
    a = [np.ascontiguousarray(x, np.float32) for x in (bias6, nga6, walk6, acc, gyro, dt)]
    L.orc_imu_preintegrate(_p(a[0]), _p(a[1]), _p(a[2]), len(a[5]), _p(a[3]), _p(a[4]), _p(a[5]), _p(out))
    return out


def pose_inertial_optimization_last_keyframe(p, pre, bRecInit=False):
    L = lib()
    f = C.c_float
    L.orc_pose_inertial_optimization_last_keyframe.argtypes = [C.c_int] + [C.c_void_p] * 5 + [f] * 5 + [C.c_void_p] * 3 + [C.c_int] + \
        [C.c_void_p] * 3
    n = len(p["hasMP"])
    state = p["state0"].astype(np.float32).copy(); outl = np.zeros(n, np.uint8); prior = np.zeros(246, np.float64)
    a = [np.ascontiguousarray(p[k]) for k in ("hasMP", "obs", "invSigma2", "Xw", "close", "Tbc12", "kfState")]
    pre = np.ascontiguousarray(pre, np.float32)
    cam = p["cam"]
    if p.get("rig28") is not None:
        L.orc_pose_inertial_optimization_last_keyframe_fisheye.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_int] + [C.c_void_p] * 3
        rig = np.ascontiguousarray(p["rig28"], np.float32)
        r = L.orc_pose_inertial_optimization_last_keyframe_fisheye(n, int(p["Nleft"]), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]),
                                                                   _p(rig), _p(a[5]), _p(a[6]), _p(pre), int(bool(bRecInit)), _p(state),
                                                                   _p(outl), _p(prior))
        return r, state, outl, prior
    r = L.orc_pose_inertial_optimization_last_keyframe(n, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), cam["fx"], cam["fy"],
                                                       cam["cx"], cam["cy"], cam["bf"], _p(a[5]), _p(a[6]), _p(pre),
                                                       int(bool(bRecInit)), _p(state), _p(outl), _p(prior))
    return r, state, outl, prior


def pose_inertial_optimization_last_frame(p, prevState, preFrame, preKF, prevPrior, bRecInit=False):
    L = lib()
    f = C.c_float
    L.orc_pose_inertial_optimization_last_frame.argtypes = [C.c_int] + [C.c_void_p] * 5 + [f] * 5 + [C.c_void_p] * 5 + [C.c_int] + \
        [C.c_void_p] * 3
    n = len(p["hasMP"])
    state = p["state0"].astype(np.float32).copy(); outl = np.zeros(n, np.uint8); prior = np.zeros(246, np.float64)
    a = [np.ascontiguousarray(p[k]) for k in ("hasMP", "obs", "invSigma2", "Xw", "close", "Tbc12")]
    b = [np.ascontiguousarray(prevState, np.float32), np.ascontiguousarray(preFrame, np.float32),
         np.ascontiguousarray(preKF, np.float32), np.ascontiguousarray(prevPrior, np.float64)]
    cam = p["cam"]
    if p.get("rig28") is not None:
        L.orc_pose_inertial_optimization_last_frame_fisheye.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 11 + [C.c_int] + [C.c_void_p] * 3
        rig = np.ascontiguousarray(p["rig28"], np.float32)
        r = L.orc_pose_inertial_optimization_last_frame_fisheye(n, int(p["Nleft"]), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(rig),
                                                                _p(a[5]), _p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]), int(bool(bRecInit)),
                                                                _p(state), _p(outl), _p(prior))
        return r, state, outl, prior
    r = L.orc_pose_inertial_optimization_last_frame(n, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), cam["fx"], cam["fy"], cam["cx"],
                                                    cam["cy"], cam["bf"], _p(a[5]), _p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]),
                                                    int(bool(bRecInit)), _p(state), _p(outl), _p(prior))
    return r, state, outl, prior


def local_inertial_ba(p, pre, bLarge=False):
    """pre: [nI, PREINT_FLOATS] preintegration records of the links."""
    L = lib()
    f = C.c_float
    vp = C.c_void_p
    L.orc_local_inertial_ba.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp] + [f] * 5 + \
        [vp, C.c_int, vp, vp]
    kf = p["kfState"].astype(np.float32).copy(); mp = p["mpPos"].astype(np.float32).copy()
    nE = len(p["eKF"]); erase = np.zeros(nE, np.uint8); stats = np.zeros(2, np.int32)
    a = [np.ascontiguousarray(p[k]) for k in ("kfKind", "mpClose", "eKF", "eMP", "eObs", "eInvSigma2", "iKF1", "iKF2", "iRobust", "iInfoScale", "Tbc12")]
    pre = np.ascontiguousarray(pre, np.float32)
    cam = p["cam"]
    if p.get("rig28") is not None:
        L.orc_local_inertial_ba_fisheye.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp,
                                                    vp, vp, C.c_int, vp, vp]
        rig = np.ascontiguousarray(p["rig28"], np.float32); er = np.ascontiguousarray(p["eRight"], np.uint8)
        r = L.orc_local_inertial_ba_fisheye(len(kf), _p(kf), _p(a[0]), len(mp), _p(mp), _p(a[1]), nE, _p(a[2]), _p(a[3]), _p(a[4]), _p(er),
                                            _p(a[5]), len(a[6]), _p(a[6]), _p(a[7]), _p(pre), _p(a[8]), _p(a[9]), _p(rig), _p(a[10]),
                                            int(bool(bLarge)), _p(erase), _p(stats))
        return r, kf, mp, erase, stats
    r = L.orc_local_inertial_ba(len(kf), _p(kf), _p(a[0]), len(mp), _p(mp), _p(a[1]), nE, _p(a[2]), _p(a[3]), _p(a[4]), _p(a[5]),
                                len(a[6]), _p(a[6]), _p(a[7]), _p(pre), _p(a[8]), _p(a[9]), cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                                cam["bf"], _p(a[10]), int(bool(bLarge)), _p(erase), _p(stats))
    return r, kf, mp, erase, stats


def undistort_points(xy, cam, dist, pcam=None):
    L = lib()
    f = C.c_float
    L.orc_undistort_points.argtypes = [C.c_int, C.c_void_p] + [f] * 4 + [C.c_void_p] + [f] * 4 + [C.c_void_p]
    L.orc_undistort_points.restype = None
    xy = np.ascontiguousarray(xy, np.float32); out = np.zeros_like(xy)
    d5 = np.zeros(5, np.float32); d5[:len(dist)] = np.asarray(dist, np.float32)
    pc = pcam or cam
    L.orc_undistort_points(len(xy), _p(xy), cam["fx"], cam["fy"], cam["cx"], cam["cy"], _p(d5), pc["fx"], pc["fy"], pc["cx"], pc["cy"], _p(out))
    return out


def stereo_from_rgbd(kp_xy, kpun_xy, depth, bf):
    L = lib()
    L.orc_stereo_from_rgbd.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_stereo_from_rgbd.restype = None
    a = [np.ascontiguousarray(x, np.float32) for x in (kp_xy, kpun_xy, depth)]
    n = len(a[0]); ur = np.zeros(n, np.float32); d = np.zeros(n, np.float32)
    L.orc_stereo_from_rgbd(n, _p(a[0]), _p(a[1]), _p(a[2]), depth.shape[1], depth.shape[0], bf, _p(ur), _p(d))
    return ur, d


def bow_vector(leaf, node_weight, node_word=None, weighting=0, scoring=0):
    L = lib()
    L.orc_bow_vector.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    leaf = np.ascontiguousarray(leaf, np.int32); w = np.ascontiguousarray(node_weight, np.float64)
    nw = None if node_word is None else np.ascontiguousarray(node_word, np.int32)
    ow = np.zeros(len(leaf), np.int32); ov = np.zeros(len(leaf), np.float64)
    k = L.orc_bow_vector(len(leaf), _p(leaf), None if nw is None else _p(nw), _p(w), weighting, scoring, _p(ow), _p(ov))
    return ow[:k], ov[:k]


# ---- Tracking.cc's host loops between the searches and PoseOptimization (checker for csrc/tracking.hip) ----
def frame_set_pose(Tcw7):
    L = lib()
    L.orc_frame_set_pose.argtypes = [C.c_void_p] * 4
    a = np.ascontiguousarray(Tcw7, np.float32)
    R, t, Ow = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    L.orc_frame_set_pose(_p(a), _p(R), _p(t), _p(Ow))
    return R, t, Ow


def pose_edges(F, invLevelSigma2, frameMP, mpXw):
    L = lib()
    L.orc_pose_edges.argtypes = [C.c_void_p] * 8
    n = F.N
    a = [np.ascontiguousarray(invLevelSigma2, np.float32), np.ascontiguousarray(frameMP, np.int32), np.ascontiguousarray(mpXw, np.float32)]
    has, obs, is2, Xw = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    L.orc_pose_edges(C.byref(F), _p(a[0]), _p(a[1]), _p(a[2]), _p(has), _p(obs), _p(is2), _p(Xw))
    return has, obs, is2, Xw


def discard_outliers(frameMP, outlier, mpHasObs, want_blocked=True, want_seen=True):
    L = lib()
    L.orc_discard_outliers.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    L.orc_discard_outliers.restype = C.c_int
    fm = np.ascontiguousarray(frameMP, np.int32).copy(); ol = np.ascontiguousarray(outlier, np.uint8).copy()
    ho = np.ascontiguousarray(mpHasObs, np.uint8)
    blk = np.zeros(len(fm), np.uint8) if want_blocked else None
    seen = np.zeros(len(ho), np.uint8) if want_seen else None
    nmap = C.c_int(0)
    nm = L.orc_discard_outliers(len(fm), _p(fm), _p(ol), len(ho), _p(ho), None if blk is None else _p(blk),
                                None if seen is None else _p(seen), C.byref(nmap))
    return nm, nmap.value, fm, blk, seen


def search_by_projection_kf_rig(CurLeft, cam8, curHasMP, Tcw7, Ow, kfKps, kfValid, Xw, maxD, minD, mpDesc, th, ORBdist, checkOri):
    """SearchByProjection(CurrentFrame, pKF, ...) with a KannalaBrandt8 rig CurrentFrame: CurLeft = OrcFrame of its LEFT features."""
    L = lib()
    L.orc_search_by_projection_kf_rig.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
    m = np.full(CurLeft.N, -1, np.int32)
    a = [np.ascontiguousarray(x) for x in (np.asarray(cam8, np.float32), curHasMP.astype(np.uint8), np.asarray(Tcw7, np.float32), np.asarray(Ow, np.float32),
                                           kfKps, kfValid.astype(np.uint8), np.asarray(Xw, np.float32), np.asarray(maxD, np.float32),
                                           np.asarray(minD, np.float32), mpDesc)]
    r = L.orc_search_by_projection_kf_rig(C.byref(CurLeft), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(kfKps), _p(a[4]), _p(a[5]), _p(a[6]), _p(a[7]),
                                          _p(a[8]), _p(a[9]), th, int(ORBdist), int(checkOri), _p(m))
    return r, m
