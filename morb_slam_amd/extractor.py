"""ORBextractor — host-side mirror of the reference class surface (include/ORBextractor.h:44-105) over the
HIP implementation.  Same constructor arguments, same accessors, same call semantics (returns monoIndex,
keypoints as cv::KeyPoint-layout records, descriptors N x 32 u8)."""
import ctypes as C

import numpy as np

from . import capi
from .capi import KP_DTYPE, check, lib, ptr, stream_arg


class ORBextractor:
    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.morb_extractor_create(C.byref(self._h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device))
        self.nfeatures, self.nlevels, self.device = nfeatures, nlevels, device
        n = nlevels
        self._sc, self._isc, self._s2, self._is2 = (np.zeros(n, np.float32) for _ in range(4))
        self._fpl = np.zeros(n, np.int32)
        check(self._L.morb_extractor_tables(self._h, ptr(self._sc), ptr(self._isc), ptr(self._s2), ptr(self._is2), ptr(self._fpl)))
        self.max_keypoints = check(self._L.morb_extractor_max_keypoints(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.morb_extractor_destroy(self._h)
            self._h = None

    __del__ = close

    # accessors, ORBextractor.h:59-74
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return float(self._L.morb_extractor_scale_factor(self._h))
    def GetScaleFactors(self): return self._sc.copy()
    def GetInverseScaleFactors(self): return self._isc.copy()
    def GetScaleSigmaSquares(self): return self._s2.copy()
    def GetInverseScaleSigmaSquares(self): return self._is2.copy()
    def features_per_level(self): return self._fpl.copy()

    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """operator()(image, mask, keypoints, descriptors, vLappingArea) -> (monoIndex, keypoints, descriptors).
        Empty image -> (-1, [], []) like the reference (ORBextractor.cc:1011)."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (ORBextractor.cc:1014)"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        stride = image.strides[0]
        cap = self.max_keypoints
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        mono = check(self._L.morb_extract(self._h, ptr(image), w, h, stride, int(vLappingArea[0]), int(vLappingArea[1]),
                                          ptr(kps), ptr(desc), cap, C.byref(n)))
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, d_images, lap=None, out=None, stream=None):
        """Device-resident batch: d_images is a CUDA/HIP torch uint8 tensor [nimg, H, W] (contiguous rows).
        Returns torch device tensors (kps as a [nimg, cap, 7] int32 view-compatible byte tensor, desc, count, mono)."""
        import torch
        nimg, h, w = d_images.shape
        assert d_images.dtype == torch.uint8 and d_images.stride(2) == 1
        cap = self.max_keypoints
        if out is None:
            dev = d_images.device
            out = (torch.empty((nimg, cap, 28), dtype=torch.uint8, device=dev),
                   torch.empty((nimg, cap, 32), dtype=torch.uint8, device=dev),
                   torch.empty((nimg,), dtype=torch.int32, device=dev),
                   torch.empty((nimg,), dtype=torch.int32, device=dev))
        kps, desc, cnt, mono = out
        lap_arr = None if lap is None else np.ascontiguousarray(lap, np.int32).reshape(nimg, 2)
        st = stream_arg(stream)
        check(self._L.morb_extract_batch(self._h, ptr(d_images), nimg, w, h, d_images.stride(1), d_images.stride(0),
                                         ptr(lap_arr), ptr(kps), ptr(desc), cap, ptr(cnt), ptr(mono), st))
        return out

    # mvImagePyramid (ORBextractor.h:76) and parity taps
    def level_size(self, lvl, img=0):
        w, h, s, p = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
        check(self._L.morb_extractor_pyramid_level(self._h, img, lvl, C.byref(p), C.byref(w), C.byref(h), C.byref(s)))
        return w.value, h.value

    def pyramid_level(self, lvl, img=0):
        w, h = self.level_size(lvl, img)
        out = np.zeros((h + 38, w + 38), np.uint8)
        check(self._L.morb_extractor_pyramid_level_host(self._h, img, lvl, ptr(out)))
        return out

    def blurred_level(self, lvl, img=0):
        w, h = self.level_size(lvl, img)
        out = np.zeros((h, w), np.uint8)
        check(self._L.morb_extractor_blurred_level_host(self._h, img, lvl, ptr(out)))
        return out

    def _kp_tap(self, fn, lvl, img):
        n = C.c_int(0)
        check(fn(self._h, img, lvl, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        check(fn(self._h, img, lvl, ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def level_candidates(self, lvl, img=0):
        return self._kp_tap(self._L.morb_extractor_level_candidates_host, lvl, img)

    def level_keypoints(self, lvl, img=0):
        return self._kp_tap(self._L.morb_extractor_level_keypoints_host, lvl, img)

    def event_after_fast(self):
        """hipEvent_t (as an int) recorded by the last extract_batch after its FAST stage (morb_extractor_event_after_fast)."""
        ev = C.c_void_p()
        check(self._L.morb_extractor_event_after_fast(self._h, C.byref(ev)))
        return ev.value

    def event_after_pyramid(self):
        """hipEvent_t (as an int) recorded by the last extract_batch behind its pyramid launches (morb_extractor_event_after_pyramid)."""
        ev = C.c_void_p()
        check(self._L.morb_extractor_event_after_pyramid(self._h, C.byref(ev)))
        return ev.value

    def check_status(self):
        """Raise if an extraction since the last check was flagged on the device (call after synchronising the batch call's stream)."""
        check(self._L.morb_extractor_status(self._h, None))

    def set_profiling(self, on=True):
        check(self._L.morb_extractor_set_profiling(self._h, 1 if on else 0))

    def stage_ms(self):
        ms = np.zeros(7, np.float32)
        n = check(self._L.morb_extractor_stage_ms(self._h, ptr(ms)))
        self.last_profile_calls = n
        return dict(zip(["pyramid", "blur", "fast", "distribute", "layout", "describe", "total"], ms.tolist()))
