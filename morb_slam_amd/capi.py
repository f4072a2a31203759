"""ctypes binding of libmorb_hip.so (include/morb_hip.h).  There is NO CPU fallback: importing this module
without the built library, or calling into it without a GPU, raises."""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
# MORB_HIP_LIB selects another build of the same HIP library (the phase-timing build of tools/fast_phases.py); never a CPU path
LIB_PATH = os.environ.get("MORB_HIP_LIB") or os.path.join(_DIR, "libmorb_hip.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

MORB_OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_EMPTY = 0, -1, -2, -3, -4, -5


class FrameParams(C.Structure):
    """morb_frame_params (include/morb_hip.h)"""
    _fields_ = [("minX", C.c_float), ("minY", C.c_float), ("maxX", C.c_float), ("maxY", C.c_float),
                ("gridInvW", C.c_float), ("gridInvH", C.c_float), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("mbf", C.c_float), ("mb", C.c_float), ("logScaleFactor", C.c_float),
                ("nlevels", C.c_int32), ("scaleFactors", C.c_float * 16), ("levelSigma2", C.c_float * 16)]


def make_frame_params(width, height, fx, fy, cx, cy, mbf, mb, scale_factors, level_sigma2, scale_factor=1.2):
    """Frame constructor bookkeeping for an undistorted camera (Frame.cc:229-241, ComputeImageBounds :859-887)."""
    import numpy as np
    p = FrameParams()
    p.minX, p.minY, p.maxX, p.maxY = 0.0, 0.0, float(width), float(height)
    p.gridInvW = float(np.float32(64.0) / np.float32(p.maxX - p.minX))
    p.gridInvH = float(np.float32(48.0) / np.float32(p.maxY - p.minY))
    p.fx, p.fy, p.cx, p.cy, p.mbf, p.mb = fx, fy, cx, cy, mbf, mb
    p.logScaleFactor = float(np.log(np.float32(scale_factor)))   # mfLogScaleFactor = log(mfScaleFactor) (float)
    p.nlevels = len(scale_factors)
    for i, v in enumerate(scale_factors):
        p.scaleFactors[i] = float(v)
        p.levelSigma2[i] = float(level_sigma2[i])
    return p


def stream_arg(stream):
    """ABI `void* stream` argument.  None = the handle's own HIP stream.  That stream is a blocking stream, i.e. ordered
    with the legacy default stream, but torch may be producing the inputs on a non-default current stream (e.g. the DMA
    of a fresh `.cuda()` upload inside a `torch.cuda.stream(...)` block), so the torch stream is drained first.  Pass a
    raw stream handle (`torch.cuda.Stream.cuda_stream`) to stay asynchronous."""
    if stream is None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        return None
    return C.c_void_p(stream)


class MorbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmorb_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the product path)")
        # torch (device memory / streams / torch.distributed plumbing) bundles its own libamdhip64.so.7; load it
        # first so libmorb_hip.so binds to the SAME HIP runtime instead of pulling a second one into the process
        # (two runtimes in one process leave the later one without a GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.morb_last_error.restype = C.c_char_p
        L.morb_device_count.restype = i
        L.morb_extractor_create.argtypes = [C.POINTER(vp), i, f, i, i, i, i]
        L.morb_extractor_destroy.argtypes = [vp]
        L.morb_extractor_destroy.restype = None
        L.morb_extractor_levels.argtypes = [vp]
        L.morb_extractor_scale_factor.argtypes = [vp]
        L.morb_extractor_scale_factor.restype = f
        L.morb_extractor_tables.argtypes = [vp, vp, vp, vp, vp, vp]
        L.morb_extractor_max_keypoints.argtypes = [vp]
        L.morb_extract.argtypes = [vp, vp, i, i, i, i, i, vp, vp, i, C.POINTER(i)]
        L.morb_extract_batch.argtypes = [vp, vp, i, i, i, i, sz, vp, vp, vp, i, vp, vp, vp]
        L.morb_extractor_pyramid_level.argtypes = [vp, i, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
        L.morb_extractor_pyramid_level_host.argtypes = [vp, i, i, vp]
        L.morb_extractor_blurred_level_host.argtypes = [vp, i, i, vp]
        L.morb_extractor_level_candidates_host.argtypes = [vp, i, i, vp, i, C.POINTER(i)]
        L.morb_extractor_level_keypoints_host.argtypes = [vp, i, i, vp, i, C.POINTER(i)]
        L.morb_extractor_set_profiling.argtypes = [vp, i]
        L.morb_extractor_stage_ms.argtypes = [vp, vp]
        L.morb_extractor_event_after_fast.argtypes = [vp, C.POINTER(vp)]
        L.morb_extractor_event_after_pyramid.argtypes = [vp, C.POINTER(vp)]
        L.morb_stream_wait_event.argtypes = [vp, vp]
        L.morb_extractor_status.argtypes = [vp, C.POINTER(i)]
        L.morb_matcher_create.argtypes = [C.POINTER(vp), i]
        L.morb_matcher_destroy.argtypes = [vp]
        L.morb_matcher_destroy.restype = None
        L.morb_matcher_sync.argtypes = [vp]
        L.morb_matcher_stream.argtypes = [vp]
        L.morb_matcher_stream.restype = vp
        L.morb_hamming_pairs.argtypes = [vp, vp, vp, i, vp, vp]
        L.morb_feature_slab_bytes.argtypes = [i, i]
        L.morb_feature_slab_bytes.restype = sz
        L.morb_feature_slab_pack.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp]
        L.morb_feature_slab_unpack.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp]
        L.morb_hamming_knn2_batch.argtypes = [vp, i, vp, vp, i, vp, vp, vp, i, vp, vp, vp, vp]
        L.morb_stereo_match_batch.argtypes = [vp, vp, i, vp, vp, vp, i, f, f, vp, vp, vp]
        L.morb_stereo_fisheye_match_batch.argtypes = [vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp]
        L.morb_bow_transform_batch.argtypes = [vp, i, vp, vp, i, vp, vp, i, i, i, vp, vp, vp]
        L.morb_search_by_bow_batch.argtypes = [vp, i, vp, vp, i, vp, vp, vp, vp, vp, i, f, i, vp, vp, vp]
        L.morb_bow_transform_tree_batch.argtypes = [vp, i, vp, vp, i, vp, vp, vp, i, i, vp, vp, vp]
        L.morb_vocabulary_load_text.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.morb_vocabulary_destroy.argtypes = [vp]
        L.morb_vocabulary_destroy.restype = None
        L.morb_vocabulary_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
        L.morb_vocabulary_arrays.argtypes = [vp] * 6
        L.morb_distinctive_descriptors_batch.argtypes = [vp, i, vp, vp, vp, vp]
        L.morb_search_by_bow_kfkf_batch.argtypes = [vp, i, vp, vp, vp, i, vp, vp, vp, vp, vp, i, f, i, vp, vp, vp]
        L.morb_search_by_bow_fisheye_batch.argtypes = [vp, i, vp, vp, vp, i, vp, vp, vp, vp, vp, i, f, i, vp, vp, vp]
        PP = C.POINTER(FrameParams)
        L.morb_is_in_frustum_batch.argtypes = [vp, PP, i, vp, vp, vp, i, vp, vp, vp, vp, vp, f] + [vp] * 8
        L.morb_search_by_projection_mps_batch.argtypes = [vp, PP, i, vp, i, vp, vp, vp, vp, vp, i] + [vp] * 11 + [f, i, f, f, vp, vp, vp]
        L.morb_is_in_frustum_kb8_batch.argtypes = [vp, PP, vp, i, vp, vp, vp, i, vp, vp, vp, vp, vp, f] + [vp] * 7
        L.morb_search_by_projection_mps_fisheye_batch.argtypes = [vp, PP, i, vp, i, vp, vp, vp, vp, vp, vp, vp, i] + [vp] * 15 + [f, i, f, f, vp, vp, vp]
        L.morb_frame_set_pose_batch.argtypes = [vp, i, vp, vp, vp, vp, vp]
        L.morb_pose_edges_batch.argtypes = [vp, PP, i, vp, i, vp, vp, vp, vp, vp, i, i] + [vp] * 7
        L.morb_track_discard_outliers_batch.argtypes = [vp, i, vp, i, vp, vp, vp, i] + [vp] * 6
        L.morb_search_by_projection_last_batch.argtypes = [vp, PP, i, vp, vp, i] + [vp] * 10 + [f, vp, vp, i, vp, vp, vp]
        L.morb_search_by_projection_last_fisheye_batch.argtypes = [vp, PP, vp, vp, i, vp, vp, vp, i] + [vp] * 9 + [f, vp, vp, i, vp, vp, vp]
        L.morb_search_by_projection_kf_batch.argtypes = [vp, PP, i, vp, vp, i] + [vp] * 11 + [f, i, i, vp, vp, vp]
        L.morb_search_by_projection_kf_rig_batch.argtypes = [vp, PP, vp, i, vp, vp, vp, i] + [vp] * 11 + [f, i, i, vp, vp, vp]
        L.morb_search_for_initialization_batch.argtypes = [vp, PP, i, vp, vp, i, vp, vp, vp, vp, i, f, i, vp, vp, vp]
        L.morb_search_for_triangulation_batch.argtypes = [vp, PP, i, vp, vp, i, i] + [vp] * 9 + [i, i, i, vp, vp, vp]
        L.morb_search_for_triangulation_fisheye_batch.argtypes = [vp, PP, i, vp, vp, vp, vp, i, i] + [vp] * 8 + [i, i, i, vp, vp, vp]
        L.morb_fuse_batch.argtypes = [vp, PP, i, vp, i] + [vp] * 9 + [i] + [vp] * 7 + [f, i, vp, vp, vp]
        L.morb_search_by_projection_sim3_batch.argtypes = [vp, PP, i, vp, i] + [vp] * 5 + [i] + [vp] * 8 + [i, f, i, vp, vp, vp]
        L.morb_search_by_sim3_batch.argtypes = [vp, PP, i, vp, vp, i] + [vp] * 17 + [f, vp, vp, vp, vp, vp]
        L.morb_search_by_projection_sim3_rig_batch.argtypes = [vp, PP, i, vp, i] + [vp] * 5 + [i] + [vp] * 8 + [i, f, i, vp, vp, vp, vp, vp]
        L.morb_search_by_sim3_rig_batch.argtypes = [vp, PP, i, vp, vp, i] + [vp] * 17 + [f, vp, vp, vp, vp, vp, vp, vp]
        L.morb_bow_vector_batch.argtypes = [vp, i, vp, vp, i, vp, vp, i, i, vp, vp, vp, vp]
        L.morb_vocabulary_weights.argtypes = [vp, vp, vp, vp]
        L.morb_undistort_keypoints_batch.argtypes = [vp, i, i, vp, vp, f, f, f, f, vp, vp, vp]
        L.morb_stereo_from_rgbd_batch.argtypes = [vp, i, i, vp, vp, vp, vp, i, i, C.c_size_t, C.c_size_t, f, vp, vp, vp]
        L.morb_image_bounds.argtypes = [i, i, f, f, f, f, vp, vp]
        L.morb_optimizer_create.argtypes = [C.POINTER(vp), i]
        L.morb_optimizer_destroy.argtypes = [vp]
        L.morb_optimizer_destroy.restype = None
        L.morb_optimizer_sync.argtypes = [vp]
        L.morb_optimizer_set_exact_order.argtypes = [vp, i]
        L.morb_optimizer_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.morb_optimizer_stream.argtypes = [vp]
        L.morb_optimizer_stream.restype = vp
        L.morb_pose_optimization_batch.argtypes = [vp, i, i, vp, vp, vp, vp, vp, f, f, f, f, f, vp, vp, vp, vp, vp]
        L.morb_imu_preintegrate_batch.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.morb_pose_inertial_optimization_last_keyframe_batch.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, f, f, f, f, f, vp, vp, vp, i,
                                                                          vp, vp, vp, vp, vp]
        L.morb_pose_inertial_optimization_last_frame_batch.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, f, f, f, f, f, vp, vp, vp, vp, vp,
                                                                       i, vp, vp, vp, vp, vp]
        L.morb_local_inertial_ba.argtypes = [vp, i, vp, vp, i, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, f, f, f, f, f, vp, i, vp, vp]
        L.morb_pose_inertial_optimization_last_keyframe_fisheye_batch.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i,
                                                                                  vp, vp, vp, vp, vp]
        L.morb_pose_inertial_optimization_last_frame_fisheye_batch.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i,
                                                                               vp, vp, vp, vp, vp]
        L.morb_local_inertial_ba_fisheye.argtypes = [vp, i, vp, vp, i, vp, vp, i, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, vp, i, vp, vp]
        L.morb_pose_optimization_fisheye_batch.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.morb_ba_problem_create_fisheye.argtypes = [vp, C.POINTER(vp), i, vp, vp, i, vp, i, vp, vp, vp, vp, vp, vp, vp, vp, i]
        L.morb_local_bundle_adjustment.argtypes = [vp, i, vp, vp, i, vp, i, vp, vp, vp, vp, f, f, f, f, f, i, vp, vp, vp]
        L.morb_ba_problem_create.argtypes = [vp, C.POINTER(vp), i, vp, vp, i, vp, i, vp, vp, vp, vp, f, f, f, f, f, i]
        L.morb_ba_problem_destroy.argtypes = [vp]
        L.morb_ba_problem_destroy.restype = None
        L.morb_ba_set_stop.argtypes = [vp, i]
        L.morb_ba_set_mode.argtypes = [vp, i]
        L.morb_ba_solve.argtypes = [vp, vp]
        L.morb_ba_results.argtypes = [vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise MorbError(rc, lib().morb_last_error().decode(errors="replace"))
    return rc


def ptr(a):
    """Host numpy array or torch tensor (host or device) -> void*."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())
