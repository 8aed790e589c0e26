"""ctypes binding of libsvo_hip.so (the C-ABI declared in include/svo.h).

There is no CPU fallback: if the HIP library has not been built the import of this module fails
loudly, and every call fails with SvoError when no MI355X is usable.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvo_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "stereo_visual_odometry_amd: %s is missing. Build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C stereo_visual_odometry_amd/csrc). "
        "There is no CPU fallback." % LIB_PATH)

# Load order: the PyTorch ROCm wheel bundles its own libamdhip64 / libhsa-runtime64.  If the system ROCm runtime (a
# dependency of libsvo_hip.so) is initialised first, torch's copy later reports "No HIP GPUs are available"; the other order
# works (measured on the MI355X box).  torch is plumbing here (device tensors in bench.py), so when it is installed it is
# imported first.  Set SVO_NO_TORCH_PRELOAD=1 to skip.
if not os.environ.get("SVO_NO_TORCH_PRELOAD"):
    try:
        import torch  # noqa: F401
    except Exception:
        pass

lib = C.CDLL(LIB_PATH)

SVO_OK, SVO_ERR_ARG, SVO_ERR_HIP, SVO_ERR_CAPACITY, SVO_ERR_STATE = 0, -1, -2, -3, -4


class SvoError(RuntimeError):
    pass


class SvoConfig(C.Structure):
    _fields_ = [
        ("bucket_start_row", C.c_int), ("buckets_along_height", C.c_int), ("buckets_along_width", C.c_int),
        ("features_per_bucket", C.c_int), ("features_threshold", C.c_int),
        ("pre_matching_feature_threshold", C.c_int), ("age_threshold", C.c_int), ("fast_threshold", C.c_int),
        ("ransac_reprojection_error", C.c_float), ("ransac_iterations", C.c_int),
        ("optical_flow_min_eig_threshold", C.c_double), ("circular_matching_success_threshold", C.c_double),
        ("max_translation_norm", C.c_double), ("max_rotation_norm", C.c_double),
        ("win_w", C.c_int), ("win_h", C.c_int), ("max_level", C.c_int), ("lk_max_count", C.c_int),
        ("lk_epsilon", C.c_double), ("ransac_confidence", C.c_float), ("max_features", C.c_int), ("channels", C.c_int),
        ("lk_float_sums", C.c_int),
    ]


class SvoFrameStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_after_detect", "second_pass", "n_into_lk", "n_after_circular", "n_after_bounds",
        "n_inliers", "ransac_iters", "fail_reason", "n_features_out", "lk_level_visits", "lk_newton_steps",
        "lk_dead_after_pass0", "lk_dead_after_pass1", "lk_dead_after_pass2")]

    def as_dict(self):
        return {f[0]: getattr(self, f[0]) for f in self._fields_}


lib.svo_last_error.restype = C.c_char_p
lib.svo_alloc_pinned.restype = C.c_void_p
lib.svo_alloc_pinned.argtypes = [C.c_size_t]
lib.svo_free_pinned.restype = None
lib.svo_free_pinned.argtypes = [C.c_void_p]
lib.svo_get_stream.restype = C.c_void_p
lib.svo_stage_cache_clear.restype = None
lib.svo_stage_cache_clear.argtypes = []
lib.svo_get_stream.argtypes = [C.c_void_p]

# every symbol include/svo.h declares (tests/test_abi.py checks the list against the header)
EXPORTS = [
    "svo_last_error", "svo_device_count", "svo_config_default", "svo_create", "svo_destroy", "svo_set_projection",
    "svo_process_batch", "svo_process", "svo_alloc_pinned", "svo_free_pinned", "svo_circular_matching", "svo_submit_batch", "svo_collect", "svo_get_features", "svo_get_last_tracks",
    "svo_get_lk_registers_left", "svo_get_last_timing", "svo_set_stage_timing", "svo_get_stage_timing", "svo_get_stream", "svo_fast_detect", "svo_fast_score_map", "svo_bucket_filter",
    "svo_append_features_from_image", "svo_build_pyramid", "svo_lk_track", "svo_circular_match",
    "svo_find_close_points", "svo_stage_cache_clear", "svo_stage_cache_clear_all", "svo_triangulate", "svo_camera_to_world", "svo_inverse_transform",
]


def check(rc):
    if rc < 0:
        raise SvoError("libsvo_hip status %d: %s" % (rc, (lib.svo_last_error() or b"").decode()))
    return rc


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def default_config(**over):
    c = SvoConfig()
    lib.svo_config_default(C.byref(c))
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def copy_config(cfg):
    c = SvoConfig()
    C.memmove(C.byref(c), C.byref(cfg), C.sizeof(SvoConfig))
    return c


def device_count():
    return lib.svo_device_count()


def u8img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim != 2:
        raise ValueError("single-channel 8-bit image expected (the ROS path delivers MONO8, src/stereo_vo.cpp:9)")
    return img


def u8frame(img):
    """A frame for the frame pipeline: (H, W) gray, or (H, W, 3) interleaved BGR as the reference CLI feeds it (svo.h: channels)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if not (img.ndim == 2 or (img.ndim == 3 and img.shape[2] == 3)):
        raise ValueError("8-bit (H, W) or (H, W, 3) image expected")
    return img
