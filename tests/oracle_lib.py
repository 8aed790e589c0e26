"""ctypes binding of the CPU oracle (oracle/libsvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker / reported baseline.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
# SVO_ORACLE_LIB selects another build of the same sources (oracle/Makefile: `native` for the timed CPU baseline,
# `asan` for the sanitizer run); the default is the portable -O2 checker
_LIB_PATH = os.environ.get("SVO_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libsvo_oracle.so")


def build_oracle(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _LIB_PATH


class OrcConfig(C.Structure):
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


class OrcFrameStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_after_detect", "second_pass", "n_into_lk", "n_after_circular", "n_after_bounds",
        "n_inliers", "ransac_iters", "fail_reason", "n_features_out", "lk_level_visits", "lk_newton_steps",
        "lk_dead_after_pass0", "lk_dead_after_pass1", "lk_dead_after_pass2")]


ORC_MAX_LEVELS = 8


class OrcPyramid(C.Structure):
    _fields_ = [
        ("nlevels", C.c_int), ("pad_x", C.c_int), ("pad_y", C.c_int),
        ("w", C.c_int * ORC_MAX_LEVELS), ("h", C.c_int * ORC_MAX_LEVELS),
        ("img", C.c_void_p * ORC_MAX_LEVELS), ("img_stride", C.c_int * ORC_MAX_LEVELS),
        ("deriv", C.c_void_p * ORC_MAX_LEVELS), ("deriv_stride", C.c_int * ORC_MAX_LEVELS),
        ("owned", C.c_void_p * (2 * ORC_MAX_LEVELS)),
    ]


class OrcBucket(C.Structure):
    _fields_ = [("max_size", C.c_int), ("n", C.c_int), ("xy", C.POINTER(C.c_float)),
                ("ages", C.POINTER(C.c_int)), ("strengths", C.POINTER(C.c_int))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_vo_create.restype = C.c_void_p
        _lib.orc_epnp.restype = C.c_double
        _lib.orc_rng_next.restype = C.c_uint32
        _lib.orc_set_threads(1)          # deterministic default: single thread (bench.py raises it for the all-cores baseline)
    return _lib


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    """threads for the oracle's data-parallel loops (n <= 0: all cores); results do not depend on it"""
    return lib().orc_set_threads(int(n))


def default_config(**over):
    c = OrcConfig()
    lib().orc_config_default(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def u8img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


# ---------------------------------------------------------------- stage wrappers
def fast_detect(img, threshold, nonmax=True):
    img = u8img(img)
    h, w = img.shape
    cap = w * h // 4 + 16
    xy = np.zeros((cap, 2), np.float32)
    resp = np.zeros(cap, np.float32)
    n = lib().orc_fast_detect(_p(img), w, h, w, int(threshold), int(nonmax), cap, _p(xy), _p(resp))
    n = min(n, cap)
    return xy[:n].copy(), resp[:n].copy()


def fast_score_map(img, threshold, nonmax=True):
    img = u8img(img)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_fast_score_map(_p(img), w, h, w, int(threshold), int(nonmax), _p(out))
    return out


def bucket_filter(img_w, img_h, xy, ages, strengths, bah=92, baw=160, start_row=4, per_bucket=1,
                  age_threshold=20, fast_threshold=20):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2).copy()
    ages = np.ascontiguousarray(ages, np.int32).copy()
    strengths = np.ascontiguousarray(strengths, np.int32).copy()
    n = len(ages)
    m = lib().orc_bucket_filter(img_w, img_h, n, _p(xy), _p(ages), _p(strengths), bah, baw, start_row, per_bucket,
                                age_threshold, fast_threshold)
    return xy[:m].copy(), ages[:m].copy(), strengths[:m].copy()


class Bucket:
    """Mirror of the reference's Bucket class for the known-answer tests (src/main.cpp:50-78)."""

    def __init__(self, max_size, age_threshold=20, fast_threshold=20):
        self.b = OrcBucket()
        self.age_threshold, self.fast_threshold = age_threshold, fast_threshold
        lib().orc_bucket_init(C.byref(self.b), max_size)

    def add_feature(self, x, y, age, strength):
        lib().orc_bucket_add_feature(C.byref(self.b), C.c_float(x), C.c_float(y), age, strength,
                                     self.age_threshold, self.fast_threshold)

    @property
    def max_size(self):
        return self.b.max_size

    def size(self):
        return self.b.n

    @property
    def ages(self):
        return [self.b.ages[i] for i in range(self.b.n)]

    @property
    def strengths(self):
        return [self.b.strengths[i] for i in range(self.b.n)]

    def __del__(self):
        try:
            lib().orc_bucket_free(C.byref(self.b))
        except Exception:
            pass


class Pyramid:
    def __init__(self, img, win=(10, 10), max_level=3):
        img = u8img(img)
        h, w = img.shape
        self.p = OrcPyramid()
        lib().orc_build_pyramid(_p(img), w, h, w, win[0], win[1], max_level, C.byref(self.p))

    @property
    def nlevels(self):
        return self.p.nlevels

    def level(self, l):
        w, h, st = self.p.w[l], self.p.h[l], self.p.img_stride[l]
        buf = (C.c_uint8 * (st * h)).from_address(self.p.img[l])
        return np.frombuffer(buf, np.uint8).reshape(h, st)[:, :w].copy()

    def deriv(self, l):
        w, h, st = self.p.w[l], self.p.h[l], self.p.deriv_stride[l]
        buf = (C.c_int16 * (st * h)).from_address(self.p.deriv[l])
        return np.frombuffer(buf, np.int16).reshape(h, st)[:, :2 * w].reshape(h, w, 2).copy()

    def __del__(self):
        try:
            lib().orc_pyramid_free(C.byref(self.p))
        except Exception:
            pass


def pyr_down(img):
    img = u8img(img)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(_p(img), w, h, w, _p(out), out.shape[1])
    return out


def lk_track(pyr_a, pyr_b, pts, win=(10, 10), max_level=3, max_count=30, epsilon=1e-4, min_eig=1e-3):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    out = np.zeros((n, 2), np.float32)
    status = np.zeros(n, np.uint8)
    lib().orc_lk_track(C.byref(pyr_a.p), C.byref(pyr_b.p), n, _p(pts), _p(out), _p(status), win[0], win[1],
                       max_level, max_count, C.c_double(epsilon), C.c_double(min_eig))
    return out, status


def circular_match(l0, r0, l1, r1, pl0, cfg):
    pl0 = np.ascontiguousarray(pl0, np.float32).reshape(-1, 2)
    n = len(pl0)
    outs = [np.zeros((n, 2), np.float32) for _ in range(4)]
    ok = np.zeros(n, np.uint8)
    lib().orc_circular_match(C.byref(l0.p), C.byref(r0.p), C.byref(l1.p), C.byref(r1.p), n, _p(pl0),
                             _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), _p(ok), C.byref(cfg))
    return outs[0], outs[1], outs[2], outs[3], ok    # pl1, pr1, pr0, pl0_circle, ok


def find_close_points(p1, p2, thr):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    ok = np.zeros(len(p1), np.uint8)
    lib().orc_find_close_points(len(p1), _p(p1), _p(p2), C.c_float(thr), _p(ok))
    return ok


def triangulate(Pl, Pr, pts_l, pts_r):
    Pl = np.ascontiguousarray(Pl, np.float32).reshape(12)
    Pr = np.ascontiguousarray(Pr, np.float32).reshape(12)
    pts_l = np.ascontiguousarray(pts_l, np.float32).reshape(-1, 2)
    pts_r = np.ascontiguousarray(pts_r, np.float32).reshape(-1, 2)
    n = len(pts_l)
    xyz = np.zeros((n, 3), np.float32)
    hom = np.zeros((n, 4), np.float32)
    lib().orc_triangulate(_p(Pl), _p(Pr), n, _p(pts_l), _p(pts_r), _p(xyz), _p(hom))
    return xyz, hom


def rodrigues_to_matrix(r):
    r = np.ascontiguousarray(r, np.float64).reshape(3)
    R = np.zeros(9, np.float64)
    J = np.zeros(27, np.float64)
    lib().orc_rodrigues_to_matrix(_p(r), _p(R), _p(J))
    return R.reshape(3, 3), J.reshape(3, 9)


def rodrigues_to_vector(R):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    r = np.zeros(3, np.float64)
    lib().orc_rodrigues_to_vector(_p(R), _p(r))
    return r


def rng_sequence(n, seed=0xFFFFFFFFFFFFFFFF):
    st = C.c_uint64(seed)
    return [lib().orc_rng_next(C.byref(st)) for _ in range(n)]


def ransac_update_num_iters(p, ep, model_points, max_iters):
    return lib().orc_ransac_update_num_iters(C.c_double(p), C.c_double(ep), model_points, max_iters)


def epnp(obj, img, fx, fy, cx, cy):
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    R = np.zeros(9); t = np.zeros(3)
    err = lib().orc_epnp(len(obj), _p(obj), _p(img), C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), _p(R), _p(t))
    return R.reshape(3, 3), t, err


def camera_to_world(K, cam_pts, world_pts, R, t, iterations=100, reproj=8.0, confidence=0.98):
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    cam = np.ascontiguousarray(cam_pts, np.float32).reshape(-1, 2)
    world = np.ascontiguousarray(world_pts, np.float32).reshape(-1, 3)
    n = len(cam)
    R = np.ascontiguousarray(R, np.float64).reshape(9).copy()
    t = np.ascontiguousarray(t, np.float64).reshape(3).copy()
    inl = np.zeros(max(n, 1), np.int32)
    nin = C.c_int(0)
    dbg = (C.c_int * 2)()
    ok = lib().orc_camera_to_world(_p(K), n, _p(cam), _p(world), _p(R), _p(t), _p(inl), C.byref(nin), iterations,
                                   C.c_float(reproj), C.c_float(confidence), dbg)
    return bool(ok), R.reshape(3, 3), t, inl[:nin.value].copy(), (dbg[0], dbg[1])


def inverse_transform(R, t):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    T = np.zeros(16)
    lib().orc_inverse_transform(_p(R), _p(t), _p(T))
    return T.reshape(4, 4)


class VisualOdometry:
    """Oracle counterpart of the reference's VisualOdometry (include/vo.h:231-380)."""

    def __init__(self, cfg=None):
        self.cfg = cfg if cfg is not None else default_config()
        self.h = C.c_void_p(lib().orc_vo_create(C.byref(self.cfg)))

    def initalize_projection_matricies(self, Pl, Pr):
        Pl = np.ascontiguousarray(Pl, np.float32).reshape(12)
        Pr = np.ascontiguousarray(Pr, np.float32).reshape(12)
        lib().orc_vo_set_projection(self.h, _p(Pl), _p(Pr))

    def stereo_callback(self, left, right):
        """(H, W) gray images, or (H, W, 3) interleaved BGR — what the reference CLI feeds."""
        left = np.ascontiguousarray(left, dtype=np.uint8); right = np.ascontiguousarray(right, dtype=np.uint8)
        assert left.shape == right.shape and (left.ndim == 2 or (left.ndim == 3 and left.shape[2] == 3))
        h, w = left.shape[:2]
        cn = 1 if left.ndim == 2 else 3
        T = np.zeros(16)
        st = OrcFrameStats()
        ok = lib().orc_vo_stereo_callback_cn(self.h, _p(left), _p(right), w, h, w * cn, cn, _p(T), C.byref(st))
        assert ok >= 0, "channel count changed between frames"
        self.stats = st
        return bool(ok), T.reshape(4, 4)

    def features(self):
        n = lib().orc_vo_num_features(self.h)
        xy = np.zeros((max(n, 1), 2), np.float32); ages = np.zeros(max(n, 1), np.int32); st = np.zeros(max(n, 1), np.int32)
        lib().orc_vo_get_features(self.h, _p(xy), _p(ages), _p(st))
        return xy[:n], ages[:n], st[:n]

    def pose_guess(self):
        R = np.zeros(9); t = np.zeros(3)
        lib().orc_vo_get_pose_guess(self.h, _p(R), _p(t))
        return R.reshape(3, 3), t

    def last_tracks(self):
        n = lib().orc_vo_get_last_tracks(self.h, None, None, None, None, None, None)
        a = [np.zeros((max(n, 1), 2), np.float32) for _ in range(4)]
        world = np.zeros((max(n, 1), 3), np.float32); inl = np.zeros(max(n, 1), np.uint8)
        lib().orc_vo_get_last_tracks(self.h, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(world), _p(inl))
        return dict(pl0=a[0][:n], pr0=a[1][:n], pl1=a[2][:n], pr1=a[3][:n], world=world[:n], inlier=inl[:n])

    def __del__(self):
        try:
            lib().orc_vo_destroy(self.h)
        except Exception:
            pass
