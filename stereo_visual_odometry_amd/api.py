"""Host-side mirror of the reference's C++ surface (include/vo.h) over the C-ABI of libsvo_hip.so.

Same names, argument meaning and failure behaviour as namespace visual_odometry:
VisualOdometry (vo.h:231-380), FeatureSet (vo.h:132-188), Bucket (vo.h:195-229) and the free
functions featureDetectionFast / deletePointsWithFailureStatus / deleteFeaturesWithFailureStatus /
findClosePoints / cameraToWorld / getInverseTransform (vo.h:393-470).  cv::Mat becomes numpy,
std::vector<cv::Point2f> becomes an (N,2) float32 array.  All arithmetic runs on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SvoConfig, SvoFrameStats, check, default_config, lib, ptr, u8frame, u8img

# the reference's constants (include/vo.h:53-127) for callers that used them by name
BUCKET_START_ROW, BUCKETS_ALONG_HEIGHT, BUCKETS_ALONG_WIDTH, FEATURES_PER_BUCKET = 4, 92, 160, 1
FEATURES_THRESHOLD, PRE_MATCHING_FEATURE_THRESHOLD, AGE_THRESHOLD, FAST_THRESHOLD = 15, 100, 20, 20
RANSAC_REPROJECTION_ERROR, RANSAC_ITERATIONS = 8.0, 100
OPTICAL_FLOW_MIN_EIG_THRESHOLD, CIRCULAR_MATCHING_SUCCESS_THRESHOLD = 0.001, 0.15
MAX_TRANSLATION_NORM, MAX_ROTATION_NORM = 0.1, 0.5


def _pts(p):
    return np.ascontiguousarray(p, np.float32).reshape(-1, 2)


# ---------------------------------------------------------------------------- free functions
def featureDetectionFast(image, fast_threshold, device=0):
    """vo.h:393-395 — returns (points (N,2) f32, response_strengths (N,) f32), raster order."""
    img = u8img(image)
    h, w = img.shape
    cap = 4096
    while True:
        xy = np.zeros((cap, 2), np.float32)
        resp = np.zeros(cap, np.float32)
        n = C.c_int(0)
        check(lib.svo_fast_detect(device, ptr(img), w, h, w, int(fast_threshold), cap, ptr(xy), ptr(resp), C.byref(n)))
        if n.value <= cap:
            return xy[:n.value].copy(), resp[:n.value].copy()
        cap = n.value


def fastScoreMap(image, fast_threshold, device=0):
    img = u8img(image)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    check(lib.svo_fast_score_map(device, ptr(img), w, h, w, int(fast_threshold), ptr(out)))
    return out


def deletePointsWithFailureStatus(point_vector, is_ok):
    """vo.h:406-407 — stable removal of entries whose flag is false (host-side list op)."""
    is_ok = np.asarray(is_ok, bool)
    pv = _pts(point_vector)
    keep = np.ones(len(pv), bool)
    keep[:len(is_ok)] = is_ok
    return pv[keep]


def deleteFeaturesWithFailureStatus(features, is_ok):
    """vo.h:416-417"""
    is_ok = np.asarray(is_ok, bool)
    keep = np.ones(features.size(), bool)
    keep[:len(is_ok)] = is_ok
    features.points = features.points[keep]
    features.ages = features.ages[keep]
    features.strengths = features.strengths[keep]


def findClosePoints(points_1, points_2, threshold, device=0):
    """vo.h:430-432 — v[i] = max(|dx|,|dy|) <= threshold."""
    p1, p2 = _pts(points_1), _pts(points_2)
    ok = np.zeros(len(p1), np.uint8)
    check(lib.svo_find_close_points(device, len(p1), ptr(p1), ptr(p2), C.c_float(threshold), ptr(ok)))
    return ok.astype(bool)


def cameraToWorld(cameraProjection, cameraPoints, worldPoints, rotation, translation,
                  iterations=RANSAC_ITERATIONS, reprojection_error=RANSAC_REPROJECTION_ERROR, confidence=0.98, device=0):
    """vo.h:452-456 — returns ((inliers, success), rotation, translation, iterations_run): the reference's pair, then its two
    in/out arguments (the updated values on success, the inputs on failure, vo.cpp:307-311), then how many RANSAC iterations
    the adaptive loop ran (a counter the reference does not expose)."""
    K = np.ascontiguousarray(cameraProjection, np.float32).reshape(-1)[:9].copy() if np.size(cameraProjection) == 9 else \
        np.ascontiguousarray(np.asarray(cameraProjection, np.float32)[:3, :3]).reshape(9)
    cam = _pts(cameraPoints)
    world = np.ascontiguousarray(worldPoints, np.float32).reshape(-1, 3)
    n = len(cam)
    R = np.ascontiguousarray(rotation, np.float64).reshape(9).copy()
    t = np.ascontiguousarray(translation, np.float64).reshape(3).copy()
    inl = np.zeros(max(n, 1), np.int32)
    nin, ok, iters = C.c_int(0), C.c_int(0), C.c_int(0)
    check(lib.svo_camera_to_world(device, ptr(K), n, ptr(cam), ptr(world), ptr(R), ptr(t), ptr(inl), C.byref(nin), C.byref(ok),
                                  int(iterations), C.c_float(reprojection_error), C.c_float(confidence), C.byref(iters)))
    return (inl[:nin.value].copy(), bool(ok.value)), R.reshape(3, 3), t.reshape(3, 1), iters.value


def getInverseTransform(rotation, translation, device=0):
    """vo.h:469-470"""
    R = np.ascontiguousarray(rotation, np.float64).reshape(9)
    t = np.ascontiguousarray(translation, np.float64).reshape(3)
    T = np.zeros(16)
    check(lib.svo_inverse_transform(device, ptr(R), ptr(t), ptr(T)))
    return T.reshape(4, 4)


def triangulatePoints(Pl, Pr, pts_l, pts_r, device=0):
    """cv::triangulatePoints + cv::convertPointsFromHomogeneous as used at vo.cpp:89-94 -> (N,3) f32."""
    Pl = np.ascontiguousarray(Pl, np.float32).reshape(12)
    Pr = np.ascontiguousarray(Pr, np.float32).reshape(12)
    a, b = _pts(pts_l), _pts(pts_r)
    xyz = np.zeros((len(a), 3), np.float32)
    check(lib.svo_triangulate(device, ptr(Pl), ptr(Pr), len(a), ptr(a), ptr(b), ptr(xyz)))
    return xyz


def buildOpticalFlowPyramid(image, win, max_level, device=0):
    """cv::buildOpticalFlowPyramid (vo.cpp:50,52,200,201) -> list of u8 level images."""
    img = u8img(image)
    h, w = img.shape
    cap = int(w * h * 1.5) + 64
    buf = np.zeros(cap, np.uint8)
    nl = C.c_int(0)
    check(lib.svo_build_pyramid(device, ptr(img), w, h, w, int(win), int(max_level), ptr(buf), C.c_int64(cap), C.byref(nl)))
    out, off = [], 0
    for _ in range(nl.value):
        out.append(buf[off:off + w * h].reshape(h, w).copy())
        off += w * h
        w, h = (w + 1) // 2, (h + 1) // 2
    return out


def calcOpticalFlowPyrLK(prev_img, next_img, prev_pts, win=10, max_level=3, max_count=30, epsilon=1e-4,
                         min_eig_threshold=OPTICAL_FLOW_MIN_EIG_THRESHOLD, device=0):
    """cv::calcOpticalFlowPyrLK as called at vo.cpp:203-215 -> (next_pts, status)."""
    a, b = u8img(prev_img), u8img(next_img)
    h, w = a.shape
    p = _pts(prev_pts)
    out = np.zeros_like(p)
    st = np.zeros(len(p), np.uint8)
    check(lib.svo_lk_track(device, ptr(a), ptr(b), w, h, w, len(p), ptr(p), ptr(out), ptr(st), int(win), int(max_level),
                           int(max_count), C.c_double(epsilon), C.c_double(min_eig_threshold)))
    return out, st


def circularMatching(cfg, l0, r0, l1, r1, points_left_t0, device=0):
    """The four LK passes + mask of VisualOdometry::circularMatching (vo.cpp:203-230) on four images,
    without the compaction -> (pl1, pr1, pr0, pl0_circle, ok)."""
    imgs = [u8img(i) for i in (l0, r0, l1, r1)]
    h, w = imgs[0].shape
    p = _pts(points_left_t0)
    n = len(p)
    outs = [np.zeros((n, 2), np.float32) for _ in range(4)]
    ok = np.zeros(n, np.uint8)
    check(lib.svo_circular_match(device, C.byref(cfg), ptr(imgs[0]), ptr(imgs[1]), ptr(imgs[2]), ptr(imgs[3]), w, h, w, n,
                                 ptr(p), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), ptr(ok)))
    return outs[0], outs[1], outs[2], outs[3], ok


# ---------------------------------------------------------------------------- FeatureSet / Bucket
class FeatureSet:
    """vo.h:132-188 — parallel arrays points / ages / strengths."""

    def __init__(self, device=0):
        self.points = np.zeros((0, 2), np.float32)
        self.ages = np.zeros(0, np.int32)
        self.strengths = np.zeros(0, np.int32)
        self.device = device

    def size(self):
        return len(self.points)

    def clear(self):
        self.__init__(self.device)

    def filterByBucketLocationInternal(self, image, buckets_along_height, buckets_along_width, bucket_start_row,
                                       features_per_bucket):
        h, w = np.asarray(image).shape[:2]
        xy = np.ascontiguousarray(self.points, np.float32).reshape(-1, 2).copy()
        ages = np.ascontiguousarray(self.ages, np.int32).copy()
        st = np.ascontiguousarray(self.strengths, np.int32).copy()
        n = C.c_int(len(ages))
        check(lib.svo_bucket_filter(self.device, w, h, C.byref(n), ptr(xy), ptr(ages), ptr(st), buckets_along_height,
                                    buckets_along_width, bucket_start_row, features_per_bucket, AGE_THRESHOLD, FAST_THRESHOLD))
        self.points, self.ages, self.strengths = xy[:n.value].copy(), ages[:n.value].copy(), st[:n.value].copy()

    def filterByBucketLocation(self, image):
        self.filterByBucketLocationInternal(image, BUCKETS_ALONG_HEIGHT, BUCKETS_ALONG_WIDTH, BUCKET_START_ROW, FEATURES_PER_BUCKET)

    def appendFeaturesFromImage(self, image, fast_threshold, cfg=None):
        """vo.h:186-187 — FAST + append (age 0) + default-grid bucket filter, fused on the GPU."""
        img = u8img(image)
        h, w = img.shape
        cfg = cfg if cfg is not None else default_config()
        rows = max(cfg.buckets_along_height - cfg.bucket_start_row, 0)
        cap = max(rows * cfg.buckets_along_width, 64, self.size())
        xy = np.zeros((cap, 2), np.float32); ages = np.zeros(cap, np.int32); st = np.zeros(cap, np.int32)
        n0 = self.size()
        xy[:n0], ages[:n0], st[:n0] = self.points, self.ages, self.strengths
        n = C.c_int(n0)
        check(lib.svo_append_features_from_image(self.device, C.byref(cfg), ptr(img), w, h, w, int(fast_threshold), cap,
                                                 C.byref(n), ptr(xy), ptr(ages), ptr(st)))
        self.points, self.ages, self.strengths = xy[:n.value].copy(), ages[:n.value].copy(), st[:n.value].copy()


class Bucket:
    """vo.h:195-229.  add_feature applies the insertion rule (feature_set.cpp:20-53) on the GPU: the bucket's current content, in
    slot order, followed by the new feature goes through a 1x1 grid of capacity max_size — the stored features refill their
    slots in the same order, then the new one meets exactly the state the reference's bucket is in (at most max_size + 1
    inputs per insertion)."""

    def __init__(self, max_size, device=0):
        self.max_size = max_size
        self.device = device
        self.features = FeatureSet(device)

    def compute_score(self, age, strength):
        q = abs(strength - FAST_THRESHOLD) // 20
        return age + (q if strength >= FAST_THRESHOLD else -q)       # C++ int division truncates toward zero

    def add_feature(self, point, age, strength):
        if not self.max_size:
            return
        fs = self.features
        fs.points = np.concatenate([np.asarray(fs.points, np.float32).reshape(-1, 2), np.array([[point[0], point[1]]], np.float32)])
        fs.ages = np.concatenate([fs.ages, np.array([age], np.int32)]); fs.strengths = np.concatenate([fs.strengths, np.array([strength], np.int32)])
        side = int(max(2, np.ceil(fs.points.max()) + 1))
        fs.filterByBucketLocationInternal(np.zeros((side, side), np.uint8), 1, 1, 0, self.max_size)

    def size(self):
        return self.features.size()


class PinnedImage:
    """An (H, W) or (H, W, 3) uint8 numpy array living in page-locked host memory (svo_alloc_pinned): frames kept in such
    buffers are read by the DMA engines in place instead of being copied to a staging buffer first (include/svo.h)."""

    def __init__(self, shape):
        self._n = int(np.prod(shape))
        self._p = lib.svo_alloc_pinned(self._n)
        if not self._p:
            raise MemoryError("svo_alloc_pinned(%d)" % self._n)
        self.array = np.ctypeslib.as_array((C.c_uint8 * self._n).from_address(self._p)).reshape(shape)

    def __del__(self):
        try:
            if self._p:
                lib.svo_free_pinned(self._p); self._p = None
        except Exception:
            pass


# ---------------------------------------------------------------------------- VisualOdometry
class BatchVisualOdometry:
    """n_seq independent VisualOdometry instances advancing in lock-step on one GPU."""

    def __init__(self, width, height, n_seq=1, cfg=None, device=0):
        self.cfg = cfg if cfg is not None else default_config()
        self.n_seq, self.width, self.height, self.device = n_seq, width, height, device
        self._h = C.c_void_p()
        check(lib.svo_create(C.byref(self.cfg), device, n_seq, width, height, C.byref(self._h)))
        self.stats = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib.svo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def initalize_projection_matricies(self, leftCameraProjection, rightCameraProjection, seq=-1):
        Pl = np.ascontiguousarray(leftCameraProjection, np.float32).reshape(12)
        Pr = np.ascontiguousarray(rightCameraProjection, np.float32).reshape(12)
        check(lib.svo_set_projection(self._h, seq, ptr(Pl), ptr(Pr)))

    def stereo_callback_batch(self, lefts, rights):
        """lists of n_seq host images -> (ok (n_seq,) bool, T (n_seq,4,4) f64)."""
        L = [u8frame(i) for i in lefts]; R = [u8frame(i) for i in rights]
        assert len(L) == self.n_seq and len(R) == self.n_seq
        cn = max(1, self.cfg.channels)
        assert all(i.shape == (self.height, self.width) + ((3,) if cn == 3 else ()) for i in L + R), "image shape / channels"
        lp = (C.c_void_p * self.n_seq)(*[i.ctypes.data for i in L])
        rp = (C.c_void_p * self.n_seq)(*[i.ctypes.data for i in R])
        T = np.zeros((self.n_seq, 16)); ok = np.zeros(self.n_seq, np.int32)
        st = (SvoFrameStats * self.n_seq)()
        check(lib.svo_process_batch(self._h, lp, rp, self.width * cn, 0, ptr(T), ptr(ok), st))
        self.stats = list(st)
        return ok.astype(bool), T.reshape(self.n_seq, 4, 4)

    def process_device(self, left_ptrs, right_ptrs, stride):
        """device pointers (ints) -> same outputs; inputs stay resident in HBM."""
        lp = (C.c_void_p * self.n_seq)(*left_ptrs); rp = (C.c_void_p * self.n_seq)(*right_ptrs)
        T = np.zeros((self.n_seq, 16)); ok = np.zeros(self.n_seq, np.int32)
        st = (SvoFrameStats * self.n_seq)()
        check(lib.svo_process_batch(self._h, lp, rp, stride, 1, ptr(T), ptr(ok), st))
        self.stats = list(st)
        return ok.astype(bool), T.reshape(self.n_seq, 4, 4)

    def submit_device(self, left_ptrs, right_ptrs, stride):
        lp = (C.c_void_p * self.n_seq)(*left_ptrs); rp = (C.c_void_p * self.n_seq)(*right_ptrs)
        check(lib.svo_submit_batch(self._h, lp, rp, stride))

    def collect(self):
        T = np.zeros((self.n_seq, 16)); ok = np.zeros(self.n_seq, np.int32)
        st = (SvoFrameStats * self.n_seq)()
        check(lib.svo_collect(self._h, ptr(T), ptr(ok), st))
        self.stats = list(st)
        return ok.astype(bool), T.reshape(self.n_seq, 4, 4)

    def set_stage_timing(self, on=True):
        """Record the stage-boundary events of every frame from now on (svo_set_stage_timing): last_timing()'s LK time and
        stage_timing() need them; off by default (they cost a lone stream ~10 us per frame)."""
        check(lib.svo_set_stage_timing(self._h, int(bool(on))))

    def last_timing(self):
        lk, fr = C.c_float(0), C.c_float(0)
        check(lib.svo_get_last_timing(self._h, C.byref(lk), C.byref(fr)))
        return lk.value, fr.value

    STAGES = ("ingest+pyramid", "detect", "lk", "compact+triangulate", "pnp")

    def stage_timing(self):
        """HIP-event milliseconds of the last collected frame per pipeline stage (svo_get_stage_timing) -> dict."""
        ms = (C.c_float * 5)()
        check(lib.svo_get_stage_timing(self._h, ms))
        return dict(zip(self.STAGES, [float(x) for x in ms]))

    def stream(self):
        return lib.svo_get_stream(self._h)

    def features(self, seq=0):
        cap = 1 << 15
        xy = np.zeros((cap, 2), np.float32); ages = np.zeros(cap, np.int32); st = np.zeros(cap, np.int32)
        n = check(lib.svo_get_features(self._h, seq, cap, ptr(xy), ptr(ages), ptr(st)))
        return xy[:n].copy(), ages[:n].copy(), st[:n].copy()

    def last_tracks(self, seq=0):
        cap = 1 << 15
        a = [np.zeros((cap, 2), np.float32) for _ in range(4)]
        world = np.zeros((cap, 3), np.float32); inl = np.zeros(cap, np.uint8)
        n = check(lib.svo_get_last_tracks(self._h, seq, cap, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(world), ptr(inl)))
        return dict(pl0=a[0][:n].copy(), pr0=a[1][:n].copy(), pl1=a[2][:n].copy(), pr1=a[3][:n].copy(),
                    world=world[:n].copy(), inlier=inl[:n].copy())


class VisualOdometry(BatchVisualOdometry):
    """vo.h:231-380 — one stereo stream.  stereo_callback(left, right) -> (success, 4x4 float64)."""

    def __init__(self, width=None, height=None, cfg=None, device=0):
        self._args = (cfg, device)
        self._created = False
        if width is not None:
            super().__init__(width, height, 1, cfg, device)
            self._created = True
        self._P = None
        self._timing = None

    def set_stage_timing(self, on=True):
        self._timing = bool(on)
        if self._created:
            super().set_stage_timing(on)

    def initalize_projection_matricies(self, leftCameraProjection, rightCameraProjection, seq=-1):
        self._P = (leftCameraProjection, rightCameraProjection)
        if self._created:
            super().initalize_projection_matricies(leftCameraProjection, rightCameraProjection)

    def stereo_callback(self, image_left, image_right):
        L, R = u8frame(image_left), u8frame(image_right)
        if not self._created:                         # the reference learns the image size (and type) from the first frame
            cfg, device = self._args
            cfg = _lib.copy_config(cfg) if cfg is not None else default_config()       # never write into the caller's struct
            cfg.channels = 3 if L.ndim == 3 else 1    # colour Mats, as the reference CLI feeds them (main.cpp:38-46)
            super().__init__(L.shape[1], L.shape[0], 1, cfg, device)
            self._created = True
            # like the reference, a first frame needs no projection matrices (vo.cpp:47-56 only caches); until
            # initalize_projection_matricies is called they are all-zero, as the reference's empty Mats effectively are
            super().initalize_projection_matricies(*(self._P if self._P is not None else (np.zeros(12, np.float32), np.zeros(12, np.float32))))
            if self._timing is not None:
                super().set_stage_timing(self._timing)
        self._check_frame(L, "left"); self._check_frame(R, "right")
        T = np.zeros(16)
        st = SvoFrameStats()
        rc = check(lib.svo_process(self._h, ptr(L), ptr(R), L.strides[0], ptr(T), C.byref(st)))
        self.stats = st
        return bool(rc), T.reshape(4, 4)

    def _check_frame(self, img, which):
        """cv::Mat carries size and type and OpenCV asserts on a mismatch; a raw pointer does not: check before the C call."""
        want = (self.height, self.width) + ((3,) if self.cfg.channels == 3 else ())
        if img.shape != want:
            raise ValueError("%s image has shape %s, the context was created for %s" % (which, img.shape, want))

    def circularMatching(self, imgLeftT1, imgRightT1, pointsLeftT0, current_features):
        """vo.h:374-379, vo.cpp:169-240.  Tracks pointsLeftT0 around T0-left -> T1-left -> T1-right -> T0-right -> T0-left against
        the pyramid pair this object cached on the device in its previous stereo_callback / circularMatching, removes every point
        (and its feature, in place) that lost an LK status or does not close the loop, and makes the T1 pyramids the cached pair
        (vo.cpp:231-232) — the state stereo_callback itself works on, as in the reference.
        Returns (pointsLeftT0, pointsRightT0, pointsLeftT1, pointsRightT1) — Python has no out-parameters."""
        p0 = _pts(pointsLeftT0)
        empty = np.zeros((0, 2), np.float32)
        if len(p0) == 0:
            return p0, empty, empty, empty                                                  # vo.cpp:179-181
        if not self._created:
            raise RuntimeError("circularMatching: no cached pyramids (call stereo_callback first)")
        l1, r1 = u8frame(imgLeftT1), u8frame(imgRightT1)
        self._check_frame(l1, "left"); self._check_frame(r1, "right")
        n = len(p0)
        outs = [np.zeros((n, 2), np.float32) for _ in range(4)]
        ok = np.zeros(n, np.uint8)
        check(lib.svo_circular_matching(self._h, ptr(l1), ptr(r1), l1.strides[0], n, ptr(p0), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]),
                                        ptr(outs[3]), ptr(ok)))
        pl1, pr1, pr0 = outs[0], outs[1], outs[2]
        keep = ok.astype(bool)
        deleteFeaturesWithFailureStatus(current_features, keep)                             # :233
        return p0[keep], pr0[keep], pl1[keep], pr1[keep]                                    # :234-238

    def matchingFeatures(self, imageLeftT0, imageRightT0, imageLeftT1, imageRightT1, currentVOFeatures):
        """vo.h:354-362, vo.cpp:315-366 -> (pointsLeftT0, pointsRightT0, pointsLeftT1, pointsRightT1); the feature set is updated in place."""
        currentVOFeatures.appendFeaturesFromImage(imageLeftT0, FAST_THRESHOLD)              # :325
        if currentVOFeatures.size() < PRE_MATCHING_FEATURE_THRESHOLD:                       # :327-332
            currentVOFeatures.appendFeaturesFromImage(imageLeftT0, FAST_THRESHOLD // 4)
        # as in the reference, imageLeftT0 only feeds FAST; the loop's T0 side is the cached pyramid pair (prime with stereo_callback)
        pl0, pr0, pl1, pr1 = self.circularMatching(imageLeftT1, imageRightT1, currentVOFeatures.points.copy(), currentVOFeatures)
        h, w = u8frame(imageLeftT1).shape[:2]
        inside = np.ones(len(pl0), bool)                                                    # :341-359
        for q in (pl0, pl1, pr0, pr1):
            inside &= ~((q[:, 0] < 0) | (q[:, 1] < 0) | (q[:, 1] >= h) | (q[:, 0] >= w))
        deleteFeaturesWithFailureStatus(currentVOFeatures, inside)                          # :360-364
        return pl0[inside], pr0[inside], pl1[inside], pr1[inside]
