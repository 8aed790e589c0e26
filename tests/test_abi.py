"""No-GPU checks of the drop-in boundary: libsvo_hip.so loads and exports every symbol include/svo.h
declares; calls that need a device fail loudly (no CPU fallback); host-only entry points work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "svo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svo_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from stereo_visual_odometry_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(_lib.lib, s), "libsvo_hip.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms


def test_config_default_matches_reference_constants():
    from stereo_visual_odometry_amd import _lib
    c = _lib.default_config()
    assert (c.bucket_start_row, c.buckets_along_height, c.buckets_along_width, c.features_per_bucket) == (4, 92, 160, 1)  # vo.h:53-65
    assert (c.features_threshold, c.pre_matching_feature_threshold, c.age_threshold, c.fast_threshold) == (15, 100, 20, 20)
    assert c.ransac_reprojection_error == 8 and c.ransac_iterations == 100
    assert c.optical_flow_min_eig_threshold == 0.001 and c.circular_matching_success_threshold == 0.15
    assert c.max_translation_norm == 0.1 and c.max_rotation_norm == 0.5
    assert (c.win_w, c.win_h, c.max_level, c.lk_max_count) == (10, 10, 3, 30) and c.lk_epsilon == 1e-4
    assert abs(c.ransac_confidence - 0.98) < 1e-7 and c.max_features == 0


def test_oracle_and_product_config_structs_agree():
    import oracle_lib as orc
    from stereo_visual_odometry_amd import _lib
    a, b = _lib.default_config(), orc.default_config()
    assert [f[0] for f in a._fields_] == [f[0] for f in b._fields_]
    for f, _ in a._fields_:
        assert getattr(a, f) == getattr(b, f), f


@pytest.mark.gpu
def test_inverse_transform_entry_point():
    from stereo_visual_odometry_amd import api
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    t = np.array([0.1, -0.2, 0.3])
    T = api.getInverseTransform(R, t)
    M = np.eye(4); M[:3, :3] = R; M[:3, 3] = t
    assert np.abs(T @ M - np.eye(4)).max() < 1e-15


def test_no_cpu_fallback():
    """Without a GPU every compute entry point must fail with SVO_ERR_HIP, never silently compute."""
    from stereo_visual_odometry_amd import _lib, api
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.SvoError):
        api.featureDetectionFast(np.zeros((32, 32), np.uint8), 20)
    with pytest.raises(_lib.SvoError):                      # even the 12-flop closed form runs on the device
        api.getInverseTransform(np.eye(3), np.zeros(3))
    with pytest.raises(_lib.SvoError):
        api.BatchVisualOdometry(64, 64, 1)
    with pytest.raises(_lib.SvoError):
        api.findClosePoints(np.zeros((1, 2)), np.zeros((1, 2)), 0.5)


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or open anything under oracle/."""
    pkg = os.path.join(ROOT, "stereo_visual_odometry_amd")
    for dp, _, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "libsvo_oracle" not in txt and "orc_" not in txt, os.path.join(dp, f)
                assert not re.search(r"#include\s+[\"<][^\">]*oracle", txt), os.path.join(dp, f)
    import subprocess
    so = os.path.join(pkg, "libsvo_hip.so")
    out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_channel_arguments_are_validated_before_any_device_work():
    from stereo_visual_odometry_amd import _lib
    for channels, win, expect in ((2, 10, "channels must be 1 or 3"), (3, 31, "not built for 3-channel")):
        cfg = _lib.default_config(win_w=win, win_h=win); cfg.channels = channels
        h = C.c_void_p()
        rc = _lib.lib.svo_create(C.byref(cfg), 0, 1, 640, 480, C.byref(h))
        assert rc == _lib.SVO_ERR_ARG and expect in _lib.lib.svo_last_error().decode() and not h.value


# ---------------------------------------------------------------------------- libsvo_rccl.so (include/svo_gather.h)
def _gather_header_symbols():
    import re
    txt = open(os.path.join(ROOT, "include", "svo_gather.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svo_[a-z_0-9]+)\s*\(", txt)))


def test_gather_library_exports_every_declared_symbol():
    """include/svo_gather.h <-> libsvo_rccl.so (no GPU needed: loading and symbol lookup only)"""
    lib = C.CDLL(os.path.join(ROOT, "stereo_visual_odometry_amd", "libsvo_rccl.so"))
    syms = _gather_header_symbols()
    assert syms == ["svo_gather_last_error", "svo_gather_pose_streams", "svo_gather_pose_streams_ragged"]
    for s in syms:
        assert hasattr(lib, s), s
    # and the core library stays free of RCCL: a single-GPU integration never loads it
    import subprocess
    deps = subprocess.run(["ldd", os.path.join(ROOT, "stereo_visual_odometry_amd", "libsvo_hip.so")], capture_output=True, text=True).stdout
    assert "rccl" not in deps
