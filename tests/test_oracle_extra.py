"""More pins for the oracle itself (no GPU): analytic properties, an independent numpy restatement of the
integer kernels, cv::RNG's published recurrence, and ground truth from the synthetic renderer."""
import numpy as np
import pytest

import oracle_lib as orc
import scenes


def np_pyr_down(img):
    """independent numpy restatement: separable [1 4 6 4 1], REFLECT_101, (sum+128)>>8"""
    a = img.astype(np.int64)
    p = np.pad(a, 2, mode="reflect")
    k = np.array([1, 4, 6, 4, 1])
    h = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(5))
    v = sum(k[i] * h[i:i + a.shape[0], :] for i in range(5))
    return ((v[::2, ::2] + 128) >> 8).astype(np.uint8)


@pytest.mark.parametrize("shape", [(64, 80), (97, 131), (33, 34)])
def test_pyr_down_matches_numpy(shape):
    img = scenes.random_texture(shape[0], shape[1], 4, smooth=0)
    assert np.array_equal(orc.pyr_down(img), np_pyr_down(img))


def test_scharr_matches_numpy_and_zero_border():
    img = scenes.random_texture(40, 50, 8, smooth=1)
    p = orc.Pyramid(img, (10, 10), 0)
    d = p.deriv(0).astype(np.int64)
    a = np.pad(img.astype(np.int64), 1, mode="reflect")
    dx = 3 * (a[:-2, 2:] - a[:-2, :-2]) + 10 * (a[1:-1, 2:] - a[1:-1, :-2]) + 3 * (a[2:, 2:] - a[2:, :-2])
    dy = 3 * (a[2:, :-2] - a[:-2, :-2]) + 10 * (a[2:, 1:-1] - a[:-2, 1:-1]) + 3 * (a[2:, 2:] - a[:-2, 2:])
    assert np.array_equal(d[..., 0], dx) and np.array_equal(d[..., 1], dy)


def test_pyramid_level_stop_rule():
    img = scenes.random_texture(376, 1241, 1, smooth=0)
    assert orc.Pyramid(img, (21, 21), 3).nlevels == 4          # 1241x376, 621x188, 311x94, 156x47
    assert orc.Pyramid(img, (21, 21), 4).nlevels == 5          # + 78x24 (> 21)
    assert orc.Pyramid(img, (21, 21), 7).nlevels == 5          # next would be 39x12 <= 21 -> stop
    small = scenes.random_texture(24, 24, 1, smooth=0)
    assert orc.Pyramid(small, (10, 10), 3).nlevels == 2        # 24, 12 ; next 6 <= 10


def test_rng_is_the_published_mwc_recurrence():
    state = 0xFFFFFFFFFFFFFFFF
    want = []
    for _ in range(20):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        want.append(state & 0xFFFFFFFF)
    assert orc.rng_sequence(20) == want


def test_ransac_update_num_iters():
    assert orc.ransac_update_num_iters(0.98, 0.0, 5, 100) == 0        # all inliers: denom < DBL_MIN -> 0
    assert orc.ransac_update_num_iters(0.98, 0.5, 5, 100) == 100      # log(0.02)/log(1-0.5^5) = 123 > 100
    assert orc.ransac_update_num_iters(0.98, 0.3, 5, 100) == 21
    assert orc.ransac_update_num_iters(0.98, 1.0, 5, 100) == 100


def test_rodrigues_round_trip_and_jacobian():
    rng = np.random.default_rng(0)
    for _ in range(20):
        r = rng.normal(size=3); r = r / np.linalg.norm(r) * rng.uniform(0.01, 3.0)      # |r| < pi
        R, J = orc.rodrigues_to_matrix(r)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-14 and abs(np.linalg.det(R) - 1) < 1e-14
        assert np.abs(orc.rodrigues_to_vector(R) - r).max() < 1e-10
        eps = 1e-7
        for i in range(3):
            d = np.zeros(3); d[i] = eps
            num = (orc.rodrigues_to_matrix(r + d)[0] - orc.rodrigues_to_matrix(r - d)[0]).reshape(9) / (2 * eps)
            assert np.abs(num - J[i]).max() < 1e-6
    R0, J0 = orc.rodrigues_to_matrix(np.zeros(3))
    assert np.array_equal(R0, np.eye(3))


def test_triangulate_analytic():
    """no reference test covers cv::triangulatePoints: pin it analytically instead."""
    from stereo_visual_odometry_amd import synthetic as syn
    Pl, Pr = syn.projection_matrices(syn.KITTI00)
    rng = np.random.default_rng(3)
    n = 500
    Z = rng.uniform(4, 90, n); X = rng.uniform(-20, 20, n); Y = rng.uniform(-3, 3, n)
    ul = 718.856 * X / Z + 607.1928; v = 718.856 * Y / Z + 185.2157; ur = ul - 386.1448 / Z
    pl = np.stack([ul, v], 1).astype(np.float32); pr = np.stack([ur, v], 1).astype(np.float32)
    xyz, hom = orc.triangulate(Pl, Pr, pl, pr)
    zz = 386.1448 / (pl[:, 0].astype(np.float64) - pr[:, 0].astype(np.float64))
    assert np.abs(xyz[:, 2] / zz - 1).max() < 1e-5
    assert np.abs(np.linalg.norm(hom, axis=1) - 1).max() < 1e-6          # unit-norm singular vector
    # reprojection into both views
    P = np.concatenate([xyz.astype(np.float64), np.ones((n, 1))], 1)
    for Pm, pts in ((Pl, pl), (Pr, pr)):
        q = P @ Pm.astype(np.float64).T
        assert np.abs(q[:, :2] / q[:, 2:3] - pts).max() < 2e-2


def test_epnp_recovers_pose_from_5_points():
    rng = np.random.default_rng(5)
    obj = np.stack([rng.uniform(-5, 5, 5), rng.uniform(-2, 2, 5), rng.uniform(8, 30, 5)], 1)
    Rt, _ = orc.rodrigues_to_matrix(np.array([0.02, -0.03, 0.01])); tt = np.array([0.1, 0.05, -0.7])
    pc = obj @ Rt.T + tt
    img = np.stack([700 * pc[:, 0] / pc[:, 2] + 600, 700 * pc[:, 1] / pc[:, 2] + 180], 1)
    R, t, err = orc.epnp(obj, img, 700, 700, 600, 180)
    assert err < 1e-6 and np.abs(R - Rt).max() < 1e-6 and np.abs(t - tt).max() < 1e-5


def test_oracle_pipeline_against_renderer_ground_truth():
    """Independent of OpenCV *and* of the oracle: the renderer's camera motion must come back out."""
    from stereo_visual_odometry_amd import synthetic as syn
    cal = dict(syn.KITTI00, width=640, height=240, cx=320.0, cy=120.0)
    seq = syn.StereoSequence(cal=cal, n_frames=4, seed=99, step=0.4, yaw_amp_deg=0.3)
    vo = orc.VisualOdometry(orc.default_config(win_w=21, win_h=21, max_translation_norm=2.0))
    vo.initalize_projection_matricies(*syn.projection_matrices(cal))
    ok, T = vo.stereo_callback(seq.left[0], seq.right[0])
    assert not ok and np.array_equal(T, np.eye(4)) and vo.stats.fail_reason == 1
    for k in range(1, 4):
        ok, T = vo.stereo_callback(seq.left[k], seq.right[k])
        gt = seq.relative_motion(k)
        assert ok and vo.stats.n_inliers > 200
        assert np.abs(T[:3, 3] - gt[:3, 3]).max() < 0.02
        assert np.abs(T[:3, :3] - gt[:3, :3]).max() < 5e-4


def test_oracle_failure_gates():
    from stereo_visual_odometry_amd import synthetic as syn
    cal = dict(syn.KITTI00, width=480, height=200, cx=240.0, cy=100.0)
    seq = syn.StereoSequence(cal=cal, n_frames=3, seed=5, step=0.4)
    vo = orc.VisualOdometry(orc.default_config())               # reference defaults: MAX_TRANSLATION_NORM = 0.1
    vo.initalize_projection_matricies(*syn.projection_matrices(cal))
    black = np.zeros_like(seq.left[0])
    assert vo.stereo_callback(black, black)[0] is False
    ok, T = vo.stereo_callback(black, black)
    assert not ok and vo.stats.fail_reason == 2 and vo.stats.second_pass == 1 and np.array_equal(T, np.eye(4))
    vo.stereo_callback(seq.left[0], seq.right[0])
    ok, T = vo.stereo_callback(seq.left[1], seq.right[1])
    # 0.4 m step: either still recovering from the stale pyramid (2) or rejected by the 0.1 m motion gate (4)
    assert not ok and vo.stats.fail_reason in (2, 4) and np.array_equal(T, np.eye(4))


def test_rng_state_table_of_the_product_equals_the_published_recurrence():
    """stereo_visual_odometry_amd/csrc/svo_rng_table.hpp holds the first 256 states of cv::RNG((uint64)-1) — k_compact draws a
    lone stream's first RANSAC chunk from it.  Every entry must be what rand.cpp's multiply-with-carry recurrence gives
    (state = (uint32)state * 4164903690 + (state >> 32)), and its low halves what the oracle's generator returns."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "stereo_visual_odometry_amd", "csrc", "svo_rng_table.hpp")).read()
    vals = [int(v, 16) for v in re.findall(r"0x([0-9A-Fa-f]{16})ull", txt)]
    assert len(vals) == 256
    st = 0xFFFFFFFFFFFFFFFF
    for v in vals:
        st = ((st & 0xFFFFFFFF) * 4164903690 + (st >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert v == st
    assert [v & 0xFFFFFFFF for v in vals[:64]] == orc.rng_sequence(64)
