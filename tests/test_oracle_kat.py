"""The reference's own known-answer tests (src/main.cpp:50-264) restated against the CPU oracle.

These pin the oracle.  test_movement (main.cpp:266-291) cannot be restated:
its cfe_cameras data is a missing blob in the reference.
"""
import numpy as np
import pytest

import oracle_lib as orc
import scenes


def test_bucket_empty():                                   # main.cpp:50-60
    b = orc.Bucket(0)
    ages = [6, 2, 3, 4, 5]; strengths = [60, 20, 30, 40, 50]
    for a, s in zip(ages, strengths):
        b.add_feature(1, 1, a, s)
    assert b.max_size == 0
    assert b.size() == 0


def test_bucket_nonempty():                                # main.cpp:62-78
    b = orc.Bucket(3)
    ages = [6, 2, 3, 4, 5]; strengths = [60, 20, 30, 40, 50]
    for a, s in zip(ages, strengths):
        b.add_feature(1, 1, a, s)
    assert b.max_size == 3
    assert b.size() == 3
    assert b.ages == [6, 4, 5]
    assert b.strengths == [60, 40, 50]


def test_featureset():                                     # main.cpp:102-127
    img = scenes.featureset_scene()
    xy, resp = orc.fast_detect(img, 1)
    ages = np.zeros(len(xy), np.int32)
    strengths = resp.astype(np.int32)
    h, w = img.shape
    fxy, fa, fs = orc.bucket_filter(w, h, xy, ages, strengths)          # appendFeaturesFromImage -> default grid
    assert (fa == 0).all()
    assert (fs <= 128).all()
    assert len(fa) == 11
    # only the isolated "hole" corner of each triangle survives; its score is 119
    assert (fs == 119).all()
    assert sorted(map(tuple, fxy.astype(int).tolist())) == [(20, 20 * (i + 1)) for i in range(11)]
    fxy2, fa2, fs2 = orc.bucket_filter(w, h, fxy, fa, fs, 1, 1, 0, 7)   # all in one 7-slot bucket
    assert len(fa2) == 7


def test_featureset_filter():                              # main.cpp:128-159
    rows = cols = 300
    bucket_height = (rows + 1) // 2
    pts = [(cols - 1.0, rows - 1.0)] * 15 + [(cols - 1.0, 0.0)] * 10 + [(0.0, float(bucket_height))] * 5
    xy = np.array(pts, np.float32); ages = np.zeros(30, np.int32); st = np.full(30, 40, np.int32)
    xy1, a1, s1 = orc.bucket_filter(cols, rows, xy, ages, st, 2, 2, 0, 11)
    assert len(a1) == 26
    xy2, a2, s2 = orc.bucket_filter(cols, rows, xy1, a1, s1, 2, 1, 0, 11)
    assert len(a2) == 21
    xy3, a3, s3 = orc.bucket_filter(cols, rows, xy1, a1, s1, 1, 2, 0, 11)
    assert len(a3) == 16


def test_find_unmoved_points():                            # main.cpp:161-172
    p1 = np.array([(i, i) for i in range(35)], np.float32)
    p2 = np.array([(i + (0 if i % 5 else 1), i + (0 if i % 7 else 1)) for i in range(35)], np.float32)
    ok = orc.find_close_points(p1, p2, 0.5)
    for i in range(35):
        assert bool(ok[i]) == bool((i % 5) and (i % 7))


def test_circular_matching():                              # main.cpp:174-209
    iL0, iR0, iL1, iR1 = scenes.circular_scene()
    cfg = orc.default_config()
    xy, resp = orc.fast_detect(iL0, cfg.fast_threshold)
    fxy, fa, fs = orc.bucket_filter(600, 600, xy, np.zeros(len(xy), np.int32), resp.astype(np.int32))
    assert len(fxy) == 121
    win = (cfg.win_w, cfg.win_h)
    pL0, pR0 = orc.Pyramid(iL0, win, cfg.max_level), orc.Pyramid(iR0, win, cfg.max_level)
    pL1, pR1 = orc.Pyramid(iL1, win, cfg.max_level), orc.Pyramid(iR1, win, cfg.max_level)
    # (a) the honest 4-image loop L0 -> L1 -> R1 -> R0 -> L0
    pl1, pr1, pr0, plc, ok = orc.circular_match(pL0, pR0, pL1, pR1, fxy, cfg)
    assert ok.sum() == 121
    assert np.abs(pl1 - (fxy + [1, 0])).max() < 0.15
    assert np.abs(pr1 - (fxy + [1, 1])).max() < 0.15
    assert np.abs(pr0 - (fxy + [0, 1])).max() < 0.15
    # (b) what the reference test really runs after its single-point call replaced the cached pyramids
    #     (SURVEY §4): L1 -> L1 -> R1 -> R1 -> L1, starting from the L0 feature positions
    pl1b, pr1b, pr0b, plcb, okb = orc.circular_match(pL1, pR1, pL1, pR1, fxy, cfg)
    assert okb.sum() == 121
    # boundary conditions: empty and single-point inputs must not crash
    orc.circular_match(pL0, pR0, pL1, pR1, np.zeros((0, 2), np.float32), cfg)
    _, _, _, _, ok1 = orc.circular_match(pL0, pR0, pL1, pR1, fxy[:1], cfg)
    assert ok1.tolist() == [1]


def test_camera_to_world():                                # main.cpp:211-264
    K, cam, world = scenes.camera_to_world_scene()
    ok, R, t, inliers, dbg = orc.camera_to_world(K, cam, world, np.eye(3), np.zeros(3))
    assert ok
    assert abs(t[0]) < 1e-6 and abs(t[1]) < 1e-6 and abs(t[2] - 1) < 1e-6
    expect = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]], np.float64)
    assert np.abs(R - expect).max() < 1e-8
    assert len(inliers) == 27
    assert inliers.tolist() == list(range(27))
