#!/usr/bin/env python3
"""Freezes what the CPU oracle produces on the 8 real run1 stereo pairs of run1_frames_0_7.npz (reference defaults:
10x10 window, maxLevel 3, K = 100) into tests/golden/run1_oracle_outputs.npz: per frame the ok flag, the 4x4 transform,
every frame counter, and the feature set / track lists after the frame.  A regression pin across rounds for BOTH sides:
tests/test_golden_outputs.py checks the oracle (CPU) and the HIP path (GPU) against it.
Data only (inputs come from the fixture, outputs from oracle/): python tests/golden/make_run1_outputs.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as orc                                              # noqa: E402
from stereo_visual_odometry_amd import synthetic as syn               # noqa: E402


def run(vo, d):
    out = {}
    for k in range(len(d["left"])):
        ok, T = vo.stereo_callback(d["left"][k], d["right"][k])
        st = vo.stats if isinstance(vo.stats, dict) else (vo.stats.as_dict() if hasattr(vo.stats, "as_dict") else
                                                          {f[0]: getattr(vo.stats, f[0]) for f in vo.stats._fields_})
        out["ok_%d" % k] = np.array(int(ok)); out["T_%d" % k] = np.asarray(T, np.float64)
        out["stats_%d" % k] = np.array([st[n] for n in STAT_NAMES], np.int64)
        xy, age, strength = vo.features()
        out["feat_xy_%d" % k] = np.asarray(xy, np.float32); out["feat_age_%d" % k] = np.asarray(age, np.int32)
        out["feat_str_%d" % k] = np.asarray(strength, np.int32)
        if k > 0:
            tr = vo.last_tracks()
            for key in ("pl0", "pr0", "pl1", "pr1"):
                out["%s_%d" % (key, k)] = np.asarray(tr[key], np.float32)
            out["inlier_%d" % k] = np.asarray(tr["inlier"], np.uint8)
    return out


STAT_NAMES = ["n_after_detect", "second_pass", "n_into_lk", "n_after_circular", "n_after_bounds", "n_inliers", "ransac_iters",
              "fail_reason", "n_features_out"]


def main():
    d = np.load(os.path.join(HERE, "run1_frames_0_7.npz"))
    vo = orc.VisualOdometry(orc.default_config())
    vo.initalize_projection_matricies(*syn.projection_matrices(syn.RUN1))
    out = run(vo, d)
    path = os.path.join(HERE, "run1_oracle_outputs.npz")
    np.savez_compressed(path, stat_names=np.array(STAT_NAMES), **out)
    print(path, os.path.getsize(path), [int(out["stats_%d" % k][2]) for k in range(8)])


if __name__ == "__main__":
    main()
