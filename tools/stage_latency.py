"""Synchronous single-sequence latency (host images in, pose out) and the per-stage HIP-event times, static and mover scene.\nusage: python tools/stage_latency.py"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_visual_odometry_amd import api, synthetic as syn
cal = syn.KITTI00; W, H = cal["width"], cal["height"]
for movers in (0.0, 0.3):
    seq = syn.StereoSequence(cal=cal, n_frames=8, seed=0x5EED0002, step=0.5, cell_px=17.6, movers=movers)
    vo = api.BatchVisualOdometry(W, H, 1, api.default_config(win_w=21, win_h=21, max_translation_norm=2.0))
    vo.initalize_projection_matricies(*syn.projection_matrices(cal))
    pp = lambda i: (i % 14) if (i % 14) < 8 else 14 - (i % 14)
    for i in range(6): vo.stereo_callback_batch([seq.left[pp(i)]], [seq.right[pp(i)]])
    n = 40; st = np.zeros(5)
    t0 = time.perf_counter()
    for i in range(6, 6 + n):
        ok, T = vo.stereo_callback_batch([seq.left[pp(i)]], [seq.right[pp(i)]])
    dt = time.perf_counter() - t0
    try: vo.set_stage_timing(True)
    except Exception: pass
    for i in range(6 + n, 6 + 2 * n):
        vo.stereo_callback_batch([seq.left[pp(i)]], [seq.right[pp(i)]]); st += np.array(list(vo.stage_timing().values()))
    print("movers %.1f: %.3f ms per call; stages us: %s" % (movers, dt / n * 1e3, np.round(st / n * 1e3, 1)))
