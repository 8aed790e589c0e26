import sys, time, numpy as np
sys.path.insert(0, '.')
from stereo_visual_odometry_amd import api, synthetic as syn
cal = syn.KITTI00; W, H = cal["width"], cal["height"]
seq = syn.StereoSequence(cal=cal, n_frames=8, seed=0x5EED0002, step=0.5, cell_px=17.6)
for B in (1, 32):
    vo = api.BatchVisualOdometry(W, H, B, api.default_config(win_w=21, win_h=21, max_translation_norm=2.0))
    vo.initalize_projection_matricies(*syn.projection_matrices(cal))
    def pp(i):
        p = i % 14
        return p if p < 8 else 14 - p
    for i in range(4):
        vo.stereo_callback_batch([seq.left[pp(i)]] * B, [seq.right[pp(i)]] * B)
    n = 20
    t0 = time.perf_counter()
    for i in range(4, 4 + n):
        ok, T = vo.stereo_callback_batch([seq.left[pp(i)]] * B, [seq.right[pp(i)]] * B)
    dt = time.perf_counter() - t0
    print("host-image synchronous svo_process_batch, B=%d: %.1f frame-pairs/s (%.2f ms per call, ok=%s)" % (B, B * n / dt, dt / n * 1e3, ok.all()))
