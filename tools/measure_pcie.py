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

# the same call with the frames in page-locked buffers (svo_alloc_pinned): the DMA reads them in place, no staging memcpy
vo = api.BatchVisualOdometry(W, H, 1, api.default_config(win_w=21, win_h=21, max_translation_norm=2.0))
vo.initalize_projection_matricies(*syn.projection_matrices(cal))
pl = [api.PinnedImage((H, W)) for _ in range(8)]; pr = [api.PinnedImage((H, W)) for _ in range(8)]
for k in range(8):
    pl[k].array[:] = seq.left[k]; pr[k].array[:] = seq.right[k]
for i in range(4):
    vo.stereo_callback_batch([pl[pp(i)].array], [pr[pp(i)].array])
n = 40
t0 = time.perf_counter()
for i in range(4, 4 + n):
    ok, T = vo.stereo_callback_batch([pl[pp(i)].array], [pr[pp(i)].array])
dt = time.perf_counter() - t0
print("same, frames in page-locked host memory, B=1: %.1f frame-pairs/s (%.3f ms per call, ok=%s)" % (n / dt, dt / n * 1e3, ok.all()))
