"""cfg1 (the reference's own run1 set: 512x288 BGR, LK 10x10, 3 levels, K=100) through VisualOdometry.stereo_callback, colour and gray:\nms per frame pair and per-stage HIP-event times, on the committed 48-frame fixture.  usage: python tools/cfg1_latency.py"""
import sys, time, lzma, io, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
from stereo_visual_odometry_amd import api
def load(p):
    return np.load(io.BytesIO(lzma.open(p).read()))
L = load('tests/golden/run1_bgr_left_0_47.npy.xz'); R = load('tests/golden/run1_bgr_right_0_47.npy.xz')
print(L.shape, L.dtype)
P = np.array([[320., 0, 256, 0], [0, 320, 144, 0], [0, 0, 1, 0]], np.float32); Pr = P.copy(); Pr[0, 3] = -32.0
for mode in ('bgr', 'gray', 'bgr, float sums (the reference\'s own LK rounding)'):
    vo = api.VisualOdometry(cfg=api.default_config(lk_float_sums=1 if 'float' in mode else 0)); vo.initalize_projection_matricies(P, Pr)
    fr = (lambda a: a) if mode.startswith('bgr') else (lambda a: np.ascontiguousarray(a[..., 1]))
    for k in range(8): vo.stereo_callback(fr(L[k]), fr(R[k]))
    t0 = time.perf_counter(); n = 0
    for rep in range(3):
        for k in range(8, 48): vo.stereo_callback(fr(L[k]), fr(R[k])); n += 1
    dt = time.perf_counter() - t0
    print(mode, '%.3f ms per frame pair, %.0f fps' % (dt / n * 1e3, n / dt), 'features', vo.stats.n_into_lk, 'inliers', vo.stats.n_inliers)
    vo.set_stage_timing(True); st = np.zeros(5); m = 0
    for k in range(8, 48): vo.stereo_callback(fr(L[k]), fr(R[k])); st += np.array(list(vo.stage_timing().values())); m += 1
    print('   stages us', np.round(st / m * 1e3, 1), 'second pass', vo.stats.second_pass, 'ransac iters', vo.stats.ransac_iters)
