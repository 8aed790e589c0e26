"""Edge cases of the C-ABI on a real GPU: strides, argument checking, ring limits, lifetime, tiny images."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc
import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from stereo_visual_odometry_amd import api as a
    assert a._lib.device_count() >= 1
    return a


def small_seq(n=3, seed=4):
    from stereo_visual_odometry_amd import synthetic as syn
    cal = dict(syn.KITTI00, width=320, height=160, cx=160.0, cy=80.0)
    return syn.StereoSequence(cal=cal, n_frames=n, seed=seed, step=0.3), cal


def test_strided_host_images_equal_packed(api):
    """svo_process with stride > width (an ROI of a wider buffer, as cv::Mat::step allows) == the packed image."""
    from stereo_visual_odometry_amd import synthetic as syn
    seq, cal = small_seq()
    Pl, Pr = syn.projection_matrices(cal)
    cfg = api.default_config(max_translation_norm=2.0)
    a = api.VisualOdometry(cfg=cfg); a.initalize_projection_matricies(Pl, Pr)
    h, w = seq.left[0].shape
    ctx = C.c_void_p()
    api.check(api.lib.svo_create(C.byref(cfg), 0, 1, w, h, C.byref(ctx)))
    api.check(api.lib.svo_set_projection(ctx, -1, api.ptr(Pl.reshape(12)), api.ptr(Pr.reshape(12))))
    stride = w + 37
    for k in range(3):
        ok_a, T_a = a.stereo_callback(seq.left[k], seq.right[k])
        L = np.full((h, stride), 255, np.uint8); R = np.full((h, stride), 255, np.uint8)
        L[:, :w] = seq.left[k]; R[:, :w] = seq.right[k]
        T = np.zeros(16)
        rc = api.check(api.lib.svo_process(ctx, api.ptr(L), api.ptr(R), stride, api.ptr(T), None))
        assert bool(rc) == ok_a and np.array_equal(T.reshape(4, 4), T_a)
    api.lib.svo_destroy(ctx)


def test_argument_errors_are_reported_not_crashes(api):
    cfg = api.default_config()
    ctx = C.c_void_p()
    assert api.lib.svo_create(C.byref(cfg), 0, 0, 320, 160, C.byref(ctx)) == api._lib.SVO_ERR_ARG          # n_seq < 1
    assert api.lib.svo_create(C.byref(cfg), 99, 1, 320, 160, C.byref(ctx)) == api._lib.SVO_ERR_ARG         # no such device
    for bad in (api.default_config(win_w=33, win_h=33), api.default_config(win_w=4, win_h=4), api.default_config(win_w=12, win_h=10),
                api.default_config(win_w=25, win_h=25, channels=3)):
        assert api.lib.svo_create(C.byref(bad), 0, 1, 320, 160, C.byref(ctx)) == api._lib.SVO_ERR_ARG      # unsupported window
        assert b"window" in api.lib.svo_last_error()
    for bad in (api.default_config(features_per_bucket=0), api.default_config(features_per_bucket=65)):
        assert api.lib.svo_create(C.byref(bad), 0, 1, 320, 160, C.byref(ctx)) == api._lib.SVO_ERR_ARG
    api.check(api.lib.svo_create(C.byref(cfg), 0, 1, 320, 160, C.byref(ctx)))
    img = np.zeros((160, 320), np.uint8); T = np.zeros(16)
    assert api.lib.svo_process(ctx, api.ptr(img), api.ptr(img), 320, api.ptr(T), None) == api._lib.SVO_ERR_STATE   # projection not set
    P = np.zeros(12, np.float32)
    api.check(api.lib.svo_set_projection(ctx, -1, api.ptr(P), api.ptr(P)))
    assert api.lib.svo_process(ctx, api.ptr(img), api.ptr(img), 100, api.ptr(T), None) == api._lib.SVO_ERR_ARG     # stride < width
    assert api.lib.svo_set_projection(ctx, 5, api.ptr(P), api.ptr(P)) == api._lib.SVO_ERR_ARG                      # seq out of range
    assert api.lib.svo_collect(ctx, None, None, None) == api._lib.SVO_ERR_STATE                                    # nothing in flight
    api.lib.svo_destroy(ctx)
    api.lib.svo_destroy(None)                                                                                      # harmless


def test_ring_of_frames_in_flight_is_bounded(api):
    import torch
    from stereo_visual_odometry_amd import synthetic as syn
    seq, cal = small_seq(n=3)
    vo = api.BatchVisualOdometry(320, 160, 2, api.default_config(max_translation_norm=2.0))
    vo.initalize_projection_matricies(*syn.projection_matrices(cal))
    L = torch.from_numpy(np.stack(seq.left)).cuda(); R = torch.from_numpy(np.stack(seq.right)).cuda()
    fb = 320 * 160
    def ptrs(k):
        k = k % 3
        return [L.data_ptr() + k * fb] * 2, [R.data_ptr() + k * fb] * 2
    torch.cuda.synchronize()
    for k in range(8):
        vo.submit_device(*ptrs(k), 320)
    with pytest.raises(api._lib.SvoError):
        vo.submit_device(*ptrs(8), 320)                       # 9th frame in flight is refused, state intact
    outs = [vo.collect() for _ in range(8)]
    assert not outs[0][0].any() and outs[1][0].all()          # frame 0 primes, frame 1 yields poses, both sequences identical
    assert np.array_equal(outs[1][1][0], outs[1][1][1])
    with pytest.raises(api._lib.SvoError):
        vo.collect()


def test_create_destroy_cycles_do_not_leak(api):
    import torch
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        vo = api.BatchVisualOdometry(640, 360, 4)
        vo.close()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20                           # no growth beyond allocator noise


def test_tiny_and_untextured_images(api):
    """32x32 frames (single pyramid level for a 21-px window is refused; 10-px window gives 2 levels) and flat frames."""
    from stereo_visual_odometry_amd import synthetic as syn
    Pl, Pr = syn.projection_matrices(dict(syn.RUN1))
    for img in (scenes.random_texture(32, 32, 2, smooth=0), np.full((32, 32), 90, np.uint8)):
        g = api.VisualOdometry(cfg=api.default_config()); g.initalize_projection_matricies(Pl, Pr)
        o = orc.VisualOdometry(orc.default_config()); o.initalize_projection_matricies(Pl, Pr)
        for k in range(3):
            ok_g, T_g = g.stereo_callback(img, img)
            ok_o, T_o = o.stereo_callback(img, img)
            assert ok_g == ok_o and g.stats.as_dict() == {f[0]: getattr(o.stats, f[0]) for f in o.stats._fields_}
            assert np.abs(T_g - T_o).max() < 1e-9
    with pytest.raises(api._lib.SvoError):
        api.BatchVisualOdometry(20, 20, 1, api.default_config(win_w=21, win_h=21))       # image not larger than the window


def test_identical_frames_give_identity_motion(api):
    from stereo_visual_odometry_amd import synthetic as syn
    seq, cal = small_seq(n=1)
    g = api.VisualOdometry(cfg=api.default_config()); g.initalize_projection_matricies(*syn.projection_matrices(cal))
    g.stereo_callback(seq.left[0], seq.right[0])
    ok, T = g.stereo_callback(seq.left[0], seq.right[0])
    assert ok and np.abs(T - np.eye(4)).max() < 1e-3


def test_timing_and_stream_accessors(api, monkeypatch):
    from stereo_visual_odometry_amd import synthetic as syn
    monkeypatch.setenv("SVO_GRAPH", "0")                             # stage events exist in launch-list mode
    seq, cal = small_seq(n=2)
    vo = api.BatchVisualOdometry(320, 160, 1, api.default_config(max_translation_norm=2.0))
    vo.initalize_projection_matricies(*syn.projection_matrices(cal))
    vo.stereo_callback_batch([seq.left[0]], [seq.right[0]])
    with pytest.raises(api._lib.SvoError):                            # stage events are recorded on request only
        vo.stage_timing()
    fr0 = C.c_float(0)
    assert api.lib.svo_get_last_timing(vo._h, None, C.byref(fr0)) == 0 and fr0.value > 0
    vo.set_stage_timing(True)
    vo.stereo_callback_batch([seq.left[1]], [seq.right[1]])
    lk, fr = vo.last_timing()
    assert 0 < lk < fr < 1000
    assert vo.stream()
    st = vo.stage_timing()                                            # svo_get_stage_timing: the five stages tile the frame
    assert list(st) == list(vo.STAGES) and all(v > 0 for v in st.values())
    assert abs(st["lk"] - lk) < 1e-3 and abs(sum(st.values()) - fr) < 0.05 * fr


def test_graph_replay_equals_the_launch_list(api, monkeypatch):
    """Small contexts replay each frame as a captured hipGraph (one per results-ring slot).  Same kernels, same order: poses,
    counters and feature sets are identical to the launch list, frame after frame (more frames than ring slots, so graphs are
    re-launched, and a host-image stride change forces a re-capture); only the stage timers are unavailable."""
    from stereo_visual_odometry_amd import synthetic as syn
    seq, cal = small_seq(n=12, seed=8)
    P = syn.projection_matrices(cal)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SVO_GRAPH", mode)
        vo = api.VisualOdometry(cfg=api.default_config(max_translation_norm=2.0)); vo.initalize_projection_matricies(*P)
        vo.set_stage_timing(True)
        res = []
        for k in range(12):
            L, R = seq.left[k], seq.right[k]
            if k >= 9:                                                # a wider buffer: another row stride
                Lw = np.zeros((L.shape[0], L.shape[1] + 16), np.uint8); Lw[:, :L.shape[1]] = L
                Rw = np.zeros_like(Lw); Rw[:, :R.shape[1]] = R
                T = np.zeros(16); st = api.SvoFrameStats()
                rc = api.check(api.lib.svo_process(vo._h, api.ptr(Lw), api.ptr(Rw), Lw.shape[1], api.ptr(T), C.byref(st)))
                ok, T, stats = bool(rc), T.reshape(4, 4), st.as_dict()
            else:
                ok, T = vo.stereo_callback(L, R); stats = vo.stats.as_dict()
            res.append((ok, T.copy(), stats, vo.features()[0].copy()))
        if mode == "1":
            with pytest.raises(api._lib.SvoError):
                vo.stage_timing()
            lk = C.c_float(0); fr = C.c_float(0)
            assert api.lib.svo_get_last_timing(vo._h, None, C.byref(fr)) == 0 and fr.value > 0      # the whole-frame time is always there
        else:
            assert all(v > 0 for v in vo.stage_timing().values())
        outs[mode] = res
    for a, b in zip(outs["1"], outs["0"]):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3])
    assert sum(r[0] for r in outs["1"]) >= 9


def test_frames_in_page_locked_memory_take_the_copy_free_path_and_give_the_same_results(api):
    """svo_alloc_pinned: images in page-locked memory with packed rows are DMA'd in place; pageable ones, padded rows, and a batch
    that mixes both go through the staging copy — all must give identical poses and counters."""
    from stereo_visual_odometry_amd import synthetic as syn
    cal = dict(syn.KITTI00, width=320, height=160, cx=160.0, cy=80.0)
    seq = syn.StereoSequence(cal=cal, n_frames=5, seed=3, step=0.3)
    P = syn.projection_matrices(cal)
    over = dict(max_translation_norm=2.0)
    ref = api.BatchVisualOdometry(320, 160, 2, api.default_config(**over)); ref.initalize_projection_matricies(*P)
    pin = api.BatchVisualOdometry(320, 160, 2, api.default_config(**over)); pin.initalize_projection_matricies(*P)
    mix = api.BatchVisualOdometry(320, 160, 2, api.default_config(**over)); mix.initalize_projection_matricies(*P)
    bufs = [api.PinnedImage((160, 320)) for _ in range(4)]
    for k in range(4):
        L = [seq.left[k], seq.left[k + 1]]; R = [seq.right[k], seq.right[k + 1]]
        ok0, T0 = ref.stereo_callback_batch(L, R)
        for b, im in zip(bufs, L + R):
            b.array[:] = im
        ok1, T1 = pin.stereo_callback_batch([bufs[0].array, bufs[1].array], [bufs[2].array, bufs[3].array])
        ok2, T2 = mix.stereo_callback_batch([bufs[0].array, L[1]], [R[0], bufs[3].array])
        for ok, T, vo in ((ok1, T1, pin), (ok2, T2, mix)):
            assert np.array_equal(ok, ok0) and np.array_equal(T, T0), k
            assert [s.as_dict() for s in vo.stats] == [s.as_dict() for s in ref.stats], k
    assert ok0.all()


def test_lk_kernel_register_budget(api):
    """The default LK build at the metric's 21x21 window is asked for six waves per SIMD (lk_min_waves: 80 registers, no scratch;
    6 x 80 = 480 of the 512 per SIMD lane): measured against five and seven.  Fewer than 96 registers then
    stay free, so the chaining of two contexts' LK launches (LkGate) must be off for it.  A change that costs the sixth wave
    silently costs the bench ~2 %; one that frees >= 96 registers would switch the chaining back on and change what the bench
    line's per-launch duration means."""
    from stereo_visual_odometry_amd._lib import lib
    lib.svo_get_lk_registers_left.restype = C.c_int
    lib.svo_get_lk_registers_left.argtypes = [C.c_void_p]
    vo = api.BatchVisualOdometry(1241, 376, 1, api.default_config(win_w=21, win_h=21))
    left = lib.svo_get_lk_registers_left(vo._h)
    assert left == 32, left                                       # 6 waves x 80
    vo.close()
    for win in (10, 31):
        vo = api.BatchVisualOdometry(1241, 376, 1, api.default_config(win_w=win, win_h=win))
        assert 0 <= lib.svo_get_lk_registers_left(vo._h) < 512
        vo.close()


def test_set_projection_per_sequence_and_all(api):
    """svo_set_projection(seq = -1) writes every sequence's record, seq = k only that one (one strided copy either way):
    a batch whose sequences get DIFFERENT baselines returns different translations for the same images."""
    from stereo_visual_odometry_amd import synthetic as syn
    seq, cal = small_seq(n=2)
    Pl, Pr = syn.projection_matrices(cal)
    Pr2 = Pr.copy(); Pr2[0, 3] *= 2.0                                  # twice the baseline: depths and the translation double
    vo = api.BatchVisualOdometry(320, 160, 3, api.default_config(max_translation_norm=5.0))
    vo.initalize_projection_matricies(Pl, Pr)                          # all three
    vo.initalize_projection_matricies(Pl, Pr2, seq=1)                  # only the middle one
    for k in range(2):
        ok, T = vo.stereo_callback_batch([seq.left[k]] * 3, [seq.right[k]] * 3)
    assert ok.all()
    assert np.array_equal(T[0], T[2]) and not np.array_equal(T[0], T[1])
    assert np.allclose(T[1][:3, 3], 2.0 * T[0][:3, 3], rtol=1e-3, atol=1e-4)


def test_run_sequences_tool_ragged_lengths(tmp_path):
    """BASELINE configs[3] in miniature: sequences of different lengths batched on one GPU, one CSV per sequence whose
    end point matches the renderer's ground truth."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "run_sequences.py"), "--lengths", "6,3,5", "--width", "480", "--height", "200",
                          "--out", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    for s, n in enumerate((6, 3, 5)):
        rows = np.loadtxt(tmp_path / ("result_seq%02d.csv" % s), delimiter=",", skiprows=1)
        assert rows.shape == (n, 4) and rows[0, 3] == 0 and rows[1:, 3].all()
        assert abs(rows[-1, 2] - 0.5 * (n - 1)) < 0.05          # 0.5 m per frame along +z
