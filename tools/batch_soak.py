"""One-off soak of the MANY-SEQUENCE path (not part of the test suite): a context of B sequences (> 8: the next frame's pyramids are
built ahead on the image stream, four pyramid slots, strided second pass, 256-thread compaction), frames submitted `depth` ahead,
against the CPU oracle frame by frame: flags, every counter and the pose; the feature set (bits) whenever no later frame is in
flight (every frame with depth 1, the last one otherwise).  The B slots replay `streams`
distinct frame streams: rendered sequences with different seeds, some of them with black frames spliced in (empty feature set ->
stale lastLeftPyramid -> recovery) at different times.   usage: python tools/batch_soak.py [frames] [B] [depth] [window]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import oracle_lib as orc
from stereo_visual_odometry_amd import api, synthetic as syn
import torch

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 3
WIN = int(sys.argv[4]) if len(sys.argv) > 4 else 21
orc.set_threads(16)
cal = dict(syn.KITTI00, width=480, height=200, cx=240.0, cy=100.0)
over = dict(win_w=WIN, win_h=WIN, max_translation_norm=2.0)
Pl, Pr = syn.projection_matrices(cal)
streams = []
for k, (seed, movers, blacks) in enumerate([(101, 0.0, ()), (102, 0.3, (7, 8)), (103, 0.3, (0, 1, 20)), (104, 0.0, (15, 16, 17, 30))]):
    sq = syn.StereoSequence(cal=cal, n_frames=NF, seed=seed, step=0.12, movers=movers)
    L, R = list(sq.left), list(sq.right)
    black = np.zeros_like(L[0])
    for b in blacks:
        if b < NF: L[b] = black; R[b] = black
    streams.append((L, R))
want = []
for L, R in streams:
    o = orc.VisualOdometry(orc.default_config(**over)); o.initalize_projection_matricies(Pl, Pr)
    per = []
    for k in range(NF):
        ok, T = o.stereo_callback(L[k], R[k])
        per.append((ok, T.copy(), {f[0]: getattr(o.stats, f[0]) for f in o.stats._fields_}, [a.copy() for a in o.features()]))
    want.append(per)
print('oracle done:', [sum(w[0] for w in per) for per in want], 'poses ok per stream;', 'second passes', [sum(w[2]['second_pass'] for w in per) for per in want], flush=True)
vo = api.BatchVisualOdometry(480, 200, B, api.default_config(**over)); vo.initalize_projection_matricies(Pl, Pr)
which = lambda i: i % len(streams)
dev = [[(torch.from_numpy(np.ascontiguousarray(L[k])).cuda(), torch.from_numpy(np.ascontiguousarray(R[k])).cuda()) for k in range(NF)] for L, R in streams]
torch.cuda.synchronize()
submit = lambda k: vo.submit_device([dev[which(i)][k][0].data_ptr() for i in range(B)], [dev[which(i)][k][1].data_ptr() for i in range(B)], 480)
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
sub = 0; bad = 0; maxdt = 0.0
for k in range(NF):
    while sub < NF and sub - k < DEPTH:
        submit(sub); sub += 1
    ok, T = vo.collect()
    for i in range(B):
        w = want[which(i)][k]; sg = vo.stats[i].as_dict()
        same = bool(ok[i]) == w[0] and sg == w[2]
        if same and i < 2 * len(streams) and sub == k + 1:           # the device's feature set is frame k's only when no later frame has been submitted
            f = vo.features(i)
            same = np.array_equal(bits(f[0]), bits(w[3][0])) and np.array_equal(f[1], w[3][1]) and np.array_equal(f[2], w[3][2])
        if not same:
            bad += 1; print('MISMATCH frame', k, 'slot', i, sg, w[2], flush=True)
        maxdt = max(maxdt, float(np.abs(T[i][:3, 3] - w[1][:3, 3]).max()))
print('frames', NF, 'slots', B, 'depth', DEPTH, 'window', WIN, ': mismatching (frame, slot) pairs', bad, 'max |dt|', maxdt)
