"""Which of the oracle's documented deviations from OpenCV (oracle/orc.h D1..D5) explains which disagreement with the
reference's OWN recording (run1/result.csv)?  Measured, not assumed.

The oracle carries run-time switches (orc_set_opencv_mode, CPU only) that revert one deviation each to what OpenCV 4.5
does.  This script replays the run1 frames the way the reference's CLI fed them (BGR, identity start)
once per switch setting and prints, per frame, the distance between the oracle's pose INCREMENT and the recorded one, plus
the per-frame counters that show whether a track or an inlier decision differs from the baseline oracle.

    python tools/deviation_ablation.py [--frames 128] [--out tests/golden/deviation_ablation.txt]

Reads /root/reference/run1 when it exists (build container: all 128 frames), else the committed 48-frame fixture.
TEST INFRASTRUCTURE: uses oracle/ only; nothing here touches the product library.
"""
import argparse
import lzma
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc                                                    # noqa: E402
from stereo_visual_odometry_amd import synthetic as syn                     # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF_RUN1 = "/root/reference/run1"

D1, D2, D3R, D4, D5, HYPOT = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20          # D3R: the FORMER deviation D3 (right singular vectors) back in force
D1_FMA, D1_W4, D1_SCALAR = 0x100, 0x200, 0x400
J12_HALVES, J12_QUARTERS, TRI_RR = 0x1000, 0x2000, 0x4000                  # experiments: candidate latency cuts of the HIP kernels

SETTINGS = [
    ("the oracle as shipped (D1, D2, D4, D5 in force)", 0),
    ("former D3 back in force (EPnP basis = right singular vectors, rounds 1-2)", D3R),
    ("D1 reverted: LK float sums, OpenCV 4.x SIMD128 order", D1),
    ("D1 variant: same, v_muladd fused (FMA3 baseline)", D1 | D1_FMA),
    ("D1 variant: OpenCV 3.x SSE2 order (4-wide A loop)", D1 | D1_W4),
    ("D1 variant: float sums without SIMD (scalar order)", D1 | D1_SCALAR),
    ("D1 reverted + former D3 back in force (= rounds 1-2 with D1 reverted)", D1 | D3R),
    ("D2 reverted: LM starts from the last hypothesis", D2),
    ("D4 reverted: hypotheses scored through R->rvec->R", D4),
    ("D5 reverted: cyclic-by-rows 12x12 Jacobi", D5),
    ("D1+D2 reverted", D1 | D2),
    ("D1+D4 reverted", D1 | D4),
    ("D1+D5 reverted", D1 | D5),
    ("D1 reverted + libm hypot in every Jacobi rotation", D1 | HYPOT),
    ("D1+D2+D4+D5 reverted + hypot", D1 | D2 | D4 | D5 | HYPOT),
    ("candidate cut: D1 reverted + 12-term sums of EPnP's Jacobi as two halves", D1 | J12_HALVES),
    ("candidate cut: D1 reverted + 12-term sums of EPnP's Jacobi as four quarters", D1 | J12_QUARTERS),
    ("candidate cut: D1 reverted + round-robin 4x4 sweep in the triangulation", D1 | TRI_RR),
    ("candidate cut: quarters, D1 in force (the default mode)", J12_QUARTERS),
]


def load_frames(n):
    if os.path.isdir(REF_RUN1):
        from PIL import Image
        bgr = lambda p: np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[..., ::-1])
        n = min(n, 128)
        return ([bgr("%s/left/frame%06d.png" % (REF_RUN1, i)) for i in range(n)],
                [bgr("%s/right/frame%06d.png" % (REF_RUN1, i)) for i in range(n)], "reference run1/ (%d frames)" % n)
    out = []
    for cam in ("left", "right"):
        with lzma.open(os.path.join(GOLD, "run1_bgr_%s_0_47.npy.xz" % cam), "rb") as f:
            out.append(np.load(f, allow_pickle=False))
    n = min(n, len(out[0]))
    return list(out[0][:n]), list(out[1][:n]), "committed fixture (%d frames)" % n


def replay(mode, left, right):
    prev = orc.lib().orc_set_opencv_mode(mode)
    try:
        vo = orc.VisualOdometry(orc.default_config())
        vo.initalize_projection_matricies(*syn.projection_matrices(syn.RUN1))
        pose, track, rows = np.eye(4), [], []
        for l, r in zip(left, right):
            ok, T = vo.stereo_callback(l, r)
            pose = pose @ T
            track.append(pose[:3, 3].copy())
            s = vo.stats
            rows.append((int(ok), s.n_into_lk, s.n_after_circular, s.n_after_bounds, s.n_inliers, s.ransac_iters))
        return np.array(track), np.array(rows)
    finally:
        orc.lib().orc_set_opencv_mode(prev)


def increments(track):
    return np.diff(np.vstack([np.zeros(3), track]), axis=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=4)
    a = ap.parse_args()
    left, right, src = load_frames(a.frames)
    ref = np.load(os.path.join(GOLD, "run1_recorded.npz"))["result_csv"][:len(left), :3]
    orc.set_threads(a.threads)
    lines = ["deviation ablation against the reference's recording run1/result.csv — input: %s" % src,
             "inc = |oracle pose increment - recorded pose increment| per frame (m); a frame 'agrees' when inc < 2e-6 m (the file prints 6 digits)",
             ""]
    base_rows = None
    watch = [14, 15, 22, 23, 25, 26, 27]
    table = []
    for name, mode in SETTINGS:
        track, rows = replay(mode, left, right)
        n = len(track)
        inc = np.linalg.norm(increments(track) - increments(ref[:n]), axis=1)
        err = np.linalg.norm(track - ref[:n], axis=1)
        if base_rows is None:
            base_rows = rows
        bad = [k for k in range(n) if inc[k] >= 2e-6]
        diff_tracks = [k for k in range(n) if rows[k][2] != base_rows[k][2] or rows[k][3] != base_rows[k][3]]
        diff_inl = [k for k in range(n) if rows[k][4] != base_rows[k][4] and k not in diff_tracks]
        lines.append("== %s (mode 0x%03x)" % (name, mode))
        lines.append("   poses ok %d/%d   frames that disagree with the recording (inc >= 2e-6 m): %d  first: %s" %
                     (rows[:, 0].sum(), n - 1, len(bad), bad[:12]))
        lines.append("   inc at frames %s: %s" % (watch, " ".join("%.2e" % inc[k] for k in watch if k < n)))
        lines.append("   cumulative error: frame 22 %.2e  frame 47 %.2e  last %.2e m   rms %.2e m" %
                     (err[min(22, n - 1)], err[min(47, n - 1)], err[-1], np.sqrt((err ** 2).mean())))
        printed = np.vectorize(lambda v: float("%.6g" % v))(track)                 # the reference's ofstream prints 6 significant digits
        same = printed == ref[:n]
        lines.append("   rows of result.csv reproduced digit for digit: %d/%d (values %d/%d); first row that differs: %s" %
                     (same.all(axis=1).sum(), n, same.sum(), same.size, [int(v) for v in np.where(~same.all(axis=1))[0][:6]]))
        lines.append("   vs baseline oracle: track counts differ at %s; inlier counts alone differ at %s" %
                     (diff_tracks[:12], diff_inl[:12]))
        lines.append("")
        table.append((name, mode, inc, err, bad))
        print("\n".join(lines[-7:]), flush=True)
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    orc.set_threads(1)
    return table


if __name__ == "__main__":
    main()
