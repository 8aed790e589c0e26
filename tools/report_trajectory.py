#!/usr/bin/env python3
"""Counterpart of the reference's visualize_data.py: reads a result.csv written by `vo N folder` /
tools/svo_cli (columns x,y,z,gtx,gty; main.cpp:346-348, 397-400), prints the end-point-error figures that script prints
and, with --plot, draws estimated vs ground-truth track with the same axis remap (x,y,z) -> (-z, x, y).

usage: python tools/report_trajectory.py result.csv [--ref other_result.csv] [--plot out.png]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_visual_odometry_amd import evaluate                      # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--ref", help="a second result.csv (e.g. the reference's own run) to report the position RMSE against")
    ap.add_argument("--plot", help="write a PNG of the top-down tracks")
    a = ap.parse_args(argv)
    rows = evaluate.read_result_csv(a.csv)
    e = evaluate.endpoint_error(rows)
    print("frames            %d" % len(rows))
    print("abs end error     %.6f m" % e["abs_error"])
    print("goal distance     %.6f m" % e["goal_distance"])
    print("relative error    %s" % ("%.4f" % e["rel_error"] if e["rel_error"] != float("inf") else "inf (goal at the origin)"))
    print("distance covered  %.6f m" % e["total_d"])
    if a.ref:
        ref = evaluate.read_result_csv(a.ref)
        print("position RMSE vs %s  %.6f m" % (a.ref, evaluate.position_rmse(rows, ref)))
    if a.plot:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        xs, ys = -rows[:, 2], rows[:, 0]                              # visualize_data.py:28
        gx, gy = rows[:, 3] - rows[0, 3], rows[:, 4] - rows[0, 4]
        fig, ax = plt.subplots(figsize=(5, 5))
        ax.plot(xs, ys, label="estimated"); ax.plot(gx, gy, label="ground truth")
        ax.set_xlabel("x [m]"); ax.set_ylabel("y [m]"); ax.axis("equal"); ax.legend()
        fig.savefig(a.plot, dpi=100)
        print("wrote", a.plot)
    return e


if __name__ == "__main__":
    main()
