"""Trajectory evaluation: the end-point-error figures the reference's visualize_data.py
prints (formula restated from visualize_data.py:9-46, plotting left out) and an un-aligned position RMSE (ATE)."""
import numpy as np


def read_result_csv(path):
    """result.csv as written by the reference CLI (main.cpp:346-348, 397-400): header x,y,z,gtx,gty."""
    return np.loadtxt(path, delimiter=",", skiprows=1).reshape(-1, 5)


def endpoint_error(rows):
    """rows: (n,5) x,y,z,gtx,gty.  Returns dict(abs_error, goal_distance, rel_error, total_d) with the reference's
    conventions: ground truth re-based on its first row, robot axes remapped (x,y,z) -> (-z, x, y)."""
    rows = np.asarray(rows, np.float64)
    x0, y0 = rows[0, 3], rows[0, 4]
    gtx = rows[:, 3] - x0
    gty = rows[:, 4] - y0
    gtx[0] = gty[0] = 0.0
    xs, ys, zs = -rows[:, 2], rows[:, 0], rows[:, 1]
    dist = float(np.linalg.norm([xs[-1] - gtx[-1], ys[-1] - gty[-1], zs[-1]]))
    goal = float(np.hypot(gtx[-1], gty[-1]))
    return dict(abs_error=dist, goal_distance=goal, rel_error=(abs(dist) / goal if goal > 0 else float("inf")),
                total_d=float(np.sqrt(abs(xs[-1] ** 2 + ys[-1] ** 2 + zs[-1] ** 2))))


def position_rmse(a, b):
    """un-aligned RMSE between two (n,3) position tracks (ATE as used in SURVEY.md §8d)."""
    a, b = np.asarray(a, np.float64)[:, :3], np.asarray(b, np.float64)[:, :3]
    n = min(len(a), len(b))
    return float(np.sqrt(((a[:n] - b[:n]) ** 2).sum(1).mean()))
