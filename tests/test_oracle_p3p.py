"""cv::solvePnPRansac's small-N branches in the oracle: 4 points -> one P3P solve (Gao et al.),
5 points -> one EPnP solve, every point an inlier, no RANSAC, no refine.  The reference holds no test or fixture for these
branches (cameraToWorld behind stereo_callback always sees > 15 points, vo.cpp:82): PARITY UNPINNED against OpenCV; pinned
here analytically — exact projections of a known pose are inverted — and, for the quartic of the P3P, symbolically."""
import numpy as np
import pytest

import oracle_lib as orc


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def scene(n, seed, K):
    rng = np.random.default_rng(seed)
    R = rodrigues(rng.normal(0, 0.2, 3)); t = rng.normal(0, 0.3, 3) + [0, 0, 1.0]
    X = np.concatenate([rng.uniform(-2, 2, (n, 2)), rng.uniform(4, 9, (n, 1))], 1).astype(np.float32)
    Xc = X.astype(np.float64) @ R.T + t
    uv = (Xc[:, :2] / Xc[:, 2:]) * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
    return R, t, X, uv.astype(np.float32)


K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], np.float32)


@pytest.mark.parametrize("seed", range(8))
def test_four_points_p3p_recovers_the_pose(seed):
    R, t, X, uv = scene(4, seed, K)
    ok, R2, t2, inl, dbg = orc.camera_to_world(K, uv, X, np.eye(3), np.zeros(3))
    assert ok and inl.tolist() == [0, 1, 2, 3] and dbg[0] == 0            # no RANSAC iteration ran
    # the image points are float32 (1e-5 px): the pose comes back to ~1e-5
    assert np.abs(R2 - R).max() < 2e-4 and np.abs(t2 - t).max() < 2e-3
    Xc = X.astype(np.float64) @ R2.T + t2
    uv2 = (Xc[:, :2] / Xc[:, 2:]) * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
    assert np.abs(uv2 - uv).max() < 1e-2                                  # all four points reproject (the 4th picked the branch)


@pytest.mark.parametrize("seed", range(4))
def test_five_points_direct_epnp_no_ransac_no_refine(seed):
    R, t, X, uv = scene(5, 100 + seed, K)
    ok, R2, t2, inl, dbg = orc.camera_to_world(K, uv, X, np.eye(3), np.zeros(3))
    assert ok and inl.tolist() == [0, 1, 2, 3, 4] and dbg[0] == 0
    assert np.abs(R2 - R).max() < 5e-3 and np.abs(t2 - t).max() < 5e-2     # EPnP on 5 points without the LM polish


def test_fewer_than_four_points_fail():
    R, t, X, uv = scene(3, 1, K)
    ok, R2, t2, inl, _ = orc.camera_to_world(K, uv, X, np.eye(3), np.zeros(3))
    assert not ok and len(inl) == 0 and np.array_equal(R2, np.eye(3))


def test_p3p_quartic_is_the_resultant_of_the_cosine_laws():
    """The coefficients A..E restated in orc_p3p.c (Gao et al. 2003, eq. for x = |PA| / |PC|) are exactly the resultant in y of
    (1-a) y^2 + (a r x - p) y + 1 - a x^2   and   b y^2 - b r x y + (b-1) x^2 + q x - 1   (the law of cosines on the three
    apex angles, normalised by |AB|^2)."""
    sp = pytest.importorskip("sympy")
    x, y, a, b, p, q, r = sp.symbols("x y a b p q r")
    res = sp.Poly(sp.expand(sp.resultant((1 - a) * y**2 + (a * r * x - p) * y + (1 - a * x**2),
                                         b * y**2 - b * r * x * y + ((b - 1) * x**2 + q * x - 1), y)), x).all_coeffs()
    a2, b2, p2, q2, r2 = a * a, b * b, p * p, q * q, r * r
    pr, ab, a_2, a_4 = p * r, a * b, 2 * a, 4 * a
    pqr = q * pr
    mine = [-2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2,
            q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab),
            q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2,
            pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2),
            1 + 2 * (b - a - ab) + b2 - b * p2 + a2]
    assert all(sp.expand(c - m) == 0 for c, m in zip(res, mine))
