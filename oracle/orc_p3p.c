/* orc_p3p.c — ORACLE (test infrastructure only): the P3P kernel cv::solvePnPRansac falls back to when it is given exactly
 * four points (calib3d solvepnp.cpp: `else if (npoints == 4) { model_points = 4; ransac_kernel_method = SOLVEPNP_P3P; }`,
 * and `if (model_points == npoints)` -> one direct solvePnP, every point an inlier, no RANSAC, no refine).
 * Reached through the reference's public cameraToWorld (include/vo.h:452-456, src/vo.cpp:301-304) only; stereo_callback never
 * gets there (vo.cpp:82 wants more than 15 tracks).
 *
 * Restated from the published algorithm OpenCV 4.5 implements in modules/calib3d/src/p3p.cpp and polynom_solver.cpp (neither
 * is on disk): X.S. Gao, X.-R. Hou, J. Tang, H.-F. Chang, "Complete Solution Classification for the Perspective-Three-Point
 * Problem", PAMI 25(8) 2003 — quartic in x = |PA| / |PC| (its coefficients are the resultant of the two law-of-cosines
 * quadratics; checked symbolically, tests/test_oracle_p3p.py), the closed-form y = |PB| / |PC|, Horn's quaternion alignment of
 * the three camera-frame points, the fourth point picks the solution with the smallest reprojection error.
 * PARITY UNPINNED: the reference holds no test or fixture for this branch; pinned analytically only (a known pose is recovered
 * from exact projections, tests/test_oracle_p3p.py). */
#include <math.h>
#include <string.h>
#include "orc.h"

/* ---- polynom_solver.cpp: real roots of degree 2 / 3 / 4 (MathWorld's closed forms) ---- */
static int solve_deg2(double a, double b, double c, double* x1, double* x2) {
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) { *x1 = -b * inv_2a; *x2 = *x1; return 1; }
    double sqrt_delta = sqrt(delta);
    *x1 = (-b + sqrt_delta) * inv_2a;
    *x2 = (-b - sqrt_delta) * inv_2a;
    return 2;
}

static int solve_deg3(double a, double b, double c, double d, double* x0, double* x1, double* x2) {
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return solve_deg2(b, c, d, x0, x1);
    }
    double inv_a = 1. / a;
    double b_a = inv_a * b, b_a2 = b_a * b_a;
    double c_a = inv_a * c;
    double d_a = inv_a * d;
    double Q = (3 * c_a - b_a2) / 9;
    double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) { *x0 = *x1 = *x2 = -b_a_3; return 3; }
        *x0 = pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {                                        /* three real roots */
        double theta = acos(R / sqrt(-Q3));
        double sqrt_Q = sqrt(-Q);
        *x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
        *x1 = 2 * sqrt_Q * cos((theta + 2 * 3.14159265358979323846) / 3.0) - b_a_3;
        *x2 = 2 * sqrt_Q * cos((theta + 4 * 3.14159265358979323846) / 3.0) - b_a_3;
        return 3;
    }
    double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    *x0 = AD + BD - b_a_3;
    return 1;
}

static int solve_deg4(double a, double b, double c, double d, double e, double x[4]) {
    if (a == 0) { x[3] = 0; return solve_deg3(b, c, d, e, &x[0], &x[1], &x[2]); }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
    if (n == 0) return 0;
    double R2 = 0.25 * b2 - c + r0, R;
    if (R2 < 0) return 0;
    R = sqrt(R2);
    double inv_R = 1. / R;
    int nb_real_roots = 0;
    double D2, E2;
    if (R < 10E-12) {
        double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            double sqrt_temp = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
            E2 = D2 - 4 * sqrt_temp;
        }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        double D = sqrt(D2);
        nb_real_roots = 2;
        double D_2 = 0.5 * D;
        x[0] = R_2 + D_2 - b_4;
        x[1] = x[0] - D;
    }
    if (E2 >= 0) {
        double E = sqrt(E2);
        double E_2 = 0.5 * E;
        if (nb_real_roots == 0) {
            x[0] = -R_2 + E_2 - b_4;
            x[1] = x[0] - E;
            nb_real_roots = 2;
        } else {
            x[2] = -R_2 + E_2 - b_4;
            x[3] = x[2] - E;
            nb_real_roots = 4;
        }
    }
    return nb_real_roots;
}

/* p3p.cpp solve_for_lengths: distances = |BC|, |AC|, |AB|; cosines of the angles BPC, APC, APB -> |PA|, |PB|, |PC| per solution */
static int solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3]) {
    double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]);
    double b = inv_d22 * (distances[1] * distances[1]);
    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;                /* the four points (P, A, B, C) are coplanar */
    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double roots[4];
    int n = solve_deg4(A, B, C, D, E, roots), nb = 0, i;
    if (n == 0) return 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    double inv_b0 = 1. / b0;
    for (i = 0; i < n; i++) {
        double x = roots[i];
        if (x <= 0) continue;
        double x2 = x * x;
        double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        double y = inv_b0 * b1;
        double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        double Z = distances[2] / sqrt(v);
        lengths[nb][0] = x * Z; lengths[nb][1] = y * Z; lengths[nb][2] = Z;
        nb++;
    }
    return nb;
}

/* p3p.cpp jacobi_4x4: eigen decomposition of a symmetric 4x4 by cyclic Jacobi rotations (A row-major, upper triangle used) */
static int jacobi_4x4(double* A, double* D, double* U) {
    double B[4], Z[4] = {0, 0, 0, 0};
    int iter, i, j, k;
    for (i = 0; i < 16; i++) U[i] = (i % 5 == 0);
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    memcpy(D, B, sizeof(B));
    for (iter = 0; iter < 50; iter++) {
        double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return 1;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (i = 0; i < 3; i++) {
            double* pAij = A + 5 * i + 1;
            for (j = i + 1; j < 4; j++) {
                double Aij = *pAij;
                double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j]))
                    *pAij = 0.0;
                else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh)) t = Aij / hh;
                    else {
                        double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh; D[i] -= hh; D[j] += hh;
                    *pAij = 0.0;
                    double c = 1.0 / sqrt(1 + t * t);
                    double s = t * c;
                    double tau = s / (1.0 + c);
                    for (k = 0; k <= i - 1; k++) { double g = A[k * 4 + i], h = A[k * 4 + j]; A[k * 4 + i] = g - s * (h + g * tau); A[k * 4 + j] = h + s * (g - h * tau); }
                    for (k = i + 1; k <= j - 1; k++) { double g = A[i * 4 + k], h = A[k * 4 + j]; A[i * 4 + k] = g - s * (h + g * tau); A[k * 4 + j] = h + s * (g - h * tau); }
                    for (k = j + 1; k < 4; k++) { double g = A[i * 4 + k], h = A[j * 4 + k]; A[i * 4 + k] = g - s * (h + g * tau); A[j * 4 + k] = h + s * (g - h * tau); }
                    for (k = 0; k < 4; k++) { double g = U[k * 4 + i], h = U[k * 4 + j]; U[k * 4 + i] = g - s * (h + g * tau); U[k * 4 + j] = h + s * (g - h * tau); }
                }
                pAij++;
            }
        }
        for (i = 0; i < 4; i++) B[i] += Z[i];
        memcpy(D, B, sizeof(B));
        memset(Z, 0, sizeof(Z));
    }
    return 0;
}

/* p3p.cpp align: rigid motion taking the three world points onto the three camera-frame points (Horn, unit quaternion) */
static int p3p_align(const double M_end[3][3], const double Xw[3][3], double R[3][3], double T[3]) {
    double C_start[3], C_end[3], s[9], Qs[16], evs[4], U[16], q[4];
    int i, j;
    for (i = 0; i < 3; i++) {
        C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
        C_start[i] = (Xw[0][i] + Xw[1][i] + Xw[2][i]) / 3;
    }
    for (j = 0; j < 3; j++)
        for (i = 0; i < 3; i++)
            s[i * 3 + j] = (Xw[0][i] * M_end[0][j] + Xw[1][i] * M_end[1][j] + Xw[2][i] * M_end[2][j]) / 3 - C_end[j] * C_start[i];
    memset(Qs, 0, sizeof(Qs));
    Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
    Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
    Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
    Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    jacobi_4x4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[0];
    for (i = 1; i < 4; i++) if (evs[i] > ev_max) { ev_max = evs[i]; i_ev = i; }
    for (i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32; R[0][1] = 2. * (q1_2 - q0_3); R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3); R[1][1] = q02 + q22 - q12 - q32; R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2); R[2][1] = 2. * (q2_3 + q0_1); R[2][2] = q02 + q32 - q12 - q22;
    for (i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
    return 1;
}

/* p3p::solve with four points: the solutions of the first three, ordered by the reprojection error of the fourth; the best one.
 * obj: 4 x 3 world points; img: 4 x 2 pixel coordinates (already "undistorted"); returns 1 and R (row-major), t, or 0. */
int orc_p3p(const double obj[12], const double img[8], double fx, double fy, double cx, double cy, double R[9], double t[3]) {
    const double inv_fx = 1. / fx, inv_fy = 1. / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double mu[4], mv[4], mk[3], distances[3], cosines[3], lengths[4][3], best_err = 0;
    double Xw[3][3];
    int i, k, n, nb = 0;
    for (i = 0; i < 4; i++) { mu[i] = inv_fx * img[2 * i] - cx_fx; mv[i] = inv_fy * img[2 * i + 1] - cy_fy; }
    for (i = 0; i < 3; i++) {
        double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm; mu[i] *= mk[i]; mv[i] *= mk[i];
        for (k = 0; k < 3; k++) Xw[i][k] = obj[3 * i + k];
    }
#define D3(a, b) sqrt((obj[3 * a] - obj[3 * b]) * (obj[3 * a] - obj[3 * b]) + (obj[3 * a + 1] - obj[3 * b + 1]) * (obj[3 * a + 1] - obj[3 * b + 1]) + \
                      (obj[3 * a + 2] - obj[3 * b + 2]) * (obj[3 * a + 2] - obj[3 * b + 2]))
    distances[0] = D3(1, 2); distances[1] = D3(0, 2); distances[2] = D3(0, 1);
#undef D3
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
    n = solve_for_lengths(lengths, distances, cosines);
    for (i = 0; i < n; i++) {
        double M_orig[3][3], Rs[3][3], ts[3];
        for (k = 0; k < 3; k++) { M_orig[k][0] = lengths[i][k] * mu[k]; M_orig[k][1] = lengths[i][k] * mv[k]; M_orig[k][2] = lengths[i][k] * mk[k]; }
        if (!p3p_align(M_orig, Xw, Rs, ts)) continue;
        double X3p = Rs[0][0] * obj[9] + Rs[0][1] * obj[10] + Rs[0][2] * obj[11] + ts[0];
        double Y3p = Rs[1][0] * obj[9] + Rs[1][1] * obj[10] + Rs[1][2] * obj[11] + ts[1];
        double Z3p = Rs[2][0] * obj[9] + Rs[2][1] * obj[10] + Rs[2][2] * obj[11] + ts[2];
        double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        double err = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
        if (nb == 0 || err < best_err) {                       /* the insertion sort of p3p.cpp keeps the first of equal errors in front */
            best_err = err;
            for (k = 0; k < 3; k++) { R[3 * k] = Rs[k][0]; R[3 * k + 1] = Rs[k][1]; R[3 * k + 2] = Rs[k][2]; t[k] = ts[k]; }
        }
        nb++;
    }
    return nb > 0;
}
