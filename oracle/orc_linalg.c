/* ORACLE (test infrastructure, see orc.h).  Small dense linear algebra in f64:
 * one-sided Jacobi SVD (the method OpenCV's cv::SVD uses for small matrices when built without
 * LAPACK, modules/core/src/lapack.cpp JacobiSVDImpl_), SVD back-substitution, least squares.
 * Only + - * / sqrt are used, in a fixed order, so the HIP kernels can reproduce the bits. */
#include "orc.h"
#include "orc_linalg.h"
#include <float.h>
#include <math.h>
#include <string.h>

/* sum of 12 terms in the order the EPnP kernel's lanes produce it (experiment switch ORC_ALT_J12_*: 0 sequential) */
static double sum12(const double* t, int mode) {
    int k; double s = 0;
    if (mode == 1) {            /* halves: (t0+..+t5) + (t6+..+t11), each from zero */
        double lo = 0, hi = 0;
        for (k = 0; k < 6; k++) lo += t[k];
        for (k = 6; k < 12; k++) hi += t[k];
        return lo + hi;
    }
    if (mode == 2) {            /* quarters: ((t0+t1+t2) + (t3+t4+t5)) + ((t6+t7+t8) + (t9+t10+t11)) */
        double q[4];
        for (k = 0; k < 4; k++) q[k] = (t[3 * k] + t[3 * k + 1]) + t[3 * k + 2];
        return (q[0] + q[1]) + (q[2] + q[3]);
    }
    for (k = 0; k < 12; k++) s += t[k];
    return s;
}

/* One Hestenes rotation of rows i < j of At (and of Vt).  Returns 1 if the pair was rotated. */
static int rotate_pair(double* At, int m, int n, double* W, double* Vt, int i, int j) {
    const double eps = DBL_EPSILON * 10;
    double* Ai = At + i * m; double* Aj = At + j * m;
    double a = W[i], p = 0, b = W[j], c, s;
    int k;
    const unsigned ocv = orc_get_opencv_mode();
    const int mode12 = (m == 12 && n == 12) ? ((ocv & ORC_ALT_J12_HALVES) ? 1 : (ocv & ORC_ALT_J12_QUARTERS) ? 2 : 0) : 0;
    if (mode12) {
        double t[12], u[12], v[12];
        for (k = 0; k < 12; k++) t[k] = Ai[k] * Aj[k];
        p = sum12(t, mode12);
        if (fabs(p) <= eps * sqrt(a * b)) return 0;
        p *= 2;
        {
            double beta = a - b, gamma = (ocv & ORC_OCV_HYPOT) ? hypot(p, beta) : sqrt(p * p + beta * beta);
            if (beta < 0) { double delta = (gamma - beta) * 0.5; s = sqrt(delta / gamma); c = p / (gamma * s * 2); }
            else { c = sqrt((gamma + beta) / (gamma * 2)); s = p / (gamma * c * 2); }
        }
        for (k = 0; k < 12; k++) {
            double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
            Ai[k] = t0; Aj[k] = t1; u[k] = t0 * t0; v[k] = t1 * t1;
        }
        W[i] = sum12(u, mode12); W[j] = sum12(v, mode12);
        if (Vt) {
            double* Vi = Vt + i * n; double* Vj = Vt + j * n;
            for (k = 0; k < n; k++) { double t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k]; Vi[k] = t0; Vj[k] = t1; }
        }
        return 1;
    }
    for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
    if (fabs(p) <= eps * sqrt(a * b)) return 0;
    p *= 2;
    /* JacobiSVDImpl_ calls hypot(p, beta) (libm); the oracle's default is the plain sqrt form the HIP kernels reproduce bit
       for bit — ORC_OCV_HYPOT (a deviation switch, orc.h) selects libm's */
    double beta = a - b, gamma = (orc_get_opencv_mode() & ORC_OCV_HYPOT) ? hypot(p, beta) : sqrt(p * p + beta * beta);
    if (beta < 0) {
        double delta = (gamma - beta) * 0.5;
        s = sqrt(delta / gamma);
        c = p / (gamma * s * 2);
    } else {
        c = sqrt((gamma + beta) / (gamma * 2));
        s = p / (gamma * c * 2);
    }
    a = b = 0;
    for (k = 0; k < m; k++) {
        double t0 = c * Ai[k] + s * Aj[k];
        double t1 = -s * Ai[k] + c * Aj[k];
        Ai[k] = t0; Aj[k] = t1;
        a += t0 * t0; b += t1 * t1;
    }
    W[i] = a; W[j] = b;
    if (Vt) {
        double* Vi = Vt + i * n; double* Vj = Vt + j * n;
        for (k = 0; k < n; k++) {
            double t0 = c * Vi[k] + s * Vj[k];
            double t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0; Vj[k] = t1;
        }
    }
    return 1;
}

/* At: n rows of length m (the transpose of the m x n input, m >= n). On exit rows of At are the
 * left singular vectors (for i < n1), W[n] descending, Vt n x n (rows = right singular vectors).
 * ordering 0: cyclic by rows (i, j>i), the order OpenCV's JacobiSVDImpl_ uses.
 * ordering 1 (n even): round-robin / Brent-Luk "circle" ordering: n-1 rounds per sweep, each round
 *   rotating n/2 DISJOINT row pairs — the pairs of a round touch disjoint data, so a parallel
 *   implementation (one lane per pair) produces bit-identical results to this sequential loop. */
void orc_jacobi_svd_ord(double* At, int m, int n, double* W, double* Vt, int n1, int ordering) {
    const double minval = DBL_MIN;
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) { double t = At[i * m + k]; sd += t * t; }
        W[i] = sd;
        if (Vt) { for (k = 0; k < n; k++) Vt[i * n + k] = 0; Vt[i * n + i] = 1; }
    }
    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        if (ordering == 0 || (n & 1)) {
            for (i = 0; i < n - 1; i++)
                for (j = i + 1; j < n; j++) changed |= rotate_pair(At, m, n, W, Vt, i, j);
        } else {
            int r, q;
            for (r = 0; r < n - 1; r++)
                for (q = 0; q < n / 2; q++) {
                    /* circle method: position 0 is fixed, positions 1..n-1 rotate by r */
                    int pa = q == 0 ? 0 : 1 + (q - 1 + r) % (n - 1);
                    int pb = 1 + (n - 1 - q - 1 + r) % (n - 1);
                    i = pa < pb ? pa : pb; j = pa < pb ? pb : pa;
                    changed |= rotate_pair(At, m, n, W, Vt, i, j);
                }
        }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) { double t = At[i * m + k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {           /* selection sort, descending */
        j = i;
        for (k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            if (Vt) {
                for (k = 0; k < m; k++) { t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
                for (k = 0; k < n; k++) { t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
            }
        }
    }
    if (!Vt) return;
    for (i = 0; i < n1; i++) {
        double sd = i < n ? W[i] : 0;
        /* OpenCV fills the left vector of an exactly-zero singular value with an orthogonalised random
           vector; the oracle leaves it zero (never consumed on this path). */
        double s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

void orc_jacobi_svd(double* At, int m, int n, double* W, double* Vt, int n1) { orc_jacobi_svd_ord(At, m, n, W, Vt, n1, 0); }

/* SVD of a row-major m x n matrix A (m >= n, n <= ORC_LA_MAX). Ut: n x m, Vt: n x n. */
void orc_svd(const double* A, int m, int n, double* W, double* Ut, double* Vt) {
    int i, j;
    for (i = 0; i < n; i++) for (j = 0; j < m; j++) Ut[i * m + j] = A[j * n + i];
    orc_jacobi_svd(Ut, m, n, W, Vt, n);
}

/* x = pinv(A) b through the SVD (cv::SVD::backSubst thresholding: w_i > 2*DBL_EPSILON * sum(w)). */
void orc_svd_solve(const double* A, int m, int n, const double* b, double* x) {
    double W[ORC_LA_MAX], Ut[ORC_LA_MAX * ORC_LA_MAX], Vt[ORC_LA_MAX * ORC_LA_MAX];
    int i, k;
    orc_svd(A, m, n, W, Ut, Vt);
    double thr = 0;
    for (i = 0; i < n; i++) thr += W[i];
    thr *= DBL_EPSILON * 2;
    for (k = 0; k < n; k++) x[k] = 0;
    for (i = 0; i < n; i++) {
        if (W[i] <= thr) continue;
        double s = 0;
        for (k = 0; k < m; k++) s += Ut[i * m + k] * b[k];
        s /= W[i];
        for (k = 0; k < n; k++) x[k] += s * Vt[i * n + k];
    }
}

/* pseudo-inverse of a 3x3 through the SVD (cvInvert(.., CV_SVD)) */
void orc_inv3_svd(const double A[9], double Ainv[9]) {
    double W[3], Ut[9], Vt[9];
    int i, j, k;
    orc_svd(A, 3, 3, W, Ut, Vt);
    double thr = (W[0] + W[1] + W[2]) * DBL_EPSILON * 2;
    for (i = 0; i < 9; i++) Ainv[i] = 0;
    for (k = 0; k < 3; k++) {
        if (W[k] <= thr) continue;
        double iw = 1 / W[k];
        for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) Ainv[i * 3 + j] += Vt[k * 3 + i] * iw * Ut[k * 3 + j];
    }
}

/* Householder QR least squares, A is m x n row-major (destroyed), b (destroyed) -> x. Returns 0 if singular. */
int orc_qr_solve(double* A, int m, int n, double* b, double* x) {
    double A1[ORC_LA_MAX], A2[ORC_LA_MAX];
    int i, j, k;
    for (k = 0; k < n; k++) {
        double eta = 0;
        for (i = k; i < m; i++) { double e = fabs(A[i * n + k]); if (eta < e) eta = e; }
        if (eta == 0) return 0;
        double sum2 = 0, inv_eta = 1. / eta;
        for (i = k; i < m; i++) { A[i * n + k] *= inv_eta; sum2 += A[i * n + k] * A[i * n + k]; }
        double sigma = sqrt(sum2);
        if (A[k * n + k] < 0) sigma = -sigma;
        A[k * n + k] += sigma;
        A1[k] = sigma * A[k * n + k];
        A2[k] = -eta * sigma;
        for (j = k + 1; j < n; j++) {
            double sum = 0;
            for (i = k; i < m; i++) sum += A[i * n + k] * A[i * n + j];
            double tau = sum / A1[k];
            for (i = k; i < m; i++) A[i * n + j] -= tau * A[i * n + k];
        }
    }
    for (j = 0; j < n; j++) {                 /* b <- Qt b */
        double tau = 0;
        for (i = j; i < m; i++) tau += A[i * n + j] * b[i];
        tau /= A1[j];
        for (i = j; i < m; i++) b[i] -= tau * A[i * n + j];
    }
    x[n - 1] = b[n - 1] / A2[n - 1];          /* x = R^-1 b */
    for (i = n - 2; i >= 0; i--) {
        double sum = 0;
        for (j = i + 1; j < n; j++) sum += A[i * n + j] * x[j];
        x[i] = (b[i] - sum) / A2[i];
    }
    return 1;
}
