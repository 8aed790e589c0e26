/* ORACLE (test infrastructure, see orc.h).  Triangulation, Rodrigues, EPnP, RANSAC-PnP, LM refine.
 * Stands in for cv::triangulatePoints + cv::convertPointsFromHomogeneous (/root/reference/src/vo.cpp:89-94),
 * cv::solvePnPRansac + cv::Rodrigues (vo.cpp:282-313) and the 4x4 inverse of vo.cpp:246-258, restating
 * OpenCV 4.5 modules/calib3d/src/{triangulate,solvepnp,ptsetreg,epnp,calibration}.cpp and
 * modules/core/src/rand.cpp as summarised in SURVEY.md Appendix A.4-A.7.  Deviations D2, D4, D5: orc.h. */
#include "orc.h"
#include "orc_linalg.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ triangulation (A.4) */
void orc_triangulate(const float Pl[12], const float Pr[12], int n, const float* pts_l, const float* pts_r,
                     float* xyz, float* homog) {
    int i;
#pragma omp parallel for
    for (i = 0; i < n; i++) {
        int k;
        double A[16], W[4], Ut[16], Vt[16];
        double xl = pts_l[2 * i], yl = pts_l[2 * i + 1], xr = pts_r[2 * i], yr = pts_r[2 * i + 1];
        for (k = 0; k < 4; k++) {
            A[0 * 4 + k] = xl * (double)Pl[8 + k] - (double)Pl[0 + k];
            A[1 * 4 + k] = yl * (double)Pl[8 + k] - (double)Pl[4 + k];
            A[2 * 4 + k] = xr * (double)Pr[8 + k] - (double)Pr[0 + k];
            A[3 * 4 + k] = yr * (double)Pr[8 + k] - (double)Pr[4 + k];
        }
        if (orc_get_opencv_mode() & ORC_ALT_TRI_RR) {
            int r, cc;
            for (r = 0; r < 4; r++) for (cc = 0; cc < 4; cc++) Ut[r * 4 + cc] = A[cc * 4 + r];
            orc_jacobi_svd_ord(Ut, 4, 4, W, Vt, 4, 1);
        } else
        orc_svd(A, 4, 4, W, Ut, Vt);
        float X = (float)Vt[12], Y = (float)Vt[13], Z = (float)Vt[14], Wh = (float)Vt[15];   /* 4xN output is CV_32F */
        if (homog) { homog[4 * i] = X; homog[4 * i + 1] = Y; homog[4 * i + 2] = Z; homog[4 * i + 3] = Wh; }
        float scale = Wh != 0.f ? 1.f / Wh : 1.f;     /* convertPointsFromHomogeneous, f32 */
        xyz[3 * i] = X * scale; xyz[3 * i + 1] = Y * scale; xyz[3 * i + 2] = Z * scale;
    }
}

/* ------------------------------------------------------------------ Rodrigues (A.7) */
void orc_rodrigues_to_matrix(const double r[3], double R[9], double J[27]) {
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    int i, k;
    if (theta < DBL_EPSILON) {
        for (i = 0; i < 9; i++) R[i] = 0;
        R[0] = R[4] = R[8] = 1;
        if (J) { memset(J, 0, sizeof(double) * 27); J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                           0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                   0, 0, 1, 0, 0, 0, -1, 0, 0,
                                   0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

void orc_rodrigues_to_vector(const double Rin[9], double r[3]) {
    double W[3], Ut[9], Vt[9], R[9];
    int i, j, k;
    orc_svd(Rin, 3, 3, W, Ut, Vt);                        /* R = U * Vt (orthonormal clean-up) */
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) {
        double s = 0;
        for (k = 0; k < 3; k++) s += Ut[k * 3 + i] * Vt[k * 3 + j];
        R[i * 3 + j] = s;
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { rx = ry = rz = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5; ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

/* ------------------------------------------------------------------ cv::RNG (A.6) */
void orc_rng_init(orc_rng* r, uint64_t seed) { r->state = seed ? seed : 0xffffffffULL; }
uint32_t orc_rng_next(orc_rng* r) {
    r->state = (uint64_t)(uint32_t)r->state * 4164903690U + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}
int orc_rng_uniform(orc_rng* r, int a, int b) { return a == b ? a : (int)(orc_rng_next(r) % (uint32_t)(b - a) + a); }

int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* ------------------------------------------------------------------ EPnP (A.6b) */
#define EPNP_MAXN 16
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

typedef struct {
    int n; double fu, fv, uc, vc;
    double pws[3 * EPNP_MAXN], us[2 * EPNP_MAXN], alphas[4 * EPNP_MAXN], pcs[3 * EPNP_MAXN];
    double cws[4][3], ccs[4][3];
} epnp_t;

static void epnp_compute_ccs_pcs(epnp_t* e, const double* betas, const double* vt) {
    int i, j, k;
    for (i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0;
    for (i = 0; i < 4; i++) {
        const double* v = vt + 12 * (11 - i);
        for (j = 0; j < 4; j++) for (k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
    for (i = 0; i < e->n; i++) {
        const double* a = e->alphas + 4 * i; double* pc = e->pcs + 3 * i;
        for (j = 0; j < 3; j++) pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
}

static void epnp_estimate_R_t(epnp_t* e, double R[9], double t[3]) {
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0}, abt[9], W[3], Ut[9], Vt[9];
    int i, j, k, n = e->n;
    for (i = 0; i < n; i++) for (j = 0; j < 3; j++) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
    for (j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    for (i = 0; i < 9; i++) abt[i] = 0;
    for (i = 0; i < n; i++) {
        const double* pc = e->pcs + 3 * i; const double* pw = e->pws + 3 * i;
        for (j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    orc_svd(abt, 3, 3, W, Ut, Vt);
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) {          /* R = U * Vt */
        double s = 0;
        for (k = 0; k < 3; k++) s += Ut[k * 3 + i] * Vt[k * 3 + j];
        R[i * 3 + j] = s;
    }
    double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7]
               - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (i = 0; i < 3; i++) t[i] = pc0[i] - dot3(R + 3 * i, pw0);
}

static double epnp_reproj_error(const epnp_t* e, const double R[9], const double t[3]) {
    double sum2 = 0.0; int i;
    for (i = 0; i < e->n; i++) {
        const double* pw = e->pws + 3 * i;
        double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
        double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / e->n;
}

static double epnp_compute_R_and_t(epnp_t* e, const double* vt, const double* betas, double R[9], double t[3]) {
    int i, j;
    epnp_compute_ccs_pcs(e, betas, vt);
    if (e->pcs[2] < 0.0) {                                   /* solve_for_sign */
        for (i = 0; i < 4; i++) for (j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
        for (i = 0; i < e->n; i++) { e->pcs[3 * i] = -e->pcs[3 * i]; e->pcs[3 * i + 1] = -e->pcs[3 * i + 1]; e->pcs[3 * i + 2] = -e->pcs[3 * i + 2]; }
    }
    epnp_estimate_R_t(e, R, t);
    return epnp_reproj_error(e, R, t);
}

static void epnp_gauss_newton(const double* L, const double* rho, double betas[4]) {
    int it, i;
    for (it = 0; it < 5; it++) {
        double A[24], B[6], X[4];
        for (i = 0; i < 6; i++) {
            const double* rl = L + 10 * i; double* ra = A + 4 * i;
            ra[0] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
            ra[1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
            ra[2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
            ra[3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
            B[i] = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] +
                             rl[3] * betas[0] * betas[2] + rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] +
                             rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] + rl[8] * betas[2] * betas[3] +
                             rl[9] * betas[3] * betas[3]);
        }
        if (!orc_qr_solve(A, 6, 4, B, X)) return;
        for (i = 0; i < 4; i++) betas[i] += X[i];
    }
}

double orc_epnp(int n, const double* obj, const double* img, double fx, double fy, double cx, double cy,
                double R[9], double t[3]) {
    epnp_t e; int i, j, k;
    if (n > EPNP_MAXN) n = EPNP_MAXN;
    e.n = n; e.fu = fx; e.fv = fy; e.uc = cx; e.vc = cy;
    memcpy(e.pws, obj, sizeof(double) * 3 * n);
    memcpy(e.us, img, sizeof(double) * 2 * n);
    /* choose_control_points */
    for (j = 0; j < 3; j++) e.cws[0][j] = 0;
    for (i = 0; i < n; i++) for (j = 0; j < 3; j++) e.cws[0][j] += e.pws[3 * i + j];
    for (j = 0; j < 3; j++) e.cws[0][j] /= n;
    {
        double ptp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dc[3], ut[9], vt3[9];
        for (i = 0; i < n; i++) {
            double d[3];
            for (j = 0; j < 3; j++) d[j] = e.pws[3 * i + j] - e.cws[0][j];
            for (j = 0; j < 3; j++) for (k = 0; k < 3; k++) ptp[3 * j + k] += d[j] * d[k];
        }
        orc_svd(ptp, 3, 3, dc, ut, vt3);
        for (i = 1; i < 4; i++) {
            double kk = sqrt(dc[i - 1] / n);
            for (j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + kk * vt3[3 * (i - 1) + j];
        }
    }
    /* compute_barycentric_coordinates */
    {
        double cc[9], ci[9];
        for (i = 0; i < 3; i++) for (j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
        orc_inv3_svd(cc, ci);
        for (i = 0; i < n; i++) {
            const double* pi = e.pws + 3 * i; double* a = e.alphas + 4 * i;
            for (j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) + ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    /* M (2n x 12) and MtM */
    double MtM[144], W[12], Ut[144], Vt[144];
    {
        double M[2 * EPNP_MAXN * 12];
        for (i = 0; i < n; i++) {
            double* M1 = M + 24 * i; double* M2 = M1 + 12; const double* as = e.alphas + 4 * i;
            double u = e.us[2 * i], v = e.us[2 * i + 1];
            for (j = 0; j < 4; j++) {
                M1[3 * j] = as[j] * e.fu; M1[3 * j + 1] = 0.0; M1[3 * j + 2] = as[j] * (e.uc - u);
                M2[3 * j] = 0.0; M2[3 * j + 1] = as[j] * e.fv; M2[3 * j + 2] = as[j] * (e.vc - v);
            }
        }
        for (i = 0; i < 12; i++) for (j = i; j < 12; j++) {
            double s = 0;
            for (k = 0; k < 2 * n; k++) s += M[12 * k + i] * M[12 * k + j];
            MtM[12 * i + j] = MtM[12 * j + i] = s;
        }
    }
    /* MtM is exactly symmetric, so it is its own transpose.  Round-robin ordering (orc_linalg.c): OpenCV uses the
       cyclic-by-rows order; the ordering is not observable through any reference test and the parallel order is the
       one the GPU kernel can run with 6 lanes per hypothesis, bit-identically (deviation D5 in orc.h). */
    memcpy(Ut, MtM, sizeof(MtM));
    /* The basis EPnP reads is the LEFT singular vectors — the normalised rows of the rotated At — as
       cvSVD(&MtM, &D, &Ut, 0, CV_SVD_MODIFY_A | CV_SVD_U_T) returns them (epnp.cpp).  Rounds 1-2 read the right ones (former
       deviation D3): same subspace, but for the two (numerically) zero singular values of a 5-point MtM a different basis OF it,
       and the beta approximations depend on the basis — measured on the reference's recording (tools/deviation_ablation.py):
       with the left vectors all 128 rows of run1/result.csv are reproduced, with the right ones an inlier decision flips at
       frame 25.  Deviation switches (orc.h): ORC_ALT_D3_RIGHT puts the old behaviour back; D5 reverted = cyclic-by-rows sweeps. */
    const unsigned ocv = orc_get_opencv_mode();
    orc_jacobi_svd_ord(Ut, 12, 12, W, Vt, 12, (ocv & ORC_OCV_D5_JACOBI_CYCLIC) ? 0 : 1);
    const double* basis = (ocv & ORC_ALT_D3_RIGHT) ? Vt : Ut;
    /* L_6x10 and rho; null-space basis v[i] = basis row 11-i */
    double L[60], rho[6];
    {
        const double* v[4]; double dv[4][6][3];
        for (i = 0; i < 4; i++) v[i] = basis + 12 * (11 - i);
        for (i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (j = 0; j < 6; j++) {
                for (k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        for (i = 0; i < 6; i++) {
            double* row = L + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
        rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]); rho[2] = dist2(e.cws[0], e.cws[3]);
        rho[3] = dist2(e.cws[1], e.cws[2]); rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);
    }
    double Betas[4][4], rep[4], Rs[4][9], ts[4][3];
    {   /* approx 1: [B11 B12 B13 B14] from columns {0,1,3,6} */
        double L4[24], b4[4];
        for (i = 0; i < 6; i++) { L4[4 * i] = L[10 * i]; L4[4 * i + 1] = L[10 * i + 1]; L4[4 * i + 2] = L[10 * i + 3]; L4[4 * i + 3] = L[10 * i + 6]; }
        orc_svd_solve(L4, 6, 4, rho, b4);
        double* be = Betas[1];
        if (b4[0] < 0) { be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
        else { be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
        epnp_gauss_newton(L, rho, be);
        rep[1] = epnp_compute_R_and_t(&e, basis, be, Rs[1], ts[1]);
    }
    {   /* approx 2: [B11 B12 B22] from columns {0,1,2} */
        double L3[18], b3[3];
        for (i = 0; i < 6; i++) { L3[3 * i] = L[10 * i]; L3[3 * i + 1] = L[10 * i + 1]; L3[3 * i + 2] = L[10 * i + 2]; }
        orc_svd_solve(L3, 6, 3, rho, b3);
        double* be = Betas[2];
        if (b3[0] < 0) { be[0] = sqrt(-b3[0]); be[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
        else { be[0] = sqrt(b3[0]); be[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
        if (b3[1] < 0) be[0] = -be[0];
        be[2] = 0.0; be[3] = 0.0;
        epnp_gauss_newton(L, rho, be);
        rep[2] = epnp_compute_R_and_t(&e, basis, be, Rs[2], ts[2]);
    }
    {   /* approx 3: [B11 B12 B22 B13 B23] from columns {0..4} */
        double L5[30], b5[5];
        for (i = 0; i < 6; i++) for (j = 0; j < 5; j++) L5[5 * i + j] = L[10 * i + j];
        orc_svd_solve(L5, 6, 5, rho, b5);
        double* be = Betas[3];
        if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
        else { be[0] = sqrt(b5[0]); be[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) be[0] = -be[0];
        be[2] = b5[3] / be[0];
        be[3] = 0.0;
        epnp_gauss_newton(L, rho, be);
        rep[3] = epnp_compute_R_and_t(&e, basis, be, Rs[3], ts[3]);
    }
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    memcpy(R, Rs[N], sizeof(double) * 9);
    memcpy(t, ts[N], sizeof(double) * 3);
    return rep[N];
}

/* ------------------------------------------------------------------ projection with Jacobians (cvProjectPoints2, no distortion) */
static void project_points(int n, const double* obj, const double rvec[3], const double tvec[3],
                           double fx, double fy, double cx, double cy, double* proj, double* J /* 2n x 6 or NULL */) {
    double R[9], dRdr[27];
    int i, j;
    orc_rodrigues_to_matrix(rvec, R, J ? dRdr : NULL);
    for (i = 0; i < n; i++) {
        double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        proj[2 * i] = x * fx + cx; proj[2 * i + 1] = y * fy + cy;
        if (J) {
            double* jx = J + 12 * i; double* jy = jx + 6;
            double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11], X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14], X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17], X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                jx[j] = fx * dxdr; jy[j] = fy * dydr;
            }
            jx[3] = fx * z; jx[4] = 0; jx[5] = fx * (-x * z);
            jy[3] = 0; jy[4] = fy * z; jy[5] = fy * (-y * z);
        }
    }
}

/* ------------------------------------------------------------------ LM refine (CvLevMarq as driven by cvFindExtrinsicCameraParams2) */
static void lm_step(const double JtJ[36], const double JtErr[6], int lambdaLg10, const double prev[6], double param[6]) {
    const double LOG10 = log(10.);
    double lambda = exp(lambdaLg10 * LOG10), A[36], x[6];
    int i;
    memcpy(A, JtJ, sizeof(A));
    for (i = 0; i < 6; i++) A[7 * i] *= 1. + lambda;
    orc_svd_solve(A, 6, 6, JtErr, x);                       /* cv::solve(.., DECOMP_SVD) */
    for (i = 0; i < 6; i++) param[i] = prev[i] - x[i];
}

int orc_pnp_refine_lm(int n, const double* obj, const double* img, double fx, double fy, double cx, double cy,
                      double rvec[3], double tvec[3]) {
    const int max_iter = 20;
    const double eps = FLT_EPSILON;
    double param[6], prev[6], JtJ[36], JtErr[6];
    double* J = (double*)malloc(sizeof(double) * (size_t)n * 12);
    double* err = (double*)malloc(sizeof(double) * (size_t)n * 2);
    int lambdaLg10 = -3, iters = 0, i, j, k;
    double prevErrNorm = DBL_MAX, errNorm = 0;
    memcpy(param, rvec, sizeof(double) * 3); memcpy(param + 3, tvec, sizeof(double) * 3);
    /* STARTED */
    project_points(n, obj, param, param + 3, fx, fy, cx, cy, err, J);
    for (k = 0; k < 2 * n; k++) err[k] -= img[k];
    for (;;) {
        /* CALC_J */
        for (i = 0; i < 6; i++) for (j = i; j < 6; j++) {
            double s = 0;
            for (k = 0; k < 2 * n; k++) s += J[6 * k + i] * J[6 * k + j];
            JtJ[6 * i + j] = JtJ[6 * j + i] = s;
        }
        for (i = 0; i < 6; i++) { double s = 0; for (k = 0; k < 2 * n; k++) s += J[6 * k + i] * err[k]; JtErr[i] = s; }
        memcpy(prev, param, sizeof(prev));
        lm_step(JtJ, JtErr, lambdaLg10, prev, param);
        if (iters == 0) { double s = 0; for (k = 0; k < 2 * n; k++) s += err[k] * err[k]; prevErrNorm = sqrt(s); }
        /* CHECK_ERR */
        for (;;) {
            double s = 0;
            project_points(n, obj, param, param + 3, fx, fy, cx, cy, err, NULL);
            for (k = 0; k < 2 * n; k++) { err[k] -= img[k]; s += err[k] * err[k]; }
            errNorm = sqrt(s);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { lm_step(JtJ, JtErr, lambdaLg10, prev, param); continue; }
            }
            break;
        }
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        {
            double dn = 0, pn = 0;
            for (i = 0; i < 6; i++) { dn += (param[i] - prev[i]) * (param[i] - prev[i]); pn += prev[i] * prev[i]; }
            if (++iters >= max_iter || sqrt(dn) / (sqrt(pn) + DBL_EPSILON) < eps) break;
        }
        prevErrNorm = errNorm;
        project_points(n, obj, param, param + 3, fx, fy, cx, cy, err, J);
        for (k = 0; k < 2 * n; k++) err[k] -= img[k];
    }
    memcpy(rvec, param, sizeof(double) * 3); memcpy(tvec, param + 3, sizeof(double) * 3);
    free(J); free(err);
    return iters;
}

/* ------------------------------------------------------------------ solvePnPRansac as used by cameraToWorld (vo.cpp:282-313) */
static void epnp_on_subset(const float K[9], const float* world, const float* cam, const int idx[5], int m,
                           double R[9], double t[3]) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5], obj[15], img[10];
    double ifx = 1. / fx, ify = 1. / fy;
    int i;
    for (i = 0; i < m; i++) {
        obj[3 * i] = world[3 * idx[i]]; obj[3 * i + 1] = world[3 * idx[i] + 1]; obj[3 * i + 2] = world[3 * idx[i] + 2];
        /* undistortPoints on CV_32FC2 with zero distortion: normalise in f64, store f32; epnp re-applies fu,uc */
        float xn = (float)(((double)cam[2 * idx[i]] - cx) * ifx), yn = (float)(((double)cam[2 * idx[i] + 1] - cy) * ify);
        img[2 * i] = (double)xn * fx + cx; img[2 * i + 1] = (double)yn * fy + cy;
    }
    orc_epnp(m, obj, img, fx, fy, cx, cy, R, t);
}

/* computeError + findInliers: f64 projection stored as f32, squared pixel distance in f32, err <= thr^2 */
static int score_model(const float K[9], int n, const float* world, const float* cam, const double R[9], const double t[3],
                       float thr2, uint8_t* mask) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    int i, good = 0;
    for (i = 0; i < n; i++) {
        double X = world[3 * i], Y = world[3 * i + 1], Z = world[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        float pu = (float)(x * fx + cx), pv = (float)(y * fy + cy);
        float du = cam[2 * i] - pu, dv = cam[2 * i + 1] - pv;
        float e = du * du + dv * dv;
        mask[i] = (uint8_t)(e <= thr2);
        good += mask[i];
    }
    return good;
}

int orc_camera_to_world(const float K[9], int n, const float* cam_pts, const float* world_pts,
                        double R[9], double t[3], int* inliers, int* n_inliers,
                        int ransac_iterations, float reproj_error, float confidence, int* dbg) {
    const int model_points = 5;
    const unsigned ocv = orc_get_opencv_mode();
    double bestR[9], bestT[3], lastR[9], lastT[3];
    int i, iter, max_good = 0, iters_run = 0;
    *n_inliers = 0;
    if (dbg) { dbg[0] = 0; dbg[1] = 0; }
    if (n < 4) return 0;                  /* CV_Assert(npoints >= 4): the reference would throw; the oracle reports failure */
    if (n <= model_points) {
        /* solvepnp.cpp: `if (model_points == npoints)` — with 4 points the kernel is P3P (model_points = 4), with 5 it is EPnP:
         * ONE direct solvePnP on all points, every point an inlier, no RANSAC and no refine.  The pose still makes the
         * Rodrigues round trip of cameraToWorld (rvec out of solvePnP, vo.cpp:308 back to a matrix). */
        double rvec[3];
        int ok;
        if (n == 4) {
            double fx = K[0], fy = K[4], cx = K[2], cy = K[5], obj[12], img[8];
            for (i = 0; i < 4; i++) {
                obj[3 * i] = world_pts[3 * i]; obj[3 * i + 1] = world_pts[3 * i + 1]; obj[3 * i + 2] = world_pts[3 * i + 2];
                /* undistortPoints(.., P = cameraMatrix) on CV_32FC2 with zero distortion: normalise and re-project in f64, store f32 */
                img[2 * i] = (double)(float)((((double)cam_pts[2 * i] - cx) * (1. / fx)) * fx + cx);
                img[2 * i + 1] = (double)(float)((((double)cam_pts[2 * i + 1] - cy) * (1. / fy)) * fy + cy);
            }
            ok = orc_p3p(obj, img, fx, fy, cx, cy, bestR, bestT);
        } else {
            int idx[5] = {0, 1, 2, 3, 4};
            epnp_on_subset(K, world_pts, cam_pts, idx, 5, bestR, bestT);
            ok = 1;
        }
        if (!ok) return 0;
        orc_rodrigues_to_vector(bestR, rvec);
        orc_rodrigues_to_matrix(rvec, R, NULL);           /* vo.cpp:308 */
        memcpy(t, bestT, sizeof(double) * 3);
        for (i = 0; i < n; i++) inliers[i] = i;
        *n_inliers = n;
        if (dbg) { dbg[0] = 0; dbg[1] = n; }
        return 1;
    }
    uint8_t* mask = (uint8_t*)malloc((size_t)n);
    uint8_t* best_mask = (uint8_t*)malloc((size_t)n);
    double thr = (double)reproj_error;
    float thr2 = (float)(thr * thr);
    {
        orc_rng rng; orc_rng_init(&rng, (uint64_t)-1);
        int niters = ransac_iterations > 1 ? ransac_iterations : 1;
        for (iter = 0; iter < niters; iter++) {
            int idx[5];
            for (i = 0; i < model_points; i++) {       /* getSubset */
                int idx_i, dup;
                do {
                    int k2;
                    idx_i = orc_rng_uniform(&rng, 0, n);
                    dup = 0;
                    for (k2 = 0; k2 < i; k2++) if (idx[k2] == idx_i) dup = 1;
                } while (dup);
                idx[i] = idx_i;
            }
            double Rm[9], tm[3];
            epnp_on_subset(K, world_pts, cam_pts, idx, 5, Rm, tm);
            if (ocv & ORC_OCV_D4_RVEC_TRIP) {      /* the model is rvec|tvec (PnPRansacCallback::runKernel); projectPoints rebuilds R from it */
                double rv[3];
                orc_rodrigues_to_vector(Rm, rv); orc_rodrigues_to_matrix(rv, Rm, NULL);
            }
            memcpy(lastR, Rm, sizeof(lastR)); memcpy(lastT, tm, sizeof(lastT));
            int good = score_model(K, n, world_pts, cam_pts, Rm, tm, thr2, mask);
            iters_run = iter + 1;
            if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
                uint8_t* tmp = mask; mask = best_mask; best_mask = tmp;
                memcpy(bestR, Rm, sizeof(bestR)); memcpy(bestT, tm, sizeof(bestT));
                max_good = good;
                niters = orc_ransac_update_num_iters((double)confidence, (double)(n - good) / n, model_points, niters);
            }
        }
    }
    if (dbg) { dbg[0] = iters_run; dbg[1] = max_good; }
    if (max_good <= 0) { free(mask); free(best_mask); return 0; }
    /* refine on the inliers: solvePnP(ITERATIVE, useExtrinsicGuess=true), starting from the best model (D2) */
    {
        double fx = K[0], fy = K[4], cx = K[2], cy = K[5], rvec[3];
        double* obj = (double*)malloc(sizeof(double) * 3 * (size_t)max_good);
        double* img = (double*)malloc(sizeof(double) * 2 * (size_t)max_good);
        int m = 0;
        for (i = 0; i < n; i++) if (best_mask[i]) {
            obj[3 * m] = world_pts[3 * i]; obj[3 * m + 1] = world_pts[3 * i + 1]; obj[3 * m + 2] = world_pts[3 * i + 2];
            img[2 * m] = cam_pts[2 * i]; img[2 * m + 1] = cam_pts[2 * i + 1];
            inliers[m] = i; m++;
        }
        if (ocv & ORC_OCV_D2_LM_FROM_LAST) {   /* rvec / tvec alias the callback's buffers: they hold the LAST hypothesis solved */
            memcpy(bestR, lastR, sizeof(bestR)); memcpy(bestT, lastT, sizeof(bestT));
        }
        orc_rodrigues_to_vector(bestR, rvec);
        orc_pnp_refine_lm(m, obj, img, fx, fy, cx, cy, rvec, bestT);
        orc_rodrigues_to_matrix(rvec, R, NULL);           /* vo.cpp:308 */
        memcpy(t, bestT, sizeof(double) * 3);
        *n_inliers = m;
        free(obj); free(img);
    }
    free(mask); free(best_mask);
    return 1;
}

/* vo.cpp:246-258 — [R t; 0 1]^-1 in closed form (R orthonormal): [Rt, -Rt t; 0 1] */
void orc_inverse_transform(const double R[9], const double t[3], double T[16]) {
    int i, j;
    for (i = 0; i < 3; i++) {
        for (j = 0; j < 3; j++) T[4 * i + j] = R[3 * j + i];
        T[4 * i + 3] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}
