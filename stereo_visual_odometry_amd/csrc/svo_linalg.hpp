// svo_linalg.hpp — small dense f64 linear algebra for the geometry kernels (device code).
// One-sided Jacobi SVD (Hestenes) with fixed compile-time sizes.  Only + - * / sqrt in a fixed
// order and no FMA contraction (-ffp-contract=off), so results are IEEE-reproducible.
#pragma once
#include <hip/hip_runtime.h>

#define SVO_DBL_EPS 2.2204460492503131e-16
#define SVO_DBL_MIN 2.2250738585072014e-308

// At: N rows of length M (transpose of an M x N matrix, M >= N).  On exit rows i < n1 of At hold the
// left singular vectors, Wv[N] the singular values (descending), Vt (N x N) the right singular vectors as rows.
// A strided view of per-thread data living in LDS as [element][thread]: element i of this thread is base[i * STRIDE].
template <int STRIDE> struct LdsVec {
    double* base;
    __device__ __forceinline__ double& operator[](int i) const { return base[i * STRIDE]; }
    __device__ __forceinline__ LdsVec operator+(int off) const { return LdsVec{base + off * STRIDE}; }
};

// c and s of one Hestenes rotation (OpenCV's JacobiSVDImpl_): two arithmetically different forms, chosen by the sign of beta.
// The lanes of a wave disagree about that sign, so written as if / else the wave walks BOTH division -> square root -> division
// chains one after the other (~90 dependent f64 instructions instead of ~45).  Here the two forms share one instruction stream:
// the same operations on selected operands, the same bits.
__device__ __forceinline__ void jacobi_cs(double p, double beta, double gamma, double& c, double& s) {
    const bool neg = beta < 0;
    const double num = neg ? (gamma - beta) * 0.5 : gamma + beta;
    const double den = neg ? gamma : gamma * 2;
    const double r = sqrt(num / den);
    const double o = p / (gamma * r * 2);
    c = neg ? o : r;
    s = neg ? r : o;
}

template <int M, int N, typename PA = double*, typename PW = double*>
__device__ void jacobi_svd(PA At, PW Wv, PA Vt, int n1) {
    const double eps = SVO_DBL_EPS * 10, minval = SVO_DBL_MIN;
    const int max_iter = M > 30 ? M : 30;
    for (int i = 0; i < N; i++) {
        double sd = 0;
        for (int k = 0; k < M; k++) { double t = At[i * M + k]; sd += t * t; }
        Wv[i] = sd;
        for (int k = 0; k < N; k++) Vt[i * N + k] = 0;
        Vt[i * N + i] = 1;
    }
#pragma unroll 1
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll 1
        for (int i = 0; i < N - 1; i++)
#pragma unroll 1
            for (int j = i + 1; j < N; j++) {
                PA Ai = At + i * M; PA Aj = At + j * M;
                double a = Wv[i], p = 0, b = Wv[j], c, s;
                for (int k = 0; k < M; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = sqrt(p * p + beta * beta);
                jacobi_cs(p, beta, gamma, c, s);
                a = b = 0;
                for (int k = 0; k < M; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                Wv[i] = a; Wv[j] = b;
                changed = true;
                PA Vi = Vt + i * N; PA Vj = Vt + j * N;
                for (int k = 0; k < N; k++) {
                    double t0 = c * Vi[k] + s * Vj[k];
                    double t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < N; i++) {
        double sd = 0;
        for (int k = 0; k < M; k++) { double t = At[i * M + k]; sd += t * t; }
        Wv[i] = sqrt(sd);
    }
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        for (int k = i + 1; k < N; k++) if (Wv[j] < Wv[k]) j = k;
        if (i != j) {
            double t = Wv[i]; Wv[i] = Wv[j]; Wv[j] = t;
            for (int k = 0; k < M; k++) { t = At[i * M + k]; At[i * M + k] = At[j * M + k]; At[j * M + k] = t; }
            for (int k = 0; k < N; k++) { t = Vt[i * N + k]; Vt[i * N + k] = Vt[j * N + k]; Vt[j * N + k] = t; }
        }
    }
    for (int i = 0; i < n1; i++) {
        double sd = i < N ? Wv[i] : 0;
        double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < M; k++) At[i * M + k] *= s;
    }
}

// The same algorithm on REGISTER arrays with every index a compile-time constant (all loops over rows / pairs unrolled; only the
// sweep loop is dynamic).  jacobi_svd above walks its row pairs with run-time indices: on private arrays that turns every
// access into a select chain or a scratch access, in LDS into a dependent ds_read.  This form executes the same rotations in the
// same order with the same arithmetic — bit-identical results — in a fraction of the instructions, for the small fixed sizes the
// geometry kernels solve over and over (3 x 3, 4 x 4, 6 x 5).  The selection sort is replayed with predicated row swaps.
template <int M, int N>
__device__ __forceinline__ void jacobi_svd_reg(double (&At)[N][M], double (&Wv)[N], double (&Vt)[N][N], bool normalize) {
    const double eps = SVO_DBL_EPS * 10, minval = SVO_DBL_MIN;
    constexpr int max_iter = M > 30 ? M : 30;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
        Wv[i] = sd;
#pragma unroll
        for (int k = 0; k < N; k++) Vt[i][k] = i == k ? 1.0 : 0.0;
    }
#pragma unroll 1
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = Wv[i], p = 0, b = Wv[j], c, s;
#pragma unroll
                for (int k = 0; k < M; k++) p += At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = sqrt(p * p + beta * beta);
                jacobi_cs(p, beta, gamma, c, s);
                a = b = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    double t0 = c * At[i][k] + s * At[j][k];
                    double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                Wv[i] = a; Wv[j] = b;
                changed = true;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    double t0 = c * Vt[i][k] + s * Vt[j][k];
                    double t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0; Vt[j][k] = t1;
                }
            }
        }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
        Wv[i] = sqrt(sd);
    }
    // selection sort, descending, the first maximum wins ties: row i <-> row j with j found at run time, applied as predicated swaps
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        int j = i; double wj = Wv[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) { const bool g = wj < Wv[k]; j = g ? k : j; wj = g ? Wv[k] : wj; }
#pragma unroll
        for (int k = i + 1; k < N; k++) {
            const bool e = j == k;
            { const double x = Wv[i], y = Wv[k]; Wv[i] = e ? y : x; Wv[k] = e ? x : y; }
#pragma unroll
            for (int c = 0; c < M; c++) { const double x = At[i][c], y = At[k][c]; At[i][c] = e ? y : x; At[k][c] = e ? x : y; }
#pragma unroll
            for (int c = 0; c < N; c++) { const double x = Vt[i][c], y = Vt[k][c]; Vt[i][c] = e ? y : x; Vt[k][c] = e ? x : y; }
        }
    }
    if (normalize) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const double sd = Wv[i];
            const double s = sd > minval ? 1 / sd : 0.;
#pragma unroll
            for (int k = 0; k < M; k++) At[i][k] *= s;
        }
    }
}

// SVD of a row-major M x N matrix A.  Ut: N x M, Vt: N x N.
template <int M, int N>
__device__ void svd_rm(const double* A, double* Wv, double* Ut, double* Vt) {
    if constexpr (M <= 4 && N <= 4) {                                // small: the register form (bit-identical, far fewer instructions)
        double At[N][M], W[N], V[N][N];
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < M; j++) At[i][j] = A[j * N + i];
        }
        jacobi_svd_reg<M, N>(At, W, V, true);
#pragma unroll
        for (int i = 0; i < N; i++) {
            Wv[i] = W[i];
#pragma unroll
            for (int j = 0; j < M; j++) Ut[i * M + j] = At[i][j];
#pragma unroll
            for (int j = 0; j < N; j++) Vt[i * N + j] = V[i][j];
        }
    } else {
        for (int i = 0; i < N; i++) for (int j = 0; j < M; j++) Ut[i * M + j] = A[j * N + i];
        jacobi_svd<M, N>(Ut, Wv, Vt, N);
    }
}

// x = pinv(A) b by SVD back-substitution; singular values <= 2*eps*sum(w) are dropped.
template <int M, int N>
__device__ void svd_solve(const double* A, const double* b, double* x) {
    double Wv[N], Ut[N * M], Vt[N * N];
    svd_rm<M, N>(A, Wv, Ut, Vt);
    double thr = 0;
    for (int i = 0; i < N; i++) thr += Wv[i];
    thr *= SVO_DBL_EPS * 2;
    for (int k = 0; k < N; k++) x[k] = 0;
    for (int i = 0; i < N; i++) {
        if (Wv[i] <= thr) continue;
        double s = 0;
        for (int k = 0; k < M; k++) s += Ut[i * M + k] * b[k];
        s /= Wv[i];
        for (int k = 0; k < N; k++) x[k] += s * Vt[i * N + k];
    }
}

__device__ inline void inv3_svd(const double A[9], double Ainv[9]) {
    double Wv[3], Ut[9], Vt[9];
    svd_rm<3, 3>(A, Wv, Ut, Vt);
    double thr = (Wv[0] + Wv[1] + Wv[2]) * SVO_DBL_EPS * 2;
    for (int i = 0; i < 9; i++) Ainv[i] = 0;
    for (int k = 0; k < 3; k++) {
        if (Wv[k] <= thr) continue;
        double iw = 1 / Wv[k];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ainv[i * 3 + j] += Vt[k * 3 + i] * iw * Ut[k * 3 + j];
    }
}

// Householder QR least squares: A (M x N, row-major, destroyed), b (destroyed) -> x.  false if singular.
template <int M, int N>
__device__ bool qr_solve(double* A, double* b, double* x) {
    double A1[N], A2[N];
    for (int k = 0; k < N; k++) {
        double eta = 0;
        for (int i = k; i < M; i++) { double e = fabs(A[i * N + k]); if (eta < e) eta = e; }
        if (eta == 0) return false;
        double sum2 = 0, inv_eta = 1. / eta;
        for (int i = k; i < M; i++) { A[i * N + k] *= inv_eta; sum2 += A[i * N + k] * A[i * N + k]; }
        double sigma = sqrt(sum2);
        if (A[k * N + k] < 0) sigma = -sigma;
        A[k * N + k] += sigma;
        A1[k] = sigma * A[k * N + k];
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < N; j++) {
            double sum = 0;
            for (int i = k; i < M; i++) sum += A[i * N + k] * A[i * N + j];
            double tau = sum / A1[k];
            for (int i = k; i < M; i++) A[i * N + j] -= tau * A[i * N + k];
        }
    }
    for (int j = 0; j < N; j++) {
        double tau = 0;
        for (int i = j; i < M; i++) tau += A[i * N + j] * b[i];
        tau /= A1[j];
        for (int i = j; i < M; i++) b[i] -= tau * A[i * N + j];
    }
    x[N - 1] = b[N - 1] / A2[N - 1];
    for (int i = N - 2; i >= 0; i--) {
        double sum = 0;
        for (int j = i + 1; j < N; j++) sum += A[i * N + j] * x[j];
        x[i] = (b[i] - sum) / A2[i];
    }
    return true;
}

// Rodrigues: rotation vector -> matrix (+ optional 3x9 Jacobian dR/dr), and matrix -> vector.
__device__ inline void rodrigues_to_matrix(const double r[3], double R[9], double* J) {
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < SVO_DBL_EPS) {
        for (int i = 0; i < 9; i++) R[i] = 0;
        R[0] = R[4] = R[8] = 1;
        if (J) { for (int i = 0; i < 27; i++) J[i] = 0; J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
        return;
    }
    double c, s;
    sincos(theta, &s, &c);                                             // one argument reduction for both
    double c1 = 1. - c, itheta = 1. / theta;
    double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                           0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                   0, 0, 1, 0, 0, 0, -1, 0, 0,
                                   0, -1, 0, 1, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
#pragma unroll
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

__device__ inline void rodrigues_to_vector(const double Rin[9], double r[3]) {
    double Wv[3], Ut[9], Vt[9], R[9];
    svd_rm<3, 3>(Rin, Wv, Ut, Vt);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += Ut[k * 3 + i] * Vt[k * 3 + j];
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
