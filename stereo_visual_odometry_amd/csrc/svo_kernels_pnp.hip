// svo_kernels_pnp.hip — stereo triangulation and RANSAC-PnP on the GPU (gfx950).
//
// Replaces, for the reference call sites in src/vo.cpp:
//   :89-94   cv::triangulatePoints + cv::convertPointsFromHomogeneous      -> k_triangulate
//   :282-313 cameraToWorld = cv::solvePnPRansac(.., ITERATIVE, useExtrinsicGuess) -> k_pnp_subsets,
//            k_pnp_epnp, k_pnp_score, k_pnp_final
//   :115-136 inlier feature update, motion gates, getInverseTransform       -> k_pnp_final
//
// RANSAC is restructured for the GPU: OpenCV's loop is sequential only through the adaptive
// iteration count, and its RNG draws do not depend on model quality.  So all K minimal subsets are
// drawn up front with the same MWC generator and seed, all K EPnP hypotheses are solved and scored
// in parallel (the hypothesis-score loop is K x N_t independent projections, popcount-reduced), and
// one thread then replays the accept / shrink rule over the K inlier counts — the same winner the
// serial loop would pick.  f64 throughout; hypothesis + scoring use only IEEE-exact operations so
// the inlier mask is bit-reproducible; the final Levenberg–Marquardt refine runs in one workgroup.
#include "svo_internal.hpp"
#include "svo_linalg.hpp"

static __device__ __forceinline__ bool seq_live(const SeqState& s) { return s.active && s.fail_reason == 0; }
static __device__ int ransac_update_num_iters(double log_num, double ep, int max_iters);

// ------------------------------------------------------------------------------------------------ triangulation
// ---- the 4 x 4 SVD of the triangulation, register-resident.  jacobi_svd<4, 4> walks its row pairs with run-time indices; on
// arrays the compiler keeps in VGPRs every such access becomes a select chain (the pair loop body was ~800 VALU instructions,
// 430 of them v_cndmask).  Here the six pairs are six template instances with compile-time indices: the same rotations in the
// same order with the same arithmetic (bit-identical results), ~230 instructions per pair.  Only the right singular vector of
// the SMALLEST singular value is needed (the null vector of the DLT system), so the descending selection sort of the generic
// routine is replayed on the four values alone to learn which row it would have moved to the last place.
template <int I, int J>
static __device__ __forceinline__ bool svd4_rotate(double (&At)[4][4], double (&Wv)[4], double (&Vt)[4][4]) {
    const double eps = SVO_DBL_EPS * 10;
    double a = Wv[I], p = 0, b = Wv[J], c, s;
#pragma unroll
    for (int k = 0; k < 4; k++) p += At[I][k] * At[J][k];
    if (fabs(p) <= eps * sqrt(a * b)) return false;
    p *= 2;
    double beta = a - b, gamma = sqrt(p * p + beta * beta);
    jacobi_cs(p, beta, gamma, c, s);
    a = b = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double t0 = c * At[I][k] + s * At[J][k];
        double t1 = -s * At[I][k] + c * At[J][k];
        At[I][k] = t0; At[J][k] = t1;
        a += t0 * t0; b += t1 * t1;
    }
    Wv[I] = a; Wv[J] = b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double t0 = c * Vt[I][k] + s * Vt[J][k];
        double t1 = -s * Vt[I][k] + c * Vt[J][k];
        Vt[I][k] = t0; Vt[J][k] = t1;
    }
    return true;
}
// A: row-major 4 x 4.  X: the row of Vt that svd_rm<4, 4> would return as Vt[12..15].
static __device__ void svd4_null_vector(const double (&A)[4][4], double (&X)[4]) {
    double At[4][4], Vt[4][4], Wv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) { At[i][j] = A[j][i]; Vt[i][j] = i == j ? 1.0 : 0.0; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += At[i][k] * At[i][k];
        Wv[i] = sd;
    }
#pragma unroll 1
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        changed |= svd4_rotate<0, 1>(At, Wv, Vt); changed |= svd4_rotate<0, 2>(At, Wv, Vt); changed |= svd4_rotate<0, 3>(At, Wv, Vt);
        changed |= svd4_rotate<1, 2>(At, Wv, Vt); changed |= svd4_rotate<1, 3>(At, Wv, Vt); changed |= svd4_rotate<2, 3>(At, Wv, Vt);
        if (!changed) break;
    }
    double w[4]; int p[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += At[i][k] * At[i][k];
        w[i] = sqrt(sd); p[i] = i;
    }
    // the selection sort of jacobi_svd (descending; the FIRST maximum wins ties), replayed on (value, original row) pairs
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int j = i; double wj = w[i];
#pragma unroll
        for (int k = i + 1; k < 4; k++) { const bool g = wj < w[k]; j = g ? k : j; wj = g ? w[k] : wj; }
        const double wi = w[i]; const int pi = p[i]; int pj = p[i];
#pragma unroll
        for (int k = i + 1; k < 4; k++) { const bool e = j == k; pj = e ? p[k] : pj; w[k] = e ? wi : w[k]; p[k] = e ? pi : p[k]; }
        w[i] = wj; p[i] = pj;
    }
    const int r = p[3];
#pragma unroll
    for (int c = 0; c < 4; c++) X[c] = r == 0 ? Vt[0][c] : r == 1 ? Vt[1][c] : r == 2 ? Vt[2][c] : Vt[3][c];
}

// The last block of every sequence does not triangulate: its first lane draws the RANSAC subsets (they depend on the
// track count only), so the serial RNG walk hides under the triangulation instead of being a launch of its own.
// `lanes` tracks per wave: 64 when many sequences fill the GPU; 16 when a single stream runs alone — the Jacobi sweeps of a wave
// last as long as its slowest lane needs, so with the GPU nearly empty fewer tracks per wave shorten the kernel (same results).
// one track: the DLT system of the two views (vo.cpp:89-91), its null vector, dehomogenised to f32 (vo.cpp:93-94)
static __device__ __forceinline__ void triangulate_point(const SeqState& s, float2 pl, float2 pr, float (&w)[3]) {
    double A[4][4], V[4];
    const double xl = pl.x, yl = pl.y, xr = pr.x, yr = pr.y;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A[0][k] = xl * (double)s.Pl[8 + k] - (double)s.Pl[0 + k];
        A[1][k] = yl * (double)s.Pl[8 + k] - (double)s.Pl[4 + k];
        A[2][k] = xr * (double)s.Pr[8 + k] - (double)s.Pr[0 + k];
        A[3][k] = yr * (double)s.Pr[8 + k] - (double)s.Pr[4 + k];
    }
    svd4_null_vector(A, V);
    const float X = (float)V[0], Y = (float)V[1], Z = (float)V[2], Wh = (float)V[3];           // 4xN result is CV_32F
    const float scale = Wh != 0.f ? 1.f / Wh : 1.f;                                             // convertPointsFromHomogeneous
    w[0] = X * scale; w[1] = Y * scale; w[2] = Z * scale;
}
// spare: the last block (bx == nblocks - 1) draws the RANSAC subsets instead of triangulating (many-sequence contexts; a
// lone stream's k_compact has drawn them already, and its triangulation shares a launch with the first EPnP chunk, k_tri_epnp)
static __device__ __forceinline__ void triangulate_body(const DevBuffers& d, int lanes, int bx, int seq, int nblocks, bool spare) {
    SeqState& s = d.st[seq];
    if (!seq_live(s)) return;
    if (spare && bx == nblocks - 1) {
        if (threadIdx.x == 0) pnp_draw_subsets(d, s, seq, pnp_first_chunk(d));
        return;
    }
    if ((int)threadIdx.x >= lanes) return;
    const int i = bx * lanes + threadIdx.x;
    if (i >= s.n_tracks) return;
    const size_t o = (size_t)seq * d.CAP + i;
    float w[3];
    triangulate_point(s, d.tl0[o], d.tr0[o], w);
    d.world[3 * o] = w[0]; d.world[3 * o + 1] = w[1]; d.world[3 * o + 2] = w[2];
}
// Two builds (here and for EPnP and the final refine below).  An LK wave holds 104 registers and four of them share a SIMD: 96
// registers stay free.  A kernel that needs more cannot start beside the LK grid of another context — it waits until that grid
// drains (traced: k_pnp_epnp 7.5 ms instead of 0.2, k_pnp_final 8.3 ms instead of 0.1) and everything behind it in the stream
// with it.  The `_lean` builds are capped at 96 registers (amdgpu_num_vgpr counts VGPR + AGPR pairs on gfx950) and spill to
// scratch: slower alone (32 sequences, one context: EPnP 202 -> 286 us, final 90 -> 362 us), but they run under the other
// context's LK instead of after it.  Used when several many-sequence contexts share the device (DevBuffers::co_resident).
__global__ __launch_bounds__(64) void k_triangulate(DevBuffers d, int lanes) { triangulate_body(d, lanes, blockIdx.x, blockIdx.y, gridDim.x, true); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(48))) void k_triangulate_lean(DevBuffers d, int lanes) { triangulate_body(d, lanes, blockIdx.x, blockIdx.y, gridDim.x, true); }
void launch_triangulate(const DevBuffers& d, hipStream_t st) {
    const bool lean = d.co_resident;
    const int lanes = d.B > SVO_LONE_MAX_SEQ ? 64 : 16;
    if (lean) hipLaunchKernelGGL(k_triangulate_lean, dim3((d.CAP + lanes - 1) / lanes + 1, d.B), dim3(64), 0, st, d, lanes);
    else hipLaunchKernelGGL(k_triangulate, dim3((d.CAP + lanes - 1) / lanes + 1, d.B), dim3(64), 0, st, d, lanes);
}

// (pnp_draw_subsets — cv::RNG + getSubset — lives in svo_internal.hpp: k_compact draws the first chunk for lone-stream contexts)
// stand-alone launch for callers that enter at launch_pnp without a triangulation before it (svo_camera_to_world)
__global__ void k_pnp_subsets(DevBuffers d) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= d.B) return;
    SeqState& s = d.st[seq];
    if (!seq_live(s)) return;
    pnp_draw_subsets(d, s, seq, pnp_first_chunk(d));
}
void launch_pnp_subsets(const DevBuffers& d, hipStream_t st) {
    hipLaunchKernelGGL(k_pnp_subsets, dim3((d.B + 63) / 64), dim3(64), 0, st, d);
}

// ------------------------------------------------------------------------------------------------ EPnP on 5 points
// 8 or 32 lanes cooperate on one hypothesis (8 or 2 hypotheses per 64-thread block); the matrices live in an LDS arena.
// The 12x12 one-sided Jacobi SVD of MtM — more than half of EPnP's time — sweeps its row pairs in round-robin order: the 6
// pairs of a round are disjoint, so they rotate in parallel (one lane per pair, or four: rotate_pair12_quads) and the result is
// bit-identical to the sequential round-robin loop.  The three beta approximations (N = 4, 2, 3 null vectors) are independent
// after L and rho and run on lanes 0..2.
#define EP_G 8                                   // lanes per hypothesis, many sequences (k_pnp_epnp_lean); 64 / EP_G hypotheses per block
#define EP_LEAN_LDS ((64 / EP_G) * EP_STRIDE * sizeof(double))
#define EP_G_LONE 16                             // lanes per hypothesis when few sequences run (k_pnp_epnp): see rotate_pair12_halves
#define EP_STRIDE 1032                           // doubles per hypothesis (+8 pad: distinct LDS banks per hypothesis)
// arena map (doubles)
#define EA_AT 0                                  // 144  MtM, then the rotating rows
#define EA_VT 144                                // 144  rows 8..11: the four basis vectors EPnP reads (normalised rows of At, moved here after the sort)
#define EA_W 288                                 // 12
#define EA_M 300                                 // 120  (dead after MtM is built)
#define EA_PWS 420                               // 15
#define EA_US 435                                // 10
#define EA_AL 445                                // 20
#define EA_CWS 465                               // 12
#define EA_L 480                                 // 60
#define EA_RHO 540                               // 6
#define EA_RES 552                               // 3 x 13: rep, R[9], t[3] of the three branches
#define EA_BR 600                                // 3 x 140 branch workspaces
#define EA_BRSZ 140

static __device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static __device__ __forceinline__ double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// one Hestenes rotation of rows i < j (12 x 12); returns true if the pair was rotated.  Only the rows of At rotate: EPnP reads
// the LEFT singular vectors (the normalised rows of the rotated At, as cvSVD(&MtM, &D, &Ut, 0, CV_SVD_MODIFY_A | CV_SVD_U_T)
// returns them, epnp.cpp), so no Vt is accumulated (round 3; measured against the reference's recording)
static __device__ bool rotate_pair12(double* At, double* Wv, int i, int j) {
    const double eps = SVO_DBL_EPS * 10;
    double* Ai = At + i * 12; double* Aj = At + j * 12;
    double a = Wv[i], p = 0, b = Wv[j], c, s;
    for (int k = 0; k < 12; k++) p += Ai[k] * Aj[k];
    if (fabs(p) <= eps * sqrt(a * b)) return false;
    p *= 2;
    double beta = a - b, gamma = sqrt(p * p + beta * beta);
    jacobi_cs(p, beta, gamma, c, s);
    a = b = 0;
    for (int k = 0; k < 12; k++) {
        double t0 = c * Ai[k] + s * Aj[k];
        double t1 = -s * Ai[k] + c * Aj[k];
        Ai[k] = t0; Aj[k] = t1;
        a += t0 * t0; b += t1 * t1;
    }
    Wv[i] = a; Wv[j] = b;
    return true;
}

// the value a DPP control brings from another lane, for a double (DPP moves 32-bit registers: the halves go separately)
template <int CTRL> static __device__ __forceinline__ double dpp_move_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// The rotation of one pair on TWO lanes (16 lanes per hypothesis), for a lone stream: the twelve elements of the two rows are
// split over neighbouring lanes (two HALVES), one instruction stream; every read of At precedes, in program order, the first
// write of the rotated rows (the lanes of a pair sit in one wave).  A lone wave per SIMD hides no latency and the compiler orders
// for register pressure, so the order is pinned with sched_barrier: loads, the row-norm square root under them, independent
// products, then the rotation software-pipelined (products of element k, sums of k-1, squares of k-2, norms of k-3 per step).
// Every sum keeps its element order 0..11: the lower half sums its six terms from zero, hands the partial to its neighbour
// (DPP row_shr:1), which continues with elements 6..11 — bit for bit the sequential sum — and, for p, hands the total back
// (row_shl:1).  (Round 2 ran four lanes per pair: a second SIDE rotated the rows of Vt; EPnP reads the left singular vectors
// since round 3, so that side is gone.)
static __device__ bool rotate_pair12_halves(double* At, double* Wv, int i, int j, int half) {
    const double eps = SVO_DBL_EPS * 10;
    const int oi = __mul24(i, 12) + half * 6, oj = __mul24(j, 12) + half * 6;
    double* X = At + oi; double* Y = At + oj;
    double a = Wv[i], b = Wv[j], c, s;
    double x[6], y[6], pr[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { x[k] = X[k]; y[k] = Y[k]; }
    __builtin_amdgcn_sched_barrier(0);
    const double lim = eps * sqrt(a * b);                             // while the rows arrive from LDS
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 6; k++) pr[k] = x[k] * y[k];
    __builtin_amdgcn_sched_barrier(0);
    double plo = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) plo += pr[k];                         // lower half: elements 0..5 from zero
    double pup = dpp_move_f64<0x111>(plo);                            // row_shr:1 -> the upper half continues its neighbour's partial
#pragma unroll
    for (int k = 0; k < 6; k++) pup += pr[k];
    const double pdn = dpp_move_f64<0x101>(pup);                      // row_shl:1 -> and hands the total back
    double p = half ? pup : pdn;
    if (fabs(p) <= lim) return false;
    p *= 2;
    double beta = a - b, gamma = sqrt(p * p + beta * beta);
    jacobi_cs(p, beta, gamma, c, s);
    double ma[6], mb[6], mc[6], md[6], t0[6], t1[6], q0[6], q1[6];
    double alo = 0, blo = 0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 9; st++) {
        if (st < 6) { ma[st] = c * x[st]; mb[st] = s * y[st]; mc[st] = -s * x[st]; md[st] = c * y[st]; }
        if (st >= 1 && st < 7) { t0[st - 1] = ma[st - 1] + mb[st - 1]; t1[st - 1] = mc[st - 1] + md[st - 1]; }
        if (st >= 2 && st < 8) { q0[st - 2] = t0[st - 2] * t0[st - 2]; q1[st - 2] = t1[st - 2] * t1[st - 2]; }
        if (st >= 3) { alo += q0[st - 3]; blo += q1[st - 3]; }
        if (st >= 2 && st < 8 && ((st - 2) & 1)) {
            X[st - 3] = t0[st - 3]; X[st - 2] = t0[st - 2]; Y[st - 3] = t1[st - 3]; Y[st - 2] = t1[st - 2];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    double aup = dpp_move_f64<0x111>(alo), bup = dpp_move_f64<0x111>(blo);
#pragma unroll
    for (int k = 0; k < 6; k++) { aup += q0[k]; bup += q1[k]; }
    double* Wo = half ? Wv : At + EA_M;                               // the upper lane holds the norms; the lower drops its partial in dead space
    Wo[i] = aup; Wo[j] = bup;
    return true;
}

// x = pinv(A) b for A: 6 x n (n = 3, 4, 5) on REGISTERS.  The three beta approximations of EPnP solve 6x4, 6x3 and 6x5 systems on
// three lanes of one wave — as three template instances they were three code paths the wave executed one after the other; here
// the system is zero-padded to five columns so that they run one instruction stream.  A zero row of At never rotates (p = 0 <= eps sqrt(a b) = 0), keeps its singular
// value 0 (<= the threshold: dropped) and leaves the identity columns of Vt alone, so the n x n part is computed exactly as the
// unpadded routine computes it, bit for bit.
static __device__ void svd_solve6_reg(const double* A, int n, const double* b, double* x) {
    constexpr int M = 6, N = 5;
    double At[N][M], Wv[N], Vt[N][N];
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < M; j++) At[i][j] = i < n ? A[j * n + i] : 0.0;
    }
    jacobi_svd_reg<M, N>(At, Wv, Vt, true);
    double thr = 0;
#pragma unroll
    for (int i = 0; i < N; i++) thr += Wv[i];
    thr *= SVO_DBL_EPS * 2;
    double xx[N] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (Wv[i] <= thr) continue;
        double s = 0;
#pragma unroll
        for (int k = 0; k < M; k++) s += At[i][k] * b[k];
        s /= Wv[i];
#pragma unroll
        for (int k = 0; k < N; k++) xx[k] += s * Vt[i][k];
    }
#pragma unroll
    for (int k = 0; k < N; k++) if (k < n) x[k] = xx[k];
}

// Householder QR least squares for the 6 x 4 Gauss-Newton step, on registers with compile-time indices (the arithmetic and its
// order are those of qr_solve<6, 4> in svo_linalg.hpp)
static __device__ __forceinline__ bool qr_solve64_reg(double (&A)[6][4], double (&b)[6], double (&x)[4]) {
    constexpr int M = 6, N = 4;
    double A1[N], A2[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        double eta = 0;
#pragma unroll
        for (int i = k; i < M; i++) { double e = fabs(A[i][k]); if (eta < e) eta = e; }
        if (eta == 0) return false;
        double sum2 = 0, inv_eta = 1. / eta;
#pragma unroll
        for (int i = k; i < M; i++) { A[i][k] *= inv_eta; sum2 += A[i][k] * A[i][k]; }
        double sigma = sqrt(sum2);
        if (A[k][k] < 0) sigma = -sigma;
        A[k][k] += sigma;
        A1[k] = sigma * A[k][k];
        A2[k] = -eta * sigma;
#pragma unroll
        for (int j = k + 1; j < N; j++) {
            double sum = 0;
#pragma unroll
            for (int i = k; i < M; i++) sum += A[i][k] * A[i][j];
            double tau = sum / A1[k];
#pragma unroll
            for (int i = k; i < M; i++) A[i][j] -= tau * A[i][k];
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        double tau = 0;
#pragma unroll
        for (int i = j; i < M; i++) tau += A[i][j] * b[i];
        tau /= A1[j];
#pragma unroll
        for (int i = j; i < M; i++) b[i] -= tau * A[i][j];
    }
    x[N - 1] = b[N - 1] / A2[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; i--) {
        double sum = 0;
#pragma unroll
        for (int j = i + 1; j < N; j++) sum += A[i][j] * x[j];
        x[i] = (b[i] - sum) / A2[i];
    }
    return true;
}

static __device__ void epnp_gauss_newton(const double* L, const double* rho, double* betas, double* ws) {
    (void)ws;
    double bt[4] = {betas[0], betas[1], betas[2], betas[3]};
#pragma unroll 1
    for (int it = 0; it < 5; it++) {
        const double b0 = bt[0], b1 = bt[1], b2 = bt[2], b3 = bt[3];
        double A[6][4], B[6], X[4];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double* rl = L + 10 * i;
            A[i][0] = 2 * rl[0] * b0 + rl[1] * b1 + rl[3] * b2 + rl[6] * b3;
            A[i][1] = rl[1] * b0 + 2 * rl[2] * b1 + rl[4] * b2 + rl[7] * b3;
            A[i][2] = rl[3] * b0 + rl[4] * b1 + 2 * rl[5] * b2 + rl[8] * b3;
            A[i][3] = rl[6] * b0 + rl[7] * b1 + rl[8] * b2 + 2 * rl[9] * b3;
            B[i] = rho[i] - (rl[0] * b0 * b0 + rl[1] * b0 * b1 + rl[2] * b1 * b1 +
                             rl[3] * b0 * b2 + rl[4] * b1 * b2 + rl[5] * b2 * b2 +
                             rl[6] * b0 * b3 + rl[7] * b1 * b3 + rl[8] * b2 * b3 +
                             rl[9] * b3 * b3);
        }
        if (!qr_solve64_reg(A, B, X)) break;
#pragma unroll
        for (int i = 0; i < 4; i++) bt[i] += X[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) betas[i] = bt[i];
}

// compute_ccs, compute_pcs, solve_for_sign, estimate_R_and_t (Arun / Horn), reprojection_error.  ws: >= 66 doubles.
static __device__ double epnp_compute_R_and_t(const double* ar, const double* betas, double* R, double* t, double* ws,
                                              double fu, double fv, double uc, double vc) {
    const int n = 5;
    const double* vt = ar + EA_VT; const double* alphas = ar + EA_AL; const double* pws = ar + EA_PWS; const double* us = ar + EA_US;
    double* ccs = ws; double* pcs = ws + 12; double* abt = ws + 27; double* Wv = ws + 36; double* Ut = ws + 39; double* Vt = ws + 48;
    double* pc0 = ws + 57; double* pw0 = ws + 60;
    for (int i = 0; i < 12; i++) ccs[i] = 0.0;
    for (int i = 0; i < 4; i++) {
        const double* v = vt + 12 * (11 - i);
        const double be = betas[i];
        for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) ccs[3 * j + k] += be * v[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
        const double* a = alphas + 4 * i; double* pc = pcs + 3 * i;
        for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[j] + a[1] * ccs[3 + j] + a[2] * ccs[6 + j] + a[3] * ccs[9 + j];
    }
    if (pcs[2] < 0.0) {
        for (int i = 0; i < 12; i++) ccs[i] = -ccs[i];
        for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
    }
    for (int j = 0; j < 3; j++) { pc0[j] = 0; pw0[j] = 0; }
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    for (int i = 0; i < 9; i++) abt[i] = 0;
    for (int i = 0; i < n; i++) {
        const double* pc = pcs + 3 * i; const double* pw = pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    svd_rm<3, 3>(abt, Wv, Ut, Vt);                                     // register form (svo_linalg.hpp)
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += Ut[k * 3 + i] * Vt[k * 3 + j];
        R[i * 3 + j] = s;
    }
    double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7]
               - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int i = 0; i < 3; i++) t[i] = pc0[i] - dot3(R + 3 * i, pw0);
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
        const double* pw = pws + 3 * i;
        double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
        double u = us[2 * i], v = us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

// phase 1 (one lane): control points, barycentric coordinates, M, MtM -> arena; W = squared row norms
static __device__ __attribute__((always_inline)) void epnp_setup(double* ar, double fu, double fv, double uc, double vc) {
    const int n = 5;
    double* pws = ar + EA_PWS; double* us = ar + EA_US; double* alphas = ar + EA_AL; double* cws = ar + EA_CWS;
    for (int j = 0; j < 3; j++) cws[j] = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) cws[j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[j] /= n;
    {
        double ptp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dc[3], ut[9], vt3[9];
        for (int i = 0; i < n; i++) {
            double dd[3];
            for (int j = 0; j < 3; j++) dd[j] = pws[3 * i + j] - cws[j];
            for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) ptp[3 * j + k] += dd[j] * dd[k];
        }
        svd_rm<3, 3>(ptp, dc, ut, vt3);
        for (int i = 1; i < 4; i++) {
            double kk = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[3 * i + j] = cws[j] + kk * vt3[3 * (i - 1) + j];
        }
    }
    {
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[3 * j + i] - cws[i];
        inv3_svd(cc, ci);
        for (int i = 0; i < n; i++) {
            const double* pi = pws + 3 * i; double* a = alphas + 4 * i;
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0]) + ci[3 * j + 1] * (pi[1] - cws[1]) + ci[3 * j + 2] * (pi[2] - cws[2]);
            a[0] = 1.0 - a[1] - a[2] - a[3];
        }
    }
    double* M = ar + EA_M;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        double* M1 = M + 24 * i; double* M2 = M1 + 12; const double* as = alphas + 4 * i;
        double u = us[2 * i], v = us[2 * i + 1];
        for (int j = 0; j < 4; j++) {
            M1[3 * j] = as[j] * fu; M1[3 * j + 1] = 0.0; M1[3 * j + 2] = as[j] * (uc - u);
            M2[3 * j] = 0.0; M2[3 * j + 1] = as[j] * fv; M2[3 * j + 2] = as[j] * (vc - v);
        }
    }
}

// phase 1b (one lane per row i): row i of MtM = M^T M (each entry summed over the 10 rows of M in order; the products commute, so
// entry (i, j) and entry (j, i) are the same bits: MtM is exactly symmetric, i.e. its own transpose, and the one-sided Jacobi
// runs in place on it) and its squared norm.
static __device__ void epnp_setup_row(double* ar, int i) {
    const double* M = ar + EA_M; double* MtM = ar + EA_AT; double* Wv = ar + EA_W;
    double mi[10];
#pragma unroll
    for (int k = 0; k < 10; k++) mi[k] = M[12 * k + i];
    double sd = 0;
#pragma unroll 2
    for (int j = 0; j < 12; j++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 10; k++) s += M[12 * k + j] * mi[k];
        MtM[12 * i + j] = s;
        sd += s * s;
    }
    Wv[i] = sd;
}

// phase 3, spread over the lanes of the hypothesis (the caller separates the steps with barriers):
//   a (lane per row)   singular values = row norms of the rotated MtM
//   b (one lane)       the descending selection sort (the FIRST maximum wins ties), on (value, row) pairs
//   c (lanes 0..3)     the four left singular vectors everything after this reads — the sorted rows 11, 10, 9, 8 of At times
//                      1 / singular value (JacobiSVDImpl_'s final normalisation) — move to rows 11..8 of the EA_VT block
//   d (lanes 0..3)     differences of the control points of null vector i;   e (lanes 0..5 + one) L (6x10) row by row, rho
static __device__ void epnp_row_norm(double* ar, int i) {
    const double* At = ar + EA_AT;
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) { double tt = At[i * 12 + k]; sd += tt * tt; }
    ar[EA_W + i] = sqrt(sd);
}
static __device__ void epnp_sort_rows(double* ar) {
    double w[12]; int id[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { w[i] = ar[EA_W + i]; id[i] = i; }
#pragma unroll
    for (int i = 0; i < 11; i++) {
        // position i takes the first maximum of w[i..11]; the displaced pair goes where the maximum was
        double best = w[i]; int bj = i;
#pragma unroll
        for (int k = i + 1; k < 12; k++) if (best < w[k]) { best = w[k]; bj = k; }
        const double wi = w[i]; const int ii = id[i];
        int bid = ii;
#pragma unroll
        for (int k = i + 1; k < 12; k++) if (k == bj) { bid = id[k]; w[k] = wi; id[k] = ii; }
        w[i] = best; id[i] = bid;
    }
    int* src = (int*)(ar + EA_M);                                     // M is dead
    src[0] = id[11]; src[1] = id[10]; src[2] = id[9]; src[3] = id[8];
}
static __device__ void epnp_null_vector_diffs(double* ar, int i) {
    const double* v = ar + EA_VT + 12 * (11 - i);
    double* dv = ar + EA_AT;                                          // 4 x 6 x 3 scratch in the dead At region
    int a = 0, b = 1;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        for (int k = 0; k < 3; k++) dv[(i * 6 + j) * 3 + k] = v[3 * a + k] - v[3 * b + k];
        b++;
        if (b > 3) { a++; b = a + 1; }
    }
}
static __device__ void epnp_L_row(double* ar, int i) {
    const double* dv = ar + EA_AT;
    double* row = ar + EA_L + 10 * i;
    const double* d0 = dv + (0 * 6 + i) * 3; const double* d1 = dv + (1 * 6 + i) * 3;
    const double* d2 = dv + (2 * 6 + i) * 3; const double* d3 = dv + (3 * 6 + i) * 3;
    row[0] = dot3(d0, d0);
    row[1] = 2.0 * dot3(d0, d1);
    row[2] = dot3(d1, d1);
    row[3] = 2.0 * dot3(d0, d2);
    row[4] = 2.0 * dot3(d1, d2);
    row[5] = dot3(d2, d2);
    row[6] = 2.0 * dot3(d0, d3);
    row[7] = 2.0 * dot3(d1, d3);
    row[8] = 2.0 * dot3(d2, d3);
    row[9] = dot3(d3, d3);
}
static __device__ void epnp_rho(double* ar) {
    const double* cws = ar + EA_CWS; double* rho = ar + EA_RHO;
    rho[0] = dist2(cws, cws + 3); rho[1] = dist2(cws, cws + 6); rho[2] = dist2(cws, cws + 9);
    rho[3] = dist2(cws + 3, cws + 6); rho[4] = dist2(cws + 3, cws + 9); rho[5] = dist2(cws + 6, cws + 9);
}

// phase 4 (lanes 0..2): beta approximation `branch` (1: N=4 null vectors, 2: N=2, 3: N=3), Gauss-Newton, R, t, error
static __device__ __attribute__((always_inline)) void epnp_branch(double* ar, int branch, double fu, double fv, double uc, double vc) {
    const double* L = ar + EA_L; const double* rho = ar + EA_RHO;
    double* ws = ar + EA_BR + (branch - 1) * EA_BRSZ;
    double* Lx = ws + 70; double* bx = ws + 100; double* be = ws + 106;       // svd workspace occupies ws[0, 60)
    // the columns of L each approximation keeps (epnp.cpp find_betas_approx_1 / _2 / _3) and how many
    const int N = branch == 1 ? 4 : branch == 2 ? 3 : 5;
    const int c3 = branch == 1 ? 6 : 3, c2 = branch == 1 ? 3 : 2;
    for (int i = 0; i < 6; i++) {
        Lx[N * i] = L[10 * i]; Lx[N * i + 1] = L[10 * i + 1]; Lx[N * i + 2] = L[10 * i + c2];
        if (N > 3) Lx[N * i + 3] = L[10 * i + c3];
        if (N > 4) Lx[N * i + 4] = L[10 * i + 4];
    }
    svd_solve6_reg(Lx, N, rho, bx);                                   // ONE code path for the three lanes, on registers
    if (branch == 1) {
        if (bx[0] < 0) { be[0] = sqrt(-bx[0]); be[1] = -bx[1] / be[0]; be[2] = -bx[2] / be[0]; be[3] = -bx[3] / be[0]; }
        else { be[0] = sqrt(bx[0]); be[1] = bx[1] / be[0]; be[2] = bx[2] / be[0]; be[3] = bx[3] / be[0]; }
    } else if (branch == 2) {
        if (bx[0] < 0) { be[0] = sqrt(-bx[0]); be[1] = (bx[2] < 0) ? sqrt(-bx[2]) : 0.0; }
        else { be[0] = sqrt(bx[0]); be[1] = (bx[2] > 0) ? sqrt(bx[2]) : 0.0; }
        if (bx[1] < 0) be[0] = -be[0];
        be[2] = 0.0; be[3] = 0.0;
    } else {
        if (bx[0] < 0) { be[0] = sqrt(-bx[0]); be[1] = (bx[2] < 0) ? sqrt(-bx[2]) : 0.0; }
        else { be[0] = sqrt(bx[0]); be[1] = (bx[2] > 0) ? sqrt(bx[2]) : 0.0; }
        if (bx[1] < 0) be[0] = -be[0];
        be[2] = bx[3] / be[0];
        be[3] = 0.0;
    }
    epnp_gauss_newton(L, rho, be, ws);
    double* res = ar + EA_RES + (branch - 1) * 13;
    res[0] = epnp_compute_R_and_t(ar, be, res + 1, res + 10, ws, fu, fv, uc, vc);
}

// Hypotheses [h0, h1).  The first chunk (h0 == 0) is always solved; later chunks only up to s.pnp_need, the bound the
// adaptive loop had reached after the first chunk (the bound only ever shrinks, so nothing beyond it can be consulted).
// OWN_TRI: the hypothesis triangulates its five points itself (lanes 0..4, the same triangulate_point, hence the same floats)
// instead of reading d.world — so that the first chunk can share a launch with the triangulation of all tracks (k_tri_epnp).
template <int G, bool OWN_TRI = false>    // lanes per hypothesis: 8 (six rotate a pair each) or 16 (twelve: two lanes per pair)
static __device__ __forceinline__ void pnp_epnp_body(const DevBuffers& d, int h0, int h1, double* arena, int bx, int seq) {
    constexpr int HPB = 64 / G;
    const SeqState& s = d.st[seq];
    if (!seq_live(s)) return;
    int hend = h0 > 0 ? (s.pnp_need < h1 ? s.pnp_need : h1) : h1;
    if (s.n_tracks == 5 && hend > 1) hend = 1;                          // five points: a single direct EPnP, no RANSAC
    if (h0 + bx * HPB >= hend) return;                                  // block-uniform
    const int g = threadIdx.x / G, q = threadIdx.x % G;
    const int h = h0 + bx * HPB + g;
    const bool valid = h < hend;
    double* ar = arena + g * EP_STRIDE;
    const double fx = s.K[0], fy = s.K[4], cx = s.K[2], cy = s.K[5];
    if (valid && q < 5) {                                               // one lane per point of the subset
        const size_t o = (size_t)seq * d.CAP;
        const int k = d.subsets[((size_t)seq * d.K + h) * 5 + q];
        const double ifx = 1. / fx, ify = 1. / fy;
        float w[3];
        if (OWN_TRI) triangulate_point(s, d.tl0[o + k], d.tr0[o + k], w);
        else { w[0] = d.world[3 * (o + k)]; w[1] = d.world[3 * (o + k) + 1]; w[2] = d.world[3 * (o + k) + 2]; }
        ar[EA_PWS + 3 * q] = w[0]; ar[EA_PWS + 3 * q + 1] = w[1]; ar[EA_PWS + 3 * q + 2] = w[2];
        const float2 c = d.tl1[o + k];
        // undistortPoints on CV_32FC2 with zero distortion (normalise in f64, store f32), then epnp re-applies fu, uc
        const float xn = (float)(((double)c.x - cx) * ifx), yn = (float)(((double)c.y - cy) * ify);
        ar[EA_US + 2 * q] = (double)xn * fx + cx; ar[EA_US + 2 * q + 1] = (double)yn * fy + cy;
    }
    __syncthreads();
    if (valid && q == 0) epnp_setup(ar, fx, fy, cx, cy);
    __syncthreads();
    if (valid) for (int i = q; i < 12; i += G) epnp_setup_row(ar, i);
    __syncthreads();
    // ---- 12 x 12 one-sided Jacobi, round-robin ordering: lane q < 6 owns pair q of every round
    bool done = !valid;
    const int qp = G == 16 ? (q >> 1) : q;                              // the pair of the round this lane works on (G = 16: lanes 12..15 idle)
    // round r pairs row pa = 1 + (qp - 1 + r) % 11 (row 0 for pair 0) with row pb = 1 + (10 - qp + r) % 11: both walk the
    // cycle 1..11 one step per round and are back where they started after the 11 rounds of a sweep
    int pa = qp, pb = 11 - qp;
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        for (int r = 0; r < 11; r++) {
            const int i = pa < pb ? pa : pb, j = pa < pb ? pb : pa;
            pa = qp == 0 ? 0 : pa == 11 ? 1 : pa + 1;
            pb = pb == 11 ? 1 : pb + 1;
            if (!done && qp < 6) {
                if (G == 16) changed |= rotate_pair12_halves(ar + EA_AT, ar + EA_W, i, j, q & 1);
                else changed |= rotate_pair12(ar + EA_AT, ar + EA_W, i, j);
            }
            __syncthreads();
        }
        const unsigned long long m = __ballot(changed);
        if (((m >> (g * G)) & ((1ull << G) - 1)) == 0) done = true;            // this hypothesis converged (no pair rotated in the sweep)
        if (__ballot(!done) == 0ull) break;
    }
    __syncthreads();
    if (valid) for (int i = q; i < 12; i += G) epnp_row_norm(ar, i);
    __syncthreads();
    if (valid && q == 0) epnp_sort_rows(ar);
    __syncthreads();
    {
        double row[12];
        const bool mover = valid && q < 4;
        if (mover) {
            const int r = ((const int*)(ar + EA_M))[q];
            const double* sp = ar + EA_AT + 12 * r;
            const double sd = ar[EA_W + r];
            const double sc = sd > SVO_DBL_MIN ? 1 / sd : 0.;           // lapack.cpp JacobiSVDImpl_: s = sd > minval ? 1/sd : 0
#pragma unroll
            for (int k = 0; k < 12; k++) row[k] = sp[k] * sc; }
        __syncthreads();
        if (mover) { double* dp = ar + EA_VT + 12 * (11 - q);
#pragma unroll
            for (int k = 0; k < 12; k++) dp[k] = row[k]; }
    }
    __syncthreads();
    if (valid && q < 4) epnp_null_vector_diffs(ar, q);
    __syncthreads();
    if (valid && q < 6) epnp_L_row(ar, q);
    if (valid && q == 6) epnp_rho(ar);
    __syncthreads();
    if (valid && q < 3) epnp_branch(ar, q + 1, fx, fy, cx, cy);
    __syncthreads();
    if (valid && q == 0) {
        const double* res = ar + EA_RES;
        int N = 1;
        if (res[13] < res[0]) N = 2;
        if (res[26] < res[(N - 1) * 13]) N = 3;
        const double* w = res + (N - 1) * 13;
        double* out = d.hyp + ((size_t)seq * d.K + h) * 12;
        for (int i = 0; i < 12; i++) out[i] = w[1 + i];
    }
}

// Two builds (see k_triangulate_lean): the register-resident linear algebra wants ~300 registers, 32 lanes per hypothesis; the
// lean build (96 registers, 8 lanes per hypothesis, arena in dynamic LDS — with a static 66 KB arena the compiler ties the
// register budget to the occupancy the LDS allows and ignores the cap) is for contexts that share the GPU with another's LK.
__global__ __launch_bounds__(64) void k_pnp_epnp(DevBuffers d, int h0, int h1) {
    __shared__ double arena[(64 / EP_G_LONE) * EP_STRIDE];
    pnp_epnp_body<EP_G_LONE>(d, h0, h1, arena, blockIdx.x, blockIdx.y);
}
// Lone stream: the triangulation of all tracks and the FIRST chunk of EPnP hypotheses in one launch.  On a nearly empty GPU the
// two ran one after the other (36 + 140 us); a hypothesis needs the world points of its own five tracks only, which its lanes
// 0..4 now compute themselves, so the hypothesis blocks start at once and the triangulation blocks (whose output the SCORING
// needs, one launch later) run beside them.  The subsets were drawn by k_compact.  blockIdx.x < n_e: hypothesis block.
__global__ __launch_bounds__(64) void k_tri_epnp(DevBuffers d, int lanes, int n_e, int c0) {
    __shared__ double arena[(64 / EP_G_LONE) * EP_STRIDE];
    if ((int)blockIdx.x < n_e) pnp_epnp_body<EP_G_LONE, true>(d, 0, c0, arena, blockIdx.x, blockIdx.y);
    else triangulate_body(d, lanes, (int)blockIdx.x - n_e, blockIdx.y, 0, false);
}
extern __shared__ double epnp_arena_dyn[];                         // (64 / EP_G) * EP_STRIDE doubles, given at launch
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(48))) void k_pnp_epnp_lean(DevBuffers d, int h0, int h1) {
    pnp_epnp_body<EP_G>(d, h0, h1, epnp_arena_dyn, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------ hypothesis scoring
static __device__ __forceinline__ bool point_is_inlier(const double* Rt, double fx, double fy, double cx, double cy,
                                                       const float* w3, float2 c, float thr2) {
    double X = w3[0], Y = w3[1], Z = w3[2];
    double x = Rt[0] * X + Rt[1] * Y + Rt[2] * Z + Rt[9];
    double y = Rt[3] * X + Rt[4] * Y + Rt[5] * Z + Rt[10];
    double z = Rt[6] * X + Rt[7] * Y + Rt[8] * Z + Rt[11];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    float pu = (float)(x * fx + cx), pv = (float)(y * fy + cy);       // projectPoints: f64 inside, f32 out
    float du = c.x - pu, dv = c.y - pv;
    float e = du * du + dv * dv;                                      // computeError, f32
    return e <= thr2;
}

__global__ __launch_bounds__(256) void k_pnp_score(DevBuffers d, int h0, int h1) {
    const int seq = blockIdx.y, h = h0 + blockIdx.x;
    const SeqState& s = d.st[seq];
    if (!seq_live(s)) return;
    if (h >= (h0 > 0 ? (s.pnp_need < h1 ? s.pnp_need : h1) : h1)) return;
    if (s.n_tracks == 5) return;                                        // direct solve: nothing is scored
    __shared__ int total;
    __shared__ double Rt[12];
    if (threadIdx.x == 0) total = 0;
    if (threadIdx.x < 12) Rt[threadIdx.x] = d.hyp[((size_t)seq * d.K + h) * 12 + threadIdx.x];
    __syncthreads();
    const size_t o = (size_t)seq * d.CAP;
    const double fx = s.K[0], fy = s.K[4], cx = s.K[2], cy = s.K[5];
    const double thr = (double)d.cfg.ransac_reprojection_error;
    const float thr2 = (float)(thr * thr);
    int cnt = 0;
    for (int i0 = 0; i0 < s.n_tracks; i0 += 256) {
        int i = i0 + threadIdx.x;
        bool in = false;
        if (i < s.n_tracks) in = point_is_inlier(Rt, fx, fy, cx, cy, d.world + 3 * (o + i), d.tl1[o + i], thr2);
        cnt += __popcll(__ballot(in));                                // wave-level popcount of the inlier ballot
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0) d.hyp_good[(size_t)seq * d.K + h] = total;
}

// ------------------------------------------------------------------------------------------------ replay + refine + pose
// RANSACUpdateNumIters(confidence, ep, modelPoints = 5, maxIters) of ptsetreg.cpp.  log(1 - confidence) is constant per context
// and comes from the host; (1 - ep)^5 is three multiplications instead of a generic f64 pow (a few hundred instructions on the one
// thread everything waits for) — within 2 ulp of it, and the result only enters through rint(num / denom).
static __device__ int ransac_update_num_iters(double log_num, double ep, int max_iters) {
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    const double x = 1. - ep, x2 = x * x;
    double denom = 1. - x2 * x2 * x;
    if (denom < SVO_DBL_MIN) return 0;
    denom = log(denom);
    return denom >= 0 || -log_num >= max_iters * (-denom) ? max_iters : (int)rint(log_num / denom);
}

// Cholesky solve of a 6x6 SPD system (the damped normal equations of the LM step)
static __device__ void chol_solve6(const double* A, const double* b, double* x) {
    // every index below is a compile-time constant: registers.  One reciprocal per pivot (6 divisions instead of 27: each f64
    // division is ~25 dependent instructions on the thread the whole block waits for); the LM refine is compared with the oracle
    // to the pose tolerance, not bit for bit (the oracle solves the same system through an SVD, as cv::solve(DECOMP_SVD) does).
    double Lm[6][6], inv[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = A[6 * i + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= Lm[i][k] * Lm[j][k];
            if (i == j) { Lm[i][i] = sqrt(s > 1e-300 ? s : 1e-300); inv[i] = 1.0 / Lm[i][i]; }
            else Lm[i][j] = s * inv[j];
        }
    }
    double y[6], xx[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= Lm[i][k] * y[k];
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= Lm[k][i] * xx[k];
        xx[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = xx[i];
}

// After the first chunk: run the accept / shrink rule over its counts and publish how many iterations the serial loop
// could still reach (RANSACPointSetRegistrator::run: niters only decreases).
__global__ void k_pnp_decide(DevBuffers d, int c0) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= d.B) return;
    SeqState& s = d.st[seq];
    if (!seq_live(s)) return;
    const int K = d.K, n = s.n_tracks;
    if (n == 5) { s.pnp_need = 1; return; }
    const int* good = d.hyp_good + (size_t)seq * K;
    int niters = K > 1 ? K : 1, max_good = 0;
    for (int it = 0; it < niters && it < c0 && it < K; it++) {
        int g = good[it];
        if (g > (max_good > 4 ? max_good : 4)) {
            max_good = g;
            niters = ransac_update_num_iters(d.ransac_log_num, (double)(n - g) / n, niters);
        }
    }
    s.pnp_need = niters < K ? niters : K;
    pnp_draw_subsets(d, s, seq, s.pnp_need);                         // the subsets of the hypotheses the loop can still reach
}

#define PF_THREADS 512                           // lone stream; the lean build runs PF_THREADS_LEAN
#define PF_THREADS_LEAN 256
#define PF_WAVES (PF_THREADS / 64)                // LDS arrays are sized for the larger block
// the value of lane (dpp-permuted) of a double: DPP works on 32-bit registers, so move the halves separately
static __device__ __forceinline__ double dpp_f64(double v, const int ctrl_unused);
template <int CTRL> static __device__ __forceinline__ double dpp_f64_t(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double dpp_f64(double v, const int ctrl) {
    switch (ctrl) {
        case 0xB1: return dpp_f64_t<0xB1>(v);
        case 0x4E: return dpp_f64_t<0x4E>(v);
        case 0x141: return dpp_f64_t<0x141>(v);
        default: return dpp_f64_t<0x140>(v);
    }
}
struct LmShared {
    double param[6], prev[6], R[9], dRdr[27], JtJ[36], JtErr[6];
    double red[PF_WAVES][28];
    double prevErrNorm;
    int lambdaLg10, iters, state, mode;
    const double* lambda_tab;
    int wave_tot[PF_WAVES]; int total;
};

// ---- wave-level sums of 28 doubles per lane, by transposition: each step halves the number of registers while doubling the
// lanes summed into them (v_permlane32_swap / v_permlane16_swap across the rows of 16, DPP mirrors inside a row), so the wave does
// ~30 additions instead of 28 x 6.  At the end lane l holds the wave total of ONE value, lm_red_slot(l) (or a duplicate: -1).
static __device__ __forceinline__ double lm_fold32(double a, double b) {
    int al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
    const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)al, (unsigned)bl, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)ah, (unsigned)bh, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);   // lanes < 32: a, the others: b
}
static __device__ __forceinline__ double lm_fold16(double a, double b) {
    int al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
    const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)al, (unsigned)bl, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)ah, (unsigned)bh, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);   // even rows: a, odd rows: b
}
template <int CTRL> static __device__ __forceinline__ double lm_fold_row(double a, double b, bool upper) {
    const double keep = upper ? b : a, send = upper ? a : b;
    return keep + dpp_f64_t<CTRL>(send);
}
static __device__ __forceinline__ double lm_wave_sums28(const double (&acc)[28], int lane) {
    double w[14], u[7], x[4], y[2];
#pragma unroll
    for (int k = 0; k < 14; k++) w[k] = lm_fold32(acc[k], acc[k + 14]);      // lanes < 32: value k, lanes >= 32: value k + 14
#pragma unroll
    for (int k = 0; k < 7; k++) u[k] = lm_fold16(w[k], w[k + 7]);            // row r: value k + 7 r
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int k = 0; k < 3; k++) x[k] = lm_fold_row<0x140>(u[k], u[k + 4], b3);   // row_mirror: lanes 8..15 of a row take u[k + 4]
    x[3] = u[3] + dpp_f64_t<0x140>(u[3]);                                     // the odd one out: both halves hold it
    y[0] = lm_fold_row<0x141>(x[0], x[2], b2);                                // row_half_mirror
    y[1] = lm_fold_row<0x141>(x[1], x[3], b2);
    double z = lm_fold_row<0x4E>(y[0], y[1], b1);                             // quad_perm [2,3,0,1]
    return z + dpp_f64_t<0xB1>(z);                                            // quad_perm [1,0,3,2]
}
static __device__ __forceinline__ int lm_red_slot(int lane) {
    const int m = ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1);                  // which x the lane ended up with
    if ((lane & 1) || (m == 3 && (lane & 8))) return -1;                      // duplicates
    const int ui = m == 3 ? 3 : m + 4 * ((lane >> 3) & 1);
    return ui + 7 * (lane >> 4);
}

// The points a thread evaluates stay the same through the whole refine: the first LM_CACHED of them live in registers.
#define LM_CACHED 4
struct LmPoints { float X[LM_CACHED], Y[LM_CACHED], Z[LM_CACHED], u[LM_CACHED], v[LM_CACHED]; bool in[LM_CACHED]; };

static __device__ __forceinline__ void lm_point(double X, double Y, double Z, double cu, double cv, const double* R, const double* dRdr,
                                                double t0, double t1, double t2, double fx, double fy, double cx, double cy, double (&acc)[28]) {
    double x = R[0] * X + R[1] * Y + R[2] * Z + t0;
    double y = R[3] * X + R[4] * Y + R[5] * Z + t1;
    double z = R[6] * X + R[7] * Y + R[8] * Z + t2;
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    double ex = x * fx + cx - cu, ey = y * fy + cy - cv;
    acc[27] += ex * ex + ey * ey;
    double jx[6], jy[6];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double dx0 = X * dRdr[9 * j + 0] + Y * dRdr[9 * j + 1] + Z * dRdr[9 * j + 2];
        double dy0 = X * dRdr[9 * j + 3] + Y * dRdr[9 * j + 4] + Z * dRdr[9 * j + 5];
        double dz0 = X * dRdr[9 * j + 6] + Y * dRdr[9 * j + 7] + Z * dRdr[9 * j + 8];
        jx[j] = fx * (z * (dx0 - x * dz0));
        jy[j] = fy * (z * (dy0 - y * dz0));
    }
    jx[3] = fx * z; jx[4] = 0; jx[5] = fx * (-x * z);
    jy[3] = 0; jy[4] = fy * z; jy[5] = fy * (-y * z);
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) acc[q++] += jx[a] * jx[b] + jy[a] * jy[b];
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += jx[a] * ex + jy[a] * ey;
}

// one evaluation of residuals and Jacobians over the inliers at sh.R / sh.dRdr / sh.param[3..5]; totals land in sh.red[0][*],
// visible to wave 0 on return (the caller's barrier publishes what thread 0 does with them)
template <int THREADS>
static __device__ __forceinline__ void lm_eval(const DevBuffers& d, const SeqState& s, size_t o, int n, const LmPoints& pts, LmShared& sh) {
    const double fx = s.K[0], fy = s.K[4], cx = s.K[2], cy = s.K[5];
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0;
    const double* R = sh.R; const double* dRdr = sh.dRdr;
    const double t0 = sh.param[3], t1 = sh.param[4], t2 = sh.param[5];
#pragma unroll
    for (int k = 0; k < LM_CACHED; k++)
        if (pts.in[k]) lm_point(pts.X[k], pts.Y[k], pts.Z[k], pts.u[k], pts.v[k], R, dRdr, t0, t1, t2, fx, fy, cx, cy, acc);
    for (int i = threadIdx.x + LM_CACHED * THREADS; i < n; i += THREADS) {              // more tracks than the registers hold
        if (!d.inlier[o + i]) continue;
        const float2 c = d.tl1[o + i];
        lm_point(d.world[3 * (o + i)], d.world[3 * (o + i) + 1], d.world[3 * (o + i) + 2], c.x, c.y, R, dRdr, t0, t1, t2, fx, fy, cx, cy, acc);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double tot = lm_wave_sums28(acc, lane);
    const int slot = lm_red_slot(lane);
    if (slot >= 0) sh.red[wv][slot] = tot;
    __syncthreads();
    if (threadIdx.x < 28) {                                            // wave 0 alone goes on: its thread 0 runs the state machine
        double t = sh.red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < THREADS / 64; w++) t += sh.red[w][threadIdx.x];
        sh.red[0][threadIdx.x] = t;
    }
    if (wv == 0) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
}

static __device__ __attribute__((always_inline)) void lm_step(LmShared& sh) {
    const double lambda = sh.lambda_tab[sh.lambdaLg10 + 16];          // exp(lambdaLg10 * log(10)), tabulated by the host
    double A[36], x[6];
    for (int i = 0; i < 36; i++) A[i] = sh.JtJ[i];
    for (int i = 0; i < 6; i++) A[7 * i] *= 1. + lambda;
    chol_solve6(A, sh.JtErr, x);
    for (int i = 0; i < 6; i++) sh.param[i] = sh.prev[i] - x[i];
    rodrigues_to_matrix(sh.param, sh.R, sh.dRdr);                       // for the evaluation of this trial point
}


// getInverseTransform (vo.cpp:246-258): [R t; 0 1]^-1 = [Rt, -Rt t; 0 1], row-major 4x4
__device__ __forceinline__ void inverse_transform(const double* R, const double* t, double* T) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * j + i];
        T[4 * i + 3] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

template <int THREADS>
static __device__ __forceinline__ void pnp_final_body(const DevBuffers& d) {
    const int seq = blockIdx.x;
    SeqState& s = d.st[seq];
    if (!seq_live(s)) return;
    __shared__ LmShared sh;
    __shared__ double bestRt[12];
    const int n = s.n_tracks, K = d.K;
    const size_t o = (size_t)seq * d.CAP;
    const bool direct = n == 5;      // model_points == npoints (solvepnp.cpp): one EPnP on all points, all inliers, no refine
    // ---- replay of RANSACPointSetRegistrator::run's accept / shrink rule over the K precomputed inlier counts
    if (threadIdx.x == 0 && direct) { s.pnp_best = 0; s.pnp_iters = 0; s.pnp_good = n; sh.total = 0; }
    if (threadIdx.x == 0 && !direct) {
        const int* good = d.hyp_good + (size_t)seq * K;
        int niters = K > 1 ? K : 1, max_good = 0, best = -1, iters_run = 0;
        for (int it = 0; it < niters && it < K; it++) {
            int g = good[it];
            iters_run = it + 1;
            if (g > (max_good > 4 ? max_good : 4)) {
                max_good = g; best = it;
                niters = ransac_update_num_iters(d.ransac_log_num, (double)(n - g) / n, niters);
            }
        }
        s.pnp_best = best; s.pnp_iters = iters_run; s.pnp_good = max_good;
        sh.total = best;
    }
    __syncthreads();
    const int best = sh.total;
    if (best < 0) { if (threadIdx.x == 0) s.fail_reason = 3; return; }        // solvePnPRansac returned false (vo.cpp:106-113)
    if (threadIdx.x < 12) bestRt[threadIdx.x] = d.hyp[((size_t)seq * K + best) * 12 + threadIdx.x];
    __syncthreads();
    // ---- inlier mask of the winning hypothesis
    {
        const double fx = s.K[0], fy = s.K[4], cx = s.K[2], cy = s.K[5];
        const double thr = (double)d.cfg.ransac_reprojection_error;
        const float thr2 = (float)(thr * thr);
        for (int i = threadIdx.x; i < n; i += THREADS)
            d.inlier[o + i] = direct ? (uint8_t)1 : (uint8_t)point_is_inlier(bestRt, fx, fy, cx, cy, d.world + 3 * (o + i), d.tl1[o + i], thr2);
    }
    __syncthreads();
    // ---- Levenberg–Marquardt refine on the inliers (solvePnP ITERATIVE, useExtrinsicGuess; CvLevMarq state machine)
    LmPoints pts;
#pragma unroll
    for (int k = 0; k < LM_CACHED; k++) {
        const int i = threadIdx.x + k * THREADS;
        pts.in[k] = i < n && d.inlier[o + i];
        const int ii = i < n ? i : 0;
        const float2 c = d.tl1[o + ii];
        pts.X[k] = d.world[3 * (o + ii)]; pts.Y[k] = d.world[3 * (o + ii) + 1]; pts.Z[k] = d.world[3 * (o + ii) + 2];
        pts.u[k] = c.x; pts.v[k] = c.y;
    }
    if (threadIdx.x == 0) {
        double rv[3];
        rodrigues_to_vector(bestRt, rv);
        sh.param[0] = rv[0]; sh.param[1] = rv[1]; sh.param[2] = rv[2];
        sh.param[3] = bestRt[9]; sh.param[4] = bestRt[10]; sh.param[5] = bestRt[11];
        sh.lambdaLg10 = -3; sh.iters = 0; sh.state = 0; sh.mode = direct ? 2 : 1; sh.prevErrNorm = 0; sh.lambda_tab = d.lm_lambda;
        rodrigues_to_matrix(sh.param, sh.R, sh.dRdr);
    }
    __syncthreads();
    // CvLevMarq's state machine (CALC_J -> step -> CHECK_ERR -> accept / raise lambda), with one change of SCHEDULE only: every
    // evaluation of a trial point computes the Jacobian sums together with the error, so an accepted step (the normal case)
    // already has J for the next CALC_J instead of re-evaluating the same point — half the evaluations, the same numbers.
    // sh.mode: 1 = the evaluation to come is the very first (at the start point), 0 = it is of a trial point, 2 = finished.
    for (int guard = 0; guard < 1000; guard++) {
        const int mode = sh.mode;
        if (mode == 2) break;
        lm_eval<THREADS>(d, s, o, n, pts, sh);
        if (threadIdx.x == 0) {
            bool take_J = false;
            if (mode == 1) {                               // CALC_J at the start point
                sh.prevErrNorm = sqrt(sh.red[0][27]);
                take_J = true;
            } else {                                       // CHECK_ERR at the trial point
                const double errNorm = sqrt(sh.red[0][27]);
                if (errNorm > sh.prevErrNorm && ++sh.lambdaLg10 <= 16) {
                    lm_step(sh);                           // same J, same start, stronger damping
                } else {
                    sh.lambdaLg10 = sh.lambdaLg10 - 1 > -16 ? sh.lambdaLg10 - 1 : -16;
                    double dn = 0, pn = 0;
                    for (int a = 0; a < 6; a++) { dn += (sh.param[a] - sh.prev[a]) * (sh.param[a] - sh.prev[a]); pn += sh.prev[a] * sh.prev[a]; }
                    if (++sh.iters >= 20 || sqrt(dn) / (sqrt(pn) + SVO_DBL_EPS) < 1.1920928955078125e-07) sh.mode = 2;
                    else { sh.prevErrNorm = errNorm; take_J = true; }
                }
            }
            if (take_J) {                                  // CALC_J: J and err at param are ready -> step
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int b = a; b < 6; b++) { sh.JtJ[6 * a + b] = sh.JtJ[6 * b + a] = sh.red[0][q++]; }
                }
                for (int a = 0; a < 6; a++) sh.JtErr[a] = sh.red[0][21 + a];
                for (int a = 0; a < 6; a++) sh.prev[a] = sh.param[a];
                lm_step(sh);
                sh.mode = 0;
            }
        }
        __syncthreads();
    }
    // ---- count inliers, update the feature set to the inliers at their T1 positions (vo.cpp:115-121)
    const int fb = s.feat_buf;
    const int chunk = (n + THREADS - 1) / THREADS;
    const int i0 = threadIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    int cnt = 0;
    for (int i = i0; i < i1; i++) cnt += d.inlier[o + i];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = cnt;
    for (int k = 1; k < 64; k <<= 1) { int t = __shfl_up(incl, k); if (lane >= k) incl += t; }
    if (lane == 63) sh.wave_tot[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int i = 0; i < (THREADS / 64); i++) { int t = sh.wave_tot[i]; sh.wave_tot[i] = acc; acc += t; } sh.total = acc; }
    __syncthreads();
    const int n_inl = sh.total;
    if (threadIdx.x == 0) {
        // success: rotation = Rodrigues(rvec), translation = tvec (vo.cpp:307-308) — before the inlier-count gate
        rodrigues_to_matrix(sh.param, s.R, nullptr);
        s.t[0] = sh.param[3]; s.t[1] = sh.param[4]; s.t[2] = sh.param[5];
        s.n_inliers = n_inl;
    }
    if (n_inl < d.cfg.features_threshold) {                                                      // vo.cpp:106-113
        // the reference builds its is_ok vector only past this gate (vo.cpp:115): on this path no inlier flags exist
        for (int i = threadIdx.x; i < n; i += THREADS) d.inlier[o + i] = 0;
        if (threadIdx.x == 0) s.fail_reason = 3;
        return;
    }
    {
        int pos = sh.wave_tot[wv] + incl - cnt;
        const int* fage = d.feat_age[fb] + o; const int* fstr = d.feat_str[fb] + o;
        float2* nxy = d.feat_xy[fb ^ 1] + o; int* nage = d.feat_age[fb ^ 1] + o; int* nstr = d.feat_str[fb ^ 1] + o;
        for (int i = i0; i < i1; i++) {
            if (!d.inlier[o + i]) continue;
            nxy[pos] = d.tl1[o + i]; nage[pos] = fage[i]; nstr[pos] = fstr[i];
            d.inl_idx[o + pos] = i;
            pos++;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s.n_feat = n_inl; s.feat_buf = fb ^ 1;
        double rv[3];
        double tn = sqrt(s.t[0] * s.t[0] + s.t[1] * s.t[1] + s.t[2] * s.t[2]);                 // vo.cpp:124
        rodrigues_to_vector(s.R, rv);                                                            // vo.cpp:125
        double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);                     // vo.cpp:126
        if (tn > d.cfg.max_translation_norm || angle > d.cfg.max_rotation_norm) { s.fail_reason = 4; return; }   // vo.cpp:129-132
        inverse_transform(s.R, s.t, s.last_T);                                                  // vo.cpp:133
        s.ok = 1;
    }
}
__global__ __launch_bounds__(PF_THREADS) void k_pnp_final(DevBuffers d) { pnp_final_body<PF_THREADS>(d); }
__global__ __launch_bounds__(PF_THREADS_LEAN) __attribute__((amdgpu_num_vgpr(48))) void k_pnp_final_lean(DevBuffers d) { pnp_final_body<PF_THREADS_LEAN>(d); }

// ------------------------------------------------------------------------------------------------ four points: P3P
// cv::solvePnPRansac with exactly four points solves ONE P3P (Gao, Hou, Tang, Chang, PAMI 2003, as OpenCV's p3p.cpp does it:
// quartic in |PA| / |PC|, closed-form |PB| / |PC|, Horn's quaternion alignment, the fourth point picks the branch), every
// point an inlier, no RANSAC and no refine (solvepnp.cpp: npoints == 4 -> model_points = 4, SOLVEPNP_P3P, model_points ==
// npoints).  Only the public cameraToWorld reaches it (vo.h:452-456); stereo_callback wants > 15 tracks (vo.cpp:82).
// One thread: the work is a few hundred flops.  pow / acos / cos make this tolerance-parity, like the LM refine.
static __device__ int p3p_solve_deg2(double a, double b, double c, double& x1, double& x2) {
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) { x1 = -b * inv_2a; x2 = x1; return 1; }
    double sd = sqrt(delta);
    x1 = (-b + sd) * inv_2a; x2 = (-b - sd) * inv_2a;
    return 2;
}
static __device__ int p3p_solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
    if (a == 0) {
        if (b == 0) { if (c == 0) return 0; x0 = -d / c; return 1; }
        x2 = 0;
        return p3p_solve_deg2(b, c, d, x0, x1);
    }
    double inv_a = 1. / a, b_a = inv_a * b, b_a2 = b_a * b_a, c_a = inv_a * c, d_a = inv_a * d;
    double Q = (3 * c_a - b_a2) / 9, R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q, D = Q3 + R * R, b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) { x0 = x1 = x2 = -b_a_3; return 3; }
        x0 = pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        double theta = acos(R / sqrt(-Q3)), sq = sqrt(-Q);
        x0 = 2 * sq * cos(theta / 3.0) - b_a_3;
        x1 = 2 * sq * cos((theta + 2 * 3.14159265358979323846) / 3.0) - b_a_3;
        x2 = 2 * sq * cos((theta + 4 * 3.14159265358979323846) / 3.0) - b_a_3;
        return 3;
    }
    double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    x0 = AD + BD - b_a_3;
    return 1;
}
static __device__ int p3p_solve_deg4(double a, double b, double c, double d, double e, double* x) {
    if (a == 0) { x[3] = 0; return p3p_solve_deg3(b, c, d, e, x[0], x[1], x[2]); }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b, r0, r1, r2;
    int n = p3p_solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
    if (n == 0) return 0;
    double R2 = 0.25 * b2 - c + r0;
    if (R2 < 0) return 0;
    double R = sqrt(R2), inv_R = 1. / R, D2, E2;
    int nb = 0;
    if (R < 10E-12) {
        double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else { double st = sqrt(temp); D2 = 0.75 * b2 - 2 * c + 2 * st; E2 = D2 - 4 * st; }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v; E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) { double D = sqrt(D2); nb = 2; x[0] = R_2 + 0.5 * D - b_4; x[1] = x[0] - D; }
    if (E2 >= 0) {
        double E = sqrt(E2);
        if (nb == 0) { x[0] = -R_2 + 0.5 * E - b_4; x[1] = x[0] - E; nb = 2; }
        else { x[2] = -R_2 + 0.5 * E - b_4; x[3] = x[2] - E; nb = 4; }
    }
    return nb;
}
static __device__ int p3p_lengths(double (*lengths)[3], const double* distances, const double* cosines) {
    double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]), b = inv_d22 * (distances[1] * distances[1]);
    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r, pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr), b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double roots[4];
    int n = p3p_solve_deg4(A, B, C, D, E, roots), nb = 0;
    if (n == 0) return 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q, inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
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
        double y = inv_b0 * b1, v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        double Z = distances[2] / sqrt(v);
        lengths[nb][0] = x * Z; lengths[nb][1] = y * Z; lengths[nb][2] = Z;
        nb++;
    }
    return nb;
}
static __device__ void p3p_jacobi4(double* A, double* D, double* U) {      // symmetric 4 x 4 eigen decomposition, cyclic Jacobi
    double B[4], Z[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++) U[i] = (i % 5 == 0);
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    for (int i = 0; i < 4; i++) D[i] = B[i];
    for (int iter = 0; iter < 50; iter++) {
        double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (int i = 0; i < 3; i++) {
            for (int j = i + 1; j < 4; j++) {
                double& aij = A[4 * i + j];
                double Aij = aij, eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j])) aij = 0.0;
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
                    aij = 0.0;
                    double c = 1.0 / sqrt(1 + t * t), s = t * c, tau = s / (1.0 + c);
                    for (int k = 0; k <= i - 1; k++) { double g = A[k * 4 + i], h = A[k * 4 + j]; A[k * 4 + i] = g - s * (h + g * tau); A[k * 4 + j] = h + s * (g - h * tau); }
                    for (int k = i + 1; k <= j - 1; k++) { double g = A[i * 4 + k], h = A[k * 4 + j]; A[i * 4 + k] = g - s * (h + g * tau); A[k * 4 + j] = h + s * (g - h * tau); }
                    for (int k = j + 1; k < 4; k++) { double g = A[i * 4 + k], h = A[j * 4 + k]; A[i * 4 + k] = g - s * (h + g * tau); A[j * 4 + k] = h + s * (g - h * tau); }
                    for (int k = 0; k < 4; k++) { double g = U[k * 4 + i], h = U[k * 4 + j]; U[k * 4 + i] = g - s * (h + g * tau); U[k * 4 + j] = h + s * (g - h * tau); }
                }
            }
        }
        for (int i = 0; i < 4; i++) { B[i] += Z[i]; D[i] = B[i]; Z[i] = 0; }
    }
}
static __device__ void p3p_align(const double (*M_end)[3], const double (*Xw)[3], double (*R)[3], double* T) {
    double C_start[3], C_end[3], s[9], Qs[16], evs[4], U[16], q[4];
    for (int i = 0; i < 3; i++) {
        C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
        C_start[i] = (Xw[0][i] + Xw[1][i] + Xw[2][i]) / 3;
    }
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            s[i * 3 + j] = (Xw[0][i] * M_end[0][j] + Xw[1][i] * M_end[1][j] + Xw[2][i] * M_end[2][j]) / 3 - C_end[j] * C_start[i];
    for (int i = 0; i < 16; i++) Qs[i] = 0;
    Qs[0] = s[0] + s[4] + s[8]; Qs[5] = s[0] - s[4] - s[8]; Qs[10] = s[4] - s[8] - s[0]; Qs[15] = s[8] - s[0] - s[4];
    Qs[4] = Qs[1] = s[5] - s[7]; Qs[8] = Qs[2] = s[6] - s[2]; Qs[12] = Qs[3] = s[1] - s[3];
    Qs[9] = Qs[6] = s[3] + s[1]; Qs[13] = Qs[7] = s[6] + s[2]; Qs[14] = Qs[11] = s[7] + s[5];
    p3p_jacobi4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[0];
    for (int i = 1; i < 4; i++) if (evs[i] > ev_max) { ev_max = evs[i]; i_ev = i; }
    for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3], q1_2 = q[1] * q[2], q1_3 = q[1] * q[3], q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32; R[0][1] = 2. * (q1_2 - q0_3); R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3); R[1][1] = q02 + q22 - q12 - q32; R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2); R[2][1] = 2. * (q2_3 + q0_1); R[2][2] = q02 + q32 - q12 - q22;
    for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
}

// one thread per sequence; expects s.n_tracks == 4
__global__ void k_pnp_p3p(DevBuffers d) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= d.B) return;
    SeqState& s = d.st[seq];
    if (!seq_live(s) || s.n_tracks != 4) return;
    const size_t o = (size_t)seq * d.CAP;
    const double fx = s.K[0], fy = s.K[4], cx = s.K[2], cy = s.K[5];
    const double inv_fx = 1. / fx, inv_fy = 1. / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double obj[12], mu[4], mv[4], mk[3], Xw[3][3], distances[3], cosines[3], lengths[4][3];
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < 3; k++) obj[3 * i + k] = d.world[3 * (o + i) + k];
        const float2 c = d.tl1[o + i];
        // undistortPoints(.., P = cameraMatrix) on CV_32FC2 with zero distortion: normalise and re-project in f64, store f32
        const double u = (double)(float)((((double)c.x - cx) * inv_fx) * fx + cx), v = (double)(float)((((double)c.y - cy) * inv_fy) * fy + cy);
        mu[i] = inv_fx * u - cx_fx; mv[i] = inv_fy * v - cy_fy;
    }
    for (int i = 0; i < 3; i++) {
        double norm = sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1);
        mk[i] = 1. / norm; mu[i] *= mk[i]; mv[i] *= mk[i];
        for (int k = 0; k < 3; k++) Xw[i][k] = obj[3 * i + k];
    }
    distances[0] = sqrt(dist2(obj + 3, obj + 6)); distances[1] = sqrt(dist2(obj, obj + 6)); distances[2] = sqrt(dist2(obj, obj + 3));
    cosines[0] = mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2];
    cosines[1] = mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2];
    cosines[2] = mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1];
    const int n = p3p_lengths(lengths, distances, cosines);
    int nb = 0;
    double best_err = 0, bestR[9], bestT[3];
    for (int i = 0; i < n; i++) {
        double M_orig[3][3], Rs[3][3], ts[3];
        for (int k = 0; k < 3; k++) { M_orig[k][0] = lengths[i][k] * mu[k]; M_orig[k][1] = lengths[i][k] * mv[k]; M_orig[k][2] = lengths[i][k] * mk[k]; }
        p3p_align(M_orig, Xw, Rs, ts);
        double X3p = Rs[0][0] * obj[9] + Rs[0][1] * obj[10] + Rs[0][2] * obj[11] + ts[0];
        double Y3p = Rs[1][0] * obj[9] + Rs[1][1] * obj[10] + Rs[1][2] * obj[11] + ts[1];
        double Z3p = Rs[2][0] * obj[9] + Rs[2][1] * obj[10] + Rs[2][2] * obj[11] + ts[2];
        double mu3p = X3p / Z3p, mv3p = Y3p / Z3p;
        double err = (mu3p - mu[3]) * (mu3p - mu[3]) + (mv3p - mv[3]) * (mv3p - mv[3]);
        if (nb == 0 || err < best_err) {
            best_err = err;
            for (int k = 0; k < 3; k++) { bestR[3 * k] = Rs[k][0]; bestR[3 * k + 1] = Rs[k][1]; bestR[3 * k + 2] = Rs[k][2]; bestT[k] = ts[k]; }
        }
        nb++;
    }
    s.pnp_iters = 0;
    if (nb == 0) { s.pnp_best = -1; s.fail_reason = 3; return; }           // solvePnP returned false
    double rv[3];
    rodrigues_to_vector(bestR, rv);                                         // rvec out of solvePnP ...
    rodrigues_to_matrix(rv, s.R, nullptr);                                  // ... and back to a matrix (vo.cpp:308)
    s.t[0] = bestT[0]; s.t[1] = bestT[1]; s.t[2] = bestT[2];
    s.pnp_best = 0; s.pnp_good = 4; s.n_inliers = 4;
    for (int i = 0; i < 4; i++) { d.inlier[o + i] = 1; d.inl_idx[o + i] = i; }
}
void launch_pnp_p3p(const DevBuffers& d, hipStream_t st) {
    hipLaunchKernelGGL(k_pnp_p3p, dim3((d.B + 63) / 64), dim3(64), 0, st, d);
}

// ---- getInverseTransform (vo.cpp:246-258) as its own one-thread launch, for the stage API ----
__global__ void k_inverse_transform(const double* __restrict__ R, const double* __restrict__ t, double* __restrict__ T) {
    if (threadIdx.x == 0 && blockIdx.x == 0) inverse_transform(R, t, T);
}
void launch_inverse_transform(const double* R, const double* t, double* T, hipStream_t st) {
    hipLaunchKernelGGL(k_inverse_transform, dim3(1), dim3(64), 0, st, R, t, T);
}

// lone-stream frame pipeline: triangulation || first EPnP chunk (k_tri_epnp); true if launched (then call launch_pnp(.., true))
bool launch_triangulate_epnp_fused(const DevBuffers& d, hipStream_t st) {
    static const bool off = getenv("SVO_TRI_EPNP_FUSED") && atoi(getenv("SVO_TRI_EPNP_FUSED")) == 0;
    if (off || d.B > SVO_LONE_MAX_SEQ || d.co_resident) return false;
    const int lanes = 16, c0 = pnp_first_chunk(d), hpb = 64 / EP_G_LONE;
    const int n_e = (c0 + hpb - 1) / hpb, n_t = (d.CAP + lanes - 1) / lanes;
    hipLaunchKernelGGL(k_tri_epnp, dim3(n_e + n_t, d.B), dim3(64), 0, st, d, lanes, n_e, c0);
    return true;
}

void launch_pnp(const DevBuffers& d, hipStream_t st, bool first_chunk_solved) {
    // the subsets were drawn by the last block of k_triangulate (stage entry points go through launch_triangulate too)
    const int c0 = pnp_first_chunk(d);
    const bool lean = d.co_resident;                                 // see k_triangulate_lean
    if (lean) {                                                      // more than the 64 KB a kernel gets without asking; per device
        static bool asked[SVO_MAX_DEVICES];
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < SVO_MAX_DEVICES && !asked[dev]) {
            (void)hipFuncSetAttribute((const void*)k_pnp_epnp_lean, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EP_LEAN_LDS);
            asked[dev] = true;
        }
    }
    const int hpb = 64 / (lean ? EP_G : EP_G_LONE);
    if (first_chunk_solved) { /* k_tri_epnp did it */ }
    else if (lean) hipLaunchKernelGGL(k_pnp_epnp_lean, dim3((c0 + hpb - 1) / hpb, d.B), dim3(64), EP_LEAN_LDS, st, d, 0, c0);
    else hipLaunchKernelGGL(k_pnp_epnp, dim3((c0 + hpb - 1) / hpb, d.B), dim3(64), 0, st, d, 0, c0);
    hipLaunchKernelGGL(k_pnp_score, dim3(c0, d.B), dim3(256), 0, st, d, 0, c0);
    if (d.K > c0) {
        hipLaunchKernelGGL(k_pnp_decide, dim3((d.B + 63) / 64), dim3(64), 0, st, d, c0);
        if (lean) hipLaunchKernelGGL(k_pnp_epnp_lean, dim3((d.K - c0 + hpb - 1) / hpb, d.B), dim3(64), EP_LEAN_LDS, st, d, c0, d.K);
        else hipLaunchKernelGGL(k_pnp_epnp, dim3((d.K - c0 + hpb - 1) / hpb, d.B), dim3(64), 0, st, d, c0, d.K);
        hipLaunchKernelGGL(k_pnp_score, dim3(d.K - c0, d.B), dim3(256), 0, st, d, c0, d.K);
    }
    if (lean) hipLaunchKernelGGL(k_pnp_final_lean, dim3(d.B), dim3(PF_THREADS_LEAN), 0, st, d);
    else hipLaunchKernelGGL(k_pnp_final, dim3(d.B), dim3(PF_THREADS), 0, st, d);
}
