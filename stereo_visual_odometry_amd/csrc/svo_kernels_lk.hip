// svo_kernels_lk.hip — pyramidal Lucas–Kanade, one wave64 per feature.
//
// Replaces cv::calcOpticalFlowPyrLK as called four times by VisualOdometry::circularMatching
// (reference src/vo.cpp:203-215) and the status / loop-closure / in-bounds masks (vo.cpp:217-230,
// 341-359).  The four passes L0->L1->R1->R0->L0 of one feature are independent of every other
// feature, so ONE launch runs the whole chain: one single-wave workgroup per feature, all pyramid
// levels and all Newton iterations inside it.
//
// MI355X mapping (no MFMA: there is no dense contraction here):
//   * the (w+3)^2 source tile of the template image is staged once per level into LDS; Scharr
//     derivatives are computed from it on the fly — no derivative pyramid ever exists in HBM;
//   * each lane owns PPL consecutive pixels of one window row, and keeps their I, Ix, Iy samples in
//     VGPRs for the whole Newton loop;
//   * the search-image window is staged as an LDS tile with a +-LK_M px guard band and only
//     re-staged when the window walks out of it;
//   * the 2x2 normal matrix and the mismatch vector are per-lane int32 partials reduced across the
//     wave with shuffles as exact int64 — order-independent, so the result is bit-identical to the
//     sequential CPU loop; the float tail runs identically on every lane.
#include "svo_internal.hpp"

#define LK_M 4                                   // guard band of the search tile (pixels)
#define LK_WBITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

template <int W> struct LkLayout {
    static constexpr int ppl() {                 // pixels per lane: smallest p with W * ceil(W/p) <= 64
        for (int p = 1; p <= W; p++) if (W * ((W + p - 1) / p) <= 64) return p;
        return W;
    }
    static constexpr int PPL = ppl();
    static constexpr int LPR = (W + PPL - 1) / PPL;   // lanes per window row
    static constexpr int SW = W + 3;                  // template source tile side
    static constexpr int DW = W + 1;                  // derivative tile side
    static constexpr int TS = W + 1 + 2 * LK_M;       // search tile side
};

__device__ __forceinline__ long long wave_sum_i64(int partial) {
    long long v = (long long)partial;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    // uniform by construction; tell the compiler so the float tail and the branches go scalar
    int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xFFFFFFFFll));
    int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ void lk_weights(float a, float b, int& iw00, int& iw01, int& iw10, int& iw11) {
    iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << LK_WBITS));
    iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << LK_WBITS));
    iw10 = __float2int_rn((1.f - a) * b * (float)(1 << LK_WBITS));
    iw11 = (1 << LK_WBITS) - iw00 - iw01 - iw10;
}

struct LkCrit { int max_count; double eps2; double min_eig; };

// One cv::calcOpticalFlowPyrLK track of a single point across all pyramid levels (LKTrackerInvoker semantics,
// SURVEY.md Appendix A.3).  (px,py) -> (outx,outy), status.  All 64 lanes call this together.
template <int W>
__device__ void lk_pass(const Geometry& g, const uint8_t* __restrict__ pyrA, const uint8_t* __restrict__ pyrB,
                        float px, float py, float& outx, float& outy, int& status, const LkCrit& crit,
                        uint8_t* S, int* D, uint8_t* Jt) {
    using LL = LkLayout<W>;
    constexpr int PPL = LL::PPL, LPR = LL::LPR, SW = LL::SW, DW = LL::DW, TS = LL::TS;
    const int lane = threadIdx.x;
    const int row = lane / LPR, seg = lane - row * LPR;
    const bool lane_on = row < W;
    const int xs = seg * PPL;
    const float half = (W - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int top = g.nlevels - 1;
    float nx = 0.f, ny = 0.f;
    status = 1;
    outx = px; outy = py;
    for (int level = top; level >= 0; --level) {
        const LevelInfo L = g.lv[level];
        const uint8_t* __restrict__ A = pyrA + L.off;
        const uint8_t* __restrict__ Bm = pyrB + L.off;
        const float scale = 1.f / (float)(1 << level);
        float ppx = px * scale, ppy = py * scale;
        if (level == top) { nx = ppx; ny = ppy; } else { nx = outx * 2.f; ny = outy * 2.f; }
        outx = nx; outy = ny;
        ppx -= half; ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (ipx < -W || ipx >= L.w || ipy < -W || ipy >= L.h) {
            if (level == 0) status = 0;
            continue;
        }
        // ---- stage the (W+3)^2 template tile; REFLECT_101 is the pyramid border of cv::buildOpticalFlowPyramid
        __syncthreads();
        for (int i = lane; i < SW * SW; i += 64) {
            int sy = i / SW, sx = i - sy * SW;
            S[i] = A[(size_t)reflect101(ipy - 1 + sy, L.h) * L.w + reflect101(ipx - 1 + sx, L.w)];
        }
        __syncthreads();
        // ---- Scharr derivatives on the (W+1)^2 grid; zero outside the image (derivBorder = CONSTANT 0)
        for (int i = lane; i < DW * DW; i += 64) {
            int dy_ = i / DW, dx_ = i - dy_ * DW;
            int gx = ipx + dx_, gy = ipy + dy_;
            int v = 0;
            if (gx >= 0 && gx < L.w && gy >= 0 && gy < L.h) {
                const uint8_t* c = S + (dy_ + 1) * SW + (dx_ + 1);
                int a00 = c[-SW - 1], a01 = c[-SW], a02 = c[-SW + 1], a10 = c[-1], a12 = c[1];
                int a20 = c[SW - 1], a21 = c[SW], a22 = c[SW + 1];
                int t0m = 3 * (a00 + a20) + 10 * a10, t0p = 3 * (a02 + a22) + 10 * a12;
                int t1m = a20 - a00, t1c = a21 - a01, t1p = a22 - a02;
                int dxv = t0p - t0m, dyv = 3 * (t1m + t1p) + 10 * t1c;
                v = (dxv & 0xFFFF) | (dyv << 16);
            }
            D[i] = v;
        }
        __syncthreads();
        // ---- patch extraction into registers + covariance partials
        int iw00, iw01, iw10, iw11;
        lk_weights(ppx - (float)ipx, ppy - (float)ipy, iw00, iw01, iw10, iw11);
        int Ir[PPL], Ixr[PPL], Iyr[PPL];
        int pA11 = 0, pA12 = 0, pA22 = 0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            int x = xs + j;
            Ir[j] = 0; Ixr[j] = 0; Iyr[j] = 0;
            if (lane_on && x < W) {
                const uint8_t* s = S + (row + 1) * SW + (x + 1);
                int ival = DESCALE(s[0] * iw00 + s[1] * iw01 + s[SW] * iw10 + s[SW + 1] * iw11, LK_WBITS - 5);
                int d00 = D[row * DW + x], d01 = D[row * DW + x + 1], d10 = D[(row + 1) * DW + x], d11 = D[(row + 1) * DW + x + 1];
                int ixval = DESCALE((int)(short)(d00 & 0xFFFF) * iw00 + (int)(short)(d01 & 0xFFFF) * iw01 +
                                    (int)(short)(d10 & 0xFFFF) * iw10 + (int)(short)(d11 & 0xFFFF) * iw11, LK_WBITS);
                int iyval = DESCALE((d00 >> 16) * iw00 + (d01 >> 16) * iw01 + (d10 >> 16) * iw10 + (d11 >> 16) * iw11, LK_WBITS);
                Ir[j] = ival; Ixr[j] = ixval; Iyr[j] = iyval;
                pA11 += ixval * ixval; pA12 += ixval * iyval; pA22 += iyval * iyval;
            }
        }
        const long long iA11 = wave_sum_i64(pA11), iA12 = wave_sum_i64(pA12), iA22 = wave_sum_i64(pA22);
        const float A11 = (float)(double)iA11 * FLT_SCALE, A12 = (float)(double)iA12 * FLT_SCALE, A22 = (float)(double)iA22 * FLT_SCALE;
        float Dt = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * W * W);
        if ((double)minEig < crit.min_eig || Dt < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        Dt = 1.f / Dt;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        int tx0 = 0, ty0 = 0;
        bool have_tile = false;
        for (int j = 0; j < crit.max_count; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -W || inx >= L.w || iny < -W || iny >= L.h) {
                if (level == 0) status = 0;
                break;
            }
            if (!have_tile || inx < tx0 || inx > tx0 + 2 * LK_M || iny < ty0 || iny > ty0 + 2 * LK_M) {
                tx0 = inx - LK_M; ty0 = iny - LK_M;
                __syncthreads();
                for (int i = lane; i < TS * TS; i += 64) {
                    int ty = i / TS, tx = i - ty * TS;
                    Jt[i] = Bm[(size_t)reflect101(ty0 + ty, L.h) * L.w + reflect101(tx0 + tx, L.w)];
                }
                __syncthreads();
                have_tile = true;
            }
            lk_weights(nx - (float)inx, ny - (float)iny, iw00, iw01, iw10, iw11);
            int pb1 = 0, pb2 = 0;
            if (lane_on) {
                const uint8_t* jp = Jt + (iny - ty0 + row) * TS + (inx - tx0 + xs);
#pragma unroll
                for (int jj = 0; jj < PPL; jj++) {
                    if (xs + jj < W) {
                        int diff = DESCALE(jp[jj] * iw00 + jp[jj + 1] * iw01 + jp[jj + TS] * iw10 + jp[jj + TS + 1] * iw11, LK_WBITS - 5) - Ir[jj];
                        pb1 += diff * Ixr[jj]; pb2 += diff * Iyr[jj];
                    }
                }
            }
            const long long ib1 = wave_sum_i64(pb1), ib2 = wave_sum_i64(pb2);
            const float b1 = (float)(double)ib1 * FLT_SCALE, b2 = (float)(double)ib2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * Dt, dy = (A12 * b1 - A11 * b2) * Dt;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= crit.eps2) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        // flags = 0 with err != NULL (vo.cpp:182,203): the level-0 error block re-checks the final window origin
        if (status && level == 0) {
            int ix = (int)floorf(outx - half), iy = (int)floorf(outy - half);
            if (ix < -W || ix >= L.w || iy < -W || iy >= L.h) status = 0;
        }
    }
}

template <int W> struct LkSmem {
    uint8_t S[(LkLayout<W>::SW * LkLayout<W>::SW + 15) & ~15];
    int D[LkLayout<W>::DW * LkLayout<W>::DW];
    uint8_t Jt[(LkLayout<W>::TS * LkLayout<W>::TS + 15) & ~15];
};

__device__ __forceinline__ LkCrit make_crit(const svo_config& c) {
    LkCrit k;
    int mc = c.lk_max_count; mc = mc < 0 ? 0 : (mc > 100 ? 100 : mc);       // TermCriteria normalisation (lkpyramid.cpp)
    double e = c.lk_epsilon; e = e < 0. ? 0. : (e > 10. ? 10. : e);
    k.max_count = mc; k.eps2 = e * e; k.min_eig = c.optical_flow_min_eig_threshold;
    return k;
}

// ---- fused circular matching: L0 -> L1 -> R1 -> R0 -> L0 + masks (vo.cpp:203-230, 341-359) ----
template <int W>
__global__ __launch_bounds__(64) void k_lk_chain(DevBuffers d) {
    __shared__ LkSmem<W> sm;
    const int seq = blockIdx.y, idx = blockIdx.x;
    SeqState& s = d.st[seq];
    if (!s.active) return;
    int n = s.n_feat;
    if (d.cfg.max_features > 0 && n > d.cfg.max_features) n = d.cfg.max_features;
    if (idx == 0 && threadIdx.x == 0) s.n_lk = n;
    if (idx >= n) return;
    const size_t o = (size_t)seq * d.CAP + idx;
    const uint8_t* L0 = d.pyr + pyr_index(d, seq, s.slot_pyr_t0, 0);
    const uint8_t* R0 = d.pyr + pyr_index(d, seq, s.slot_pyr_t0, 1);
    const uint8_t* L1 = d.pyr + pyr_index(d, seq, s.slot_t1, 0);
    const uint8_t* R1 = d.pyr + pyr_index(d, seq, s.slot_t1, 1);
    const LkCrit crit = make_crit(d.cfg);
    const float2 p0 = d.feat_xy[s.feat_buf][o];                      // pointsLeftT0 = currentVOFeatures.points (vo.cpp:338)
    float2 p1, p2, p3, p4; int st0, st1, st2, st3;
    lk_pass<W>(d.geom, L0, L1, p0.x, p0.y, p1.x, p1.y, st0, crit, sm.S, sm.D, sm.Jt);   // vo.cpp:203
    lk_pass<W>(d.geom, L1, R1, p1.x, p1.y, p2.x, p2.y, st1, crit, sm.S, sm.D, sm.Jt);   // vo.cpp:206
    lk_pass<W>(d.geom, R1, R0, p2.x, p2.y, p3.x, p3.y, st2, crit, sm.S, sm.D, sm.Jt);   // vo.cpp:209
    lk_pass<W>(d.geom, R0, L0, p3.x, p3.y, p4.x, p4.y, st3, crit, sm.S, sm.D, sm.Jt);   // vo.cpp:213
    if (threadIdx.x == 0) {
        const float thr = (float)d.cfg.circular_matching_success_threshold;            // findClosePoints takes a float32 (vo.h:432)
        float ex = fabsf(p0.x - p4.x), ey = fabsf(p0.y - p4.y);
        float off = (ex < ey) ? ey : ex;
        int circ = st0 && st1 && st2 && st3 && !(off > thr);                            // vo.cpp:227-230
        const float Wf = (float)d.geom.W, Hf = (float)d.geom.H;
        const float2 q[4] = {p0, p1, p3, p2};
        int inb = 1;
        for (int k = 0; k < 4; k++)
            if ((q[k].x < 0) || (q[k].y < 0) || (q[k].y >= Hf) || (q[k].x >= Wf)) inb = 0;   // vo.cpp:344-359
        d.pl0[o] = p0; d.pl1[o] = p1; d.pr1[o] = p2; d.pr0[o] = p3; d.plc[o] = p4;
        d.okmask[o] = (uint8_t)(circ | (inb << 1));
    }
}

// ---- single pass, for the cv::calcOpticalFlowPyrLK-shaped stage API ----
template <int W>
__global__ __launch_bounds__(64) void k_lk_single(DevBuffers d, int slotA, int camA, int slotB, int camB, int n,
                                                  const float2* prev, float2* next, uint8_t* status) {
    __shared__ LkSmem<W> sm;
    const int idx = blockIdx.x;
    if (idx >= n) return;
    const uint8_t* A = d.pyr + pyr_index(d, 0, slotA, camA);
    const uint8_t* Bp = d.pyr + pyr_index(d, 0, slotB, camB);
    const LkCrit crit = make_crit(d.cfg);
    float2 p = prev[idx], q; int st;
    lk_pass<W>(d.geom, A, Bp, p.x, p.y, q.x, q.y, st, crit, sm.S, sm.D, sm.Jt);
    if (threadIdx.x == 0) { next[idx] = q; status[idx] = (uint8_t)st; }
}

#define LK_FOR_EACH_WINDOW(X) X(7) X(10) X(15) X(21) X(31)

bool lk_window_supported(int win) {
#define CHK(Wn) if (win == Wn) return true;
    LK_FOR_EACH_WINDOW(CHK)
#undef CHK
    return false;
}

void launch_lk_chain(const DevBuffers& d, int grid_n, hipStream_t st) {
    if (grid_n < 1) grid_n = 1;
    if (grid_n > d.CAP) grid_n = d.CAP;
    dim3 g(grid_n, d.B);
#define LAUNCH(Wn) if (d.cfg.win_w == Wn) { hipLaunchKernelGGL(k_lk_chain<Wn>, g, dim3(64), 0, st, d); return; }
    LK_FOR_EACH_WINDOW(LAUNCH)
#undef LAUNCH
}

void launch_lk_single(const DevBuffers& d, int slotA, int camA, int slotB, int camB, int n, const float2* prev, float2* next,
                      uint8_t* status, hipStream_t st) {
    if (n <= 0) return;
#define LAUNCH(Wn) if (d.cfg.win_w == Wn) { hipLaunchKernelGGL(k_lk_single<Wn>, dim3(n), dim3(64), 0, st, d, slotA, camA, slotB, camB, n, prev, next, status); return; }
    LK_FOR_EACH_WINDOW(LAUNCH)
#undef LAUNCH
}
