/* ORACLE (test infrastructure, see orc.h).  Optical-flow pyramid + pyramidal Lucas–Kanade.
 * Stands in for cv::buildOpticalFlowPyramid (/root/reference/src/vo.cpp:50,52,200,201) and
 * cv::calcOpticalFlowPyrLK (vo.cpp:203-215), restating OpenCV 4.5 modules/video/src/lkpyramid.cpp
 * and modules/imgproc/src/pyramids.cpp as summarised in SURVEY.md Appendix A.2/A.3.
 * Deviation D1 (orc.h): A and b are accumulated as exact int64 sums. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

/* pyrDown, 8-bit: separable [1 4 6 4 1], (sum + 128) >> 8, source border REFLECT_101. */
void orc_pyr_down(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dstride) {
    int dw = (sw + 1) / 2, dh = (sh + 1) / 2, y;
#pragma omp parallel
    {
    int x, k;
    int* rows = (int*)malloc(sizeof(int) * (size_t)dw * 5);
#pragma omp for
    for (y = 0; y < dh; y++) {
        for (k = 0; k < 5; k++) {
            int sy = reflect101(2 * y + k - 2, sh);
            const uint8_t* s = src + (size_t)sy * sstride;
            int* r = rows + (size_t)k * dw;
            for (x = 0; x < dw; x++) {
                int x0 = reflect101(2 * x - 2, sw), x1 = reflect101(2 * x - 1, sw), x2 = 2 * x,
                    x3 = reflect101(2 * x + 1, sw), x4 = reflect101(2 * x + 2, sw);
                r[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        for (x = 0; x < dw; x++) {
            int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[x] + rows[4 * dw + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
    }
}

/* calcSharrDeriv: int16 interleaved (dx,dy); REFLECT_101 at the image edges. */
void orc_scharr(const uint8_t* src, int w, int h, int sstride, int16_t* dst, int dstride) {
    int y;
#pragma omp parallel
    {
    int x;
    int* t0 = (int*)malloc(sizeof(int) * (size_t)(w + 2));
    int* t1 = (int*)malloc(sizeof(int) * (size_t)(w + 2));
#pragma omp for
    for (y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * sstride;
        const uint8_t* r1 = src + (size_t)y * sstride;
        const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * sstride;
        for (x = 0; x < w; x++) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        /* border of the intermediate rows: REFLECT_101 in x */
        t0[0] = t0[1 + reflect101(-1, w)]; t0[w + 1] = t0[1 + reflect101(w, w)];
        t1[0] = t1[1 + reflect101(-1, w)]; t1[w + 1] = t1[1 + reflect101(w, w)];
        int16_t* d = dst + (size_t)y * dstride;
        for (x = 0; x < w; x++) {
            d[2 * x]     = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
    }
}

static void make_level(orc_pyramid* p, int lvl, int w, int h) {
    int pw = w + 2 * p->pad_x, ph = h + 2 * p->pad_y;
    uint8_t* ibuf = (uint8_t*)malloc((size_t)pw * ph);
    int16_t* dbuf = (int16_t*)calloc((size_t)pw * ph * 2, sizeof(int16_t));   /* deriv border = CONSTANT 0 */
    p->owned[2 * lvl] = ibuf; p->owned[2 * lvl + 1] = dbuf;
    p->w[lvl] = w; p->h[lvl] = h;
    p->img_stride[lvl] = pw; p->deriv_stride[lvl] = pw * 2;
    p->img[lvl] = ibuf + (size_t)p->pad_y * pw + p->pad_x;
    p->deriv[lvl] = dbuf + ((size_t)p->pad_y * pw + p->pad_x) * 2;
}

static void fill_border_reflect101(orc_pyramid* p, int lvl) {
    int w = p->w[lvl], h = p->h[lvl], st = p->img_stride[lvl], x, y;
    uint8_t* im = p->img[lvl];
    for (y = -p->pad_y; y < h + p->pad_y; y++) {
        int sy = reflect101(y, h);
        for (x = -p->pad_x; x < w + p->pad_x; x++) {
            if (x >= 0 && x < w && y >= 0 && y < h) continue;
            im[(ptrdiff_t)y * st + x] = im[(ptrdiff_t)sy * st + reflect101(x, w)];
        }
    }
}

void orc_build_pyramid(const uint8_t* img, int w, int h, int stride, int win_w, int win_h, int max_level, orc_pyramid* p) {
    int lvl, y;
    memset(p, 0, sizeof(*p));
    p->pad_x = win_w; p->pad_y = win_h;
    if (max_level > ORC_MAX_LEVELS - 1) max_level = ORC_MAX_LEVELS - 1;
    int cw = w, ch = h;
    for (lvl = 0; lvl <= max_level; lvl++) {
        make_level(p, lvl, cw, ch);
        if (lvl == 0) {
            for (y = 0; y < h; y++) memcpy(p->img[0] + (size_t)y * p->img_stride[0], img + (size_t)y * stride, (size_t)w);
        } else {
            orc_pyr_down(p->img[lvl - 1], p->w[lvl - 1], p->h[lvl - 1], p->img_stride[lvl - 1], p->img[lvl], p->img_stride[lvl]);
        }
        fill_border_reflect101(p, lvl);
        orc_scharr(p->img[lvl], cw, ch, p->img_stride[lvl], p->deriv[lvl], p->deriv_stride[lvl]);
        p->nlevels = lvl + 1;
        cw = (cw + 1) / 2; ch = (ch + 1) / 2;
        if (cw <= win_w || ch <= win_h) break;     /* lkpyramid.cpp: stop when the NEXT level would not exceed the window */
    }
}

void orc_pyramid_free(orc_pyramid* p) {
    int i;
    for (i = 0; i < 2 * ORC_MAX_LEVELS; i++) { free(p->owned[i]); p->owned[i] = NULL; }
    p->nlevels = 0;
}

/* debug counters (test infrastructure): [0] level visits that reached the Newton loop, [1] Newton iterations, [2] level visits total */
long long orc_lk_counters[4] = {0, 0, 0, 0};
/* optional per-point work arrays for the pass in progress (set by orc_circular_match_cn, NULL otherwise): += visits / steps of point i */
static int* g_pt_visits = NULL;
static int* g_pt_steps = NULL;
orc_lk_chain_stats orc_last_chain_stats = {0, 0, {0, 0, 0}};

#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* ---- deviation switches (orc.h) ---- */
static unsigned g_ocv_mode = 0;
unsigned orc_set_opencv_mode(unsigned mask) { unsigned old = g_ocv_mode; g_ocv_mode = mask; return old; }
unsigned orc_get_opencv_mode(void) { return g_ocv_mode; }

/* D1 reverted: the sums of LKTrackerInvoker in FLOAT, in the order OpenCV 4.5's lkpyramid.cpp produces on x86
 * (`#if CV_SIMD128 && !CV_NEON`, acctype = itemtype = float).  A window row is E = w*cn interleaved elements
 * (element e = pixel e/cn, channel e%cn).  Patches are stored per plane: element e of row y is [(c*wh + y)*ww + px].
 *   A:  `for (; x <= E - 8; x += 8)`: elements x..x+3 then x+4..x+7 go to float lanes 0..3 of qA11/qA12/qA22 by
 *       v_muladd(fx, fy, q) (unfused at an SSE baseline); the scalar tail adds (float)(ix*iy) to iA; after the rows
 *       iA += v_reduce_sum(q) = (q0 + q2) + (q1 + q3).
 *   b:  per 8 elements v_dotprod pairs element k with k+4 in int32, v_cvt_f32, added to qb0 (k = 0, 1) / qb1 (k = 2, 3),
 *       lanes (bx, by, bx, by); at the end s = qb0 + qb1, ib1 += s0 + s2, ib2 += s1 + s3.
 * Variants (orc.h): FMA, the 3.x SSE2 form (A in steps of 4, reduce ((q0+q1)+q2)+q3), no SIMD. */
static inline float mul_add_f(float a, float b, float c, int fma) { return fma ? fmaf(a, b, c) : a * b + c; }

static void d1_sum_A(unsigned mode, int cn, int ww, int wh, const int16_t* dIw, float* A11, float* A12, float* A22) {
    const int E = ww * cn, fma = (mode & ORC_OCV_D1_FMA) != 0;
    const int step = (mode & ORC_OCV_D1_SCALAR) ? 0 : (mode & ORC_OCV_D1_W4) ? 4 : 8;
    const int nsimd = step ? (E / step) * step : 0;
    float q11[4] = {0, 0, 0, 0}, q12[4] = {0, 0, 0, 0}, q22[4] = {0, 0, 0, 0}, t11 = 0, t12 = 0, t22 = 0;
    int y, e;
    for (y = 0; y < wh; y++)
        for (e = 0; e < E; e++) {
            const int16_t* d = dIw + ((size_t)((e % cn) * wh + y) * ww + e / cn) * 2;
            int ix = d[0], iy = d[1];
            if (e < nsimd) {
                float fx = (float)ix, fy = (float)iy; int L = e & 3;
                q22[L] = mul_add_f(fy, fy, q22[L], fma); q12[L] = mul_add_f(fx, fy, q12[L], fma); q11[L] = mul_add_f(fx, fx, q11[L], fma);
            } else { t11 += (float)(ix * ix); t12 += (float)(ix * iy); t22 += (float)(iy * iy); }
        }
    if (mode & ORC_OCV_D1_W4) {
        t11 += q11[0] + q11[1] + q11[2] + q11[3]; t12 += q12[0] + q12[1] + q12[2] + q12[3]; t22 += q22[0] + q22[1] + q22[2] + q22[3];
    } else {
        t11 += (q11[0] + q11[2]) + (q11[1] + q11[3]); t12 += (q12[0] + q12[2]) + (q12[1] + q12[3]); t22 += (q22[0] + q22[2]) + (q22[1] + q22[3]);
    }
    *A11 = t11; *A12 = t12; *A22 = t22;
}

static void d1_sum_b(unsigned mode, int cn, int ww, int wh, const int16_t* dIw, const int16_t* diffw, float* b1, float* b2) {
    const int E = ww * cn;
    const int nsimd = (mode & ORC_OCV_D1_SCALAR) ? 0 : (E / 8) * 8;
    float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0}, t1 = 0, t2 = 0;
    int y, e, k;
    for (y = 0; y < wh; y++) {
        int it[8], ix[8], iy[8];
        for (e = 0; e < E; e++) {
            size_t at = (size_t)((e % cn) * wh + y) * ww + e / cn;
            int dv = diffw[at], dx = dIw[2 * at], dy = dIw[2 * at + 1];
            if (e < nsimd) {
                k = e & 7; it[k] = dv; ix[k] = dx; iy[k] = dy;
                if (k == 7) {
                    qb0[0] += (float)(it[0] * ix[0] + it[4] * ix[4]); qb0[1] += (float)(it[0] * iy[0] + it[4] * iy[4]);
                    qb0[2] += (float)(it[1] * ix[1] + it[5] * ix[5]); qb0[3] += (float)(it[1] * iy[1] + it[5] * iy[5]);
                    qb1[0] += (float)(it[2] * ix[2] + it[6] * ix[6]); qb1[1] += (float)(it[2] * iy[2] + it[6] * iy[6]);
                    qb1[2] += (float)(it[3] * ix[3] + it[7] * ix[7]); qb1[3] += (float)(it[3] * iy[3] + it[7] * iy[7]);
                }
            } else { t1 += (float)(dv * dx); t2 += (float)(dv * dy); }
        }
    }
    {
        float s0 = qb0[0] + qb1[0], s1 = qb0[1] + qb1[1], s2 = qb0[2] + qb1[2], s3 = qb0[3] + qb1[3];
        t1 += s0 + s2; t2 += s1 + s3;
    }
    *b1 = t1; *b2 = t2;
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }     /* round-half-even under the default mode */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* One pyramid level of LKTrackerInvoker for all points. */
static void lk_level(int cn, const orc_pyramid* const* Apl, const orc_pyramid* const* Bpl, int level, int max_level, int n,
                     const float* prev_pts, float* next_pts, uint8_t* status,
                     int ww, int wh, int max_count, double epsilon, double min_eig_threshold) {
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;
    const orc_pyramid* A = Apl[0]; const orc_pyramid* B = Bpl[0];          /* all planes share the geometry */
    const int stepI = A->img_stride[level];
    const int dstep = A->deriv_stride[level];
    const int stepJ = B->img_stride[level];
    const int colsI = A->w[level], rowsI = A->h[level], colsJ = B->w[level], rowsJ = B->h[level];
    const float FLT_SCALE = 1.f / (1 << 20);
    const float scale = (float)(1. / (1 << level));
    int i;

    /* points are independent (cv::parallel_for_ over points in calcOpticalFlowPyrLK): one patch buffer per thread */
#pragma omp parallel
    {
    long long cnt_visit = 0, cnt_step = 0, cnt_all = 0;                  /* per-thread, added to the globals once below */
    int16_t* Iw = (int16_t*)malloc(sizeof(int16_t) * (size_t)ww * wh * 4 * cn);
    int16_t* dIw = Iw + (size_t)ww * wh * cn;
    int16_t* diffw = Iw + (size_t)ww * wh * cn * 3;       /* mismatch patch, used by the D1-reverted float sums only */
    const unsigned ocv = g_ocv_mode;
    int x, y, j, pc;
#pragma omp for schedule(dynamic, 16)
    for (i = 0; i < n; i++) {
        float ppx = prev_pts[2 * i] * scale, ppy = prev_pts[2 * i + 1] * scale;
        float npx, npy;
        if (level == max_level) { npx = ppx; npy = ppy; }
        else { npx = next_pts[2 * i] * 2.f; npy = next_pts[2 * i + 1] * 2.f; }
        next_pts[2 * i] = npx; next_pts[2 * i + 1] = npy;

        cnt_all++;
        ppx -= half_x; ppy -= half_y;
        int ipx = cv_floor_f(ppx), ipy = cv_floor_f(ppy);
        if (ipx < -ww || ipx >= colsI || ipy < -wh || ipy >= rowsI) {
            if (level == 0) status[i] = 0;
            continue;
        }
        float a = ppx - ipx, b = ppy - ipy;
        int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
        int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t iA11 = 0, iA12 = 0, iA22 = 0;
        for (pc = 0; pc < cn; pc++)                       /* window sums run over every channel (lkpyramid.cpp: x < winSize.width*cn) */
        for (y = 0; y < wh; y++) {
            const uint8_t* src = Apl[pc]->img[level] + (ptrdiff_t)(y + ipy) * stepI + ipx;
            const int16_t* ds = Apl[pc]->deriv[level] + (ptrdiff_t)(y + ipy) * dstep + ipx * 2;
            int16_t* Ip = Iw + (size_t)(pc * wh + y) * ww;
            int16_t* dp = dIw + (size_t)(pc * wh + y) * ww * 2;
            for (x = 0; x < ww; x++, ds += 2, dp += 2) {
                int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11, W_BITS - 5);
                int ixval = DESCALE(ds[0] * iw00 + ds[2] * iw01 + ds[dstep] * iw10 + ds[dstep + 2] * iw11, W_BITS);
                int iyval = DESCALE(ds[1] * iw00 + ds[3] * iw01 + ds[dstep + 1] * iw10 + ds[dstep + 3] * iw11, W_BITS);
                Ip[x] = (int16_t)ival; dp[0] = (int16_t)ixval; dp[1] = (int16_t)iyval;
                iA11 += (int64_t)(ixval * ixval); iA12 += (int64_t)(ixval * iyval); iA22 += (int64_t)(iyval * iyval);
            }
        }
        /* |sum| < 2^53: int64 -> double is exact, double -> float rounds once (nearest-even) */
        float A11 = (float)(double)iA11 * FLT_SCALE, A12 = (float)(double)iA12 * FLT_SCALE, A22 = (float)(double)iA22 * FLT_SCALE;
        if (ocv & ORC_OCV_D1_LK_FLOAT) {
            d1_sum_A(ocv, cn, ww, wh, dIw, &A11, &A12, &A22);
            A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
        }
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
        if ((double)minEig < min_eig_threshold || D < 1.1920928955078125e-07f /* FLT_EPSILON */) {
            if (level == 0) status[i] = 0;
            continue;
        }
        D = 1.f / D;
        cnt_visit++;
        if (g_pt_visits) g_pt_visits[i]++;
        npx -= half_x; npy -= half_y;
        float pdx = 0.f, pdy = 0.f;
        for (j = 0; j < max_count; j++) {
            int inx = cv_floor_f(npx), iny = cv_floor_f(npy);
            if (inx < -ww || inx >= colsJ || iny < -wh || iny >= rowsJ) {
                if (level == 0) status[i] = 0;
                break;
            }
            cnt_step++;
            if (g_pt_steps) g_pt_steps[i]++;
            a = npx - inx; b = npy - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t ib1 = 0, ib2 = 0;
            for (pc = 0; pc < cn; pc++)
            for (y = 0; y < wh; y++) {
                const uint8_t* Jp = Bpl[pc]->img[level] + (ptrdiff_t)(y + iny) * stepJ + inx;
                const int16_t* Ip = Iw + (size_t)(pc * wh + y) * ww;
                const int16_t* dp = dIw + (size_t)(pc * wh + y) * ww * 2;
                for (x = 0; x < ww; x++, dp += 2) {
                    int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                    ib1 += (int64_t)(diff * dp[0]); ib2 += (int64_t)(diff * dp[1]);
                    diffw[(size_t)(pc * wh + y) * ww + x] = (int16_t)diff;
                }
            }
            float b1 = (float)(double)ib1 * FLT_SCALE, b2 = (float)(double)ib2 * FLT_SCALE;
            if (ocv & ORC_OCV_D1_LK_FLOAT) {
                d1_sum_b(ocv, cn, ww, wh, dIw, diffw, &b1, &b2);
                b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            }
            float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            npx += dx; npy += dy;
            next_pts[2 * i] = npx + half_x; next_pts[2 * i + 1] = npy + half_y;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                next_pts[2 * i] -= dx * 0.5f; next_pts[2 * i + 1] -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        /* flags=0 and err!=NULL (vo.cpp:182,203): the level-0 error block re-checks the final window origin */
        if (status[i] && level == 0) {
            float fx = next_pts[2 * i] - half_x, fy = next_pts[2 * i + 1] - half_y;
            int ix = cv_floor_f(fx), iy = cv_floor_f(fy);
            if (ix < -ww || ix >= colsJ || iy < -wh || iy >= rowsJ) status[i] = 0;
        }
    }
    free(Iw);
#pragma omp atomic
    orc_lk_counters[0] += cnt_visit;
#pragma omp atomic
    orc_lk_counters[1] += cnt_step;
#pragma omp atomic
    orc_lk_counters[2] += cnt_all;
    }
}

void orc_lk_track_cn(int cn, const orc_pyramid* const* prevs, const orc_pyramid* const* nexts, int n, const float* prev_pts,
                  float* next_pts, uint8_t* status, int win_w, int win_h, int max_level,
                  int max_count, double epsilon, double min_eig_threshold) {
    int level, i;
    const orc_pyramid* prev = prevs[0]; const orc_pyramid* next = nexts[0];
    if (max_level > prev->nlevels - 1) max_level = prev->nlevels - 1;
    if (max_level > next->nlevels - 1) max_level = next->nlevels - 1;
    /* TermCriteria normalisation (lkpyramid.cpp) */
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (epsilon < 0.) epsilon = 0.;
    if (epsilon > 10.) epsilon = 10.;
    epsilon *= epsilon;
    for (i = 0; i < n; i++) status[i] = 1;
    for (level = max_level; level >= 0; level--)
        lk_level(cn, prevs, nexts, level, max_level, n, prev_pts, next_pts, status, win_w, win_h, max_count, epsilon, min_eig_threshold);
}

/* vo.cpp:169-240 without the compaction (the mask is returned) */
void orc_lk_track(const orc_pyramid* prev, const orc_pyramid* next, int n, const float* prev_pts,
                  float* next_pts, uint8_t* status, int win_w, int win_h, int max_level,
                  int max_count, double epsilon, double min_eig_threshold) {
    orc_lk_track_cn(1, &prev, &next, n, prev_pts, next_pts, status, win_w, win_h, max_level, max_count, epsilon, min_eig_threshold);
}

void orc_extract_plane(const uint8_t* img, int w, int h, int stride, int cn, int k, uint8_t* plane) {
    int x, y;
    for (y = 0; y < h; y++) {
        const uint8_t* r = img + (size_t)y * stride + k;
        uint8_t* d = plane + (size_t)y * w;
        for (x = 0; x < w; x++) d[x] = r[(size_t)x * cn];
    }
}

void orc_circular_match_cn(int cn, const orc_pyramid* const* l0, const orc_pyramid* const* r0, const orc_pyramid* const* l1, const orc_pyramid* const* r1,
                        int n, const float* pl0, float* pl1, float* pr1, float* pr0, float* pl0_circle,
                        uint8_t* ok, const orc_config* cfg) {
    uint8_t* st = (uint8_t*)malloc((size_t)n * 4 + 4);
    int* work = (int*)calloc((size_t)n * 8 + 8, sizeof(int));       /* [pass][visits | steps][point] */
    int i, k;
#define PASS(K, A, B, FROM, TO) g_pt_visits = work + (size_t)(2 * (K)) * n; g_pt_steps = work + (size_t)(2 * (K) + 1) * n; \
    orc_lk_track_cn(cn, A, B, n, FROM, TO, st + (size_t)(K) * n, cfg->win_w, cfg->win_h, cfg->max_level, cfg->lk_max_count, cfg->lk_epsilon, cfg->optical_flow_min_eig_threshold)
    PASS(0, l0, l1, pl0, pl1);          /* :203 */
    PASS(1, l1, r1, pl1, pr1);          /* :206 */
    PASS(2, r1, r0, pr1, pr0);          /* :209 */
    PASS(3, r0, l0, pr0, pl0_circle);   /* :213 */
#undef PASS
    g_pt_visits = g_pt_steps = NULL;
    orc_find_close_points(n, pl0, pl0_circle, (float)cfg->circular_matching_success_threshold, ok);   /* :217-219 (float32 threshold parameter) */
    for (i = 0; i < n; i++) ok[i] = (uint8_t)(st[i] && st[n + i] && st[2 * n + i] && st[3 * n + i] && ok[i]);  /* :227-230 */
    /* Work accounting of the frame pipeline (orc_frame_stats): a feature whose status is 0 after pass k is deleted at
       vo.cpp:233-238 whatever the later passes return, so the HIP path does not run them; the counters count the passes up to and
       including the first one that failed, and dead_after_pass[k] the features that first failed in pass k = 0, 1, 2. */
    memset(&orc_last_chain_stats, 0, sizeof(orc_last_chain_stats));
    for (i = 0; i < n; i++)
        for (k = 0; k < 4; k++) {
            orc_last_chain_stats.level_visits += work[(size_t)(2 * k) * n + i];
            orc_last_chain_stats.newton_steps += work[(size_t)(2 * k + 1) * n + i];
            if (!st[(size_t)k * n + i]) { if (k < 3) orc_last_chain_stats.dead_after_pass[k]++; break; }
        }
    free(work);
    free(st);
}

void orc_circular_match(const orc_pyramid* l0, const orc_pyramid* r0, const orc_pyramid* l1, const orc_pyramid* r1,
                        int n, const float* pl0, float* pl1, float* pr1, float* pr0, float* pl0_circle,
                        uint8_t* ok, const orc_config* cfg) {
    orc_circular_match_cn(1, &l0, &r0, &l1, &r1, n, pl0, pl1, pr1, pr0, pl0_circle, ok, cfg);
}
