/* ORACLE (test infrastructure, see orc.h).  FAST-9/16 corner detector with 3x3 non-max suppression.
 * Stands in for cv::FAST(image, keypoints, threshold, true) called at
 * /root/reference/src/feature_set.cpp:61, restating OpenCV 4.5 modules/features2d/src/fast.cpp
 * (FAST_t<16>) and fast_score.cpp (cornerScore<16>) as summarised in SURVEY.md Appendix A.1. */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

/* Bresenham circle r=3, OpenCV order (fast_score.cpp offsets16). */
static const int CIRC[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* Is (x,y) a FAST-9 corner?  >= 9 contiguous circle pixels all < v-t or all > v+t (count > K=8). */
static int is_corner(const uint8_t* p, const int off[25], int t) {
    int v = p[0], k, count;
    int vt = v - t;
    for (k = 0, count = 0; k < 25; k++) {
        if (p[off[k]] < vt) { if (++count > 8) return 1; } else count = 0;
    }
    vt = v + t;
    for (k = 0, count = 0; k < 25; k++) {
        if (p[off[k]] > vt) { if (++count > 8) return 1; } else count = 0;
    }
    return 0;
}

/* cornerScore<16>: max(t, best 9-arc min |diff|) - 1 */
static int corner_score(const uint8_t* p, const int off[25], int t) {
    int d[25], k, v = p[0];
    for (k = 0; k < 25; k++) d[k] = v - p[off[k]];
    int a0 = t;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1], j;
        for (j = 2; j <= 8; j++) if (d[k + j] < a) a = d[k + j];
        int m0 = a < d[k] ? a : d[k];
        int m1 = a < d[k + 9] ? a : d[k + 9];
        if (m0 > a0) a0 = m0;
        if (m1 > a0) a0 = m1;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1], j;
        for (j = 2; j <= 8; j++) if (d[k + j] > b) b = d[k + j];
        int m0 = b > d[k] ? b : d[k];
        int m1 = b > d[k + 9] ? b : d[k + 9];
        if (m0 < b0) b0 = m0;
        if (m1 < b0) b0 = m1;
    }
    return -b0 - 1;
}

/* raw[] = u8 corner score at every tested pixel (0 = not a corner); tested rows 3..h-4, cols 3..w-4 */
static void fast_raw_scores(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* raw) {
    int off[25], k, x, y;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    for (k = 0; k < 16; k++) off[k] = CIRC[k][0] + CIRC[k][1] * stride;
    for (k = 16; k < 25; k++) off[k] = off[k - 16];
    memset(raw, 0, (size_t)w * h);
#pragma omp parallel for private(x)
    for (y = 3; y < h - 3; y++)
        for (x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (is_corner(p, off, threshold))
                raw[(size_t)y * w + x] = (uint8_t)corner_score(p, off, threshold);
        }
}

void orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, uint8_t* score) {
    uint8_t* raw = (uint8_t*)malloc((size_t)w * h);
    int x, y;
    fast_raw_scores(img, w, h, stride, threshold, raw);
    if (!nonmax) { memcpy(score, raw, (size_t)w * h); free(raw); return; }
    memset(score, 0, (size_t)w * h);
    /* strict '>' against all 8 neighbours; raw is 0 outside the tested band so the band edge behaves
       like OpenCV's zeroed ring-buffer rows (fast.cpp: memset(curr,0,cols) and the i==rows-3 pass). */
    for (y = 3; y < h - 3; y++)
        for (x = 3; x < w - 3; x++) {
            const uint8_t* r = raw + (size_t)y * w + x;
            int s = r[0];
            if (!s) continue;   /* not a corner (corner scores are >= threshold-1; a 0 score corner cannot win '>' anyway) */
            if (s > r[-1] && s > r[1] && s > r[-w - 1] && s > r[-w] && s > r[-w + 1] &&
                s > r[w - 1] && s > r[w] && s > r[w + 1])
                score[(size_t)y * w + x] = (uint8_t)s;
        }
    free(raw);
}

int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax,
                    int cap, float* xy, float* resp) {
    int n = 0, x, y;
    if (w < 7 || h < 7) return 0;
    uint8_t* raw = (uint8_t*)malloc((size_t)w * h);
    uint8_t* isc = NULL;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    fast_raw_scores(img, w, h, stride, threshold, raw);
    if (threshold == 0 || !nonmax) {
        /* a corner can legitimately score 0 only when threshold==0 (score = max(t,..)-1 >= t-1 would be -1 -> u8 255
           in OpenCV's cast; keep a separate corner mask so such pixels are still visited). */
        int off[25], k;
        isc = (uint8_t*)calloc((size_t)w * h, 1);
        for (k = 0; k < 16; k++) off[k] = CIRC[k][0] + CIRC[k][1] * stride;
        for (k = 16; k < 25; k++) off[k] = off[k - 16];
        for (y = 3; y < h - 3; y++)
            for (x = 3; x < w - 3; x++)
                isc[(size_t)y * w + x] = (uint8_t)is_corner(img + (size_t)y * stride + x, off, threshold);
    }
    for (y = 3; y < h - 3; y++)
        for (x = 3; x < w - 3; x++) {
            const uint8_t* r = raw + (size_t)y * w + x;
            int s = r[0];
            int corner = isc ? isc[(size_t)y * w + x] : (s != 0);
            if (!corner) continue;
            if (!nonmax || (s > r[-1] && s > r[1] && s > r[-w - 1] && s > r[-w] && s > r[-w + 1] &&
                            s > r[w - 1] && s > r[w] && s > r[w + 1])) {
                if (n < cap) { xy[2 * n] = (float)x; xy[2 * n + 1] = (float)y; resp[n] = (float)s; }
                n++;
            }
        }
    free(raw);
    free(isc);
    return n;
}
