/* ORACLE (test infrastructure, see orc.h).  Bucket / FeatureSet bucketing.
 * Restates /root/reference/src/feature_set.cpp:1-53 (Bucket) and :91-152 (filterByBucketLocation*). */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

void orc_bucket_init(orc_bucket* b, int max_size) {
    b->max_size = max_size; b->n = 0;
    int c = max_size > 0 ? max_size : 1;
    b->xy = (float*)malloc(sizeof(float) * 2 * c);
    b->ages = (int*)malloc(sizeof(int) * c);
    b->strengths = (int*)malloc(sizeof(int) * c);
}
void orc_bucket_free(orc_bucket* b) { free(b->xy); free(b->ages); free(b->strengths); b->xy = NULL; b->ages = b->strengths = NULL; }

/* feature_set.cpp:16-18 — C++ int division truncates toward zero, as C does. */
int orc_bucket_compute_score(int age, int strength, int fast_threshold) {
    return age + (strength - fast_threshold) / 20;
}

/* feature_set.cpp:20-53 */
void orc_bucket_add_feature(orc_bucket* b, float x, float y, int age, int strength,
                            int age_threshold, int fast_threshold) {
    if (!b->max_size) return;                       /* :23 */
    if (age < age_threshold) {                      /* :26 */
        if (b->n < b->max_size) {                   /* :28 */
            b->xy[2 * b->n] = x; b->xy[2 * b->n + 1] = y;
            b->ages[b->n] = age; b->strengths[b->n] = strength; b->n++;
        } else {
            const int score = orc_bucket_compute_score(age, strength, fast_threshold);
            int score_min = orc_bucket_compute_score(b->ages[0], b->strengths[0], fast_threshold);
            int score_min_idx = 0, i;
            for (i = 1; i < b->n; i++) {            /* :37-44: FIRST minimum (strict <) */
                int s = orc_bucket_compute_score(b->ages[i], b->strengths[i], fast_threshold);
                if (s < score_min) { score_min = s; score_min_idx = i; }
            }
            if (score > score_min) {                /* :45: strictly greater */
                b->xy[2 * score_min_idx] = x; b->xy[2 * score_min_idx + 1] = y;
                b->ages[score_min_idx] = age; b->strengths[score_min_idx] = strength;
            }
        }
    }
}

/* feature_set.cpp:95-147 */
int orc_bucket_filter(int img_w, int img_h, int n, float* xy, int* ages, int* strengths,
                      int buckets_along_height, int buckets_along_width, int bucket_start_row,
                      int features_per_bucket, int age_threshold, int fast_threshold) {
    int bucket_height = (img_h + buckets_along_height - 1) / buckets_along_height;   /* :91-93,:103 */
    int bucket_width  = (img_w + buckets_along_width - 1) / buckets_along_width;
    int nb = buckets_along_height * buckets_along_width, i, r, c;
    orc_bucket* buckets = (orc_bucket*)malloc(sizeof(orc_bucket) * (size_t)nb);
    for (r = 0; r < buckets_along_height; r++)
        for (c = 0; c < buckets_along_width; c++)
            orc_bucket_init(&buckets[r * buckets_along_width + c], r >= bucket_start_row ? features_per_bucket : 0); /* :108-118 */
    for (i = 0; i < n; i++) {
        /* :122-124 — float / int -> float division, then truncation to int */
        int bh = (int)(xy[2 * i + 1] / (float)bucket_height);
        int bw = (int)(xy[2 * i] / (float)bucket_width);
        int idx = bh * buckets_along_width + bw;
        /* the reference indexes the vector unchecked (UB when out of range); the oracle drops such points */
        if (bh < 0 || bh >= buckets_along_height || bw < 0 || bw >= buckets_along_width) continue;
        orc_bucket_add_feature(&buckets[idx], xy[2 * i], xy[2 * i + 1], ages[i], strengths[i], age_threshold, fast_threshold);
    }
    int m = 0;
    for (i = 0; i < nb; i++) {                       /* :132-146 bucket-raster order */
        orc_bucket* b = &buckets[i];
        int k;
        for (k = 0; k < b->n; k++) {
            xy[2 * m] = b->xy[2 * k]; xy[2 * m + 1] = b->xy[2 * k + 1];
            ages[m] = b->ages[k]; strengths[m] = b->strengths[k]; m++;
        }
        orc_bucket_free(b);
    }
    free(buckets);
    return m;
}

/* vo.cpp:265-280 */
void orc_find_close_points(int n, const float* p1, const float* p2, float threshold, uint8_t* ok) {
    int i;
    for (i = 0; i < n; i++) {
        float dx = __builtin_fabsf(p1[2 * i] - p2[2 * i]), dy = __builtin_fabsf(p1[2 * i + 1] - p2[2 * i + 1]);
        float off = (dx < dy) ? dy : dx;             /* std::max(a,b) = (a<b)?b:a */
        ok[i] = (uint8_t)(off > threshold ? 0 : 1);
    }
}
