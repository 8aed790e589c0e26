/* ORACLE (test infrastructure, see orc.h).  Per-frame orchestrator.
 * Restates /root/reference/src/vo.cpp:8-26 (initalize_projection_matricies), :41-137 (stereo_callback),
 * :144-168 (delete*WithFailureStatus), :169-240 (circularMatching), :315-366 (matchingFeatures)
 * and /root/reference/src/feature_set.cpp:75-89 (appendFeaturesFromImage), quirks included
 * (SURVEY.md Appendix B-2, B-3, B-4, B-8, B-9). */
#include "orc.h"
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

/* worker threads for the data-parallel loops (LK points, image rows); returns the count in effect. n <= 0 = all cores */
int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n <= 0) n = omp_get_num_procs();
    omp_set_num_threads(n);
    return n;
#else
    (void)n; return 1;
#endif
}

void orc_config_default(orc_config* c) {
    c->bucket_start_row = 4; c->buckets_along_height = 92; c->buckets_along_width = 160;
    c->features_per_bucket = 1; c->features_threshold = 15; c->pre_matching_feature_threshold = 100;
    c->age_threshold = 20; c->fast_threshold = 20; c->ransac_reprojection_error = 8.f;
    c->ransac_iterations = 100; c->optical_flow_min_eig_threshold = 0.001;
    c->circular_matching_success_threshold = .15; c->max_translation_norm = .1; c->max_rotation_norm = .5;
    c->win_w = 10; c->win_h = 10; c->max_level = 3; c->lk_max_count = 30; c->lk_epsilon = 0.0001;
    c->ransac_confidence = 0.98f; c->max_features = 0; c->channels = 1; c->lk_float_sums = 0;
}

struct orc_vo {
    orc_config cfg;
    int frame_id;
    int w, h;
    int cn;                          /* channels of the images being fed (1, or 3 = the reference CLI's BGR input) */
    uint8_t *imgL0, *imgR0;          /* imageLeftT0_, imageRightT0_ (vo.h:239), interleaved, row stride w * cn */
    orc_pyramid pyrL0[ORC_MAX_CN], pyrR0[ORC_MAX_CN];   /* lastLeftPyramid, lastRightPyramid (vo.h:257-258), one per colour plane */
    int have_pyr;
    /* currentVOFeatures (vo.h:245) */
    int nf, capf; float* fxy; int* fage; int* fstr;
    double R[9], t[3], last_transform[16];   /* vo.h:266-268 */
    float Pl[12], Pr[12], K[9];
    /* last frame's tracks, for parity introspection */
    int nt; float *tl0, *tr0, *tl1, *tr1, *tworld; uint8_t* tinl;
};

static void ensure_cap(orc_vo* vo, int n) {
    if (n <= vo->capf) return;
    int c = n + 1024;
    vo->fxy = (float*)realloc(vo->fxy, sizeof(float) * 2 * (size_t)c);
    vo->fage = (int*)realloc(vo->fage, sizeof(int) * (size_t)c);
    vo->fstr = (int*)realloc(vo->fstr, sizeof(int) * (size_t)c);
    vo->capf = c;
}

orc_vo* orc_vo_create(const orc_config* cfg) {
    orc_vo* vo = (orc_vo*)calloc(1, sizeof(orc_vo));
    int i;
    if (cfg) vo->cfg = *cfg; else orc_config_default(&vo->cfg);
    for (i = 0; i < 9; i++) vo->R[i] = (i % 4 == 0);
    for (i = 0; i < 16; i++) vo->last_transform[i] = (i % 5 == 0);
    return vo;
}

static void free_tracks(orc_vo* vo) {
    free(vo->tl0); free(vo->tr0); free(vo->tl1); free(vo->tr1); free(vo->tworld); free(vo->tinl);
    vo->tl0 = vo->tr0 = vo->tl1 = vo->tr1 = vo->tworld = NULL; vo->tinl = NULL; vo->nt = 0;
}

void orc_vo_destroy(orc_vo* vo) {
    if (!vo) return;
    free(vo->imgL0); free(vo->imgR0);
    if (vo->have_pyr) { int k; for (k = 0; k < vo->cn; k++) { orc_pyramid_free(&vo->pyrL0[k]); orc_pyramid_free(&vo->pyrR0[k]); } }
    free(vo->fxy); free(vo->fage); free(vo->fstr);
    free_tracks(vo);
    free(vo);
}

void orc_vo_set_projection(orc_vo* vo, const float Pl[12], const float Pr[12]) {
    int i, j;
    memcpy(vo->Pl, Pl, sizeof(float) * 12); memcpy(vo->Pr, Pr, sizeof(float) * 12);
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) vo->K[3 * i + j] = Pl[4 * i + j];     /* vo.cpp:16-25 */
}

/* vo.cpp:144-168: stable removal; works on interleaved arrays of `elems` floats/ints per entry */
static int compact_f(float* a, int elems, int n, const uint8_t* ok) {
    int i, m = 0, k;
    for (i = 0; i < n; i++) if (ok[i]) { for (k = 0; k < elems; k++) a[elems * m + k] = a[elems * i + k]; m++; }
    return m;
}
static int compact_i(int* a, int n, const uint8_t* ok) {
    int i, m = 0;
    for (i = 0; i < n; i++) if (ok[i]) a[m++] = a[i];
    return m;
}
static void compact_features(orc_vo* vo, const uint8_t* ok, int n_ok_len) {
    /* deleteFeaturesWithFailureStatus iterates over is_ok.size() entries (vo.cpp:157) */
    int m;
    compact_f(vo->fxy, 2, n_ok_len, ok); compact_i(vo->fage, n_ok_len, ok); m = compact_i(vo->fstr, n_ok_len, ok);
    /* entries beyond is_ok.size() (none on this path) would be kept */
    vo->nf = m + (vo->nf - n_ok_len);
}

/* feature_set.cpp:75-89 */
/* img: the interleaved image; cv::FAST has no channel check and scans the first w BYTES of every row (Appendix B-1) */
static void append_features_from_image(orc_vo* vo, const uint8_t* img, int w, int h, int fast_threshold) {
    const orc_config* c = &vo->cfg;
    int cap = w * h / 4 + 16, i;
    float* xy = (float*)malloc(sizeof(float) * 2 * (size_t)cap);
    float* resp = (float*)malloc(sizeof(float) * (size_t)cap);
    int n = orc_fast_detect(img, w, h, w * vo->cn, fast_threshold, 1, cap, xy, resp);
    if (n > cap) n = cap;
    ensure_cap(vo, vo->nf + n);
    for (i = 0; i < n; i++) {
        vo->fxy[2 * (vo->nf + i)] = xy[2 * i]; vo->fxy[2 * (vo->nf + i) + 1] = xy[2 * i + 1];
        vo->fage[vo->nf + i] = 0;                       /* :84 */
        vo->fstr[vo->nf + i] = (int)resp[i];            /* :86-87 float -> int */
    }
    vo->nf += n;
    vo->nf = orc_bucket_filter(w, h, vo->nf, vo->fxy, vo->fage, vo->fstr, c->buckets_along_height, c->buckets_along_width,
                               c->bucket_start_row, c->features_per_bucket, c->age_threshold, c->fast_threshold);   /* :88 */
    free(xy); free(resp);
}

/* cn single-channel pyramids of an interleaved image (buildOpticalFlowPyramid on a cn-channel Mat works per channel) */
static void build_plane_pyramids(const uint8_t* img, int w, int h, int cn, const orc_config* c, orc_pyramid* out) {
    int k;
    if (cn == 1) { orc_build_pyramid(img, w, h, w, c->win_w, c->win_h, c->max_level, &out[0]); return; }
    uint8_t* plane = (uint8_t*)malloc((size_t)w * h);
    for (k = 0; k < cn; k++) {
        orc_extract_plane(img, w, h, w * cn, cn, k, plane);
        orc_build_pyramid(plane, w, h, w, c->win_w, c->win_h, c->max_level, &out[k]);
    }
    free(plane);
}

int orc_vo_stereo_callback(orc_vo* vo, const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                           double T_out[16], orc_frame_stats* st) {
    return orc_vo_stereo_callback_cn(vo, left, right, w, h, stride, vo->cfg.channels == 3 ? 3 : 1, T_out, st);
}

int orc_vo_stereo_callback_cn(orc_vo* vo, const uint8_t* left, const uint8_t* right, int w, int h, int stride, int cn,
                              double T_out[16], orc_frame_stats* st) {
    const orc_config* c = &vo->cfg;
    orc_frame_stats local; int i, y, k;
    if (cn != 1 && cn != ORC_MAX_CN) return -1;
    if (vo->frame_id > 0 && cn != vo->cn) return -1;
    if (!st) st = &local;
    memset(st, 0, sizeof(*st));
    memcpy(T_out, vo->last_transform, sizeof(double) * 16);          /* fail_result (vo.cpp:43-44) */
    /* contiguous copies of the inputs (T1) */
    const size_t rowb = (size_t)w * cn;
    uint8_t* L1 = (uint8_t*)malloc(rowb * h);
    uint8_t* R1 = (uint8_t*)malloc(rowb * h);
    for (y = 0; y < h; y++) { memcpy(L1 + (size_t)y * rowb, left + (size_t)y * stride, rowb); memcpy(R1 + (size_t)y * rowb, right + (size_t)y * stride, rowb); }

    if (vo->frame_id == 0) {                                          /* vo.cpp:47-56 */
        vo->w = w; vo->h = h; vo->cn = cn; vo->imgL0 = L1; vo->imgR0 = R1;
        build_plane_pyramids(L1, w, h, cn, c, vo->pyrL0);
        build_plane_pyramids(R1, w, h, cn, c, vo->pyrR0);
        vo->have_pyr = 1; vo->frame_id++;
        st->fail_reason = 1; st->n_features_out = vo->nf;
        return 0;
    }
    vo->frame_id++;
    free_tracks(vo);

    /* ---- matchingFeatures (vo.cpp:315-366) ---- */
    append_features_from_image(vo, vo->imgL0, w, h, c->fast_threshold);             /* :325 — FAST on the PREVIOUS left image */
    st->n_after_detect = vo->nf;
    if (vo->nf < c->pre_matching_feature_threshold) {                                /* :327-332 */
        append_features_from_image(vo, vo->imgL0, w, h, c->fast_threshold / 4);
        st->second_pass = 1; st->n_after_detect = vo->nf;
    }
    if (c->max_features > 0 && vo->nf > c->max_features) vo->nf = c->max_features;  /* build preset, see orc.h */
    int n = vo->nf;
    st->n_into_lk = n;
    float* pl0 = (float*)malloc(sizeof(float) * 2 * (size_t)(n + 1));
    float* pl1 = (float*)malloc(sizeof(float) * 2 * (size_t)(n + 1));
    float* pr1 = (float*)malloc(sizeof(float) * 2 * (size_t)(n + 1));
    float* pr0 = (float*)malloc(sizeof(float) * 2 * (size_t)(n + 1));
    float* plc = (float*)malloc(sizeof(float) * 2 * (size_t)(n + 1));
    uint8_t* ok = (uint8_t*)malloc((size_t)n + 1);
    memcpy(pl0, vo->fxy, sizeof(float) * 2 * (size_t)n);                             /* :338 */
    int nt = 0;
    if (n > 0) {                                                                     /* circularMatching :179-181 */
        orc_pyramid pl1p[ORC_MAX_CN], pr1p[ORC_MAX_CN];
        const orc_pyramid *l0[ORC_MAX_CN], *r0[ORC_MAX_CN], *l1[ORC_MAX_CN], *r1[ORC_MAX_CN];
        build_plane_pyramids(L1, w, h, cn, c, pl1p);                                 /* :200 */
        build_plane_pyramids(R1, w, h, cn, c, pr1p);                                 /* :201 */
        for (k = 0; k < cn; k++) { l0[k] = &vo->pyrL0[k]; r0[k] = &vo->pyrR0[k]; l1[k] = &pl1p[k]; r1[k] = &pr1p[k]; }
        {
            const long long v0 = orc_lk_counters[0], s0 = orc_lk_counters[1];       /* (one VisualOdometry at a time per process: test infrastructure) */
            {   /* cfg.lk_float_sums mirrors svo_config.lk_float_sums: deviation D1 reverted for this object's LK passes */
                const unsigned prev_mode = orc_get_opencv_mode();
                if (c->lk_float_sums) orc_set_opencv_mode(prev_mode | ORC_OCV_D1_LK_FLOAT);
                orc_circular_match_cn(cn, l0, r0, l1, r1, n, pl0, pl1, pr1, pr0, plc, ok, c);   /* :203-230 */
                orc_set_opencv_mode(prev_mode);
            }
            (void)v0; (void)s0;
            st->lk_level_visits = (int)orc_last_chain_stats.level_visits; st->lk_newton_steps = (int)orc_last_chain_stats.newton_steps;
            for (k = 0; k < 3; k++) st->lk_dead_after_pass[k] = orc_last_chain_stats.dead_after_pass[k];
        }
        for (k = 0; k < cn; k++) {
            orc_pyramid_free(&vo->pyrL0[k]); orc_pyramid_free(&vo->pyrR0[k]);
            vo->pyrL0[k] = pl1p[k]; vo->pyrR0[k] = pr1p[k];                          /* :231-232 */
        }
        compact_features(vo, ok, n);                                                 /* :233 */
        compact_f(pl0, 2, n, ok); compact_f(pl1, 2, n, ok); compact_f(pr1, 2, n, ok); nt = compact_f(pr0, 2, n, ok);   /* :234-238 */
        st->n_after_circular = nt;
        /* in-image bounds over the four point sets (:341-359) */
        for (i = 0; i < nt; i++) {
            const float* P[4] = {pl0 + 2 * i, pl1 + 2 * i, pr0 + 2 * i, pr1 + 2 * i};
            int k, good = 1;
            for (k = 0; k < 4; k++)
                if ((P[k][0] < 0) || (P[k][1] < 0) || (P[k][1] >= h) || (P[k][0] >= w)) good = 0;
            ok[i] = (uint8_t)good;
        }
        compact_features(vo, ok, nt);                                                /* :360 */
        compact_f(pl0, 2, nt, ok); compact_f(pl1, 2, nt, ok); compact_f(pr0, 2, nt, ok); nt = compact_f(pr1, 2, nt, ok);
    }
    st->n_after_bounds = nt;
    for (i = 0; i < vo->nf; i++) vo->fage[i] += 1;                                   /* vo.cpp:70-72 */
    free(vo->imgL0); free(vo->imgR0); vo->imgL0 = L1; vo->imgR0 = R1;               /* :74-75 */

    int result = 0;
    vo->nt = nt; vo->tl0 = pl0; vo->tr0 = pr0; vo->tl1 = pl1; vo->tr1 = pr1;
    vo->tworld = (float*)calloc((size_t)nt * 3 + 3, sizeof(float));
    vo->tinl = (uint8_t*)calloc((size_t)nt + 1, 1);
    free(plc); free(ok);
    if (nt <= (4 > c->features_threshold ? 4 : c->features_threshold)) {             /* :82-84 */
        st->fail_reason = 2; st->n_features_out = vo->nf;
        return 0;
    }
    orc_triangulate(vo->Pl, vo->Pr, nt, pl0, pr0, vo->tworld, NULL);                 /* :89-94 */
    int* inl = (int*)malloc(sizeof(int) * (size_t)nt);
    int n_inl = 0, dbg[2];
    int success = orc_camera_to_world(vo->K, nt, pl1, vo->tworld, vo->R, vo->t, inl, &n_inl,
                                      c->ransac_iterations, c->ransac_reprojection_error, c->ransac_confidence, dbg);   /* :101-104 */
    st->n_inliers = n_inl; st->ransac_iters = dbg[0];
    if (n_inl < c->features_threshold || !success) {                                 /* :106-113 */
        st->fail_reason = 3; st->n_features_out = vo->nf; free(inl);
        return 0;
    }
    for (i = 0; i < n_inl; i++) vo->tinl[inl[i]] = 1;                                /* :115-119 */
    memcpy(vo->fxy, pl1, sizeof(float) * 2 * (size_t)nt);                            /* :120 */
    compact_features(vo, vo->tinl, nt);                                              /* :121 */
    free(inl);
    st->n_features_out = vo->nf;
    {
        double rv[3];
        double tn = sqrt(vo->t[0] * vo->t[0] + vo->t[1] * vo->t[1] + vo->t[2] * vo->t[2]);    /* :124 */
        orc_rodrigues_to_vector(vo->R, rv);                                                     /* :125 */
        double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);                    /* :126 */
        if (tn > c->max_translation_norm || angle > c->max_rotation_norm) {                     /* :129-132 */
            st->fail_reason = 4;
            return 0;
        }
    }
    orc_inverse_transform(vo->R, vo->t, vo->last_transform);                         /* :133-135 */
    memcpy(T_out, vo->last_transform, sizeof(double) * 16);
    result = 1;
    return result;
}

int orc_vo_num_features(const orc_vo* vo) { return vo->nf; }
void orc_vo_get_features(const orc_vo* vo, float* xy, int* ages, int* strengths) {
    memcpy(xy, vo->fxy, sizeof(float) * 2 * (size_t)vo->nf);
    memcpy(ages, vo->fage, sizeof(int) * (size_t)vo->nf);
    memcpy(strengths, vo->fstr, sizeof(int) * (size_t)vo->nf);
}
void orc_vo_get_pose_guess(const orc_vo* vo, double R[9], double t[3]) { memcpy(R, vo->R, sizeof(double) * 9); memcpy(t, vo->t, sizeof(double) * 3); }
int orc_vo_get_last_tracks(const orc_vo* vo, float* pl0, float* pr0, float* pl1, float* pr1, float* world, uint8_t* inlier) {
    size_t n = (size_t)vo->nt;
    if (pl0) memcpy(pl0, vo->tl0, sizeof(float) * 2 * n);
    if (pr0) memcpy(pr0, vo->tr0, sizeof(float) * 2 * n);
    if (pl1) memcpy(pl1, vo->tl1, sizeof(float) * 2 * n);
    if (pr1) memcpy(pr1, vo->tr1, sizeof(float) * 2 * n);
    if (world) memcpy(world, vo->tworld, sizeof(float) * 3 * n);
    if (inlier) memcpy(inlier, vo->tinl, n);
    return vo->nt;
}
