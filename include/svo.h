/*
 * svo.h — C-ABI of the MI355X-native stereo-VO per-frame front end (libsvo_hip.so).
 *
 * This is the drop-in boundary for the hot path of Alex7Li/stereo_visual_odometry:
 *   VisualOdometry::stereo_callback  (reference src/vo.cpp:41-137, decl include/vo.h:333-334)
 * and the stage functions its own tests call directly (src/main.cpp:110,125,168,196,243).
 * The reference has no FFI today (plain C++ class API, include/vo.h:46-472); every entry point
 * below names the reference interface it replaces.  Plain pointers and sizes only — no C++ types,
 * no torch types.  All functions return SVO_OK (0) or a negative svo_status; nothing throws.
 *
 * Everything behind this header runs on the GPU (hand-written HIP kernels for gfx950).  There is
 * no CPU fallback: if no HIP device is usable the calls fail with SVO_ERR_HIP.
 *
 * Pointer convention: unless a parameter is documented as "device", pointers are HOST memory and
 * the call synchronises before returning.
 */
#ifndef SVO_H
#define SVO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SVO_OK = 0,
    SVO_ERR_ARG = -1,      /* bad argument (null pointer, size mismatch, unsupported window ...) */
    SVO_ERR_HIP = -2,      /* HIP runtime error / no device; svo_last_error() has the text */
    SVO_ERR_CAPACITY = -3, /* input exceeds the capacity the context was created with */
    SVO_ERR_STATE = -4     /* call order violated (e.g. projection not set) */
} svo_status;

/* The reference's compile-time constants (include/vo.h:53-127, 251-252; src/vo.cpp:183-184, 295)
 * as a run-time struct.  svo_config_default() fills the reference values. */
typedef struct {
    int bucket_start_row;                 /* vo.h:53   4   */
    int buckets_along_height;             /* vo.h:60   92  */
    int buckets_along_width;              /* vo.h:61   160 */
    int features_per_bucket;              /* vo.h:65   1   (1 .. 64; 1 takes the fused argmax path, larger capacities the general Bucket::add_feature walk) */
    int features_threshold;               /* vo.h:71   15  */
    int pre_matching_feature_threshold;   /* vo.h:78   100 */
    int age_threshold;                    /* vo.h:84   20  */
    int fast_threshold;                   /* vo.h:90   20  */
    float ransac_reprojection_error;      /* vo.h:97   8   */
    int ransac_iterations;                /* vo.h:102  100 */
    double optical_flow_min_eig_threshold;/* vo.h:108  1e-3 */
    double circular_matching_success_threshold; /* vo.h:115 0.15 */
    double max_translation_norm;          /* vo.h:121  0.1 */
    double max_rotation_norm;             /* vo.h:127  0.5 */
    int win_w, win_h;                     /* vo.h:251  10x10 (square, 5 .. 31; 3-channel contexts 5 .. 21) */
    int max_level;                        /* vo.h:252  3 */
    int lk_max_count;                     /* vo.cpp:183 30 */
    double lk_epsilon;                    /* vo.cpp:184 1e-4 */
    float ransac_confidence;              /* vo.cpp:295 0.98f */
    int max_features;                     /* build preset (not in the reference): 0 = unlimited */
    int channels;                         /* 1 (default): 8-bit single-channel images, what the ROS path delivers (stereo_vo.cpp:9).
                                           * 3: interleaved 8-bit BGR, what the reference CLI really feeds (main.cpp:38-46 returns the
                                           * colour Mats): cv::FAST then walks the first `width` BYTES of every row (it has no channel
                                           * check), pyramids and LK are 3-channel.  Frame pipeline only (svo_process*, svo_submit_batch);
                                           * strides are in bytes.  Reproduces the trajectory the reference recorded for run1/. */
    int lk_float_sums;                    /* 0 (default): the LK normal equations are summed as exact integers (order-independent, fastest).
                                           * 1: they are summed in FLOAT in the lane order of OpenCV 4.x's SIMD128 code (lkpyramid.cpp,
                                           * `#if CV_SIMD128 && !CV_NEON`), i.e. with OpenCV's own rounding.  Measured on the reference's
                                           * recording: mode 1 reproduces run1/result.csv digit for digit (with channels = 3), mode 0 flips
                                           * borderline tracks at 4 of its first 24 frames (1e-5 .. 1e-3 m).  2.4 times the default's LK
                                           * time at the 21x21 window (the float chains are serial); frame pipeline and svo_circular_matching only. */
} svo_config;

/* Per-frame counters — the numbers the reference prints at vo.cpp:226,239,326,331,365,108-110,128-130. */
typedef struct {
    int n_after_detect;     /* vo.cpp:326 / :331 */
    int second_pass;        /* vo.cpp:327-332 ran */
    int n_into_lk;
    int n_after_circular;   /* vo.cpp:239 */
    int n_after_bounds;     /* vo.cpp:365 */
    int n_inliers;          /* vo.cpp:103 */
    int ransac_iters;       /* iterations the adaptive RANSAC loop would have run */
    int fail_reason;        /* 0 ok, 1 first frame, 2 too few tracks (vo.cpp:82), 3 RANSAC fail / few inliers (:106), 4 motion gate (:129) */
    int n_features_out;     /* size of currentVOFeatures on return */
    int lk_level_visits;    /* (feature, pass, level) visits of the LK passes that reached the Newton loop.  A feature runs its passes
                             * up to and including the first one that returns status 0: vo.cpp:227-238 deletes it whatever the later
                             * passes return, so the frame pipeline does not run them */
    int lk_newton_steps;    /* Newton iterations of those passes (the work term of the LK flop model) */
    int lk_dead_after_pass[3]; /* features whose status first became 0 in pass 0 (L0->L1), 1 (L1->R1), 2 (R1->R0) */
} svo_frame_stats;

typedef struct svo_context svo_context;

const char* svo_last_error(void);
int  svo_device_count(void);
void svo_config_default(svo_config* cfg);

/* ------------------------------------------------------------------------------------------------
 * Frame pipeline.  One context = n_seq independent VisualOdometry instances (reference: one object
 * per stream, include/vo.h:231) on one GPU, sharing image size and config; they advance in lock-step
 * through svo_process_batch.  n_seq = 1 is the reference's single-instance use.
 * ---------------------------------------------------------------------------------------------- */

/* replaces: VisualOdometry::VisualOdometry()  (vo.h:277, vo.cpp:5) */
int svo_create(const svo_config* cfg, int device, int n_seq, int width, int height, svo_context** out);
void svo_destroy(svo_context* ctx);

/* replaces: VisualOdometry::initalize_projection_matricies(Pl, Pr)  (vo.h:307-309, vo.cpp:8-26).
 * Pl, Pr: 3x4 float32 row-major.  seq = -1 sets every sequence. */
int svo_set_projection(svo_context* ctx, int seq, const float Pl[12], const float Pr[12]);

/* replaces: VisualOdometry::stereo_callback(left, right) -> pair<bool, Mat_<double>>  (vo.h:333-334, vo.cpp:41-137)
 * for all n_seq sequences at once.  left/right: n_seq pointers to 8-bit single-channel images
 * (width x height, row stride `stride` bytes); images_on_device != 0 means they are device pointers.
 * T_out: n_seq x 16 doubles (4x4 row-major): the pair's .second (last good transform on failure).
 * ok_out: n_seq ints: the pair's .first.  stats: n_seq entries or NULL. */
int svo_process_batch(svo_context* ctx, const uint8_t* const* left, const uint8_t* const* right, int stride,
                      int images_on_device, double* T_out, int* ok_out, svo_frame_stats* stats);

/* Page-locked host memory for the caller's image buffers (hipHostMalloc underneath; any hipHostMalloc'ed / hipHostRegister'ed
 * memory works the same).  svo_process / svo_process_batch given host images COPY them first (SURVEY.md 8b "Ownership": inputs
 * are borrowed for the call only); images that lie in page-locked memory with packed rows (stride == width * channels) are read
 * by the DMA engines in place, which saves the staging memcpy — ~40 us per 1241x376 pair, 0.54 -> 0.50 ms per synchronous call.
 * The reference has no counterpart (cv::Mat data is ordinary heap memory). */
void* svo_alloc_pinned(size_t bytes);
void  svo_free_pinned(void* p);

/* n_seq == 1 convenience with the reference's callback shape (also the ROS-callback shape, src/stereo_vo.cpp:61-62).
 * Returns 1 (pose produced), 0 (no pose this frame; T_out = last good transform) or a negative svo_status. */
int svo_process(svo_context* ctx, const uint8_t* left, const uint8_t* right, int stride, double T_out[16], svo_frame_stats* stats);

/* replaces: VisualOdometry::circularMatching(imgLeftT1, imgRightT1, pointsLeftT0, pointsRightT0, pointsLeftT1, pointsRightT1, features)
 * (vo.h:374-379, vo.cpp:169-240) as the MEMBER call it is in the reference: the T0 side is the pyramid pair this context
 * cached in its last svo_process / svo_circular_matching (vo.h:257-258; prime it with svo_process, as main.cpp:191 does), the
 * pyramids of the given T1 images become the cached pair (vo.cpp:231-232) — the next svo_process tracks against them.
 * Without the compaction: n points in, n points out per list, ok[n] = all four LK statuses && loop closure (vo.cpp:217-230).
 * n == 0 returns before anything is cached (vo.cpp:179-181).  n_seq == 1 contexts; images in the context's format. */
int svo_circular_matching(svo_context* ctx, const uint8_t* left_t1, const uint8_t* right_t1, int stride, int n,
                          const float* pl0, float* pl1, float* pr1, float* pr0, float* pl0_circle, uint8_t* ok);

/* Asynchronous form for throughput: enqueue one frame for every sequence and return immediately
 * (device pointers only; the images must stay valid until the matching svo_collect).
 * Results are queued in order; svo_collect blocks for the oldest outstanding frame. At most 8 in flight. */
/* Diagnostics: VGPRs one SIMD has left beside a full complement of this context's LK waves (-1 unknown).  Several many-sequence
 * contexts on one device overlap their f64 kernels with each other's LK kernel only when this is >= 96.  At the metric's window
 * (w = 21, single channel) it is 32 since round 3 — six waves of 80 registers: the faster LK build wins over the overlap, DESIGN.md
 * section 2 — so such contexts simply take turns; builds that leave >= 96 (w = 31, 3-channel contexts) keep the overlap scheme.  A
 * test pins the figure, so that a change to the LK kernel that alters the regime fails loudly.
 * NOTE on locality: creating or destroying ANOTHER context with more than 8 sequences on the same device changes which builds of
 * the PnP / triangulation kernels this context launches from its next frame on (full-register alone; 96-register when the
 * device is shared and the figure above is >= 96) and chains its LK launches behind the other's.  Results are identical either way. */
int svo_get_lk_registers_left(svo_context* ctx);

int svo_submit_batch(svo_context* ctx, const uint8_t* const* left_dev, const uint8_t* const* right_dev, int stride);
int svo_collect(svo_context* ctx, double* T_out, int* ok_out, svo_frame_stats* stats);

/* Introspection (parity tests): currentVOFeatures (vo.h:245) of one sequence, and the last frame's
 * compacted tracks.  Arrays may be NULL.  Returns the count or a negative status.  inlier[] is the is_ok vector of vo.cpp:115-119:
 * all zero when the frame failed before it was built (RANSAC failure or fewer inliers than features_threshold, vo.cpp:106-113). */
int svo_get_features(svo_context* ctx, int seq, int cap, float* xy, int* ages, int* strengths);
int svo_get_last_tracks(svo_context* ctx, int seq, int cap, float* pl0, float* pr0, float* pl1, float* pr1,
                        float* world, uint8_t* inlier);
/* Timing: HIP-event milliseconds of the dominant kernel (the fused LK chain) in the last processed frame, and of the whole frame.
 * With SVO_GRAPH=1 in the environment a context replays each frame as a captured hipGraph (one per results-ring slot; off by
 * default: measured slightly slower than the launch list on MI355X): stage events are then not recorded and lk_ms /
 * svo_get_stage_timing fail with SVO_ERR_STATE; frame_ms is always available. */
int svo_get_last_timing(svo_context* ctx, float* lk_ms, float* frame_ms);
/* The four stage-boundary events of a frame cost a lone stream ~10 us per frame (measured, one sequence), so they are recorded
 * only on request: svo_set_stage_timing(ctx, 1), or SVO_STAGE_TIMING=1 in the environment when the context is created.  While
 * off, lk_ms and svo_get_stage_timing fail with SVO_ERR_STATE; frame_ms is always available. */
int svo_set_stage_timing(svo_context* ctx, int on);
/* Per-stage HIP-event milliseconds of the last collected frame (all sequences of the context together), in pipeline order:
 * ms[0] ingest + pyramids (vo.cpp:74-75, 200-201)   ms[1] FAST + bucketing, both passes (vo.cpp:325-332)
 * ms[2] the four LK passes + masks (vo.cpp:203-230, 341-359)   ms[3] compaction + triangulation (vo.cpp:233-238, 360-364, 89-94)
 * ms[4] RANSAC-PnP, inlier update, gates, result record (vo.cpp:101-136).  The reference has no timers.
 * A lone-stream context (<= 8 sequences) runs independent stages in shared launches: ms[0] then covers ingest + pyramids AND the
 * first detection pass (ms[1]: only the second pass), ms[3] covers compaction, triangulation AND the first EPnP chunk. */
int svo_get_stage_timing(svo_context* ctx, float ms[5]);
void* svo_get_stream(svo_context* ctx);   /* hipStream_t the context launches on */

/* ------------------------------------------------------------------------------------------------
 * Stage-level entry points (host arrays in / out, one call = upload + kernel(s) + download).
 * They run the same kernels as the frame pipeline and exist so the reference's own unit tests
 * (src/main.cpp:50-264) and the parity tests can exercise each stage alone.  Their device buffers are kept per calling thread
 * and reused while device, image size and the buffer-shaping part of the configuration (bucket grid, LK window, levels, channels,
 * at most the RANSAC iteration count they were allocated for) stay the same — other parameters are taken over in place.
 * svo_stage_cache_clear() releases the calling thread's buffers.  A thread that exits hands its buffers to the next thread that
 * needs some (they are not freed at thread exit, where the HIP runtime may be gone): svo_stage_cache_clear_all() frees every
 * cached context that no live thread holds (plus the caller's) and returns how many it freed.
 * ---------------------------------------------------------------------------------------------- */

void svo_stage_cache_clear(void);
int svo_stage_cache_clear_all(void);

/* replaces: featureDetectionFast(image, fast_threshold, response_strengths)  (vo.h:393-395, feature_set.cpp:55-68)
 * i.e. cv::FAST(.., nonmaxSuppression=true).  xy: cap x 2, resp: cap.  *n_out = total found (may exceed cap). */
int svo_fast_detect(int device, const uint8_t* img, int w, int h, int stride, int threshold,
                    int cap, float* xy, float* resp, int* n_out);
/* the NMS-surviving score map (h*w bytes, 0 where no keypoint) — test hook for the FAST kernel */
int svo_fast_score_map(int device, const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score);

/* replaces: FeatureSet::filterByBucketLocationInternal(image, bah, baw, start_row, per_bucket)
 * (vo.h:168-172, feature_set.cpp:95-147) incl. Bucket::add_feature / compute_score (feature_set.cpp:16-53).
 * In place on (xy, ages, strengths); *n_io is the count in and out. */
int svo_bucket_filter(int device, int img_w, int img_h, int* n_io, float* xy, int* ages, int* strengths,
                      int buckets_along_height, int buckets_along_width, int bucket_start_row,
                      int features_per_bucket, int age_threshold, int fast_threshold);

/* replaces: FeatureSet::appendFeaturesFromImage(image, fast_threshold) with the default grid
 * (vo.h:186-187, feature_set.cpp:75-89): FAST + append (age 0) + bucket filter, fused on the GPU
 * (bucket winners are picked with 64-bit atomicMax keys straight from the FAST kernel). cap = array capacity. */
int svo_append_features_from_image(int device, const svo_config* cfg, const uint8_t* img, int w, int h, int stride,
                                   int fast_threshold, int cap, int* n_io, float* xy, int* ages, int* strengths);

/* replaces: cv::buildOpticalFlowPyramid(img, pyr, winSize, maxLevel)  (vo.cpp:50,52,200,201).
 * Returns the levels as tightly packed u8 images concatenated in `levels_out` (level l is
 * w_l*h_l bytes, w_l=(w_{l-1}+1)/2); n_levels_out <= max_level+1 (stops when the next level would
 * not exceed the window; win >= 7).  Derivatives are not materialised (they are fused into the LK kernel). */
int svo_build_pyramid(int device, const uint8_t* img, int w, int h, int stride, int win, int max_level,
                      uint8_t* levels_out, int64_t levels_cap, int* n_levels_out);

/* replaces: cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, prevPts, nextPts, status, err, winSize, maxLevel, termcrit, 0, minEig)
 * (vo.cpp:203-215) on two images (pyramids are built internally). */
int svo_lk_track(int device, const uint8_t* prev_img, const uint8_t* next_img, int w, int h, int stride,
                 int n, const float* prev_pts, float* next_pts, uint8_t* status,
                 int win, int max_level, int max_count, double epsilon, double min_eig_threshold);

/* replaces: VisualOdometry::circularMatching  (vo.h:374-379, vo.cpp:169-240) without the compaction:
 * the four LK passes L0->L1->R1->R0->L0, fused in one kernel, plus the status / loop-closure mask
 * (vo.cpp:217-230).  Outputs n points each and ok[n]. */
int svo_circular_match(int device, const svo_config* cfg, const uint8_t* l0, const uint8_t* r0,
                       const uint8_t* l1, const uint8_t* r1, int w, int h, int stride,
                       int n, const float* pl0, float* pl1, float* pr1, float* pr0, float* pl0_circle, uint8_t* ok);

/* replaces: findClosePoints(points_1, points_2, threshold)  (vo.h:430-432, vo.cpp:265-280) */
int svo_find_close_points(int device, int n, const float* p1, const float* p2, float threshold, uint8_t* ok);

/* replaces: cv::triangulatePoints + cv::convertPointsFromHomogeneous  (vo.cpp:89-94). xyz: n x 3 float32. */
int svo_triangulate(int device, const float Pl[12], const float Pr[12], int n, const float* pts_l, const float* pts_r, float* xyz);

/* replaces: cameraToWorld(K, cameraPoints, worldPoints, rotation, translation) -> pair<inliers, success>
 * (vo.h:452-456, vo.cpp:282-313) = cv::solvePnPRansac(.., useExtrinsicGuess, iterations, reprojErr, confidence, inliers, ITERATIVE).
 * K 3x3 f32; R (3x3 f64) and t (3 f64) in/out; inliers: int32[n]; *success = the pair's .second.
 * Small inputs follow cv::solvePnPRansac: n == 5 is one direct EPnP solve and n == 4 one direct P3P solve (every point an
 * inlier, no RANSAC, no refine); n < 4 is SVO_ERR_ARG (OpenCV asserts npoints >= 4). */
int svo_camera_to_world(int device, const float K[9], int n, const float* cam_pts, const float* world_pts,
                        double R[9], double t[3], int* inliers, int* n_inliers, int* success,
                        int ransac_iterations, float reproj_error, float confidence, int* iters_run);

/* replaces: getInverseTransform(rotation, translation)  (vo.h:469-470, vo.cpp:246-258): [R t; 0 1]^-1, 4x4 row-major.
 * Runs the device function the frame pipeline ends with (one tiny launch). */
int svo_inverse_transform(int device, const double R[9], const double t[3], double T[16]);

#ifdef __cplusplus
}
#endif
#endif
