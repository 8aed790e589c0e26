// svo_internal.hpp — shared declarations for the HIP translation units of libsvo_hip.so.
// Product code (gfx950 only).  Never includes anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/svo.h"

#define SVO_MAX_LEVELS 8
#define SVO_RING 8            // results ring / max frames in flight
#define SVO_MAX_WIN 31
#define SVO_PYR_SLOTS 4       // T1, imageLeftT0_, lastLeftPyramid (which may be a stale third one, vo.cpp:179-181) + the next frame's, built ahead

// A pyramid level is stored WITH its REFLECT_101 border, as cv::buildOpticalFlowPyramid stores it (withDerivatives = false,
// pyrBorder = BORDER_REFLECT_101): `pad` pixels on every side, rows `stride` bytes apart.  The LK kernel then reads every window
// it may visit (origin in [-win, size), lkpyramid.cpp) with plain loads — no per-byte border arithmetic in its loops.
// off: byte offset of pixel (0, 0) of the level inside one pyramid buffer.
struct LevelInfo { int w, h, off, stride; };
struct Geometry {
    int W, H;
    int nlevels;                              // levels actually built (cv::buildOpticalFlowPyramid stop rule)
    int pad;                                  // border pixels on every side of every level (lk_pad_for(win): covers the widest read of the LK kernel)
    LevelInfo lv[SVO_MAX_LEVELS];
    int pyr_bytes;                            // bytes of one pyramid (all levels with their borders; strides are multiples of 16)
};

// Device-resident state of one sequence = the members of the reference's VisualOdometry (include/vo.h:233-269).
struct SeqState {
    int frame_id;                             // vo.h:234
    int slot_t1, slot_img_t0, slot_pyr_t0;    // which of the SVO_PYR_SLOTS pyramid slots holds T1 / imageLeftT0_ / lastLeftPyramid
    int slot_next;                            // slot the NEXT frame's pyramids were built in ahead of time (image stream of a many-sequence context), else unused
    int active;                               // this frame runs matching (frame_id > 0 when the frame began)
    int feat_buf;                             // which half of the feature double-buffer is current
    int n_feat;                               // currentVOFeatures.size()
    int n_old;                                // feature count offered to the bucket grid by the current detection pass
    int do_second;                            // second FAST pass at threshold/4 required (vo.cpp:327)
    int n_lk;                                 // points entering circularMatching
    int n_tracks;                             // tracks after circular + bounds compaction
    int n_circ;                               // tracks after circular mask only (vo.cpp:239)
    int fail_reason;
    int pnp_best, pnp_iters, pnp_good;
    int pnp_need;                             // hypotheses that may still be consulted by the adaptive RANSAC loop
    int pnp_drawn;                            // RANSAC subsets drawn so far this frame
    unsigned long long pnp_rng;               // cv::RNG state after them (the later subsets are only drawn if the loop can reach them)
    int n_inliers;
    int ok;
    double R[9], t[3], last_T[16];            // vo.h:266-268
    float Pl[12], Pr[12], K[9];               // vo.h:273, :236
    svo_frame_stats stats;
};

struct FrameResult { double T[16]; int ok; svo_frame_stats stats; };

// All device buffers of a context (B sequences, capacity CAP features each).
struct DevBuffers {
    int B, CAP, NB;                            // NB = buckets_along_height * buckets_along_width
    int CN;                                    // image channels: 1, or 3 (interleaved BGR in, one pyramid per colour plane inside)
    int K;                                     // ransac_iterations
    Geometry geom;
    svo_config cfg;
    int bucket_h, bucket_w;
    float lk_mineig_cut;                       // LK: numerators below this fail the minimum-eigenvalue test (see lk_mineig_cut())
    SeqState* st;                              // [B]
    uint8_t* pyr;                              // [B][SVO_PYR_SLOTS][2 cams][CN planes][pyr_bytes]
    uint8_t* fastimg;                          // CN == 3 only: [B][3 slots][W*H] the first W bytes of every interleaved left row —
                                               // the single-channel 'image' cv::FAST sees in a BGR Mat
    float2* feat_xy[2]; int* feat_age[2]; int* feat_str[2];   // [B][CAP] each, double-buffered
    unsigned long long* bucket_keys;           // [B][NB]
    int* bucket_rowcnt;                        // [B][buckets_along_height] occupied buckets per grid row (counted at first fill)
    int* emit_ticket;                          // [B] blocks of k_bucket_emit that have finished; keys, row counts and tickets are all zero between passes
    int co_resident;                           // several many-sequence contexts share this device: launch the 96-register builds
    // features_per_bucket > 1 only (the general Bucket::add_feature walk; the default capacity 1 is an argmax and needs none of it):
    int KPCAP;                                 // candidate capacity per sequence = CAP (existing tracks) + keypoints one FAST pass can return
    uint8_t* score;                            // [B][W*H] NMS-surviving FAST scores of the pass
    int* kp_rows;                              // [B][H] keypoints per image row -> exclusive offsets
    float2* cand_xy; int* cand_age; int* cand_str;   // [B][KPCAP] the pass's input in the reference's order: tracks, then keypoints in raster order
    int* n_cand;                               // [B]
    float2* slot_xy; int* slot_age; int* slot_str;   // [B][NB][features_per_bucket]
    int* slot_n;                               // [B][NB]
    float2 *pl0, *pl1, *pr1, *pr0, *plc;       // [B][CAP] raw LK outputs
    uint8_t* okmask;                           // [B][CAP] bit0 circular ok, bit1 in-bounds
    unsigned* lk_work;                         // [B][CAP] per feature: Newton steps << 8 | (1 + first pass with status 0, or 0) << 6 | level visits (summed by k_compact)
    float2 *tl0, *tr0, *tl1, *tr1;             // [B][CAP] compacted tracks
    float* world;                              // [B][CAP][3]
    uint8_t* inlier;                           // [B][CAP]
    int* subsets;                              // [B][K][5]
    double* hyp;                               // [B][K][12]  (R row-major, t)
    int* hyp_good;                             // [B][K]
    int* inl_idx;                              // [B][CAP]
    double ransac_log_num;                     // log(max(1 - confidence, DBL_MIN)): the numerator of RANSACUpdateNumIters, computed by the host
    const double* lm_lambda;                   // [33] 10^k, k = -16..16 (the damping factors CvLevMarq can reach), computed on the host
    FrameResult* results;                      // [SVO_RING][B]
    const uint8_t** img_ptrs;                  // [SVO_RING][2][B] source image pointers: pinned host memory, read in place
};

// plane 0 of the pyramid of (sequence, slot, camera); plane k follows at + k * geom.pyr_bytes
__host__ __device__ inline size_t pyr_index(const DevBuffers& d, int seq, int slot, int cam) {
    return ((size_t)(seq * SVO_PYR_SLOTS + slot) * 2 + cam) * (size_t)d.CN * (size_t)d.geom.pyr_bytes;
}
__host__ __device__ inline size_t fastimg_index(const DevBuffers& d, int seq, int slot) {
    return (size_t)(seq * SVO_PYR_SLOTS + slot) * (size_t)d.geom.W * (size_t)d.geom.H;
}

// Hypotheses of the first RANSAC chunk (always solved).  16 when many sequences share the GPU; 32 for a lone stream: the GPU is
// empty then, a wider chunk costs no time, and the adaptive loop (11-27 iterations with 30 % outliers) rarely needs a second
// EPnP launch — which would be another 145 us on the critical path.
#define SVO_LONE_MAX_SEQ 8      // contexts of up to this many sequences are tuned for latency (wider first RANSAC chunk, 16 lanes per hypothesis ...)
#define SVO_MAX_DEVICES 64
__host__ __device__ inline int pnp_first_chunk(const DevBuffers& d) { const int c = d.B <= SVO_LONE_MAX_SEQ ? 32 : 16; return d.K < c ? d.K : c; }

// ---- launchers (each enqueues on `s`; none synchronises) ----
void launch_ingest(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs /* [2][B] device-readable array */, int stride_bytes, hipStream_t s,
                   bool begin_frame /* also run the per-frame reset of stereo_callback */);
void launch_pyramid(const DevBuffers& d, hipStream_t s);                 // levels 1.. of the T1 slot from its level 0, then the borders of all levels
void launch_pad_pyramid(const DevBuffers& d, hipStream_t s);             // REFLECT_101 borders of every level of the T1 slot (both cameras, every plane)
int lk_pad_for(int win);                                                 // border width the LK kernel's reads need at this window (svo_kernels_lk.hip)
void launch_ingest_pyramid(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs, int stride_bytes, hipStream_t s, bool begin_frame);   // both, fewer launches
// The same for the NEXT frame, on another stream, while the current frame is still being processed: the pyramids go into the slot no
// field of the sequence state refers to (SeqState::slot_next) and nothing else of the state is touched; launch_frame_begin (on the
// frame's own stream, after that work) then performs the per-frame reset with that slot as T1.  Single-channel contexts with >= 2 levels.
bool ingest_ahead_applies(const DevBuffers& d);
void launch_ingest_pyramid_ahead(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs, int stride_bytes, hipStream_t s);
void launch_frame_begin(const DevBuffers& d, hipStream_t s);
void launch_detect(const DevBuffers& d, int pass, int th_override, hipStream_t s);   // pass 0: FAST_THRESHOLD, pass 1: /4 if needed; th_override >= 0 replaces it
// grid_n = max features that can enter LK; early_out: a feature stops at its first pass with status 0 (frame pipeline) or runs all four (member call)
bool launch_lk_chain(const DevBuffers& d, int grid_n, hipStream_t s, int early_out);   // false: no kernel built for this (window, lanes, channels) — nothing ran
void launch_compact(const DevBuffers& d, hipStream_t s);
void launch_triangulate(const DevBuffers& d, hipStream_t s);
void launch_pnp(const DevBuffers& d, hipStream_t s, bool first_chunk_solved = false);   // expects the subsets drawn (launch_triangulate / k_compact do it)
bool launch_triangulate_epnp_fused(const DevBuffers& d, hipStream_t s);   // lone stream: triangulation || first EPnP chunk in one launch; false = not applicable
void launch_pnp_subsets(const DevBuffers& d, hipStream_t s);
void launch_pnp_p3p(const DevBuffers& d, hipStream_t s);                // exactly four points: one P3P, no RANSAC (stage API only)
void launch_inverse_transform(const double* R, const double* t, double* T, hipStream_t s);   // device pointers
// the front of a lone stream's frame as two fused launches (ingest + level 1 || FAST pass 0; levels 2-3 || emit), then the second
// detection pass; false = not applicable to this context, nothing was launched
bool launch_front_fused(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs, int stride, hipStream_t s);
void launch_frame_end(const DevBuffers& d, int ring_slot, hipStream_t s);

// stage helpers
void launch_fast_score_map(const uint8_t* img_dev, int w, int h, int threshold, uint8_t* score_dev, hipStream_t s);
void launch_score_compact(const uint8_t* score_dev, int w, int h, int cap, int* row_counts_dev, float2* xy_dev, float* resp_dev, int* n_dev, hipStream_t s);
void launch_bucket_general(int img_w, int img_h, int n, const float2* xy, const int* ages, const int* strs,
                           int bah, int baw, int start_row, int per_bucket, int age_thr, int fast_thr,
                           float2* slot_xy, int* slot_age, int* slot_str, int* slot_n,
                           float2* out_xy, int* out_age, int* out_str, int* n_out, hipStream_t s);
void launch_lk_single(const DevBuffers& d, int slotA, int camA, int slotB, int camB, int n, const float2* prev, float2* next,
                      uint8_t* status, hipStream_t s);
void launch_find_close(int n, const float2* a, const float2* b, float thr, uint8_t* ok, hipStream_t s);
bool lk_window_supported(int win);
int lk_registers_left(const DevBuffers& d);   // VGPRs per SIMD lane beside a full set of this context's LK waves (-1: unknown)
bool lk_window_supported_cn(int win, int cn);
float lk_mineig_cut(int win, double min_eig_threshold);

// ------------------------------------------------------------------------------------------------ subsets (cv::RNG, getSubset)
// All K 5-subsets of one sequence, drawn with cv::RNG's multiply-with-carry recurrence from the seed (uint64)-1; the number
// of draws never depends on model quality.  uniform(0, n) = next() % n: the remainder is taken through the 64-bit reciprocal
// ceil(2^64 / n) (exact for 32-bit operands: the error term x e / (n 2^64) stays below 2^-32 < 1/n), 6 instructions instead
// of the 32-bit division sequence.
// Draws subsets [s.pnp_drawn, upto) and leaves the generator state in s.pnp_rng: the first chunk is drawn beside the
// triangulation, the rest only as far as the adaptive loop can still reach (k_pnp_decide) — with a static scene that is never.
static __device__ inline void pnp_draw_subsets(const DevBuffers& d, SeqState& s, int seq, int upto) {
    const unsigned n = (unsigned)s.n_tracks;
    if (n < 2) return;
    if (upto > d.K) upto = d.K;
    int* out = d.subsets + (size_t)seq * d.K * 5;
    if (n == 5) {                                                    // model_points == npoints: one direct solve on all five (solvepnp.cpp)
        for (int i = 0; i < 5; i++) out[i] = i;
        s.pnp_drawn = d.K;
        return;
    }
    unsigned long long state = s.pnp_drawn == 0 ? 0xFFFFFFFFFFFFFFFFull : s.pnp_rng;      // RNG rng((uint64)-1)
    const unsigned long long recip = 0xFFFFFFFFFFFFFFFFull / n + 1ull;
    for (int it = s.pnp_drawn; it < upto; it++) {
        int idx[5];
        for (int i = 0; i < 5; i++) {
            int v; bool dup;
            do {
                state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32);
                const unsigned x = (unsigned)state;
                v = (int)(x - (unsigned)__umul64hi((unsigned long long)x, recip) * n);
                dup = false;
                for (int k = 0; k < i; k++) dup |= (idx[k] == v);
            } while (dup);
            idx[i] = v;
        }
        for (int i = 0; i < 5; i++) out[it * 5 + i] = idx[i];
    }
    if (upto > s.pnp_drawn) s.pnp_drawn = upto;
    s.pnp_rng = state;
}

// The first chunk of a lone stream, drawn by the 64 lanes of ONE wave (k_compact's first) from the table of raw generator states
// (svo_rng_table.hpp): lane h takes subset h at its no-duplicate position (raw draws 5h .. 5h+4), which is right for every subset
// up to the first one that meets a duplicate (getSubset redraws and the stream shifts); from that subset on lane 0 walks the
// stream serially, exactly as pnp_draw_subsets does (from the table while it lasts).  Same subsets, same generator state
// afterwards.  With ~900 tracks no subset of the 32 has a duplicate in two frames of three.  Call with all 64 lanes.
#include "svo_rng_table.hpp"
static __device__ inline void pnp_draw_first_chunk_wave(const DevBuffers& d, SeqState& s, int seq, int n_tracks, int upto) {
    const int lane = threadIdx.x & 63;
    const unsigned n = (unsigned)n_tracks;
    if (upto > d.K) upto = d.K;
    if (n < 6 || upto > 32 || upto * 5 + 8 > SVO_RNG_TABLE_N) {                 // n == 5: the direct solve; tiny sets: not worth a second path
        if (lane == 0) { s.n_tracks = n_tracks; pnp_draw_subsets(d, s, seq, upto); }
        return;
    }
    int* out = d.subsets + (size_t)seq * d.K * 5;
    const unsigned long long recip = 0xFFFFFFFFFFFFFFFFull / n + 1ull;
    int v[5]; bool dup = false;
    if (lane < upto) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const unsigned x = (unsigned)SVO_RNG_STATES[lane * 5 + i];
            v[i] = (int)(x - (unsigned)__umul64hi((unsigned long long)x, recip) * n);
        }
#pragma unroll
        for (int i = 1; i < 5; i++)
#pragma unroll
            for (int k = 0; k < i; k++) dup |= v[i] == v[k];
    }
    const unsigned long long dm = __ballot(dup);
    const int first_bad = dm ? __ffsll((long long)dm) - 1 : upto;              // subsets [0, first_bad) stand as drawn
    if (lane < first_bad) {
#pragma unroll
        for (int i = 0; i < 5; i++) out[lane * 5 + i] = v[i];
    }
    if (lane == 0) {
        int t = first_bad * 5;                                                 // raw draws consumed so far
        unsigned long long state = t > 0 ? SVO_RNG_STATES[t - 1] : 0xFFFFFFFFFFFFFFFFull;
        for (int it = first_bad; it < upto; it++) {
            int idx[5];
            for (int i = 0; i < 5; i++) {
                int w; bool dd;
                do {
                    state = t < SVO_RNG_TABLE_N ? SVO_RNG_STATES[t] : (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32);
                    t++;
                    const unsigned x = (unsigned)state;
                    w = (int)(x - (unsigned)__umul64hi((unsigned long long)x, recip) * n);
                    dd = false;
                    for (int k = 0; k < i; k++) dd |= (idx[k] == w);
                } while (dd);
                idx[i] = w;
            }
            for (int i = 0; i < 5; i++) out[it * 5 + i] = idx[i];
        }
        s.pnp_drawn = upto;
        s.pnp_rng = state;
    }
}
