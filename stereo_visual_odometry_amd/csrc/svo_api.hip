// svo_api.hip — host side of libsvo_hip.so: context, frame pipeline, C-ABI (include/svo.h).
// The per-frame orchestration of VisualOdometry::stereo_callback (reference src/vo.cpp:41-137) is
// expressed as a fixed sequence of kernel launches on one HIP stream; every data-dependent decision
// (second FAST pass, too-few-tracks gate, RANSAC failure, motion gate, stale-pyramid quirk) is taken
// on the device from SeqState, so a frame needs no host round trip until its pose is read back.
#include "svo_internal.hpp"
#include <stdio.h>
#include <string.h>
#include <stddef.h>
#include <stdlib.h>
#include <string>
#include <atomic>
#include <mutex>
#include <vector>
#include <math.h>

static thread_local std::string g_err;
extern "C" const char* svo_last_error(void) { return g_err.c_str(); }

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            g_err = _b;                                                                           \
            return SVO_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

static int fail_arg(const char* msg) { g_err = msg; return SVO_ERR_ARG; }

// The LK kernel of a many-sequence context fills the whole GPU.  When several such contexts share a device (bench.py interleaves
// two, so that one's latency-bound PnP kernels run under the other's LK), their LK launches are chained through one event per
// device: two LK grids resident together only halve each other's CUs, and HIP-event durations of either would include the other.
struct LkGate { std::mutex mu; hipEvent_t ev = nullptr; bool armed = false; std::atomic<int> contexts{0}; };
static LkGate g_lk_gate[SVO_MAX_DEVICES];
static bool lk_gated(const svo_context* c);

extern "C" int svo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void svo_config_default(svo_config* c) {
    c->bucket_start_row = 4; c->buckets_along_height = 92; c->buckets_along_width = 160;
    c->features_per_bucket = 1; c->features_threshold = 15; c->pre_matching_feature_threshold = 100;
    c->age_threshold = 20; c->fast_threshold = 20; c->ransac_reprojection_error = 8.f;
    c->ransac_iterations = 100; c->optical_flow_min_eig_threshold = 0.001;
    c->circular_matching_success_threshold = .15; c->max_translation_norm = .1; c->max_rotation_norm = .5;
    c->win_w = 10; c->win_h = 10; c->max_level = 3; c->lk_max_count = 30; c->lk_epsilon = 0.0001;
    c->ransac_confidence = 0.98f; c->max_features = 0; c->channels = 1; c->lk_float_sums = 0;
}

// cv::buildOpticalFlowPyramid's level rule: level 0 always, stop as soon as
// the NEXT level would have width <= win or height <= win.
static void make_geometry(Geometry& g, int W, int H, int win, int max_level, int pad) {
    memset(&g, 0, sizeof(g));
    g.W = W; g.H = H; g.pad = pad;
    if (max_level > SVO_MAX_LEVELS - 1) max_level = SVO_MAX_LEVELS - 1;
    int w = W, h = H, base = 0;
    for (int l = 0; l <= max_level; l++) {
        const int stride = (w + 2 * pad + 15) & ~15;
        g.lv[l].w = w; g.lv[l].h = h; g.lv[l].stride = stride; g.lv[l].off = base + pad * stride + pad;
        base += stride * (h + 2 * pad);
        g.nlevels = l + 1;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win || h <= win) break;
    }
    g.pyr_bytes = base;
}

struct svo_context {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuffers d = {};
    std::vector<void*> allocs;
    // host side of the results ring
    FrameResult* h_results = nullptr;            // pinned [SVO_RING][B]
    const uint8_t** h_ptrs = nullptr;            // pinned [SVO_RING][2][B]
    hipEvent_t ev_done[SVO_RING] = {}, ev_f0[SVO_RING] = {}, ev_lk0[SVO_RING] = {}, ev_lk1[SVO_RING] = {};
    hipEvent_t ev_pyr[SVO_RING] = {}, ev_tri[SVO_RING] = {};   // stage boundaries: pyramids built / world points triangulated
    // many-sequence contexts build the NEXT frame's pyramids on a second stream while the current frame is in its LK kernel (issue_frame)
    hipStream_t img_stream = nullptr;
    hipEvent_t ev_img[SVO_RING] = {}, ev_begin = nullptr;
    bool begin_recorded = false;
    bool staged_inputs = false;                  // this frame's images were copied in on `stream` (host-image calls): the image stream must wait for them
    int head = 0, tail = 0, inflight = 0;        // ring indices: head = next to enqueue, tail = oldest outstanding
    int last_slot = -1;
    uint8_t* staging = nullptr;                  // device [2][B][W*H] for host-image calls
    uint8_t* h_staging = nullptr;                // pinned host mirror of `staging`
    uint8_t* h_upload = nullptr;                 // pinned [6][W*H]: the stage entry points' image uploads (slot x camera)
    bool projection_set = false;
    int lk_grid = 0;
    int lk_hint = 0;                             // feature count seen in the last collected frame (sizes the LK grid; 0 = unknown)
    // hipGraph replay of the frame's launch list (one executable graph per results-ring slot: the slot fixes the pointer table,
    // the result record and the copies; re-captured when the stride or the LK grid size changes)
    bool use_graph = false;
    bool capturing = false;                      // inside hipStreamBeginCapture / EndCapture
    bool counted = false;                        // this context is in its device's LkGate count
    int lk_room = -1;                            // lk_registers_left(d), asked once
    int k_alloc = 0;                             // RANSAC hypotheses the PnP buffers were allocated for
    bool stage_timing = false;                   // record the four stage-boundary events of a frame (svo_set_stage_timing; SVO_STAGE_TIMING=1)
    hipGraphExec_t gexec[SVO_RING] = {};
    int g_stride[SVO_RING] = {}, g_gn[SVO_RING] = {}, g_co[SVO_RING] = {};   // what the slot's graph was captured with (stride, LK grid, co-resident builds)
    bool staged_slot[SVO_RING] = {};             // the slot's stage events were recorded (launch-list mode only)
};

template <typename T>
static int dev_alloc(svo_context* c, T** p, size_t count) {
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    HIPCHK(hipMemsetAsync(q, 0, count * sizeof(T) > 0 ? count * sizeof(T) : 16, c->stream));
    c->allocs.push_back(q);
    *p = (T*)q;
    return SVO_OK;
}

static int ctx_create(const svo_config* cfg_in, int device, int n_seq, int width, int height, int cap_override, svo_context** out) {
    if (!out) return fail_arg("out is null");
    *out = nullptr;
    svo_config cfg;
    if (cfg_in) cfg = *cfg_in; else svo_config_default(&cfg);
    if (n_seq < 1 || width < 16 || height < 16) return fail_arg("n_seq >= 1 and width/height >= 16 required");
    if (cfg.win_w != cfg.win_h || !lk_window_supported(cfg.win_w)) return fail_arg("unsupported LK window (square, 5 .. 31)");
    if (width <= cfg.win_w || height <= cfg.win_h) return fail_arg("image must be larger than the LK window");
    if (cfg.features_per_bucket < 1 || cfg.features_per_bucket > 64) return fail_arg("features_per_bucket must be 1 .. 64");
    if (cfg.buckets_along_height < 1 || cfg.buckets_along_width < 1 || cfg.ransac_iterations < 1) return fail_arg("bad bucket grid / ransac_iterations");
    if (cfg.channels == 0) cfg.channels = 1;
    if (cfg.channels != 1 && cfg.channels != 3) return fail_arg("channels must be 1 or 3");
    if (cfg.lk_float_sums != 0 && cfg.lk_float_sums != 1) return fail_arg("lk_float_sums must be 0 or 1");
    if (!lk_window_supported_cn(cfg.win_w, cfg.channels)) return fail_arg("this LK window is not built for 3-channel input (5 .. 21 are)");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail_arg("no such HIP device");
    HIPCHK(hipSetDevice(device));
    svo_context* c = new svo_context();
    struct Undo { svo_context* c; ~Undo() { if (c) svo_destroy(c); } } undo{c};     // every early return below frees what exists so far
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    DevBuffers& d = c->d;
    d.B = n_seq; d.cfg = cfg; d.K = cfg.ransac_iterations; d.CN = cfg.channels; c->k_alloc = d.K;
    d.NB = cfg.buckets_along_height * cfg.buckets_along_width;
    d.bucket_h = (height + cfg.buckets_along_height - 1) / cfg.buckets_along_height;     // feature_set.cpp:91-93,103-104
    d.bucket_w = (width + cfg.buckets_along_width - 1) / cfg.buckets_along_width;
    int rows = cfg.buckets_along_height - cfg.bucket_start_row; if (rows < 0) rows = 0;
    d.CAP = rows * cfg.buckets_along_width * cfg.features_per_bucket;
    if (d.CAP < 64) d.CAP = 64;
    if (cap_override > d.CAP) d.CAP = cap_override;
    c->lk_grid = (cfg.max_features > 0 && cfg.max_features < d.CAP) ? cfg.max_features : d.CAP;
    make_geometry(d.geom, width, height, cfg.win_w, cfg.max_level, lk_pad_for(cfg.win_w));
    d.lk_mineig_cut = lk_mineig_cut(cfg.win_w, cfg.optical_flow_min_eig_threshold);
    {
        double pc = (double)cfg.ransac_confidence; pc = pc > 0. ? pc : 0.; pc = pc < 1. ? pc : 1.;
        d.ransac_log_num = log(1. - pc > 2.2250738585072014e-308 ? 1. - pc : 2.2250738585072014e-308);
    }
    const size_t B = n_seq, CAP = d.CAP;
    int rc;
#define ALLOC(ptr, count) if ((rc = dev_alloc(c, &(ptr), (count))) != SVO_OK) return rc;
    ALLOC(d.st, B);
    ALLOC(d.pyr, B * 2 * SVO_PYR_SLOTS * (size_t)d.CN * (size_t)d.geom.pyr_bytes + 256);   // + slack: the LK kernel's unaligned dword loads may read a few bytes past a row
    if (d.CN == 3) ALLOC(d.fastimg, B * SVO_PYR_SLOTS * (size_t)width * (size_t)height + 256);
    for (int k = 0; k < 2; k++) { ALLOC(d.feat_xy[k], B * CAP); ALLOC(d.feat_age[k], B * CAP); ALLOC(d.feat_str[k], B * CAP); }
    ALLOC(d.bucket_keys, B * (size_t)d.NB);
    ALLOC(d.bucket_rowcnt, B * (size_t)cfg.buckets_along_height); ALLOC(d.emit_ticket, B);
    if (cfg.features_per_bucket > 1) {
        d.KPCAP = d.CAP + ((width + 1) / 2) * ((height + 1) / 2);       // strict 3x3 NMS: at most one keypoint per 2x2 block
        const size_t KC = (size_t)d.KPCAP, SL = (size_t)d.NB * cfg.features_per_bucket;
        ALLOC(d.score, B * (size_t)width * height); ALLOC(d.kp_rows, B * (size_t)height);
        ALLOC(d.cand_xy, B * KC); ALLOC(d.cand_age, B * KC); ALLOC(d.cand_str, B * KC); ALLOC(d.n_cand, B);
        ALLOC(d.slot_xy, B * SL); ALLOC(d.slot_age, B * SL); ALLOC(d.slot_str, B * SL); ALLOC(d.slot_n, B * (size_t)d.NB);
    }
    ALLOC(d.pl0, B * CAP); ALLOC(d.pl1, B * CAP); ALLOC(d.pr1, B * CAP); ALLOC(d.pr0, B * CAP); ALLOC(d.plc, B * CAP);
    ALLOC(d.okmask, B * CAP); ALLOC(d.lk_work, B * CAP);
    ALLOC(d.tl0, B * CAP); ALLOC(d.tr0, B * CAP); ALLOC(d.tl1, B * CAP); ALLOC(d.tr1, B * CAP);
    ALLOC(d.world, B * CAP * 3); ALLOC(d.inlier, B * CAP); ALLOC(d.inl_idx, B * CAP);
    ALLOC(d.subsets, B * (size_t)d.K * 5); ALLOC(d.hyp, B * (size_t)d.K * 12); ALLOC(d.hyp_good, B * (size_t)d.K);
    {
        double* lam = nullptr;
        ALLOC(lam, 33);
        double h[33];
        const double LOG10 = log(10.);
        for (int k = -16; k <= 16; k++) h[k + 16] = exp(k * LOG10);      // CvLevMarq: lambda = exp(lambdaLg10 * log(10))
        HIPCHK(hipMemcpyAsync(lam, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));                          // h is a stack array
        d.lm_lambda = lam;
    }
#undef ALLOC
    // the results ring lives in pinned HOST memory that the device can write: k_frame_end stores the B records there directly
    // (180 B per sequence over PCIe) instead of a device buffer plus a copy operation per frame
    HIPCHK(hipHostMalloc((void**)&c->h_results, sizeof(FrameResult) * SVO_RING * B, hipHostMallocMapped));
    memset(c->h_results, 0, sizeof(FrameResult) * SVO_RING * B);
    { void* dv = nullptr; HIPCHK(hipHostGetDevicePointer(&dv, c->h_results, 0)); d.results = (FrameResult*)dv; }
    HIPCHK(hipHostMalloc((void**)&c->h_ptrs, sizeof(uint8_t*) * SVO_RING * 2 * B, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void**)&d.img_ptrs, (void*)c->h_ptrs, 0));   // read in place by k_ingest: no per-frame upload
    for (int i = 0; i < SVO_RING; i++) {
        HIPCHK(hipEventCreate(&c->ev_done[i])); HIPCHK(hipEventCreate(&c->ev_f0[i]));
        HIPCHK(hipEventCreate(&c->ev_lk0[i])); HIPCHK(hipEventCreate(&c->ev_lk1[i]));
        HIPCHK(hipEventCreate(&c->ev_pyr[i])); HIPCHK(hipEventCreate(&c->ev_tri[i]));
    }
    // initial state: rotation = I, translation = 0, last_transform = I (vo.h:266-268); no slots in use
    std::vector<SeqState> hs(B);
    memset(hs.data(), 0, sizeof(SeqState) * B);
    for (size_t i = 0; i < B; i++) {
        hs[i].slot_img_t0 = -1; hs[i].slot_pyr_t0 = -1;
        for (int k = 0; k < 9; k++) hs[i].R[k] = (k % 4 == 0);
        for (int k = 0; k < 16; k++) hs[i].last_T[k] = (k % 5 == 0);
    }
    HIPCHK(hipMemcpyAsync(d.st, hs.data(), sizeof(SeqState) * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    {
        // SVO_GRAPH=1 replays every frame as a captured hipGraph.  Measured on MI355X, one sequence, synchronous, host images
        // (scratch/graph_ab.py, same box): 0.718 ms per frame pair with the launch list, 0.737 ms with the graph — the ~25 launches
        // are issued ahead of the GPU anyway and the graph's dispatch is not cheaper on this runtime — so the launch list stays
        // the default and the graph is the option.
        const char* e = getenv("SVO_GRAPH");
        c->use_graph = e ? atoi(e) != 0 : false;
        c->lk_room = lk_registers_left(c->d);
        if (device >= 0 && device < SVO_MAX_DEVICES && n_seq > SVO_LONE_MAX_SEQ) {
            std::lock_guard<std::mutex> lock(g_lk_gate[device].mu);
            g_lk_gate[device].contexts++; c->counted = true;
        }
        const char* t = getenv("SVO_STAGE_TIMING");
        c->stage_timing = t ? atoi(t) != 0 : false;
    }
    undo.c = nullptr;
    *out = c;
    return SVO_OK;
}

extern "C" int svo_create(const svo_config* cfg, int device, int n_seq, int width, int height, svo_context** out) {
    return ctx_create(cfg, device, n_seq, width, height, 0, out);
}

// Several many-sequence contexts on one device.  Two decisions, both read ONCE per frame (issue_frame); `contexts` is an atomic:
// contexts are created and destroyed on other threads while this one enqueues frames.
//  * lk_gated: the f64 kernels run as 96-register builds under the OTHER context's LK grid — only if those builds fit beside it
//    (lk_registers_left >= 96: w = 31 and the 3-channel builds; not w = 21 since round 3, nor w = 10 / 15).
//  * lk_chained: the contexts' LK launches wait for each other (one event per device).  Always, when the device is shared: two
//    LK grids resident together only share the CUs — measured the same whole-job rate either way at w = 21 (18 370 vs 18 390
//    frame-pairs/s) — and unchained each launch's duration contains a part of the other's (12.7 ms per launch chained, 13.3-16.4
//    unchained, varying from run to run), which makes the per-launch figure of the dominant kernel, the one the bench line's
//    roofline is computed from, meaningless.  SVO_LK_GATE=0 switches both off (measurement).
static bool lk_shared_device(const svo_context* c) {
    static const bool off = getenv("SVO_LK_GATE") && atoi(getenv("SVO_LK_GATE")) == 0;
    return !off && c->counted && g_lk_gate[c->device].contexts.load(std::memory_order_relaxed) > 1;
}
static bool lk_gated(const svo_context* c) { return c->lk_room >= 96 && lk_shared_device(c); }
static bool lk_chained(const svo_context* c) { return lk_shared_device(c); }

extern "C" void svo_destroy(svo_context* c) {
    if (!c) return;
    if (c->counted) {
        LkGate& g = g_lk_gate[c->device];
        std::lock_guard<std::mutex> lock(g.mu);
        if (--g.contexts == 0 && g.ev) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); (void)hipEventDestroy(g.ev); g.ev = nullptr; g.armed = false; }
    }
    // teardown is best effort: errors here have nowhere to go, the calls are (void)ed on purpose
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->staging) (void)hipFree(c->staging);
    if (c->h_staging) (void)hipHostFree(c->h_staging);
    if (c->h_upload) (void)hipHostFree(c->h_upload);
    if (c->h_results) (void)hipHostFree(c->h_results);
    if (c->h_ptrs) (void)hipHostFree((void*)c->h_ptrs);
    for (int i = 0; i < SVO_RING; i++) {
        if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
        if (c->ev_f0[i]) (void)hipEventDestroy(c->ev_f0[i]);
        if (c->ev_lk0[i]) (void)hipEventDestroy(c->ev_lk0[i]);
        if (c->ev_lk1[i]) (void)hipEventDestroy(c->ev_lk1[i]);
        if (c->ev_pyr[i]) (void)hipEventDestroy(c->ev_pyr[i]);
        if (c->ev_tri[i]) (void)hipEventDestroy(c->ev_tri[i]);
    }
    for (int i = 0; i < SVO_RING; i++) if (c->gexec[i]) (void)hipGraphExecDestroy(c->gexec[i]);
    for (int i = 0; i < SVO_RING; i++) if (c->ev_img[i]) (void)hipEventDestroy(c->ev_img[i]);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->img_stream) (void)hipStreamDestroy(c->img_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void* svo_get_stream(svo_context* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int svo_get_lk_registers_left(svo_context* c) { return c ? c->lk_room : -1; }

extern "C" int svo_set_projection(svo_context* c, int seq, const float Pl[12], const float Pr[12]) {
    if (!c || !Pl || !Pr) return fail_arg("null argument");
    if (seq < -1 || seq >= c->d.B) return fail_arg("seq out of range");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    // Pl, Pr, K are adjacent members of SeqState: one strided 2-D copy writes them into every selected sequence's record
    // (row = the 33 floats, pitch = sizeof(SeqState)) instead of three blocking copies per sequence
    static_assert(offsetof(SeqState, Pr) == offsetof(SeqState, Pl) + sizeof(float) * 12 && offsetof(SeqState, K) == offsetof(SeqState, Pl) + sizeof(float) * 24,
                  "Pl, Pr, K must be contiguous");
    const int s0 = seq < 0 ? 0 : seq, ns = seq < 0 ? c->d.B : 1;
    std::vector<float> rows((size_t)ns * 33);
    for (int s = 0; s < ns; s++) {
        float* r = rows.data() + (size_t)s * 33;
        memcpy(r, Pl, sizeof(float) * 12); memcpy(r + 12, Pr, sizeof(float) * 12);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[24 + 3 * i + j] = Pl[4 * i + j];   // K = Pl[:, :3]  (vo.cpp:16-25)
    }
    HIPCHK(hipMemcpy2D((char*)(c->d.st + s0) + offsetof(SeqState, Pl), sizeof(SeqState), rows.data(), sizeof(float) * 33, sizeof(float) * 33, (size_t)ns,
                       hipMemcpyHostToDevice));
    c->projection_set = true;
    return SVO_OK;
}

static int stage_host_images(svo_context* c, const uint8_t* const* left, const uint8_t* const* right, int stride,
                             std::vector<const uint8_t*>& lp, std::vector<const uint8_t*>& rp);

// The launch list of one frame (vo.cpp:41-137 as kernels), between the pointer-table upload and the result download.
// with_events: record the stage-boundary events (not inside a graph capture).
static int issue_frame(svo_context* c, int slot, int stride, int gn, bool with_events, int shares = -1) {
    DevBuffers& d = c->d;
    const int B = d.B;
    hipStream_t s = c->stream;
    static const bool force_lean = getenv("SVO_FORCE_LEAN") && atoi(getenv("SVO_FORCE_LEAN")) != 0;      // test knob: every context takes them
    const bool shares_device = shares < 0 ? lk_gated(c) : shares != 0;   // read once per frame: both uses below see the same answer
    d.co_resident = (shares_device || force_lean) ? 1 : 0;             // picks the 96-register builds of the f64 kernels (svo_kernels_pnp.hip)
    const uint8_t** dp = d.img_ptrs + (size_t)slot * 2 * B;         // the slot's pointer table: pinned host memory the kernel reads in place
    if (launch_front_fused(d, dp, stride, s)) {                        // lone stream: ingest + pyramid beside detection, two launches
        if (with_events) HIPCHK(hipEventRecord(c->ev_pyr[slot], s));   // stage timers: ms[0] = the fused front, ms[1] ~ 0
    } else if (!c->capturing && ingest_ahead_applies(d)) {
        // Many sequences: this frame's pyramids are built on the IMAGE stream, which only waits for the previous frame's reset — so,
        // with frames in flight, they are built while the previous frame sits in its LK kernel.  That kernel fills six of a SIMD's
        // eight wave slots and 480 of its 512 registers (svo_kernels_lk.hip): the ingest, pyramid and border kernels (11-17
        // registers) are the ones that still fit beside it.  The frame's own stream then resets the state and goes on with detection.
        if (!c->img_stream) {
            HIPCHK(hipStreamCreateWithFlags(&c->img_stream, hipStreamNonBlocking));   // (highest priority measured: same rate, but the two contexts' LK launches then run in lock-step)
            HIPCHK(hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming));
            for (int i = 0; i < SVO_RING; i++) HIPCHK(hipEventCreateWithFlags(&c->ev_img[i], hipEventDisableTiming));
        }
        if (c->begin_recorded) HIPCHK(hipStreamWaitEvent(c->img_stream, c->ev_begin, 0));   // the fields k_pick_next reads are those of the frame in flight
        if (c->staged_inputs) HIPCHK(hipStreamWaitEvent(c->img_stream, c->ev_f0[slot], 0));   // host-image call: the H2D copies were queued on `stream` before this frame's start event
        launch_ingest_pyramid_ahead(d, dp, stride, c->img_stream);
        HIPCHK(hipEventRecord(c->ev_img[slot], c->img_stream));
        HIPCHK(hipStreamWaitEvent(s, c->ev_img[slot], 0));
        launch_frame_begin(d, s);
        HIPCHK(hipEventRecord(c->ev_begin, s)); c->begin_recorded = true;
        if (with_events) HIPCHK(hipEventRecord(c->ev_pyr[slot], s));
        launch_detect(d, 0, -1, s);
        launch_detect(d, 1, -1, s);
    } else {
        launch_ingest_pyramid(d, dp, stride, s, true);                // + the per-frame reset
        if (with_events) HIPCHK(hipEventRecord(c->ev_pyr[slot], s));
        launch_detect(d, 0, -1, s);
        launch_detect(d, 1, -1, s);
    }
    const bool gated = !c->capturing && (shares_device || lk_chained(c));   // chained LK launches (a captured graph cannot wait for another stream's event)
    if (gated) {
        LkGate& g = g_lk_gate[c->device];
        std::lock_guard<std::mutex> lock(g.mu);
        if (g.armed) HIPCHK(hipStreamWaitEvent(s, g.ev, 0));
    }
    if (with_events) HIPCHK(hipEventRecord(c->ev_lk0[slot], s));
    if (!launch_lk_chain(d, gn, s, 1)) { g_err = "no LK kernel is built for this window / lanes-per-feature / channel count"; return SVO_ERR_STATE; }
    if (with_events) HIPCHK(hipEventRecord(c->ev_lk1[slot], s));
    if (gated) {
        LkGate& g = g_lk_gate[c->device];
        std::lock_guard<std::mutex> lock(g.mu);
        if (!g.ev) HIPCHK(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(g.ev, s));
        g.armed = true;
    }
    launch_compact(d, s);
    const bool tri_epnp = launch_triangulate_epnp_fused(d, s);          // lone stream: the first EPnP chunk runs beside the triangulation
    if (!tri_epnp) launch_triangulate(d, s);
    if (with_events) HIPCHK(hipEventRecord(c->ev_tri[slot], s));       // (fused: the stage timers count that chunk with the triangulation)
    launch_pnp(d, s, tri_epnp);
    launch_frame_end(d, slot, s);      // writes the result records straight into the pinned host ring (d.results is host memory mapped into the device)
    return SVO_OK;
}

// Enqueue one frame for all sequences.  ptrs: host array [2][B] of DEVICE image pointers.
static int enqueue_frame(svo_context* c, const uint8_t* const* left_dev, const uint8_t* const* right_dev, int stride) {
    if (c->inflight >= SVO_RING) { g_err = "too many frames in flight (collect first)"; return SVO_ERR_STATE; }
    DevBuffers& d = c->d;
    const int slot = c->head, B = d.B;
    const uint8_t** hp = c->h_ptrs + (size_t)slot * 2 * B;
    for (int i = 0; i < B; i++) { hp[i] = left_dev[i]; hp[B + i] = right_dev[i]; }
    hipStream_t s = c->stream;
    // LK grid sized from the last feature counts the host has seen (+30 %); the kernel strides, so an underestimate is only slower
    int gn = c->lk_grid;
    if (c->lk_hint > 0) {
        int h = c->lk_hint + c->lk_hint / 3 + 64;
        if (c->use_graph) h = (h + 511) / 512 * 512;                 // coarse steps: the graph is re-captured when this changes
        if (h < gn) gn = h;
    }
    HIPCHK(hipEventRecord(c->ev_f0[slot], s));
    bool replayed = false;
    if (c->use_graph) {
        // the captured launch list bakes in which builds of the f64 kernels run: a context captured while it had the device to
        // itself must be re-captured once another many-sequence context exists (and back), or it would keep the full-register
        // builds that cannot start beside the other's LK grid
        static const bool force_lean_g = getenv("SVO_FORCE_LEAN") && atoi(getenv("SVO_FORCE_LEAN")) != 0;
        const int shares_now = lk_gated(c) ? 1 : 0;
        const int co_now = (shares_now || force_lean_g) ? 1 : 0;
        if (!c->gexec[slot] || c->g_stride[slot] != stride || c->g_gn[slot] != gn || c->g_co[slot] != co_now) {
            if (c->gexec[slot]) { (void)hipGraphExecDestroy(c->gexec[slot]); c->gexec[slot] = nullptr; }
            hipGraph_t g = nullptr;
            bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                c->capturing = true;
                const int rc = issue_frame(c, slot, stride, gn, false, shares_now);
                c->capturing = false;
                ok = (hipStreamEndCapture(s, &g) == hipSuccess) && rc == SVO_OK && g;
            }
            if (ok) ok = hipGraphInstantiate(&c->gexec[slot], g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
            if (!ok) {                                                // capture is an optimisation: without it the same launches are issued directly
                (void)hipGetLastError();
                c->gexec[slot] = nullptr; c->use_graph = false;
            } else { c->g_stride[slot] = stride; c->g_gn[slot] = gn; c->g_co[slot] = co_now; }
        }
        if (c->use_graph) { HIPCHK(hipGraphLaunch(c->gexec[slot], s)); replayed = true; }
    }
    if (!replayed) { const int rc = issue_frame(c, slot, stride, gn, c->stage_timing); if (rc != SVO_OK) return rc; }
    c->staged_slot[slot] = !replayed && c->stage_timing;
    c->staged_inputs = false;
    HIPCHK(hipEventRecord(c->ev_done[slot], s));
    HIPCHK(hipGetLastError());
    c->head = (c->head + 1) % SVO_RING; c->inflight++;
    return SVO_OK;
}

static int collect_frame(svo_context* c, double* T_out, int* ok_out, svo_frame_stats* stats) {
    if (c->inflight <= 0) { g_err = "nothing to collect"; return SVO_ERR_STATE; }
    const int slot = c->tail, B = c->d.B;
    HIPCHK(hipEventSynchronize(c->ev_done[slot]));
    const FrameResult* r = c->h_results + (size_t)slot * B;
    for (int i = 0; i < B; i++) {
        if (T_out) memcpy(T_out + 16 * i, r[i].T, sizeof(double) * 16);
        if (ok_out) ok_out[i] = r[i].ok;
        if (stats) stats[i] = r[i].stats;
    }
    {
        int mx = 0;
        for (int i = 0; i < B; i++) if (r[i].stats.n_after_detect > mx) mx = r[i].stats.n_after_detect;
        if (mx > 0) c->lk_hint = mx;
    }
    c->last_slot = slot;
    c->tail = (c->tail + 1) % SVO_RING; c->inflight--;
    return SVO_OK;
}

extern "C" int svo_submit_batch(svo_context* c, const uint8_t* const* left_dev, const uint8_t* const* right_dev, int stride) {
    if (!c || !left_dev || !right_dev) return fail_arg("null argument");
    if (!c->projection_set) { g_err = "svo_set_projection must be called first"; return SVO_ERR_STATE; }
    if (stride < c->d.geom.W * c->d.CN) return fail_arg("stride < width * channels");
    HIPCHK(hipSetDevice(c->device));
    return enqueue_frame(c, left_dev, right_dev, stride);
}

extern "C" int svo_collect(svo_context* c, double* T_out, int* ok_out, svo_frame_stats* stats) {
    if (!c) return fail_arg("null context");
    HIPCHK(hipSetDevice(c->device));
    return collect_frame(c, T_out, ok_out, stats);
}

extern "C" int svo_process_batch(svo_context* c, const uint8_t* const* left, const uint8_t* const* right, int stride,
                                 int images_on_device, double* T_out, int* ok_out, svo_frame_stats* stats) {
    if (!c || !left || !right) return fail_arg("null argument");
    if (!c->projection_set) { g_err = "svo_set_projection must be called first"; return SVO_ERR_STATE; }
    if (stride < c->d.geom.W * c->d.CN) return fail_arg("stride < width * channels");
    if (c->inflight != 0) { g_err = "svo_process_batch with frames in flight"; return SVO_ERR_STATE; }
    HIPCHK(hipSetDevice(c->device));
    const int W = c->d.geom.W;
    int rc;
    if (images_on_device) {
        rc = enqueue_frame(c, left, right, stride);
    } else {
        // the caller's buffers are only borrowed for the duration of the call: copy to the device first (SURVEY.md §8b "Ownership")
        const size_t rowb = (size_t)W * c->d.CN;
        std::vector<const uint8_t*> lp, rp;
        if ((rc = stage_host_images(c, left, right, stride, lp, rp)) != SVO_OK) return rc;
        rc = enqueue_frame(c, lp.data(), rp.data(), (int)rowb);
    }
    if (rc != SVO_OK) return rc;
    return collect_frame(c, T_out, ok_out, stats);
}

extern "C" int svo_process(svo_context* c, const uint8_t* left, const uint8_t* right, int stride, double T_out[16], svo_frame_stats* stats) {
    if (!c) return fail_arg("null context");
    if (c->d.B != 1) return fail_arg("svo_process needs a context created with n_seq == 1");
    int ok = 0;
    const uint8_t* l[1] = {left}; const uint8_t* r[1] = {right};
    int rc = svo_process_batch(c, l, r, stride, 0, T_out, &ok, stats);
    return rc != SVO_OK ? rc : ok;
}

// Is p page-locked host memory the DMA engines can read in place (hipHostMalloc / hipHostRegister / svo_alloc_pinned)?
static bool host_pointer_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // older runtimes: an error for pageable memory
    return a.type == hipMemoryTypeHost;
}

extern "C" void* svo_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void svo_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

// Host images -> pinned staging -> device staging (one contiguous H2D copy); fills lp / rp with the device addresses.
// Images that already live in page-locked memory with packed rows skip the staging copy: the DMA reads them in place (the call
// is synchronous, the caller's buffer outlives it) — 40 us of host memcpy less per KITTI-sized pair on the single-stream path.
static int stage_host_images(svo_context* c, const uint8_t* const* left, const uint8_t* const* right, int stride,
                             std::vector<const uint8_t*>& lp, std::vector<const uint8_t*>& rp) {
    const int B = c->d.B, W = c->d.geom.W, H = c->d.geom.H;
    const size_t rowb = (size_t)W * c->d.CN, img = rowb * H;
    if (!c->staging) HIPCHK(hipMalloc((void**)&c->staging, img * 2 * B));
    if (!c->h_staging) HIPCHK(hipHostMalloc((void**)&c->h_staging, img * 2 * B));
    // rows are packed into pinned memory on the CPU (handles any stride), then contiguous async H2D copies (a 2-D copy from
    // pageable memory degenerates into per-row transfers: 3.5 ms per 1241x376 image).  All left images first, so that their
    // DMA runs while the CPU packs the right ones.
    lp.resize(B); rp.resize(B);
    c->staged_inputs = true;
    for (int cam = 0; cam < 2; cam++) {
        const uint8_t* const* src = cam ? right : left;
        bool all_direct = (size_t)stride == rowb;
        for (int i = 0; i < B && all_direct; i++) { if (!src[i]) return fail_arg("null image pointer"); all_direct = host_pointer_is_pinned(src[i]); }
        for (int i = 0; i < B; i++) {
            if (!src[i]) return fail_arg("null image pointer");
            (cam ? rp : lp)[i] = c->staging + img * (cam * B + i);
            if (all_direct) {
                HIPCHK(hipMemcpyAsync(c->staging + img * (cam * B + i), src[i], img, hipMemcpyHostToDevice, c->stream));
                continue;
            }
            uint8_t* h = c->h_staging + img * (cam * B + i);
            if ((size_t)stride == rowb) memcpy(h, src[i], img);
            else for (int y = 0; y < H; y++) memcpy(h + (size_t)y * rowb, src[i] + (size_t)y * stride, rowb);
        }
        if (!all_direct) HIPCHK(hipMemcpyAsync(c->staging + img * B * cam, c->h_staging + img * B * cam, img * B, hipMemcpyHostToDevice, c->stream));
    }
    return SVO_OK;
}

static int read_state(svo_context* c, int seq, SeqState* hs);
static int set_state(svo_context* c, const SeqState& hs);

// VisualOdometry::circularMatching as a member call on the context's own state (vo.h:374-379, vo.cpp:169-240 without the
// compaction): the T0 side is the pyramid pair the context cached (lastLeftPyramid / lastRightPyramid, vo.h:257-258), the
// T1 pyramids are built from the given images and BECOME the cached pair (vo.cpp:231-232) — so the next stereo_callback
// tracks against them, exactly as in the reference, where both entry points share those members.  Empty input returns
// before anything is cached (vo.cpp:179-181).  currentVOFeatures, the cached T0 images and frame_id are left alone.
extern "C" int svo_circular_matching(svo_context* c, const uint8_t* left_t1, const uint8_t* right_t1, int stride, int n,
                                     const float* pl0, float* pl1, float* pr1, float* pr0, float* pl0_circle, uint8_t* ok) {
    if (!c || !left_t1 || !right_t1 || n < 0) return fail_arg("bad arguments");
    if (n > 0 && (!pl0 || !pl1 || !pr1 || !pr0 || !pl0_circle || !ok)) return fail_arg("null arrays");
    if (c->d.B != 1) return fail_arg("svo_circular_matching needs a context created with n_seq == 1");
    if (stride < c->d.geom.W * c->d.CN) return fail_arg("stride < width * channels");
    if (c->inflight != 0) { g_err = "svo_circular_matching with frames in flight"; return SVO_ERR_STATE; }
    if (n == 0) return SVO_OK;                                                    // vo.cpp:179-181
    if (n > c->d.CAP) { g_err = "more points than the context's feature capacity"; return SVO_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(c->device));
    SeqState hs; int rc = read_state(c, 0, &hs); if (rc != SVO_OK) return rc;
    if (hs.frame_id < 1 || hs.slot_pyr_t0 < 0) { g_err = "no cached pyramids: call svo_process first (the reference primes them in stereo_callback, vo.cpp:47-56)"; return SVO_ERR_STATE; }
    const SeqState keep = hs;
    // every error return below leaves the context's tracking state as it was: the guard writes `keep` back unless the call
    // completes (then the state with the new cached pyramids is written instead)
    struct Restore {
        svo_context* c; const SeqState* st; bool armed = true;
        ~Restore() { if (armed) { (void)hipMemcpyAsync(c->d.st, st, sizeof(SeqState), hipMemcpyHostToDevice, c->stream); (void)hipStreamSynchronize(c->stream); } }
    } restore{c, &keep};
    int t1 = 0;
    for (int k = 0; k < 3; k++) if (k != hs.slot_img_t0 && k != hs.slot_pyr_t0) { t1 = k; break; }
    // the points go into the idle half of the feature double-buffer (scratch between frames); the state is put back below
    hs.slot_t1 = t1; hs.active = 1; hs.feat_buf = keep.feat_buf ^ 1; hs.n_feat = n;
    HIPCHK(hipMemcpyAsync(c->d.feat_xy[hs.feat_buf], pl0, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    if ((rc = set_state(c, hs)) != SVO_OK) return rc;
    std::vector<const uint8_t*> lp, rp;
    const uint8_t* l[1] = {left_t1}; const uint8_t* r[1] = {right_t1};
    if ((rc = stage_host_images(c, l, r, stride, lp, rp)) != SVO_OK) return rc;
    const uint8_t** hp = c->h_ptrs; hp[0] = lp[0]; hp[1] = rp[0];
    launch_ingest_pyramid(c->d, c->d.img_ptrs, c->d.geom.W * c->d.CN, c->stream, false);          // vo.cpp:200-201
    if (!launch_lk_chain(c->d, n, c->stream, 0)) { g_err = "no LK kernel is built for this window / lanes-per-feature / channel count"; return SVO_ERR_STATE; }   // vo.cpp:203-230 (the caller gets every pass's raw points)
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(pl1, c->d.pl1, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pr1, c->d.pr1, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pr0, c->d.pr0, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pl0_circle, c->d.plc, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(ok, c->d.okmask, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) ok[i] &= 1;                                       // bit0 = status0..3 && loop closure (vo.cpp:227-230)
    SeqState out = keep;
    out.slot_pyr_t0 = t1;                                                         // lastLeftPyramid = pyramidl1 (vo.cpp:231-232)
    rc = set_state(c, out);
    if (rc == SVO_OK) restore.armed = false;
    return rc;
}

extern "C" int svo_set_stage_timing(svo_context* c, int on) {
    if (!c) return fail_arg("null context");
    c->stage_timing = on != 0;
    return SVO_OK;
}

extern "C" int svo_get_last_timing(svo_context* c, float* lk_ms, float* frame_ms) {
    if (!c || c->last_slot < 0) return fail_arg("no frame collected yet");
    HIPCHK(hipSetDevice(c->device));
    const int s = c->last_slot;
    if (lk_ms) {
        if (!c->staged_slot[s]) { g_err = "no stage events for this frame: call svo_set_stage_timing(ctx, 1) first (and SVO_GRAPH must be off)"; return SVO_ERR_STATE; }
        HIPCHK(hipEventElapsedTime(lk_ms, c->ev_lk0[s], c->ev_lk1[s]));
    }
    if (frame_ms) HIPCHK(hipEventElapsedTime(frame_ms, c->ev_f0[s], c->ev_done[s]));
    return SVO_OK;
}

extern "C" int svo_get_stage_timing(svo_context* c, float ms[5]) {
    if (!c || !ms || c->last_slot < 0) return fail_arg("no frame collected yet");
    HIPCHK(hipSetDevice(c->device));
    const int s = c->last_slot;
    if (!c->staged_slot[s]) { g_err = "no stage events for this frame: call svo_set_stage_timing(ctx, 1) first (and SVO_GRAPH must be off)"; return SVO_ERR_STATE; }
    hipEvent_t ev[6] = {c->ev_f0[s], c->ev_pyr[s], c->ev_lk0[s], c->ev_lk1[s], c->ev_tri[s], c->ev_done[s]};
    for (int i = 0; i < 5; i++) HIPCHK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    return SVO_OK;
}

static int read_state(svo_context* c, int seq, SeqState* hs) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(hs, c->d.st + seq, sizeof(SeqState), hipMemcpyDeviceToHost));
    return SVO_OK;
}

extern "C" int svo_get_features(svo_context* c, int seq, int cap, float* xy, int* ages, int* strengths) {
    if (!c || seq < 0 || seq >= c->d.B) return fail_arg("bad context / seq");
    SeqState hs; int rc = read_state(c, seq, &hs); if (rc != SVO_OK) return rc;
    int n = hs.n_feat < cap ? hs.n_feat : cap;
    const size_t o = (size_t)seq * c->d.CAP;
    if (n > 0) {
        if (xy) HIPCHK(hipMemcpy(xy, c->d.feat_xy[hs.feat_buf] + o, sizeof(float2) * n, hipMemcpyDeviceToHost));
        if (ages) HIPCHK(hipMemcpy(ages, c->d.feat_age[hs.feat_buf] + o, sizeof(int) * n, hipMemcpyDeviceToHost));
        if (strengths) HIPCHK(hipMemcpy(strengths, c->d.feat_str[hs.feat_buf] + o, sizeof(int) * n, hipMemcpyDeviceToHost));
    }
    return hs.n_feat;
}

extern "C" int svo_get_last_tracks(svo_context* c, int seq, int cap, float* pl0, float* pr0, float* pl1, float* pr1, float* world, uint8_t* inlier) {
    if (!c || seq < 0 || seq >= c->d.B) return fail_arg("bad context / seq");
    SeqState hs; int rc = read_state(c, seq, &hs); if (rc != SVO_OK) return rc;
    int n = hs.n_tracks < cap ? hs.n_tracks : cap;
    const size_t o = (size_t)seq * c->d.CAP;
    if (n > 0) {
        if (pl0) HIPCHK(hipMemcpy(pl0, c->d.tl0 + o, sizeof(float2) * n, hipMemcpyDeviceToHost));
        if (pr0) HIPCHK(hipMemcpy(pr0, c->d.tr0 + o, sizeof(float2) * n, hipMemcpyDeviceToHost));
        if (pl1) HIPCHK(hipMemcpy(pl1, c->d.tl1 + o, sizeof(float2) * n, hipMemcpyDeviceToHost));
        if (pr1) HIPCHK(hipMemcpy(pr1, c->d.tr1 + o, sizeof(float2) * n, hipMemcpyDeviceToHost));
        if (world) HIPCHK(hipMemcpy(world, c->d.world + 3 * o, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
        if (inlier) HIPCHK(hipMemcpy(inlier, c->d.inlier + o, (size_t)n, hipMemcpyDeviceToHost));
    }
    return hs.n_tracks;
}

// ================================================================================================
// Stage-level entry points
// ================================================================================================
struct DevTmp {                                   // RAII-ish scratch allocations for the stage calls
    std::vector<void*> p;
    ~DevTmp() { for (void* q : p) (void)hipFree(q); }
    template <typename T> hipError_t get(T** out, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
        if (e == hipSuccess) { p.push_back(q); *out = (T*)q; }
        return e;
    }
};

static int use_device(int device) {
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail_arg("no such HIP device");
    HIPCHK(hipSetDevice(device));
    return SVO_OK;
}

struct CtxGuard { svo_context* c = nullptr; ~CtxGuard() { svo_destroy(c); } };

// The stage entry points need a context's worth of device buffers (pyramid slots, track arrays, RANSAC workspaces).  Creating
// and destroying one per call costs milliseconds of hipMalloc / hipFree, fine for a test and slow for a caller that uses them
// as an API (e.g. calcOpticalFlowPyrLK per frame), so the last one is kept per thread and reused when device, image size and
// configuration match and its capacity suffices.  Every call overwrites the whole sequence record, so nothing leaks from one
// call into the next.  svo_stage_cache_clear() frees it (it is deliberately not freed at thread exit: the HIP runtime may be
// gone by then).
static bool cfg_equal(const svo_config& a, const svo_config& b) {
    return a.bucket_start_row == b.bucket_start_row && a.buckets_along_height == b.buckets_along_height && a.buckets_along_width == b.buckets_along_width &&
           a.features_per_bucket == b.features_per_bucket && a.features_threshold == b.features_threshold &&
           a.pre_matching_feature_threshold == b.pre_matching_feature_threshold && a.age_threshold == b.age_threshold && a.fast_threshold == b.fast_threshold &&
           a.ransac_reprojection_error == b.ransac_reprojection_error && a.ransac_iterations == b.ransac_iterations &&
           a.optical_flow_min_eig_threshold == b.optical_flow_min_eig_threshold &&
           a.circular_matching_success_threshold == b.circular_matching_success_threshold && a.max_translation_norm == b.max_translation_norm &&
           a.max_rotation_norm == b.max_rotation_norm && a.win_w == b.win_w && a.win_h == b.win_h && a.max_level == b.max_level &&
           a.lk_max_count == b.lk_max_count && a.lk_epsilon == b.lk_epsilon && a.ransac_confidence == b.ransac_confidence &&
           a.max_features == b.max_features && a.channels == b.channels && a.lk_float_sums == b.lk_float_sums;
}
// the fields of a configuration that size or shape a context's device buffers (everything else is a parameter kernels read)
static bool cfg_same_shape(const svo_config& a, const svo_config& b) {
    return a.bucket_start_row == b.bucket_start_row && a.buckets_along_height == b.buckets_along_height && a.buckets_along_width == b.buckets_along_width &&
           a.features_per_bucket == b.features_per_bucket && a.win_w == b.win_w && a.win_h == b.win_h && a.max_level == b.max_level && a.channels == b.channels;
}
// A cached stage context takes a new configuration IN PLACE when only parameters changed (thresholds, iteration counts up to the
// allocated number, confidence, termination criteria, ...): callers that alternate such parameters between calls keep their
// buffers instead of paying a context's worth of hipMalloc / hipFree per call.
static bool stage_reconfigure(svo_context* c, const svo_config& cfg) {
    if (!cfg_same_shape(c->d.cfg, cfg) || cfg.ransac_iterations < 1 || cfg.ransac_iterations > c->k_alloc) return false;
    DevBuffers& d = c->d;
    d.cfg = cfg; d.K = cfg.ransac_iterations;
    c->lk_grid = (cfg.max_features > 0 && cfg.max_features < d.CAP) ? cfg.max_features : d.CAP;
    d.lk_mineig_cut = lk_mineig_cut(cfg.win_w, cfg.optical_flow_min_eig_threshold);
    double pc = (double)cfg.ransac_confidence; pc = pc > 0. ? pc : 0.; pc = pc < 1. ? pc : 1.;
    d.ransac_log_num = log(1. - pc > 2.2250738585072014e-308 ? 1. - pc : 2.2250738585072014e-308);
    c->lk_room = lk_registers_left(d);
    return true;
}
struct StageCache { svo_context* c = nullptr; svo_config cfg; int device = -1, w = 0, h = 0; };
// One cached context per calling thread.  The thread_local slot holds the pointer for speed; a mutex-protected registry owns the
// contexts, so that (a) a thread that exits hands its context back (its slot's destructor marks the entry free — the context is
// then REUSED by the next thread that needs one of the same shape, or freed by svo_stage_cache_clear_all), and (b) nothing is
// destroyed from a thread-exit destructor, where the HIP runtime may already be gone.  Short-lived worker threads therefore
// cost at most one context per concurrently LIVE thread, not one per thread ever started.
struct StageEntry { StageCache sc; bool in_use = false; };
static std::mutex g_stage_mu;
static std::vector<StageEntry*> g_stage_all;
struct StageSlot {
    StageEntry* e = nullptr;
    ~StageSlot() { if (e) { std::lock_guard<std::mutex> lock(g_stage_mu); e->in_use = false; e = nullptr; } }
};
static thread_local StageSlot g_stage_slot;
static StageCache& stage_cache_of_this_thread(const svo_config& cfg, int device, int w, int h, int cap) {
    if (!g_stage_slot.e) {
        std::lock_guard<std::mutex> lock(g_stage_mu);
        StageEntry* pick = nullptr;
        for (StageEntry* e : g_stage_all)                              // an orphan of the same shape first, then any orphan
            if (!e->in_use && e->sc.c && e->sc.device == device && e->sc.w == w && e->sc.h == h && e->sc.c->d.CAP >= cap) { pick = e; break; }
        if (!pick) for (StageEntry* e : g_stage_all) if (!e->in_use) { pick = e; break; }
        if (!pick) { pick = new StageEntry(); g_stage_all.push_back(pick); }
        pick->in_use = true;
        g_stage_slot.e = pick;
    }
    (void)cfg;
    return g_stage_slot.e->sc;
}
static int stage_ctx(const svo_config& cfg_in, int device, int w, int h, int cap, svo_context** out) {
    svo_config cfg = cfg_in;
    if (cfg.channels == 0) cfg.channels = 1;
    StageCache& sc = stage_cache_of_this_thread(cfg, device, w, h, cap);
    if (sc.c && sc.device == device && sc.w == w && sc.h == h && sc.c->d.CAP >= cap) {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamSynchronize(sc.c->stream));
        if (cfg_equal(sc.cfg, cfg) || stage_reconfigure(sc.c, cfg)) {
            sc.cfg = cfg;
            *out = sc.c;
            return SVO_OK;
        }
    }
    if (sc.c) { svo_destroy(sc.c); sc.c = nullptr; }
    int rc = ctx_create(&cfg, device, 1, w, h, cap, &sc.c);
    if (rc != SVO_OK) { sc.c = nullptr; return rc; }
    sc.cfg = cfg; sc.device = device; sc.w = w; sc.h = h;
    *out = sc.c;
    return SVO_OK;
}
extern "C" void svo_stage_cache_clear(void) {                       // this thread's cached context
    if (g_stage_slot.e && g_stage_slot.e->sc.c) { svo_destroy(g_stage_slot.e->sc.c); g_stage_slot.e->sc.c = nullptr; }
}
extern "C" int svo_stage_cache_clear_all(void) {                    // every cached context no live call is using (exited threads' too); returns how many were freed
    std::lock_guard<std::mutex> lock(g_stage_mu);
    int freed = 0;
    for (StageEntry* e : g_stage_all)
        if (e->sc.c && (!e->in_use || e == g_stage_slot.e)) { svo_destroy(e->sc.c); e->sc.c = nullptr; freed++; }
    return freed;
}

// Level 0 of (slot, cam) <- a host image.  Rows are packed into pinned memory first: a 2-D copy from pageable memory degenerates
// into per-row transfers (3.5 ms per 1241x376 image), the contiguous pinned copy takes ~20 us.  One pinned buffer per
// (slot, camera), so the copies of one call never overwrite each other before they have run.
static int upload_image(svo_context* c, int slot, int cam, const uint8_t* img, int stride) {
    const size_t W = c->d.geom.W, H = c->d.geom.H;
    if (!c->h_upload) HIPCHK(hipHostMalloc((void**)&c->h_upload, W * H * 6));
    uint8_t* h = c->h_upload + W * H * (size_t)(slot * 2 + cam);
    if ((size_t)stride == W) memcpy(h, img, W * H);
    else for (size_t y = 0; y < H; y++) memcpy(h + y * W, img + y * (size_t)stride, W);
    const LevelInfo& L0 = c->d.geom.lv[0];
    uint8_t* dst = c->d.pyr + pyr_index(c->d, 0, slot, cam) + L0.off;
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)L0.stride, h, W, W, H, hipMemcpyHostToDevice, c->stream));   // from pinned memory: one strided DMA
    return SVO_OK;
}
static int set_state(svo_context* c, const SeqState& hs) {
    HIPCHK(hipMemcpyAsync(c->d.st, &hs, sizeof(SeqState), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));      // hs is a stack object
    return SVO_OK;
}
static int build_pyramid_in_slot(svo_context* c, SeqState& hs, int slot) {
    hs.slot_t1 = slot;
    int rc = set_state(c, hs); if (rc != SVO_OK) return rc;
    launch_pyramid(c->d, c->stream);
    return SVO_OK;
}

extern "C" int svo_fast_score_map(int device, const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score) {
    if (!img || !score || w < 7 || h < 7 || stride < w) return fail_arg("bad image arguments");
    int rc = use_device(device); if (rc != SVO_OK) return rc;
    DevTmp t; uint8_t *dimg, *dsc;
    HIPCHK(t.get(&dimg, (size_t)w * h)); HIPCHK(t.get(&dsc, (size_t)w * h));
    HIPCHK(hipMemcpy2D(dimg, w, img, stride, w, h, hipMemcpyHostToDevice));
    launch_fast_score_map(dimg, w, h, threshold, dsc, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(score, dsc, (size_t)w * h, hipMemcpyDeviceToHost));
    return SVO_OK;
}

extern "C" int svo_fast_detect(int device, const uint8_t* img, int w, int h, int stride, int threshold,
                               int cap, float* xy, float* resp, int* n_out) {
    if (!img || !n_out || w < 7 || h < 7 || stride < w || cap < 0) return fail_arg("bad arguments");
    int rc = use_device(device); if (rc != SVO_OK) return rc;
    DevTmp t; uint8_t *dimg, *dsc; int *rows, *dn; float2* dxy; float* dresp;
    HIPCHK(t.get(&dimg, (size_t)w * h)); HIPCHK(t.get(&dsc, (size_t)w * h)); HIPCHK(t.get(&rows, (size_t)h));
    HIPCHK(t.get(&dn, 1)); HIPCHK(t.get(&dxy, (size_t)cap)); HIPCHK(t.get(&dresp, (size_t)cap));
    HIPCHK(hipMemcpy2D(dimg, w, img, stride, w, h, hipMemcpyHostToDevice));
    launch_fast_score_map(dimg, w, h, threshold, dsc, 0);
    launch_score_compact(dsc, w, h, cap, rows, dxy, dresp, dn, 0);
    HIPCHK(hipGetLastError());
    int n = 0;
    HIPCHK(hipMemcpy(&n, dn, sizeof(int), hipMemcpyDeviceToHost));
    int m = n < cap ? n : cap;
    if (m > 0 && xy) HIPCHK(hipMemcpy(xy, dxy, sizeof(float2) * m, hipMemcpyDeviceToHost));
    if (m > 0 && resp) HIPCHK(hipMemcpy(resp, dresp, sizeof(float) * m, hipMemcpyDeviceToHost));
    *n_out = n;
    return SVO_OK;
}

extern "C" int svo_bucket_filter(int device, int img_w, int img_h, int* n_io, float* xy, int* ages, int* strengths,
                                 int bah, int baw, int start_row, int per_bucket, int age_thr, int fast_thr) {
    if (!n_io || *n_io < 0 || bah < 1 || baw < 1 || per_bucket < 0 || img_w < 1 || img_h < 1) return fail_arg("bad arguments");
    int n = *n_io;
    if (n > 0 && (!xy || !ages || !strengths)) return fail_arg("null arrays");
    int rc = use_device(device); if (rc != SVO_OK) return rc;
    if (n == 0 || per_bucket == 0) { *n_io = 0; return SVO_OK; }
    const int nb = bah * baw;
    DevTmp t; float2 *dxy, *sxy, *oxy; int *dag, *dst_, *sag, *sst, *sn, *oag, *ost, *dn;
    const size_t outcap = (size_t)n;
    HIPCHK(t.get(&dxy, (size_t)n)); HIPCHK(t.get(&dag, (size_t)n)); HIPCHK(t.get(&dst_, (size_t)n));
    HIPCHK(t.get(&sxy, (size_t)nb * per_bucket)); HIPCHK(t.get(&sag, (size_t)nb * per_bucket)); HIPCHK(t.get(&sst, (size_t)nb * per_bucket));
    HIPCHK(t.get(&sn, (size_t)nb)); HIPCHK(t.get(&oxy, outcap)); HIPCHK(t.get(&oag, outcap)); HIPCHK(t.get(&ost, outcap)); HIPCHK(t.get(&dn, 1));
    HIPCHK(hipMemcpy(dxy, xy, sizeof(float2) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dag, ages, sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dst_, strengths, sizeof(int) * n, hipMemcpyHostToDevice));
    launch_bucket_general(img_w, img_h, n, dxy, dag, dst_, bah, baw, start_row, per_bucket, age_thr, fast_thr, sxy, sag, sst, sn, oxy, oag, ost, dn, 0);
    HIPCHK(hipGetLastError());
    int m = 0;
    HIPCHK(hipMemcpy(&m, dn, sizeof(int), hipMemcpyDeviceToHost));
    if (m > 0) {
        HIPCHK(hipMemcpy(xy, oxy, sizeof(float2) * m, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ages, oag, sizeof(int) * m, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(strengths, ost, sizeof(int) * m, hipMemcpyDeviceToHost));
    }
    *n_io = m;
    return SVO_OK;
}

extern "C" int svo_append_features_from_image(int device, const svo_config* cfg_in, const uint8_t* img, int w, int h, int stride,
                                              int fast_threshold, int cap, int* n_io, float* xy, int* ages, int* strengths) {
    if (!img || !n_io || *n_io < 0 || stride < w || cap < *n_io) return fail_arg("bad arguments");
    svo_config cfg; if (cfg_in) cfg = *cfg_in; else svo_config_default(&cfg);
    cfg.channels = 1;                                                             // stage entry points are single-channel
    svo_context* c = nullptr; int rc = stage_ctx(cfg, device, w, h, *n_io, &c); if (rc != SVO_OK) return rc;
    const int n = *n_io;
    if (n > 0) {
        if (!xy || !ages || !strengths) return fail_arg("null arrays");
        HIPCHK(hipMemcpyAsync(c->d.feat_xy[0], xy, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d.feat_age[0], ages, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d.feat_str[0], strengths, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    }
    rc = upload_image(c, 0, 0, img, stride); if (rc != SVO_OK) return rc;
    SeqState hs; memset(&hs, 0, sizeof(hs));
    hs.frame_id = 1; hs.active = 1; hs.slot_img_t0 = 0; hs.slot_pyr_t0 = 0; hs.slot_t1 = 1; hs.n_feat = n; hs.n_old = n; hs.feat_buf = 0;
    rc = set_state(c, hs); if (rc != SVO_OK) return rc;
    // Only detection pass 0 runs here.  When it keeps fewer than pre_matching_feature_threshold features its last emit block
    // offers them to the grid for a pass 1 that this entry point never launches, and the cached context would carry those keys
    // and row counts into the next call (phantom / wrong features): the grid state is cleared per call.
    HIPCHK(hipMemsetAsync(c->d.bucket_keys, 0, sizeof(unsigned long long) * (size_t)c->d.NB, c->stream));
    HIPCHK(hipMemsetAsync(c->d.bucket_rowcnt, 0, sizeof(int) * (size_t)c->d.cfg.buckets_along_height, c->stream));
    HIPCHK(hipMemsetAsync(c->d.emit_ticket, 0, sizeof(int), c->stream));
    launch_detect(c->d, 0, fast_threshold, c->stream);
    HIPCHK(hipGetLastError());
    int m = svo_get_features(c, 0, cap, xy, ages, strengths);
    if (m < 0) return m;
    *n_io = m;
    return SVO_OK;
}

extern "C" int svo_build_pyramid(int device, const uint8_t* img, int w, int h, int stride, int win, int max_level,
                                 uint8_t* levels_out, int64_t levels_cap, int* n_levels_out) {
    if (!img || !levels_out || !n_levels_out || stride < w) return fail_arg("bad arguments");
    svo_config cfg; svo_config_default(&cfg);
    cfg.win_w = cfg.win_h = lk_window_supported(win) ? win : 10; cfg.max_level = max_level;
    CtxGuard g; int rc = ctx_create(&cfg, device, 1, w, h, 0, &g.c); if (rc != SVO_OK) return rc;
    svo_context* c = g.c;
    const int pad = c->d.geom.pad;
    make_geometry(c->d.geom, w, h, win, max_level, pad);  // honour the caller's window for the level-stop rule
    // (pyramid buffers were sized with a window >= 7, which never yields fewer bytes than a larger window)
    Geometry chk; make_geometry(chk, w, h, cfg.win_w, max_level, pad);
    if (c->d.geom.pyr_bytes > chk.pyr_bytes) return fail_arg("window too small for this entry point");
    rc = upload_image(c, 0, 0, img, stride); if (rc != SVO_OK) return rc;
    SeqState hs; memset(&hs, 0, sizeof(hs));
    rc = build_pyramid_in_slot(c, hs, 0); if (rc != SVO_OK) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t off = 0;
    for (int l = 0; l < c->d.geom.nlevels; l++) {
        const LevelInfo& L = c->d.geom.lv[l];
        int64_t sz = (int64_t)L.w * L.h;
        if (off + sz > levels_cap) return fail_arg("levels_out too small");
        HIPCHK(hipMemcpy2D(levels_out + off, (size_t)L.w, c->d.pyr + pyr_index(c->d, 0, 0, 0) + L.off, (size_t)L.stride, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
        off += sz;
    }
    *n_levels_out = c->d.geom.nlevels;
    return SVO_OK;
}

extern "C" int svo_lk_track(int device, const uint8_t* prev_img, const uint8_t* next_img, int w, int h, int stride,
                            int n, const float* prev_pts, float* next_pts, uint8_t* status,
                            int win, int max_level, int max_count, double epsilon, double min_eig_threshold) {
    if (!prev_img || !next_img || n < 0 || stride < w) return fail_arg("bad arguments");
    if (n > 0 && (!prev_pts || !next_pts || !status)) return fail_arg("null arrays");
    svo_config cfg; svo_config_default(&cfg);
    cfg.win_w = cfg.win_h = win; cfg.max_level = max_level; cfg.lk_max_count = max_count; cfg.lk_epsilon = epsilon;
    cfg.optical_flow_min_eig_threshold = min_eig_threshold;
    svo_context* c = nullptr; int rc = stage_ctx(cfg, device, w, h, n, &c); if (rc != SVO_OK) return rc;
    if (n == 0) return SVO_OK;
    if ((rc = upload_image(c, 0, 0, prev_img, stride)) != SVO_OK) return rc;
    if ((rc = upload_image(c, 1, 0, next_img, stride)) != SVO_OK) return rc;
    SeqState hs; memset(&hs, 0, sizeof(hs));
    if ((rc = build_pyramid_in_slot(c, hs, 0)) != SVO_OK) return rc;
    if ((rc = build_pyramid_in_slot(c, hs, 1)) != SVO_OK) return rc;
    HIPCHK(hipMemcpyAsync(c->d.pl0, prev_pts, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    launch_lk_single(c->d, 0, 0, 1, 0, n, c->d.pl0, c->d.pl1, c->d.okmask, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(next_pts, c->d.pl1, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(status, c->d.okmask, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SVO_OK;
}

extern "C" int svo_circular_match(int device, const svo_config* cfg_in, const uint8_t* l0, const uint8_t* r0,
                                  const uint8_t* l1, const uint8_t* r1, int w, int h, int stride,
                                  int n, const float* pl0, float* pl1, float* pr1, float* pr0, float* pl0_circle, uint8_t* ok) {
    if (!l0 || !r0 || !l1 || !r1 || n < 0 || stride < w) return fail_arg("bad arguments");
    if (n > 0 && (!pl0 || !pl1 || !pr1 || !pr0 || !pl0_circle || !ok)) return fail_arg("null arrays");
    svo_config cfg; if (cfg_in) cfg = *cfg_in; else svo_config_default(&cfg);
    cfg.max_features = 0; cfg.channels = 1;
    svo_context* c = nullptr; int rc = stage_ctx(cfg, device, w, h, n, &c); if (rc != SVO_OK) return rc;
    if (n == 0) return SVO_OK;                                                    // vo.cpp:179-181
    if ((rc = upload_image(c, 0, 0, l0, stride)) != SVO_OK) return rc;
    if ((rc = upload_image(c, 0, 1, r0, stride)) != SVO_OK) return rc;
    if ((rc = upload_image(c, 1, 0, l1, stride)) != SVO_OK) return rc;
    if ((rc = upload_image(c, 1, 1, r1, stride)) != SVO_OK) return rc;
    SeqState hs; memset(&hs, 0, sizeof(hs));
    if ((rc = build_pyramid_in_slot(c, hs, 0)) != SVO_OK) return rc;
    hs.frame_id = 1; hs.active = 1; hs.slot_img_t0 = 0; hs.slot_pyr_t0 = 0; hs.n_feat = n; hs.feat_buf = 0;
    if ((rc = build_pyramid_in_slot(c, hs, 1)) != SVO_OK) return rc;             // leaves slot_t1 = 1
    HIPCHK(hipMemcpyAsync(c->d.feat_xy[0], pl0, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    if (!launch_lk_chain(c->d, n, c->stream, 0)) { g_err = "no LK kernel is built for this window / lanes-per-feature / channel count"; return SVO_ERR_STATE; }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(pl1, c->d.pl1, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pr1, c->d.pr1, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pr0, c->d.pr0, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pl0_circle, c->d.plc, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(ok, c->d.okmask, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) ok[i] &= 1;                                       // bit0 = status0..3 && loop closure (vo.cpp:227-230)
    return SVO_OK;
}

extern "C" int svo_find_close_points(int device, int n, const float* p1, const float* p2, float threshold, uint8_t* ok) {
    if (n < 0 || (n > 0 && (!p1 || !p2 || !ok))) return fail_arg("bad arguments");
    int rc = use_device(device); if (rc != SVO_OK) return rc;
    if (n == 0) return SVO_OK;
    DevTmp t; float2 *a, *b; uint8_t* o;
    HIPCHK(t.get(&a, (size_t)n)); HIPCHK(t.get(&b, (size_t)n)); HIPCHK(t.get(&o, (size_t)n));
    HIPCHK(hipMemcpy(a, p1, sizeof(float2) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b, p2, sizeof(float2) * n, hipMemcpyHostToDevice));
    launch_find_close(n, a, b, threshold, o, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(ok, o, (size_t)n, hipMemcpyDeviceToHost));
    return SVO_OK;
}

extern "C" int svo_triangulate(int device, const float Pl[12], const float Pr[12], int n, const float* pts_l, const float* pts_r, float* xyz) {
    if (!Pl || !Pr || n < 0 || (n > 0 && (!pts_l || !pts_r || !xyz))) return fail_arg("bad arguments");
    svo_config cfg; svo_config_default(&cfg);
    svo_context* c = nullptr; int rc = stage_ctx(cfg, device, 64, 64, n, &c); if (rc != SVO_OK) return rc;
    if (n == 0) return SVO_OK;
    SeqState hs; memset(&hs, 0, sizeof(hs));
    hs.active = 1; hs.fail_reason = 0; hs.n_tracks = n;
    memcpy(hs.Pl, Pl, sizeof(float) * 12); memcpy(hs.Pr, Pr, sizeof(float) * 12);
    HIPCHK(hipMemcpyAsync(c->d.tl0, pts_l, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d.tr0, pts_r, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    if ((rc = set_state(c, hs)) != SVO_OK) return rc;
    launch_triangulate(c->d, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(xyz, c->d.world, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SVO_OK;
}

extern "C" int svo_camera_to_world(int device, const float K[9], int n, const float* cam_pts, const float* world_pts,
                                   double R[9], double t[3], int* inliers, int* n_inliers, int* success,
                                   int ransac_iterations, float reproj_error, float confidence, int* iters_run) {
    if (!K || !R || !t || !n_inliers || !success || n < 0 || (n > 0 && (!cam_pts || !world_pts))) return fail_arg("bad arguments");
    *n_inliers = 0; *success = 0; if (iters_run) *iters_run = 0;
    if (n < 4) return fail_arg("cameraToWorld needs at least 4 points (cv::solvePnPRansac asserts npoints >= 4)");
    svo_config cfg; svo_config_default(&cfg);
    cfg.ransac_iterations = ransac_iterations > 1 ? ransac_iterations : 1;
    cfg.ransac_reprojection_error = reproj_error; cfg.ransac_confidence = confidence;
    cfg.features_threshold = 0;                                                  // cameraToWorld itself has no inlier-count gate
    cfg.max_translation_norm = 1e300; cfg.max_rotation_norm = 1e300;             // nor motion gates
    svo_context* c = nullptr; int rc = stage_ctx(cfg, device, 64, 64, n, &c); if (rc != SVO_OK) return rc;
    SeqState hs; memset(&hs, 0, sizeof(hs));
    hs.active = 1; hs.fail_reason = 0; hs.n_tracks = n; hs.n_feat = n; hs.feat_buf = 0;
    memcpy(hs.K, K, sizeof(float) * 9); memcpy(hs.R, R, sizeof(double) * 9); memcpy(hs.t, t, sizeof(double) * 3);
    HIPCHK(hipMemcpyAsync(c->d.tl1, cam_pts, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d.world, world_pts, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
    if ((rc = set_state(c, hs)) != SVO_OK) return rc;
    if (n == 4) launch_pnp_p3p(c->d, c->stream);                                 // solvepnp.cpp: npoints == 4 -> one direct P3P
    else { launch_pnp_subsets(c->d, c->stream); launch_pnp(c->d, c->stream); }   // n == 5: one direct EPnP (handled on the device)
    HIPCHK(hipGetLastError());
    if ((rc = read_state(c, 0, &hs)) != SVO_OK) return rc;
    if (iters_run) *iters_run = hs.pnp_iters;
    if (hs.pnp_best < 0) return SVO_OK;                                          // success = false: R, t untouched (vo.cpp:307-311)
    memcpy(R, hs.R, sizeof(double) * 9); memcpy(t, hs.t, sizeof(double) * 3);
    *success = 1; *n_inliers = hs.n_inliers;
    if (inliers && hs.n_inliers > 0)
        HIPCHK(hipMemcpy(inliers, c->d.inl_idx, sizeof(int) * hs.n_inliers, hipMemcpyDeviceToHost));
    return SVO_OK;
}

extern "C" int svo_inverse_transform(int device, const double R[9], const double t[3], double T[16]) {
    if (!R || !t || !T) return fail_arg("null argument");
    int rc = use_device(device); if (rc != SVO_OK) return rc;
    DevTmp tmp; double* buf;
    HIPCHK(tmp.get(&buf, 9 + 3 + 16));
    HIPCHK(hipMemcpy(buf, R, sizeof(double) * 9, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf + 9, t, sizeof(double) * 3, hipMemcpyHostToDevice));
    launch_inverse_transform(buf, buf + 9, buf + 12, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(T, buf + 12, sizeof(double) * 16, hipMemcpyDeviceToHost));
    return SVO_OK;
}
