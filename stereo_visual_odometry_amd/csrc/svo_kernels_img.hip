// svo_kernels_img.hip — frame bookkeeping, ingest, pyrDown, FAST-9/16 + NMS, bucketing, compaction.
// gfx950 (wave64).  All integer / byte work: HBM-bound by design, no MFMA.
//
// Behaviour follows the reference call sites (cited per kernel); the arithmetic of the OpenCV
// functions they call is restated independently, not translated.
#include "svo_internal.hpp"

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// ------------------------------------------------------------------------------------------------
// frame begin / end: the slot bookkeeping of stereo_callback (vo.cpp:47-56, 74-75) and of the
// pyramid cache in circularMatching (vo.cpp:179-181 vs 231-232: the cache is NOT refreshed when
// there were no points to match — the "stale pyramid" quirk).
// ------------------------------------------------------------------------------------------------
// The per-frame reset, run by one thread per sequence inside the ingest kernel (begin_frame).  The T1 slot is a function of two
// fields the reset does not write, so every ingest block derives it for itself.  (The bucket keys need no clearing: every
// detection pass leaves them zero, see k_bucket_emit.)
__device__ __forceinline__ int free_slot(const SeqState& s) {
    int t1 = 0;
    for (int c = 0; c < 3; c++) if (c != s.slot_img_t0 && c != s.slot_pyr_t0) { t1 = c; break; }
    return t1;
}
// the slot no field refers to while a frame is in flight: where the NEXT frame's pyramids can be built ahead of time.  Read at any
// moment of the current frame (even while k_frame_end rewrites the fields: the set in use only shrinks there) it is free.
__device__ __forceinline__ int next_slot(const SeqState& s) {
    const int a = s.slot_img_t0, b = s.slot_pyr_t0, t = s.slot_t1;
    for (int c = 0; c < SVO_PYR_SLOTS; c++) if (c != a && c != b && c != t) return c;
    return 0;
}
__device__ __forceinline__ void frame_begin(SeqState& s, int t1 = -1) {
    s.n_old = s.n_feat;
    s.active = s.frame_id > 0;
    s.slot_t1 = t1 >= 0 ? t1 : free_slot(s);
    s.do_second = 0; s.n_lk = 0; s.n_tracks = 0; s.n_circ = 0; s.n_inliers = 0; s.ok = 0;
    s.pnp_best = -1; s.pnp_iters = 0; s.pnp_good = 0; s.pnp_drawn = 0;
    s.fail_reason = s.active ? 0 : 1;
    svo_frame_stats z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    s.stats = z;
}

__global__ void k_frame_end(DevBuffers d, int ring_slot) {
    int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= d.B) return;
    SeqState& s = d.st[seq];
    s.slot_img_t0 = s.slot_t1;                                   // vo.cpp:48-49 / 74-75
    if (!s.active || s.n_lk > 0) s.slot_pyr_t0 = s.slot_t1;      // vo.cpp:50-53 / 231-232 (skipped on the early return :179-181)
    s.frame_id++;
    FrameResult& r = d.results[(size_t)ring_slot * d.B + seq];
    for (int i = 0; i < 16; i++) r.T[i] = s.last_T[i];           // fail_result = last_transform (vo.cpp:43-44); on success last_T was just updated
    r.ok = s.ok;
    s.stats.fail_reason = s.fail_reason;
    s.stats.n_features_out = s.n_feat;
    s.stats.n_into_lk = s.n_lk;
    s.stats.n_after_circular = s.n_circ;
    s.stats.n_after_bounds = s.n_tracks;
    s.stats.n_inliers = s.n_inliers;
    s.stats.ransac_iters = s.pnp_iters;
    r.stats = s.stats;
}

void launch_frame_end(const DevBuffers& d, int ring_slot, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_end, dim3((d.B + 63) / 64), dim3(64), 0, st, d, ring_slot);
}

// ------------------------------------------------------------------------------------------------
// ingest: copy the caller's two images into level 0 of the T1 pyramid slot (the deep copies of
// vo.cpp:74-75 / the level-0 copy of cv::buildOpticalFlowPyramid).  One thread = 4 pixels of a row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ingest(DevBuffers d, const uint8_t* const* srcs, int stride, int begin_frame) {
    const int seq = blockIdx.z, cam = blockIdx.y;
    const int W = d.geom.W, H = d.geom.H;
    const int quads_per_row = (W + 3) >> 2;
    const int total = quads_per_row * H;
    const uint8_t* src = srcs[cam * d.B + seq];
    const int slot = begin_frame ? free_slot(d.st[seq]) : d.st[seq].slot_t1;
    if (begin_frame && blockIdx.x == 0 && cam == 0 && threadIdx.x == 0) frame_begin(d.st[seq]);
    uint8_t* dst = d.pyr + pyr_index(d, seq, slot, cam) + d.geom.lv[0].off;
    const int dstride = d.geom.lv[0].stride;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        int y = q / quads_per_row, x = (q - y * quads_per_row) << 2;
        const uint8_t* sp = src + (size_t)y * stride + x;
        uint8_t* dp = dst + (size_t)y * dstride + x;
        int n = W - x < 4 ? W - x : 4;
        for (int k = 0; k < n; k++) dp[k] = sp[k];
    }
}

// Colour form: interleaved BGR rows -> three planes (level 0 of the three per-plane pyramids) and, for the left camera, the
// W x H byte image made of the first W bytes of every row, which is what cv::FAST scans in a 3-channel Mat.
__global__ __launch_bounds__(256) void k_ingest_bgr(DevBuffers d, const uint8_t* const* srcs, int stride, int begin_frame) {
    const int seq = blockIdx.z, cam = blockIdx.y;
    const int W = d.geom.W, H = d.geom.H;
    const int total = W * H;
    const uint8_t* src = srcs[cam * d.B + seq];
    const int slot = begin_frame ? free_slot(d.st[seq]) : d.st[seq].slot_t1;
    if (begin_frame && blockIdx.x == 0 && cam == 0 && threadIdx.x == 0) frame_begin(d.st[seq]);
    uint8_t* p0 = d.pyr + pyr_index(d, seq, slot, cam) + d.geom.lv[0].off;
    const int dstride = d.geom.lv[0].stride;
    uint8_t* fi = d.fastimg + fastimg_index(d, seq, slot);
    const size_t pb = (size_t)d.geom.pyr_bytes;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const uint8_t* sp = src + (size_t)y * stride;
        const size_t o = (size_t)y * dstride + x;
        p0[o] = sp[3 * x]; p0[pb + o] = sp[3 * x + 1]; p0[2 * pb + o] = sp[3 * x + 2];
        if (cam == 0) fi[i] = sp[x];
    }
}

// begin_frame: the kernel also performs the per-frame reset of stereo_callback (frame pipeline); the stage entry points pass false
void launch_ingest(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs, int stride, hipStream_t st, bool begin_frame) {
    if (d.CN == 3) {
        int gx = (d.geom.W * d.geom.H + 255) / 256; if (gx > 2048) gx = 2048;
        hipLaunchKernelGGL(k_ingest_bgr, dim3(gx, 2, d.B), dim3(256), 0, st, d, left_right_dev_ptrs, stride, (int)begin_frame);
        return;
    }
    int total = ((d.geom.W + 3) >> 2) * d.geom.H;
    int gx = (total + 255) / 256; if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_ingest, dim3(gx, 2, d.B), dim3(256), 0, st, d, left_right_dev_ptrs, stride, (int)begin_frame);
}

// ------------------------------------------------------------------------------------------------
// pyrDown (the level loop of cv::buildOpticalFlowPyramid, vo.cpp:50,52,200,201):
// dst(x,y) = (sum_{i,j} k_i k_j src(2x+i-2, 2y+j-2) + 128) >> 8, k = [1 4 6 4 1], REFLECT_101.
// 32x8 output tile per 256-thread block, source tile staged in LDS, separable in two LDS passes.
// ------------------------------------------------------------------------------------------------
#define PD_TW 32
#define PD_TH 8
__global__ __launch_bounds__(256) void k_pyrdown(DevBuffers d, int level, int ahead) {
    const int plane = blockIdx.z % d.CN, sc = blockIdx.z / d.CN;          // every colour plane is its own pyramid
    const int seq = sc / 2, cam = sc & 1;
    const LevelInfo ls = d.geom.lv[level - 1], ld = d.geom.lv[level];
    uint8_t* base = d.pyr + pyr_index(d, seq, ahead ? d.st[seq].slot_next : d.st[seq].slot_t1, cam) + (size_t)plane * d.geom.pyr_bytes;
    const uint8_t* src = base + ls.off;
    uint8_t* dst = base + ld.off;
    constexpr int SW = 2 * PD_TW + 3, SH = 2 * PD_TH + 3;        // 67 x 19 source tile
    __shared__ __attribute__((aligned(4))) uint8_t tile[SH][SW + 1];
    __shared__ unsigned short hrow[SH][PD_TW];                   // horizontal pass result (<= 16*255)
    const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
    const int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;
    if (sx0 >= 0 && sy0 >= 0 && sx0 + SW + 1 <= ls.w && sy0 + SH <= ls.h) {
        // interior tile (almost all of them): 17 unaligned dword loads per source row, no border arithmetic
        static_assert((SW + 1) % 4 == 0, "tile rows are whole dwords");
        struct __attribute__((packed, aligned(1))) UD { unsigned v; };
        constexpr int DPR = (SW + 1) / 4;
        for (int i = threadIdx.x; i < DPR * SH; i += 256) {
            int ty = i / DPR, c = i - ty * DPR;
            const unsigned v = reinterpret_cast<const UD*>(src + (size_t)(sy0 + ty) * ls.stride + sx0 + 4 * c)->v;
            *reinterpret_cast<unsigned*>(&tile[ty][4 * c]) = v;
        }
    } else {
        for (int i = threadIdx.x; i < SW * SH; i += 256) {
            int ty = i / SW, tx = i - ty * SW;
            tile[ty][tx] = src[(size_t)reflect101(sy0 + ty, ls.h) * ls.stride + reflect101(sx0 + tx, ls.w)];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SH * PD_TW; i += 256) {
        int ty = i / PD_TW, x = i - ty * PD_TW;
        const uint8_t* r = &tile[ty][2 * x];
        hrow[ty][x] = (unsigned short)(r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4]);
    }
    __syncthreads();
    {
        int y = threadIdx.x / PD_TW, x = threadIdx.x - y * PD_TW;
        int gx = ox + x, gy = oy + y;
        if (gx < ld.w && gy < ld.h) {
            int v = hrow[2 * y + 2][x] * 6 + (hrow[2 * y + 1][x] + hrow[2 * y + 3][x]) * 4 + hrow[2 * y][x] + hrow[2 * y + 4][x];
            dst[(size_t)gy * ld.stride + gx] = (uint8_t)((v + 128) >> 8);
        }
    }
}

// ---- fewer launches for the front of a frame (a lone stream is launch-bound there: a 5 us kernel every 4.5 us of host time) ----
// k_ingest_pyr1: ingest and the first pyrDown in one launch (single-channel contexts).  A block stages the 67 x 19 source tile (at 32 x 8) of
// its 32 x 8 level-1 outputs straight from the caller's image, writes the 64 x 16 level-0 pixels it owns and the level-1 tile.
// (bodies take their block coordinates as arguments so that k_front_a / k_front_b below can run two of them in one launch)
// TW x TH = the block's tile of level 1 (2 TW x 2 TH pixels of level 0).  Lone streams: 32 x 8 (many blocks for one image).  Many-sequence
// contexts: 64 x 16 — at a thousand images per launch the kernel was bound by block turnover (962 000 blocks of 1 KB of output each:
// 0.68 ms per 512 sequences = 1.6 TB/s), not by LDS or HBM; four times the work per block and half the halo.
template <int TW, int TH>
static __device__ __forceinline__ void ingest_pyr1_body(const DevBuffers& d, const uint8_t* const* srcs, int stride, int begin_frame, int bx, int by, int bz) {
    const int seq = bz >> 1, cam = bz & 1;
    const LevelInfo ls = d.geom.lv[0], ld = d.geom.lv[1];
    const uint8_t* src = srcs[cam * d.B + seq];
    // begin_frame: 0 = into the T1 slot as it stands (stage entry points), 1 = the frame pipeline's own ingest (free slot + the per-frame
    // reset), 2 = ahead of the frame (image stream): into SeqState::slot_next, which k_pick_next chose; nothing of the state is written
    const int slot = begin_frame == 2 ? d.st[seq].slot_next : begin_frame ? free_slot(d.st[seq]) : d.st[seq].slot_t1;
    if (begin_frame == 1 && bx == 0 && by == 0 && cam == 0 && threadIdx.x == 0) frame_begin(d.st[seq]);
    uint8_t* base = d.pyr + pyr_index(d, seq, slot, cam);
    uint8_t* l0 = base + ls.off; uint8_t* dst = base + ld.off;
    constexpr int SW = 2 * TW + 3, SH = 2 * TH + 3;              // 67 x 19 source tile at 32 x 8
    static_assert((SW + 1) % 4 == 0 && TW % 2 == 0, "rows of whole dwords");
    __shared__ __attribute__((aligned(4))) uint8_t tile[SH][SW + 1];
    __shared__ __attribute__((aligned(4))) unsigned short hrow[SH][TW];
    const int ox = bx * TW, oy = by * TH;
    const int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;
    struct __attribute__((packed, aligned(1))) UD { unsigned v; };
    if (sx0 >= 0 && sy0 >= 0 && sx0 + SW + 1 <= ls.w && sy0 + SH <= ls.h) {
        constexpr int DPR = (SW + 1) / 4;
        for (int i = threadIdx.x; i < DPR * SH; i += 256) {
            int ty = i / DPR, c = i - ty * DPR;
            const unsigned v = reinterpret_cast<const UD*>(src + (size_t)(sy0 + ty) * stride + sx0 + 4 * c)->v;
            *reinterpret_cast<unsigned*>(&tile[ty][4 * c]) = v;
        }
        __syncthreads();
        // the block's own 2 TW x 2 TH level-0 pixels, a dword at a time (tile bytes 4c + 2 .. 4c + 5: two aligned LDS dwords, shifted)
        for (int i = threadIdx.x; i < (TW / 2) * (2 * TH); i += 256) {
            const int ty = i / (TW / 2), c = i - ty * (TW / 2);
            const unsigned lo = *reinterpret_cast<const unsigned*>(&tile[ty + 2][4 * c]), hi = *reinterpret_cast<const unsigned*>(&tile[ty + 2][4 * c + 4]);
            UD u; u.v = __builtin_amdgcn_alignbyte(hi, lo, 2);
            *reinterpret_cast<UD*>(l0 + (size_t)(2 * oy + ty) * ls.stride + 2 * ox + 4 * c) = u;
        }
    } else {
        for (int i = threadIdx.x; i < SW * SH; i += 256) {
            int ty = i / SW, tx = i - ty * SW;
            tile[ty][tx] = src[(size_t)reflect101(sy0 + ty, ls.h) * stride + reflect101(sx0 + tx, ls.w)];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * TW * 2 * TH; i += 256) {
            const int ty = i / (2 * TW), tx = i - ty * (2 * TW);
            const int gx = 2 * ox + tx, gy = 2 * oy + ty;
            if (gx < ls.w && gy < ls.h) l0[(size_t)gy * ls.stride + gx] = tile[ty + 2][tx + 2];   // in range: the tile holds the pixel itself
        }
    }
    // horizontal 1-4-6-4-1 pass, TWO outputs per thread: outputs 2j and 2j + 1 of a row read tile bytes 4j .. 4j + 6 = two aligned
    // LDS dwords (ten byte reads before), and leave as one dword of hrow
    for (int i = threadIdx.x; i < SH * (TW / 2); i += 256) {
        const int ty = i / (TW / 2), j = i - ty * (TW / 2);
        const unsigned a = *reinterpret_cast<const unsigned*>(&tile[ty][4 * j]), b = *reinterpret_cast<const unsigned*>(&tile[ty][4 * j + 4]);
        const unsigned r0 = a & 255u, r1 = (a >> 8) & 255u, r2 = (a >> 16) & 255u, r3 = a >> 24, r4 = b & 255u, r5 = (b >> 8) & 255u, r6 = (b >> 16) & 255u;
        const unsigned h0 = r2 * 6 + (r1 + r3) * 4 + r0 + r4, h1 = r4 * 6 + (r3 + r5) * 4 + r2 + r6;
        *reinterpret_cast<unsigned*>(&hrow[ty][2 * j]) = h0 | (h1 << 16);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * (TW / 2); i += 256) {   // vertical pass, two outputs per thread: five LDS dwords, one 2-byte store
        const int y = i / (TW / 2), x = 2 * (i - y * (TW / 2));
        const int gx = ox + x, gy = oy + y;
        if (gx < ld.w && gy < ld.h) {
            unsigned q[5];
#pragma unroll
            for (int k = 0; k < 5; k++) q[k] = *reinterpret_cast<const unsigned*>(&hrow[2 * y + k][x]);
            const unsigned v0 = (q[2] & 0xFFFFu) * 6 + ((q[1] & 0xFFFFu) + (q[3] & 0xFFFFu)) * 4 + (q[0] & 0xFFFFu) + (q[4] & 0xFFFFu);
            const unsigned v1 = (q[2] >> 16) * 6 + ((q[1] >> 16) + (q[3] >> 16)) * 4 + (q[0] >> 16) + (q[4] >> 16);
            uint8_t* o = dst + (size_t)gy * ld.stride + gx;
            const unsigned b0 = (v0 + 128) >> 8, b1 = (v1 + 128) >> 8;
            if (gx + 1 < ld.w) *reinterpret_cast<unsigned short*>(o) = (unsigned short)(b0 | (b1 << 8));   // gx is even, rows are 16-byte aligned
            else o[0] = (uint8_t)b0;
        }
    }
}
#define IG_TW 64
#define IG_TH 16
template <int TW, int TH>
__global__ __launch_bounds__(256) void k_ingest_pyr1(DevBuffers d, const uint8_t* const* srcs, int stride, int begin_frame) {
    ingest_pyr1_body<TW, TH>(d, srcs, stride, begin_frame, blockIdx.x, blockIdx.y, blockIdx.z);
}
static void launch_ingest_pyr1(const DevBuffers& d, const uint8_t* const* ptrs, int stride, int begin_frame, hipStream_t st) {
    if (d.B > SVO_LONE_MAX_SEQ) {
        dim3 g((d.geom.lv[1].w + IG_TW - 1) / IG_TW, (d.geom.lv[1].h + IG_TH - 1) / IG_TH, d.B * 2);
        hipLaunchKernelGGL((k_ingest_pyr1<IG_TW, IG_TH>), g, dim3(256), 0, st, d, ptrs, stride, begin_frame);
    } else {
        dim3 g((d.geom.lv[1].w + PD_TW - 1) / PD_TW, (d.geom.lv[1].h + PD_TH - 1) / PD_TH, d.B * 2);
        hipLaunchKernelGGL((k_ingest_pyr1<PD_TW, PD_TH>), g, dim3(256), 0, st, d, ptrs, stride, begin_frame);
    }
}

// k_pyrdown2: levels l+1 AND l+2 from level l in one launch.  A block owns a 16 x 8 tile of level l+2, i.e. 32 x 16 of level l+1;
// it computes the 35 x 19 level-(l+1) pixels its own tile needs (the 3-pixel rim is recomputed by the neighbours: +30 % work on
// levels that are 1/16 and 1/64 of the image) from a 73 x 41 source tile.  Rim positions outside level l+1 are never read:
// REFLECT_101 folds them onto positions inside the tile.
#define P2_TW 16
#define P2_TH 8
static __device__ __forceinline__ void pyrdown2_body(const DevBuffers& d, int level, int bx, int by, int bz, int ahead = 0) {
    const int plane = bz % d.CN, sc = bz / d.CN;
    const int seq = sc / 2, cam = sc & 1;
    const LevelInfo ls = d.geom.lv[level], lm = d.geom.lv[level + 1], ld = d.geom.lv[level + 2];
    uint8_t* base = d.pyr + pyr_index(d, seq, ahead ? d.st[seq].slot_next : d.st[seq].slot_t1, cam) + (size_t)plane * d.geom.pyr_bytes;
    const uint8_t* src = base + ls.off;
    uint8_t* mid = base + lm.off; uint8_t* dst = base + ld.off;
    constexpr int MW = 2 * P2_TW + 3, MH = 2 * P2_TH + 3;        // 35 x 19 of the middle level
    constexpr int SW = 2 * MW + 3, SH = 2 * MH + 3;              // 73 x 41 of the source level
    __shared__ __attribute__((aligned(4))) uint8_t tile[SH][SW + 3];
    __shared__ unsigned short hrow[SH][MW + 1];
    __shared__ uint8_t mtile[MH][MW + 1];
    __shared__ unsigned short hrow2[MH][P2_TW];
    const int ox = bx * P2_TW, oy = by * P2_TH;                  // level l+2
    const int mx0 = 2 * ox - 2, my0 = 2 * oy - 2;                // level l+1
    const int sx0 = 2 * mx0 - 2, sy0 = 2 * my0 - 2;              // level l
    if (sx0 >= 0 && sy0 >= 0 && sx0 + SW + 3 <= ls.w && sy0 + SH <= ls.h) {
        // tiles inside the source level (nearly all of them): rows of 19 unaligned dwords instead of 73 reflected byte loads
        struct __attribute__((packed, aligned(1))) UD { unsigned v; };
        constexpr int DPR = (SW + 3) / 4;
        for (int i = threadIdx.x; i < DPR * SH; i += 256) {
            const int ty = i / DPR, c = i - ty * DPR;
            *reinterpret_cast<unsigned*>(&tile[ty][4 * c]) = reinterpret_cast<const UD*>(src + (size_t)(sy0 + ty) * ls.stride + sx0 + 4 * c)->v;
        }
    } else {
        for (int i = threadIdx.x; i < SW * SH; i += 256) {
            int ty = i / SW, tx = i - ty * SW;
            tile[ty][tx] = src[(size_t)reflect101(sy0 + ty, ls.h) * ls.stride + reflect101(sx0 + tx, ls.w)];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SH * MW; i += 256) {
        int ty = i / MW, x = i - ty * MW;
        const uint8_t* r = &tile[ty][2 * x];
        hrow[ty][x] = (unsigned short)(r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MH * MW; i += 256) {
        int y = i / MW, x = i - y * MW;
        int v = hrow[2 * y + 2][x] * 6 + (hrow[2 * y + 1][x] + hrow[2 * y + 3][x]) * 4 + hrow[2 * y][x] + hrow[2 * y + 4][x];
        const uint8_t m = (uint8_t)((v + 128) >> 8);
        mtile[y][x] = m;
        const int gx = mx0 + x, gy = my0 + y;
        if (x >= 2 && x < 2 + 2 * P2_TW && y >= 2 && y < 2 + 2 * P2_TH && gx < lm.w && gy < lm.h) mid[(size_t)gy * lm.stride + gx] = m;   // the owned 32 x 16
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MH * P2_TW; i += 256) {
        int ty = i / P2_TW, x = i - ty * P2_TW;
        // row / column of the middle tile that holds this (possibly folded) position; outputs beyond the level's edge (never
        // stored) may fold outside the tile: clamped, so that no index leaves the array
        int gy = reflect101(my0 + ty, lm.h) - my0;
        gy = gy < 0 ? 0 : gy > MH - 1 ? MH - 1 : gy;
        int t[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int gx = reflect101(mx0 + 2 * x + k, lm.w) - mx0;
            gx = gx < 0 ? 0 : gx > MW - 1 ? MW - 1 : gx;
            t[k] = mtile[gy][gx];
        }
        hrow2[ty][x] = (unsigned short)(t[2] * 6 + (t[1] + t[3]) * 4 + t[0] + t[4]);
    }
    __syncthreads();
    if (threadIdx.x < P2_TW * P2_TH) {
        int y = threadIdx.x / P2_TW, x = threadIdx.x - y * P2_TW;
        int gx = ox + x, gy = oy + y;
        if (gx < ld.w && gy < ld.h) {
            int v = hrow2[2 * y + 2][x] * 6 + (hrow2[2 * y + 1][x] + hrow2[2 * y + 3][x]) * 4 + hrow2[2 * y][x] + hrow2[2 * y + 4][x];
            dst[(size_t)gy * ld.stride + gx] = (uint8_t)((v + 128) >> 8);
        }
    }
}
__global__ __launch_bounds__(256) void k_pyrdown2(DevBuffers d, int level, int ahead) { pyrdown2_body(d, level, blockIdx.x, blockIdx.y, blockIdx.z, ahead); }

// k_pad_pyramid: the REFLECT_101 border of every level of the T1 slot — what cv::buildOpticalFlowPyramid's copyMakeBorder leaves
// around each level (pyrBorder = BORDER_REFLECT_101).  Pixel (x, y) outside the level takes level(reflect101(y), reflect101(x)) — the
// same index function the LK kernel's per-byte border path used before the border was materialised, so the values it sees are the
// same bytes.  One thread per border DWORD (row starts, the pad and pixel (0, 0) are 4-byte aligned): the ring of a level is cut
// into its top and bottom bands (pad rows of the padded width) and, per image row, the left band and the right band — the latter
// from the aligned x at or below the level's width, so up to three pixels of the level itself are rewritten with their own values.
// The four source bytes of a dword are consecutive ascending (one unaligned dword load) or descending (load + byte swap) except
// where a dword straddles a fold of a level narrower than the pad: those go byte by byte.  (First version: one thread, one byte
// load and one byte store per border pixel — 0.39 ms per 512 sequences; the image stream's kernels run in the gap between two LK
// launches, so their time is whole-job time.)
__global__ __launch_bounds__(256) void k_pad_pyramid(DevBuffers d, int ahead) {
    const int plane = blockIdx.z % d.CN, sc = blockIdx.z / d.CN;
    const int seq = sc / 2, cam = sc & 1, level = blockIdx.y;
    const LevelInfo L = d.geom.lv[level];
    const int P = d.geom.pad, w = L.w, h = L.h;
    uint8_t* img = d.pyr + pyr_index(d, seq, ahead ? d.st[seq].slot_next : d.st[seq].slot_t1, cam) + (size_t)plane * d.geom.pyr_bytes + L.off;
    const int rowdw = (w + 2 * P + 3) >> 2;                       // dwords of a band row, from x = -P (the last one may reach into the row's stride padding)
    const int xr0 = w & ~3, ldw = P >> 2, sdw = ldw + ((w + P - xr0 + 3) >> 2);   // left + right dwords of an image row
    const int band = P * rowdw, total = 2 * band + h * sdw;
    struct __attribute__((packed, aligned(1))) UD { unsigned v; };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int x, y;
        if (i < 2 * band) { const int j = i < band ? i : i - band, r = j / rowdw; x = (j - r * rowdw) * 4 - P; y = i < band ? r - P : h + r; }
        else { const int j = i - 2 * band; y = j / sdw; const int k = j - y * sdw; x = k < ldw ? 4 * k - P : xr0 + 4 * (k - ldw); }
        const uint8_t* srow = img + (size_t)reflect101(y, h) * L.stride;
        const int s0 = reflect101(x, w), s1 = reflect101(x + 1, w), s2 = reflect101(x + 2, w), s3 = reflect101(x + 3, w);
        unsigned v;
        if (s1 == s0 + 1 && s2 == s0 + 2 && s3 == s0 + 3) v = reinterpret_cast<const UD*>(srow + s0)->v;
        else if (s1 == s0 - 1 && s2 == s0 - 2 && s3 == s0 - 3) v = __builtin_bswap32(reinterpret_cast<const UD*>(srow + s3)->v);
        else v = (unsigned)srow[s0] | ((unsigned)srow[s1] << 8) | ((unsigned)srow[s2] << 16) | ((unsigned)srow[s3] << 24);
        *reinterpret_cast<unsigned*>(img + (ptrdiff_t)y * L.stride + x) = v;
    }
}
static void launch_pad_pyramid_into(const DevBuffers& d, hipStream_t st, int ahead) {
    const LevelInfo& L0 = d.geom.lv[0];
    const int P = d.geom.pad, ring0 = 2 * P * (L0.w + 2 * P) + 2 * P * L0.h;
    int gx = (ring0 + 4 * 256 - 1) / (4 * 256); if (gx < 1) gx = 1; if (gx > 64) gx = 64;      // ~one border dword per thread at level 0 (the smaller levels stride less)
    hipLaunchKernelGGL(k_pad_pyramid, dim3(gx, d.geom.nlevels, d.B * 2 * d.CN), dim3(256), 0, st, d, ahead);
}
void launch_pad_pyramid(const DevBuffers& d, hipStream_t st) { launch_pad_pyramid_into(d, st, 0); }

// levels first .. nlevels-1 from level first-1: pairs of levels per launch where two remain
static void launch_pyramid_from(const DevBuffers& d, int first, hipStream_t st, int ahead = 0) {
    int l = first;
    while (l < d.geom.nlevels) {
        if (l + 1 < d.geom.nlevels) {
            dim3 g((d.geom.lv[l + 1].w + P2_TW - 1) / P2_TW, (d.geom.lv[l + 1].h + P2_TH - 1) / P2_TH, d.B * 2 * d.CN);
            hipLaunchKernelGGL(k_pyrdown2, g, dim3(256), 0, st, d, l - 1, ahead);
            l += 2;
        } else {
            dim3 g((d.geom.lv[l].w + PD_TW - 1) / PD_TW, (d.geom.lv[l].h + PD_TH - 1) / PD_TH, d.B * 2 * d.CN);
            hipLaunchKernelGGL(k_pyrdown, g, dim3(256), 0, st, d, l, ahead);
            l += 1;
        }
    }
}
void launch_pyramid(const DevBuffers& d, hipStream_t st) { launch_pyramid_from(d, 1, st); launch_pad_pyramid(d, st); }
// ingest + all pyramid levels of the T1 slot (vo.cpp:74-75, 200-201): single-channel contexts fuse the ingest with the first level
void launch_ingest_pyramid(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs, int stride, hipStream_t st, bool begin_frame) {
    if (d.CN == 1 && d.geom.nlevels >= 2) {
        launch_ingest_pyr1(d, left_right_dev_ptrs, stride, (int)begin_frame, st);
        launch_pyramid_from(d, 2, st);
        launch_pad_pyramid(d, st);
        return;
    }
    launch_ingest(d, left_right_dev_ptrs, stride, st, begin_frame);
    launch_pyramid(d, st);
}

// ---- the next frame's pyramids, ahead of the frame (many-sequence contexts; svo_api.hip issue_frame) ----
// k_pick_next: one thread per sequence names the slot (a single decision per sequence: the blocks of the ingest that follows must
// all write the same one, whatever k_frame_end of the frame in flight does to the fields meanwhile).
__global__ void k_pick_next(DevBuffers d) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq < d.B) d.st[seq].slot_next = next_slot(d.st[seq]);
}
__global__ void k_frame_begin(DevBuffers d) {
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq < d.B) frame_begin(d.st[seq], d.st[seq].slot_next);
}
bool ingest_ahead_applies(const DevBuffers& d) {
    static const bool off = getenv("SVO_INGEST_AHEAD") && atoi(getenv("SVO_INGEST_AHEAD")) == 0;
    return !off && d.B > SVO_LONE_MAX_SEQ && d.CN == 1 && d.geom.nlevels >= 2;
}
void launch_ingest_pyramid_ahead(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs, int stride, hipStream_t st) {
    hipLaunchKernelGGL(k_pick_next, dim3((d.B + 63) / 64), dim3(64), 0, st, d);
    launch_ingest_pyr1(d, left_right_dev_ptrs, stride, 2, st);
    launch_pyramid_from(d, 2, st, 1);
    launch_pad_pyramid_into(d, st, 1);
}
void launch_frame_begin(const DevBuffers& d, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_begin, dim3((d.B + 63) / 64), dim3(64), 0, st, d);
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 score + strict 3x3 NMS (cv::FAST(img, kps, th, true) at feature_set.cpp:61).
// 64x16 output tile per 256-thread block: 72x24 pixels staged in LDS (3-px circle halo + 1-px NMS
// halo), u8 scores for the 66x18 ring-extended tile in LDS, then NMS.  Survivors either go straight
// into the bucket grid with one 64-bit atomicMax (frame pipeline) or into a dense score map (stage API).
// ------------------------------------------------------------------------------------------------
#define FT_W 64
#define FT_H 16
#define FT_PW (FT_W + 8)
#define FT_PH (FT_H + 8)
#define FT_SW (FT_W + 2)
#define FT_SH (FT_H + 2)

// Bresenham circle r=3 in OpenCV order
#define CIRC_DX {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1}
#define CIRC_DY {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3}

// >= 9 contiguous set bits in a circular 16-bit mask
__device__ __forceinline__ bool run9(unsigned m) {
    unsigned m2 = m | (m << 16);
    unsigned a = m2 & (m2 >> 1);
    unsigned b = a & (a >> 2);
    unsigned c = b & (b >> 4);
    return ((c & (m2 >> 8)) & 0xFFFFu) != 0;
}

// score of the pixel at LDS position (py,px): 0 if not a corner, else max(t, best 9-arc min|diff|) - 1
__device__ __forceinline__ int fast_score(const uint8_t (*pix)[FT_PW + 4], int py, int px, int t) {
    constexpr int cdx[16] = CIRC_DX, cdy[16] = CIRC_DY;
    int v = pix[py][px];
    int dd[16];
    unsigned mb = 0, md = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int p = pix[py + cdy[k]][px + cdx[k]];
        dd[k] = v - p;
        md |= (unsigned)(p < v - t) << k;       // darker than centre
        mb |= (unsigned)(p > v + t) << k;       // brighter
    }
    if (!run9(mb) && !run9(md)) return 0;
    // sliding 9-window minimum / maximum over the circular sequence
    int best = t, worst = -t;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn = dd[k], mx = dd[k];
#pragma unroll
        for (int j = 1; j < 9; j++) { int e = dd[(k + j) & 15]; mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
        best = mn > best ? mn : best;           // a0 = max(a0, arc-min of (v - p))
        worst = mx < worst ? mx : worst;        // b0 = min(b0, arc-max of (v - p))
    }
    // cornerScore: a0 from the "centre brighter" side, then b0 = min(-a0, arc-max...) ; score = -b0 - 1
    int b0 = -best;
    b0 = worst < b0 ? worst : b0;
    return -b0 - 1;
}

struct BucketGrid { int bucket_h, bucket_w, bah, baw, start_row, age_thr, fast_thr; };

__device__ __forceinline__ unsigned long long make_bucket_key(int score, unsigned order, int strength) {
    int s = score + 32768; s = s < 1 ? 1 : (s > 65535 ? 65535 : s);
    return ((unsigned long long)s << 48) | ((unsigned long long)(0xFFFFFFFFu - order) << 16) | (unsigned long long)(strength & 0xFFFF);
}

// Bucket::add_feature for capacity 1 as one 64-bit atomicMax; the first candidate of a bucket also counts it into its grid row,
// which lets k_bucket_emit place a row's winners without scanning the rows before it.
__device__ __forceinline__ void bucket_offer(const DevBuffers& d, int seq, int bh, int bw, unsigned long long key) {
    const unsigned long long old = atomicMax(&d.bucket_keys[(size_t)seq * d.NB + bh * d.cfg.buckets_along_width + bw], key);
    if (old == 0ull) atomicAdd(&d.bucket_rowcnt[(size_t)seq * d.cfg.buckets_along_height + bh], 1);
}

// the tracks the feature set already holds, offered to the grid (feature_set.cpp:20-53, 122-124) — the first pass's k_fast blocks
// share them out before they scan their tiles, the second pass's offer is made by the last block of the first pass's emit
static __device__ __forceinline__ void offer_tracks(const DevBuffers& d, int seq, int fb, int n, int first, int step) {
    const float2* xy = d.feat_xy[fb] + (size_t)seq * d.CAP;
    const int* age = d.feat_age[fb] + (size_t)seq * d.CAP;
    const int* str = d.feat_str[fb] + (size_t)seq * d.CAP;
    for (int i = first; i < n; i += step) {
        float2 p = xy[i];
        int a = age[i], st = str[i];
        int bh = (int)(p.y / (float)d.bucket_h), bw = (int)(p.x / (float)d.bucket_w);   // feature_set.cpp:122-123
        if (p.x < 0.f || p.y < 0.f || bh < d.cfg.bucket_start_row || bh >= d.cfg.buckets_along_height || bw >= d.cfg.buckets_along_width) continue;
        if (a >= d.cfg.age_threshold) continue;                                           // feature_set.cpp:26
        int score = a + (st - d.cfg.fast_threshold) / 20;
        bucket_offer(d, seq, bh, bw, make_bucket_key(score, (unsigned)i, st));
    }
}
// MODE 0: frame pipeline, survivors go straight to the bucket keys (features_per_bucket == 1).  MODE 1: one image -> score map
// (stage API).  MODE 2: frame pipeline -> per-sequence score map (features_per_bucket > 1: the general walk needs the keypoint list).
template <int MODE>
static __device__ __forceinline__ void fast_body(const uint8_t* img_single, int w_single, int h_single, uint8_t* score_out,
                                                 const DevBuffers& d, int pass, int threshold, int bx, int by, int bz, int gdx, int gdy) {
    constexpr bool TO_BUCKETS = MODE == 0;
    __shared__ __attribute__((aligned(4))) uint8_t pix[FT_PH][FT_PW + 4];
    __shared__ uint8_t sc[FT_SH][FT_SW + 2];
    __shared__ unsigned short cand[FT_SH * FT_SW];               // screened pixels of the tile (order is irrelevant)
    __shared__ int ncand;
    const int seq = bz;
    int W, H, istride; const uint8_t* img;                       // istride: row pitch of img (a pyramid level 0 carries its border)
    if (MODE != 1) {
        const SeqState& s = d.st[seq];
        // Pass 0 reads nothing the per-frame reset writes (`active` is frame_id > 0, `n_old` is n_feat until the first emit), so
        // that it may run in the SAME launch as the ingest blocks one of whose threads performs that reset (k_front_a).
        if (pass == 0 ? !(s.frame_id > 0) : !s.do_second) return;
        W = d.geom.W; H = d.geom.H;
        // FAST runs on the PREVIOUS left image (vo.cpp:325); for a BGR context on the byte image cv::FAST really scans
        img = d.CN == 3 ? d.fastimg + fastimg_index(d, seq, s.slot_img_t0) : d.pyr + pyr_index(d, seq, s.slot_img_t0, 0) + d.geom.lv[0].off;
        istride = d.CN == 3 ? W : d.geom.lv[0].stride;
        if (MODE == 2) score_out = d.score + (size_t)seq * W * H;
        if (TO_BUCKETS && pass == 0)                                 // the existing tracks enter the grid here (no launch of their own)
            offer_tracks(d, seq, s.feat_buf, s.n_feat, (by * gdx + bx) * 256 + threadIdx.x, gdx * gdy * 256);
    } else { W = w_single; H = h_single; img = img_single; istride = w_single; }
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    const int x0 = bx * FT_W, y0 = by * FT_H;
    if (x0 >= 4 && y0 >= 4 && x0 - 4 + FT_PW <= W && y0 - 4 + FT_PH <= H) {
        // interior tile: 18 unaligned dword loads per row instead of 72 guarded byte loads
        static_assert(FT_PW % 4 == 0, "tile rows are whole dwords");
        struct __attribute__((packed, aligned(1))) UD { unsigned v; };
        constexpr int DPR = FT_PW / 4;
        for (int i = threadIdx.x; i < DPR * FT_PH; i += 256) {
            int py = i / DPR, c = i - py * DPR;
            const unsigned v = reinterpret_cast<const UD*>(img + (size_t)(y0 - 4 + py) * istride + (x0 - 4) + 4 * c)->v;
            *reinterpret_cast<unsigned*>(&pix[py][4 * c]) = v;
        }
    } else {
        for (int i = threadIdx.x; i < FT_PH * FT_PW; i += 256) {
            int py = i / FT_PW, px = i - py * FT_PW;
            int gx = x0 - 4 + px, gy = y0 - 4 + py;
            pix[py][px] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? img[(size_t)gy * istride + gx] : (uint8_t)0;
        }
    }
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();
    // Screening: a 9-arc of the 16-pixel circle contains at least one pixel of every antipodal pair, so a corner needs a
    // darker (or a brighter) member in each of the pairs (0,8) and (4,12).  Typically ~7 % of the pixels pass (th = 20);
    // only those are queued for the full 16-pixel test, which then runs on densely packed waves.
    for (int i = threadIdx.x; i < FT_SH * FT_SW; i += 256) {
        int sy = i / FT_SW, sx = i - sy * FT_SW;
        int gx = x0 - 1 + sx, gy = y0 - 1 + sy;
        sc[sy][sx] = 0;
        if (gx >= 3 && gx < W - 3 && gy >= 3 && gy < H - 3) {
            const int py = sy + 3, px = sx + 3;
            const int v = pix[py][px], lo = v - threshold, hi = v + threshold;
            const int a = pix[py + 3][px], b = pix[py - 3][px], c = pix[py][px + 3], e = pix[py][px - 3];   // circle 0, 8, 4, 12
            const bool dark = (a < lo || b < lo) && (c < lo || e < lo);
            const bool bright = (a > hi || b > hi) && (c > hi || e > hi);
            if (dark || bright) cand[atomicAdd(&ncand, 1)] = (unsigned short)i;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ncand; k += 256) {
        const int i = cand[k];
        int sy = i / FT_SW, sx = i - sy * FT_SW;
        sc[sy][sx] = (uint8_t)fast_score(pix, sy + 3, sx + 3, threshold);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FT_W * FT_H; i += 256) {
        int oy = i / FT_W, ox = i - oy * FT_W;
        int gx = x0 + ox, gy = y0 + oy;
        if (gx >= W || gy >= H) continue;
        int s = sc[oy + 1][ox + 1];
        bool keep = s != 0 &&
                    s > sc[oy + 1][ox] && s > sc[oy + 1][ox + 2] &&
                    s > sc[oy][ox] && s > sc[oy][ox + 1] && s > sc[oy][ox + 2] &&
                    s > sc[oy + 2][ox] && s > sc[oy + 2][ox + 1] && s > sc[oy + 2][ox + 2];
        if (TO_BUCKETS) {
            if (keep) {
                // feature_set.cpp:83-87 (age 0, strength = response) + :122-124 bucket index + Bucket::add_feature (:20-53)
                int bh = (int)((float)gy / (float)d.bucket_h), bw = (int)((float)gx / (float)d.bucket_w);
                if (bh >= d.cfg.bucket_start_row && bh < d.cfg.buckets_along_height && bw < d.cfg.buckets_along_width && 0 < d.cfg.age_threshold) {
                    int score = 0 + (s - d.cfg.fast_threshold) / 20;                       // feature_set.cpp:16-18
                    const unsigned n_old = pass == 0 ? (unsigned)d.st[seq].n_feat : (unsigned)d.st[seq].n_old;   // pass 0: the set is still the old one
                    unsigned order = n_old + (unsigned)(gy * W + gx);                      // raster rank keeps cv::FAST's output order
                    bucket_offer(d, seq, bh, bw, make_bucket_key(score, order, s));
                }
            }
        } else {
            score_out[(size_t)gy * W + gx] = keep ? (uint8_t)s : (uint8_t)0;
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_fast(const uint8_t* img_single, int w_single, int h_single, uint8_t* score_out,
                                              DevBuffers d, int pass, int threshold) {
    fast_body<MODE>(img_single, w_single, h_single, score_out, d, pass, threshold, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// The SECOND detection pass of a many-sequence context: launched every frame, needed by a sequence that kept < 100 features (vo.cpp:327)
// — on a healthy scene by none.  One block per tile and sequence (123 000 blocks at 256 sequences) costs 0.15-0.25 ms on the frame's
// critical chain just to look at do_second and leave; here a sequence gets FAST_STRIDED_BLOCKS blocks that walk its tiles.
#define FAST_STRIDED_BLOCKS 32
__global__ __launch_bounds__(256) void k_fast_strided(DevBuffers d, int pass, int threshold, int fx, int fy) {
    const SeqState& s = d.st[blockIdx.z];
    if (pass == 0 ? !(s.frame_id > 0) : !s.do_second) return;
    for (int t = blockIdx.x; t < fx * fy; t += gridDim.x) {
        fast_body<0>(nullptr, 0, 0, nullptr, d, pass, threshold, t % fx, t / fx, blockIdx.z, fx, fy);
        __syncthreads();                                              // the tile arrays in LDS are reused by the next tile
    }
}

void launch_fast_score_map(const uint8_t* img_dev, int w, int h, int threshold, uint8_t* score_dev, hipStream_t st) {
    DevBuffers dummy = {};
    dim3 g((w + FT_W - 1) / FT_W, (h + FT_H - 1) / FT_H, 1);
    hipLaunchKernelGGL(k_fast<1>, g, dim3(256), 0, st, img_dev, w, h, score_dev, dummy, 0, threshold);
}

// raster-ordered compaction of a score map into keypoints (cv::FAST output order + KeyPoint::convert, feature_set.cpp:62-66)
__global__ void k_score_row_count(const uint8_t* score, int w, int h, int* row_counts) {
    int y = blockIdx.x;
    int cnt = 0;
    for (int x = threadIdx.x; x < w; x += 64) cnt += score[(size_t)y * w + x] != 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (threadIdx.x == 0) row_counts[y] = cnt;
}
__global__ void k_scan_rows(int* row_counts, int h, int* n_out) {          // single thread; h is a few hundred
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        for (int y = 0; y < h; y++) { int c = row_counts[y]; row_counts[y] = acc; acc += c; }
        *n_out = acc;
    }
}
__global__ void k_score_row_emit(const uint8_t* score, int w, int h, const int* row_off, int cap, float2* xy, float* resp) {
    int y = blockIdx.x;
    int base = row_off[y];
    for (int x0 = 0; x0 < w; x0 += 64) {
        int x = x0 + threadIdx.x;
        int s = x < w ? score[(size_t)y * w + x] : 0;
        unsigned long long m = __ballot(s != 0);
        if (s) {
            int idx = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
            if (idx < cap) { xy[idx] = make_float2((float)x, (float)y); resp[idx] = (float)s; }
        }
        base += __popcll(m);
    }
}
void launch_score_compact(const uint8_t* score_dev, int w, int h, int cap, int* row_counts_dev, float2* xy_dev, float* resp_dev, int* n_dev, hipStream_t st) {
    hipLaunchKernelGGL(k_score_row_count, dim3(h), dim3(64), 0, st, score_dev, w, h, row_counts_dev);
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(64), 0, st, row_counts_dev, h, n_dev);
    hipLaunchKernelGGL(k_score_row_emit, dim3(h), dim3(64), 0, st, score_dev, w, h, row_counts_dev, cap, xy_dev, resp_dev);
}

// ------------------------------------------------------------------------------------------------
// Bucketing for features_per_bucket == 1 (the reference default, vo.h:65).
// Bucket::add_feature's "first in, replace the minimum iff strictly better" with one slot is
// argmax(score) with ties -> lowest input index, i.e. one 64-bit atomicMax per candidate with
// key = (score, ~input_index, strength).  Input order = existing tracks first, then the new FAST
// hits in raster order (feature_set.cpp:83-87).  Emission is bucket-raster order (:132-146).
// ------------------------------------------------------------------------------------------------
// Emit the winners in bucket-raster order (feature_set.cpp:132-146): one block per (grid row, sequence).  The rows' occupancy
// counts (bucket_offer) give a block its first output position; inside the row a ballot scan orders the winners.  Each block
// zeroes the keys it has read and the LAST block of a sequence (ticket) zeroes the row counts and publishes the new feature set,
// so keys, counts and tickets are all zero again when the next pass or frame starts — no clearing launch anywhere.  If the first
// pass kept too few features (vo.cpp:327) that last block also offers the new set to the grid for the second pass.
#define EMIT_THREADS 256
#define EMIT_WAVES (EMIT_THREADS / 64)
static __device__ __forceinline__ void bucket_emit_body(const DevBuffers& d, int pass, int row, int seq, int n_rows) {
    SeqState& s = d.st[seq];
    if (pass == 0 ? !s.active : !s.do_second) return;
    __shared__ int sh_before[EMIT_WAVES], sh_all[EMIT_WAVES], sh_cnt[EMIT_WAVES], sh_last;
    const int fb = s.feat_buf, n_old = s.n_old, W = d.geom.W;
    const int baw = d.cfg.buckets_along_width, bah = d.cfg.buckets_along_height;
    unsigned long long* keys = d.bucket_keys + (size_t)seq * d.NB + (size_t)row * baw;
    int* rowcnt = d.bucket_rowcnt + (size_t)seq * bah;
    const float2* oxy = d.feat_xy[fb] + (size_t)seq * d.CAP;
    const int* oage = d.feat_age[fb] + (size_t)seq * d.CAP;
    const int* ostr = d.feat_str[fb] + (size_t)seq * d.CAP;
    float2* nxy = d.feat_xy[fb ^ 1] + (size_t)seq * d.CAP;
    int* nage = d.feat_age[fb ^ 1] + (size_t)seq * d.CAP;
    int* nstr = d.feat_str[fb ^ 1] + (size_t)seq * d.CAP;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // winners in the rows before this one, and in all rows
    int before = 0, all = 0;
    for (int r = threadIdx.x; r < bah; r += EMIT_THREADS) { const int c = rowcnt[r]; all += c; if (r < row) before += c; }
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
    if (lane == 0) { sh_before[wv] = before; sh_all[wv] = all; }
    __syncthreads();
    before = 0; all = 0;
    for (int w = 0; w < EMIT_WAVES; w++) { before += sh_before[w]; all += sh_all[w]; }
    int run = before;
    for (int b0 = 0; b0 < baw; b0 += EMIT_THREADS) {
        const int bw = b0 + threadIdx.x;
        unsigned long long k = 0ull;
        if (bw < baw) { k = keys[bw]; if (k != 0ull) keys[bw] = 0ull; }
        const unsigned long long m = __ballot(k != 0ull);
        __syncthreads();                                              // sh_cnt of the previous chunk has been read
        if (lane == 0) sh_cnt[wv] = __popcll(m);
        __syncthreads();
        int pos = run + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; w++) pos += sh_cnt[w];
        for (int w = 0; w < EMIT_WAVES; w++) run += sh_cnt[w];
        if (k != 0ull && pos < d.CAP) {
            const unsigned order = 0xFFFFFFFFu - (unsigned)((k >> 16) & 0xFFFFFFFFull);
            if (order < (unsigned)n_old) { nxy[pos] = oxy[order]; nage[pos] = oage[order]; nstr[pos] = ostr[order]; }
            else {
                const unsigned pixi = order - (unsigned)n_old;
                const int y = (int)(pixi / (unsigned)W), x = (int)(pixi - (unsigned)y * (unsigned)W);
                nxy[pos] = make_float2((float)x, (float)y); nage[pos] = 0; nstr[pos] = (int)(k & 0xFFFFull);
            }
        }
    }
    // ---- the last block of the sequence to get here publishes the result.  Every block has read the state it needs by now, so
    // the ticket needs no fence — except when a second pass follows (every block can tell: it knows the total), where the last
    // block reads the features the others wrote.  (An agent-scope fence writes back and invalidates the XCD's L2: thousands of
    // them per frame cost the LK kernel of the other context a third of its speed.)
    const int total = all < d.CAP ? all : d.CAP;
    const bool second = pass == 0 && total < d.cfg.pre_matching_feature_threshold;               // vo.cpp:327
    if (second) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) sh_last = atomicAdd(&d.emit_ticket[seq], 1) == n_rows - 1;
    __syncthreads();
    if (!sh_last) return;
    if (second) __threadfence();
    for (int r = threadIdx.x; r < bah; r += EMIT_THREADS) rowcnt[r] = 0;
    if (threadIdx.x == 0) {
        d.emit_ticket[seq] = 0;
        s.n_feat = total; s.feat_buf = fb ^ 1;
        s.stats.n_after_detect = total;
        if (pass == 0) { s.do_second = second; if (second) s.n_old = total; }
        else s.stats.second_pass = 1;
    }
    if (second) {                                                     // block-uniform
        __threadfence();
        __syncthreads();
        offer_tracks(d, seq, fb ^ 1, total, threadIdx.x, EMIT_THREADS);
    }
}
__global__ __launch_bounds__(EMIT_THREADS) void k_bucket_emit(DevBuffers d, int pass) { bucket_emit_body(d, pass, blockIdx.x, blockIdx.y, gridDim.x); }
// the same with EMIT_STRIDED_BLOCKS blocks per sequence walking the grid rows (the second pass of many-sequence contexts, see
// k_fast_strided): the ticket still counts ROWS, so the block that finishes the last row publishes
#define EMIT_STRIDED_BLOCKS 8
__global__ __launch_bounds__(EMIT_THREADS) void k_bucket_emit_strided(DevBuffers d, int pass, int n_rows) {
    const SeqState& s = d.st[blockIdx.y];
    if (pass == 0 ? !s.active : !s.do_second) return;
    for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
        bucket_emit_body(d, pass, row, blockIdx.y, n_rows);
        __syncthreads();
    }
}

// ---- the front of a lone stream's frame in two launches instead of four.  Ingest + pyramid and detection are independent
// chains (FAST runs on the PREVIOUS left image, vo.cpp:325): on a nearly empty GPU their kernels ran one after the other, each
// a few microseconds of work behind a launch.  k_front_a = {ingest + level 1 (with the per-frame reset)  ||  FAST pass 0},
// k_front_b = {levels 2 and 3  ||  emit of pass 0}: blocks of both kinds in one grid, told apart by their linear index; the
// two halves of a launch touch disjoint data (pass 0 of k_fast reads no field the reset writes).  Lone-stream contexts only
// (SVO_LONE_MAX_SEQ): with many sequences every kernel fills the GPU by itself and the separate launches stay.
static_assert(EMIT_THREADS == 256, "k_front_b runs emit blocks beside 256-thread pyramid blocks");
__global__ __launch_bounds__(256) void k_front_a(DevBuffers d, const uint8_t* const* srcs, int stride, int ax, int ay, int n_a, int fx, int fy, int threshold) {
    const int i = blockIdx.x;
    if (i < n_a) { ingest_pyr1_body<PD_TW, PD_TH>(d, srcs, stride, 1, i % ax, (i / ax) % ay, i / (ax * ay)); return; }
    const int j = i - n_a;
    fast_body<0>(nullptr, 0, 0, nullptr, d, 0, threshold, j % fx, (j / fx) % fy, j / (fx * fy), fx, fy);
}
__global__ __launch_bounds__(256) void k_front_b(DevBuffers d, int px, int py, int n_p, int n_rows) {
    const int i = blockIdx.x;
    if (i < n_p) { pyrdown2_body(d, 1, i % px, (i / px) % py, i / (px * py)); return; }
    const int j = i - n_p;
    bucket_emit_body(d, 0, j % n_rows, j / n_rows, n_rows);
}
// true if the fused front applies to this context (then it has been launched): single-channel, four pyramid levels or more,
// one feature per bucket, a lone stream
bool launch_front_fused(const DevBuffers& d, const uint8_t* const* left_right_dev_ptrs, int stride, hipStream_t st) {
    static const bool off = getenv("SVO_FRONT_FUSED") && atoi(getenv("SVO_FRONT_FUSED")) == 0;
    if (off || d.B > SVO_LONE_MAX_SEQ || d.CN != 1 || d.geom.nlevels < 4 || d.cfg.features_per_bucket != 1) return false;
    const int ax = (d.geom.lv[1].w + PD_TW - 1) / PD_TW, ay = (d.geom.lv[1].h + PD_TH - 1) / PD_TH, n_a = ax * ay * d.B * 2;
    const int fx = (d.geom.W + FT_W - 1) / FT_W, fy = (d.geom.H + FT_H - 1) / FT_H, n_f = fx * fy * d.B;
    hipLaunchKernelGGL(k_front_a, dim3(n_a + n_f), dim3(256), 0, st, d, left_right_dev_ptrs, stride, ax, ay, n_a, fx, fy, d.cfg.fast_threshold);
    const int px = (d.geom.lv[3].w + P2_TW - 1) / P2_TW, py = (d.geom.lv[3].h + P2_TH - 1) / P2_TH, n_p = px * py * d.B * 2;
    const int n_rows = d.cfg.buckets_along_height;
    hipLaunchKernelGGL(k_front_b, dim3(n_p + n_rows * d.B), dim3(256), 0, st, d, px, py, n_p, n_rows);
    launch_pyramid_from(d, 4, st);                                    // a fifth level and beyond (cfg3)
    launch_pad_pyramid(d, st);
    launch_detect(d, 1, -1, st);                                      // the second pass exits at once unless needed (vo.cpp:327-332)
    return true;
}

// ------------------------------------------------------------------------------------------------
// features_per_bucket > 1 inside the frame pipeline: FeatureSet::appendFeaturesFromImage as written (feature_set.cpp:75-89):
// the pass's input list = existing tracks, then cv::FAST's keypoints in raster order (age 0, strength = response), walked in
// that order through Bucket::add_feature (fill, then replace the first minimum iff strictly better, :20-53), emitted in
// bucket-raster order (:132-146).  One thread per bucket walks the whole list: O(N x buckets), the price of a capacity the
// reference itself only uses in its unit tests (main.cpp:125, 152-157); the default capacity 1 never comes here.
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ bool pass_runs(const SeqState& s, int pass) { return pass == 0 ? s.active : s.do_second; }

__global__ void k_gen_row_count(DevBuffers d, int pass) {
    const int seq = blockIdx.y, y = blockIdx.x, W = d.geom.W, H = d.geom.H;
    if (!pass_runs(d.st[seq], pass)) return;
    const uint8_t* score = d.score + (size_t)seq * W * H;
    int cnt = 0;
    for (int x = threadIdx.x; x < W; x += 64) cnt += score[(size_t)y * W + x] != 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (threadIdx.x == 0) d.kp_rows[(size_t)seq * H + y] = cnt;
}
__global__ void k_gen_scan_rows(DevBuffers d, int pass) {          // one lane per sequence; H is a few hundred
    const int seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= d.B) return;
    SeqState& s = d.st[seq];
    if (!pass_runs(s, pass)) return;
    int* rows = d.kp_rows + (size_t)seq * d.geom.H;
    int acc = s.n_feat;                                              // keypoints are appended after the existing tracks
    for (int y = 0; y < d.geom.H; y++) { int c = rows[y]; rows[y] = acc; acc += c; }
    d.n_cand[seq] = acc < d.KPCAP ? acc : d.KPCAP;
    s.n_old = s.n_feat;
}
__global__ void k_gen_emit_candidates(DevBuffers d, int pass) {
    const int seq = blockIdx.y, W = d.geom.W, H = d.geom.H;
    const SeqState& s = d.st[seq];
    if (!pass_runs(s, pass)) return;
    const size_t co = (size_t)seq * d.KPCAP, fo = (size_t)seq * d.CAP;
    if ((int)blockIdx.x == H) {                                      // the extra block copies the existing tracks to the head of the list
        const int fb = s.feat_buf;
        for (int i = threadIdx.x; i < s.n_feat; i += 64) {
            d.cand_xy[co + i] = d.feat_xy[fb][fo + i]; d.cand_age[co + i] = d.feat_age[fb][fo + i]; d.cand_str[co + i] = d.feat_str[fb][fo + i];
        }
        return;
    }
    const int y = blockIdx.x;
    const uint8_t* score = d.score + (size_t)seq * W * H;
    int base = d.kp_rows[(size_t)seq * H + y];
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + threadIdx.x;
        const int sc = x < W ? score[(size_t)y * W + x] : 0;
        const unsigned long long m = __ballot(sc != 0);
        if (sc) {
            const int idx = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
            if (idx < d.KPCAP) { d.cand_xy[co + idx] = make_float2((float)x, (float)y); d.cand_age[co + idx] = 0; d.cand_str[co + idx] = sc; }   // feature_set.cpp:83-87
        }
        base += __popcll(m);
    }
}
__global__ void k_gen_bucket_walk(DevBuffers d, int pass) {
    const int seq = blockIdx.y, b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.NB || !pass_runs(d.st[seq], pass)) return;
    const int baw = d.cfg.buckets_along_width, bah = d.cfg.buckets_along_height, per = d.cfg.features_per_bucket;
    const int cap = (b / baw) >= d.cfg.bucket_start_row ? per : 0;                     // feature_set.cpp:113-116
    const size_t co = (size_t)seq * d.KPCAP, so = ((size_t)seq * d.NB + b) * per;
    float2* sx = d.slot_xy + so; int* sa = d.slot_age + so; int* ss = d.slot_str + so;
    const int n = d.n_cand[seq], fthr = d.cfg.fast_threshold;
    int cnt = 0;
    for (int i = 0; i < n && cap > 0; i++) {
        const float2 p = d.cand_xy[co + i];
        const int bh = (int)(p.y / (float)d.bucket_h), bw = (int)(p.x / (float)d.bucket_w);   // :122-123
        if (bh < 0 || bh >= bah || bw < 0 || bw >= baw || bh * baw + bw != b) continue;
        const int a = d.cand_age[co + i], st = d.cand_str[co + i];
        if (a >= d.cfg.age_threshold) continue;                                        // :26
        if (cnt < cap) { sx[cnt] = p; sa[cnt] = a; ss[cnt] = st; cnt++; }
        else {
            const int score = a + (st - fthr) / 20;
            int smin = sa[0] + (ss[0] - fthr) / 20, imin = 0;
            for (int k = 1; k < cnt; k++) { const int c = sa[k] + (ss[k] - fthr) / 20; if (c < smin) { smin = c; imin = k; } }
            if (score > smin) { sx[imin] = p; sa[imin] = a; ss[imin] = st; }
        }
    }
    d.slot_n[(size_t)seq * d.NB + b] = cnt;
}
#define GEN_EMIT_THREADS 1024
#define GEN_EMIT_WAVES (GEN_EMIT_THREADS / 64)
__global__ __launch_bounds__(GEN_EMIT_THREADS) void k_gen_bucket_emit(DevBuffers d, int pass) {
    const int seq = blockIdx.x;
    SeqState& s = d.st[seq];
    if (!pass_runs(s, pass)) return;
    __shared__ int wave_tot[GEN_EMIT_WAVES];
    __shared__ int s_total;
    const int fb = s.feat_buf, nb = d.NB, per = d.cfg.features_per_bucket;
    const int* sn = d.slot_n + (size_t)seq * nb;
    float2* nxy = d.feat_xy[fb ^ 1] + (size_t)seq * d.CAP; int* nage = d.feat_age[fb ^ 1] + (size_t)seq * d.CAP; int* nstr = d.feat_str[fb ^ 1] + (size_t)seq * d.CAP;
    const int chunk = (nb + GEN_EMIT_THREADS - 1) / GEN_EMIT_THREADS;
    const int b0 = threadIdx.x * chunk < nb ? threadIdx.x * chunk : nb, b1 = (b0 + chunk < nb) ? b0 + chunk : nb;
    int cnt = 0;
    for (int b = b0; b < b1; b++) cnt += sn[b];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int i = 0; i < GEN_EMIT_WAVES; i++) { int t = wave_tot[i]; wave_tot[i] = acc; acc += t; } s_total = acc; }
    __syncthreads();
    int pos = wave_tot[wv] + incl - cnt;
    for (int b = b0; b < b1; b++) {
        const size_t so = ((size_t)seq * nb + b) * per;
        for (int k = 0; k < sn[b]; k++, pos++)
            if (pos < d.CAP) { nxy[pos] = d.slot_xy[so + k]; nage[pos] = d.slot_age[so + k]; nstr[pos] = d.slot_str[so + k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = s_total < d.CAP ? s_total : d.CAP;
        s.n_feat = total; s.feat_buf = fb ^ 1;
        s.stats.n_after_detect = total;
        if (pass == 0) s.do_second = total < d.cfg.pre_matching_feature_threshold;   // vo.cpp:327
        else s.stats.second_pass = 1;
    }
}
static void launch_detect_general(const DevBuffers& d, int pass, int th, hipStream_t st) {
    const int W = d.geom.W, H = d.geom.H;
    dim3 g((W + FT_W - 1) / FT_W, (H + FT_H - 1) / FT_H, d.B);
    hipLaunchKernelGGL(k_fast<2>, g, dim3(256), 0, st, (const uint8_t*)nullptr, 0, 0, (uint8_t*)nullptr, d, pass, th);
    hipLaunchKernelGGL(k_gen_row_count, dim3(H, d.B), dim3(64), 0, st, d, pass);
    hipLaunchKernelGGL(k_gen_scan_rows, dim3((d.B + 63) / 64), dim3(64), 0, st, d, pass);
    hipLaunchKernelGGL(k_gen_emit_candidates, dim3(H + 1, d.B), dim3(64), 0, st, d, pass);
    hipLaunchKernelGGL(k_gen_bucket_walk, dim3((d.NB + 255) / 256, d.B), dim3(256), 0, st, d, pass);
    hipLaunchKernelGGL(k_gen_bucket_emit, dim3(d.B), dim3(GEN_EMIT_THREADS), 0, st, d, pass);
}

void launch_detect(const DevBuffers& d, int pass, int th_override, hipStream_t st) {
    int th = pass == 0 ? d.cfg.fast_threshold : d.cfg.fast_threshold / 4;            // vo.cpp:325 / :329-330
    if (th_override >= 0) th = th_override;
    if (d.cfg.features_per_bucket > 1) { launch_detect_general(d, pass, th, st); return; }
    // pass 0: k_fast offers the existing tracks itself; pass 1 finds them offered (and n_old set) by the last block of pass 0's emit
    dim3 g((d.geom.W + FT_W - 1) / FT_W, (d.geom.H + FT_H - 1) / FT_H, d.B);
    static const bool strided_off = getenv("SVO_SECOND_PASS_STRIDED") && atoi(getenv("SVO_SECOND_PASS_STRIDED")) == 0;
    if (pass == 1 && d.B > SVO_LONE_MAX_SEQ && !strided_off) {
        hipLaunchKernelGGL(k_fast_strided, dim3(FAST_STRIDED_BLOCKS, 1, d.B), dim3(256), 0, st, d, pass, th, (int)g.x, (int)g.y);
        hipLaunchKernelGGL(k_bucket_emit_strided, dim3(EMIT_STRIDED_BLOCKS, d.B), dim3(EMIT_THREADS), 0, st, d, pass, d.cfg.buckets_along_height);
        return;
    }
    hipLaunchKernelGGL(k_fast<0>, g, dim3(256), 0, st, (const uint8_t*)nullptr, 0, 0, (uint8_t*)nullptr, d, pass, th);
    hipLaunchKernelGGL(k_bucket_emit, dim3(d.cfg.buckets_along_height, d.B), dim3(EMIT_THREADS), 0, st, d, pass);
}

// ------------------------------------------------------------------------------------------------
// General bucketing (any features_per_bucket): one thread per bucket walks the inputs in order and
// applies Bucket::add_feature verbatim in behaviour (feature_set.cpp:20-53).  O(N * buckets): only
// the reference's unit tests use per-bucket capacities other than 1 (main.cpp:125,152-157).
// ------------------------------------------------------------------------------------------------
__global__ void k_bucket_general(int img_w, int img_h, int n, const float2* xy, const int* ages, const int* strs,
                                 int bah, int baw, int start_row, int per_bucket, int age_thr, int fast_thr,
                                 float2* slot_xy, int* slot_age, int* slot_str, int* slot_n) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bah * baw) return;
    int bucket_h = (img_h + bah - 1) / bah, bucket_w = (img_w + baw - 1) / baw;      // feature_set.cpp:91-93,103-104
    int cap = (b / baw) >= start_row ? per_bucket : 0;                               // :113-116
    float2* sx = slot_xy + (size_t)b * per_bucket; int* sa = slot_age + (size_t)b * per_bucket; int* ss = slot_str + (size_t)b * per_bucket;
    int cnt = 0;
    for (int i = 0; i < n && cap > 0; i++) {
        float2 p = xy[i];
        int bh = (int)(p.y / (float)bucket_h), bw = (int)(p.x / (float)bucket_w);
        if (bh < 0 || bh >= bah || bw < 0 || bw >= baw || bh * baw + bw != b) continue;
        int a = ages[i], st = strs[i];
        if (a >= age_thr) continue;
        if (cnt < cap) { sx[cnt] = p; sa[cnt] = a; ss[cnt] = st; cnt++; }
        else {
            int score = a + (st - fast_thr) / 20;
            int smin = sa[0] + (ss[0] - fast_thr) / 20, imin = 0;
            for (int k = 1; k < cnt; k++) { int c = sa[k] + (ss[k] - fast_thr) / 20; if (c < smin) { smin = c; imin = k; } }
            if (score > smin) { sx[imin] = p; sa[imin] = a; ss[imin] = st; }
        }
    }
    slot_n[b] = cnt;
}
__global__ void k_bucket_general_emit(int nb, int per_bucket, const float2* slot_xy, const int* slot_age, const int* slot_str, const int* slot_n,
                                      float2* out_xy, int* out_age, int* out_str, int* n_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;                                 // test-sized inputs: serial emit keeps bucket-raster order
    int m = 0;
    for (int b = 0; b < nb; b++)
        for (int k = 0; k < slot_n[b]; k++) {
            out_xy[m] = slot_xy[(size_t)b * per_bucket + k]; out_age[m] = slot_age[(size_t)b * per_bucket + k]; out_str[m] = slot_str[(size_t)b * per_bucket + k]; m++;
        }
    *n_out = m;
}
void launch_bucket_general(int img_w, int img_h, int n, const float2* xy, const int* ages, const int* strs,
                           int bah, int baw, int start_row, int per_bucket, int age_thr, int fast_thr,
                           float2* slot_xy, int* slot_age, int* slot_str, int* slot_n,
                           float2* out_xy, int* out_age, int* out_str, int* n_out, hipStream_t st) {
    int nb = bah * baw;
    hipLaunchKernelGGL(k_bucket_general, dim3((nb + 255) / 256), dim3(256), 0, st, img_w, img_h, n, xy, ages, strs, bah, baw, start_row,
                       per_bucket, age_thr, fast_thr, slot_xy, slot_age, slot_str, slot_n);
    hipLaunchKernelGGL(k_bucket_general_emit, dim3(1), dim3(64), 0, st, nb, per_bucket, slot_xy, slot_age, slot_str, slot_n, out_xy, out_age, out_str, n_out);
}

// ------------------------------------------------------------------------------------------------
// Stable compaction after circular matching + in-bounds mask (deletePointsWithFailureStatus /
// deleteFeaturesWithFailureStatus, vo.cpp:144-168, called at :233-238 and :360-364), fused with
// ages[i] += 1 (vo.cpp:70-72) and the "too few tracks" gate (vo.cpp:82-84).
// One 256-thread block per sequence, order-preserving prefix sum.
// ------------------------------------------------------------------------------------------------
#define SCAN_THREADS 1024
#define SCAN_WAVES (SCAN_THREADS / 64)
__global__ __launch_bounds__(SCAN_THREADS) void k_compact(DevBuffers d) {
    // One block per sequence walks the tracks in rounds of SCAN_THREADS: coalesced loads, one track per thread, a ballot gives a
    // survivor its rank inside the wave, LDS the survivors of the lower waves, a running total those of the earlier rounds — the
    // output keeps the input order (order is semantics: RANSAC samples rows).  (Round 1: 256 threads x a serial chunk of 8
    // tracks each, two passes of dependent loads: 19 us at one sequence.)
    const int seq = blockIdx.x;
    SeqState& s = d.st[seq];
    if (!s.active) return;
    __shared__ int sh_cnt[SCAN_WAVES], sh_c[SCAN_WAVES];
    const int n = s.n_lk, fb = s.feat_buf;
    const size_t o = (size_t)seq * d.CAP;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nthr = blockDim.x, nwv = nthr >> 6;                    // 1024 threads for a lone stream (fewest rounds), 256 for many sequences (below)
    const float2* fxy = d.feat_xy[fb] + o; const int* fage = d.feat_age[fb] + o; const int* fstr = d.feat_str[fb] + o;
    float2* nxy = d.feat_xy[fb ^ 1] + o; int* nage = d.feat_age[fb ^ 1] + o; int* nstr = d.feat_str[fb ^ 1] + o;
    int run = 0, cntc = 0;
    unsigned visits = 0, steps = 0;                                  // svo_frame_stats.lk_level_visits / lk_newton_steps
    unsigned dead0 = 0, dead1 = 0, dead2 = 0;                        // svo_frame_stats.lk_dead_after_pass: features that first failed in pass 0, 1, 2
    for (int base = 0; base < n; base += nthr) {
        const int i = base + threadIdx.x;
        uint8_t m = 0;
        if (i < n) {
            m = d.okmask[o + i];
            const unsigned wk = d.lk_work[o + i]; visits += wk & 0x3Fu; steps += wk >> 8;
            const unsigned dk = (wk >> 6) & 3u; dead0 += dk == 1u; dead1 += dk == 2u; dead2 += dk == 3u;
        }
        const bool keep = m == 3;
        cntc += m & 1;
        const unsigned long long bal = __ballot(keep);
        __syncthreads();                                              // the previous round's counts have been read
        if (lane == 0) sh_cnt[wv] = __popcll(bal);
        __syncthreads();
        int pos = run + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < nwv; w++) { const int c = sh_cnt[w]; if (w < wv) pos += c; run += c; }
        if (keep) {
            d.tl0[o + pos] = d.pl0[o + i]; d.tl1[o + pos] = d.pl1[o + i];
            d.tr1[o + pos] = d.pr1[o + i]; d.tr0[o + pos] = d.pr0[o + i];
            nxy[pos] = fxy[i]; nage[pos] = fage[i] + 1; nstr[pos] = fstr[i];
        }
    }
    for (int k = 32; k > 0; k >>= 1) { visits += __shfl_xor(visits, k); steps += __shfl_xor(steps, k); cntc += __shfl_xor(cntc, k);
                                     dead0 += __shfl_xor(dead0, k); dead1 += __shfl_xor(dead1, k); dead2 += __shfl_xor(dead2, k); }
    if (lane == 0 && (visits | steps)) { atomicAdd(&s.stats.lk_level_visits, (int)visits); atomicAdd(&s.stats.lk_newton_steps, (int)steps); }
    if (lane == 0 && (dead0 | dead1 | dead2)) {
        atomicAdd(&s.stats.lk_dead_after_pass[0], (int)dead0); atomicAdd(&s.stats.lk_dead_after_pass[1], (int)dead1); atomicAdd(&s.stats.lk_dead_after_pass[2], (int)dead2);
    }
    if (lane == 0) sh_c[wv] = cntc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total_c = 0;
        for (int w = 0; w < nwv; w++) total_c += sh_c[w];
        if (n > 0) {
            s.n_tracks = run; s.n_circ = total_c; s.n_feat = run; s.feat_buf = fb ^ 1;
        } else {
            s.n_tracks = 0; s.n_circ = 0; s.n_feat = 0;           // empty feature set: circularMatching returned early (vo.cpp:179-181)
        }
        int thr = d.cfg.features_threshold > 4 ? d.cfg.features_threshold : 4;
        if (s.n_tracks <= thr) s.fail_reason = 2;                  // vo.cpp:82-84
    }
    // a lone stream's first RANSAC chunk shares a launch with the triangulation (k_tri_epnp): its subsets — a function of the track
    // count alone, which every thread of this block knows by now — are drawn here, one launch earlier, by the first wave
    // (many-sequence contexts: the spare block of k_triangulate)
    if (d.B <= SVO_LONE_MAX_SEQ && wv == 0 && n > 0) {
        const int thr = d.cfg.features_threshold > 4 ? d.cfg.features_threshold : 4;
        if (run > thr && s.active) pnp_draw_first_chunk_wave(d, s, seq, run, pnp_first_chunk(d));
    }
}
void launch_compact(const DevBuffers& d, hipStream_t st) {
    // A 1024-thread block needs four free wave slots on every SIMD of one CU at the same moment.  Beside other resident kernels — the
    // tail of an LK grid, the image stream's kernels of a many-sequence context — that moment comes late: traces show this kernel
    // waiting 0.8-1.6 ms for 20 us of work.  Many-sequence contexts launch it with 256 threads (four times the rounds, no waiting).
    const int threads = d.B > SVO_LONE_MAX_SEQ ? 256 : SCAN_THREADS;
    hipLaunchKernelGGL(k_compact, dim3(d.B), dim3(threads), 0, st, d);
}

// findClosePoints (vo.cpp:265-280) as a stand-alone stage
__global__ void k_find_close(int n, const float2* a, const float2* b, float thr, uint8_t* ok) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float dx = fabsf(a[i].x - b[i].x), dy = fabsf(a[i].y - b[i].y);
    float off = (dx < dy) ? dy : dx;
    ok[i] = off > thr ? 0 : 1;
}
void launch_find_close(int n, const float2* a, const float2* b, float thr, uint8_t* ok, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_find_close, dim3((n + 255) / 256), dim3(256), 0, st, n, a, b, thr, ok);
}
