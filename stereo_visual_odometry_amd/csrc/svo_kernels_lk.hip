// svo_kernels_lk.hip — pyramidal Lucas–Kanade, one wave64 per feature.
//
// Replaces cv::calcOpticalFlowPyrLK as called four times by VisualOdometry::circularMatching
// (reference src/vo.cpp:203-215) and the status / loop-closure / in-bounds masks (vo.cpp:217-230,
// 341-359).  The four passes L0->L1->R1->R0->L0 of one feature are independent of every other
// feature, so ONE launch runs the whole chain: one single-wave workgroup walks features, and for each
// runs all pyramid levels and all Newton iterations of all four passes.
//
// MI355X mapping (no MFMA: there is no dense contraction here).  Round-1 profiling of a first version
// that staged tiles in LDS and reduced with __shfl_xor showed the kernel LDS-issue-bound
// (SQ_WAIT_INST_LDS = 53 % of wave cycles; 64-bit shuffles lower to ds_bpermute and byte-granular
// ds_read_u8 traffic saturated the LDS pipe).  This version keeps everything in registers:
//   * each lane owns PPL consecutive pixels of one window row; it loads the 4 x (PPL+3) source bytes
//     it needs straight from the (L2 / Infinity-Cache resident) pyramid with unaligned dword loads,
//     computes the Scharr derivatives in registers (no derivative pyramid in HBM, no LDS tile) and
//     keeps I, Ix, Iy of its pixels in VGPRs for the whole Newton loop;
//   * the search-image window (2 rows x (PPL+1) bytes per lane) is also held in VGPRs and only
//     re-loaded when the INTEGER window position changes — in the sub-pixel phase of the Newton
//     iteration (most steps) the loop touches no memory at all;
//   * windows that cross the image border take a per-byte REFLECT_101 path (rare);
//   * the 2x2 normal matrix and the mismatch vector are per-lane int32 partials, split into 16-bit
//     halves and reduced with DPP row operations + v_readlane into SGPRs — exact integers, so the
//     result is order-independent and bit-identical to the sequential CPU loop, and the float tail
//     and every branch are wave-uniform.
#include "svo_internal.hpp"
#include <stdlib.h>
#include <string.h>

#define LK_WBITS 14
// Newton-loop form per lanes-per-feature: 0 = epochs, 1 = flat
#ifndef LK_LOOP_FORM
#define LK_LOOP_FORM(G) ((G) >= 32 ? 0 : 1)
#endif
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// G lanes cooperate on one feature (G = 16: one DPP row per feature, 4 features per wave; G = 64: the whole wave).
// The window is cut into NSEG = W * LPR row segments of PPL pixels; each lane owns SPL consecutive segments.
template <int W, int G> struct LkLayout {
    static constexpr int ppl() {                 // pixels per segment: smallest p with W * ceil(W/p) <= 64
        for (int p = 1; p <= W; p++) if (W * ((W + p - 1) / p) <= 64) return p;
        return W;
    }
    static constexpr int PPL = ppl();
    static constexpr int LPR = (W + PPL - 1) / PPL;   // segments per window row
    static constexpr int EXT = LPR * PPL;             // columns covered by the segments of a row (>= W)
    static constexpr int NSEG = W * LPR;
    static constexpr int SPL = (NSEG + G - 1) / G;    // segments per lane
    static constexpr int NS = PPL + 3;                // template source bytes per segment per row
    static constexpr int NB = PPL + 1;                // search-window bytes per segment per row
};

// Border width the kernel's reads need around every pyramid level at window `win` (Geometry::pad): the template reads rows and
// columns origin - 1 .. origin + EXT + 1 (+ up to 3 bytes of the last dword) with origin in [-win, size), the search window
// origin .. origin + EXT.  Same PPL / EXT recipe as LkLayout (which does not depend on the lanes per feature).
int lk_pad_for(int win) {
    if (win < 1) win = 1;
    int ppl = win;
    for (int p = 1; p <= win; p++) if (win * ((win + p - 1) / p) <= 64) { ppl = p; break; }
    const int ext = ((win + ppl - 1) / ppl) * ppl;
    return (ext + 5 + 3) & ~3;
}

// one doubling step of the in-row DPP reduction (S = 0..3: lane pairs, quads, half rows, rows)
template <int S>
__device__ __forceinline__ int dpp_row_step(int v) {
    if (S == 0) return v + __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    if (S == 1) return v + __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    if (S == 2) return v + __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    return v + __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);               // row_mirror: every lane holds its row's sum
}
// rows -> group: every lane of the group ends up with the group total
template <int G>
__device__ __forceinline__ int dpp_cross_rows(int v) {
    if (G == 64) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
        v = __builtin_amdgcn_readlane(v, 63);
    }
    if (G == 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = r[0] + r[1];
    }
    return v;
}
// Largest per-lane |partial| of PIX window pixels: |diff| <= 255 * 32, |Ix|, |Iy| <= 4080 (Scharr of u8).  PRE = number of
// doubling steps such a partial survives in int32; the split into 16-bit halves can wait that long.
constexpr int lk_presplit_steps(int pix) {
    long long m = (long long)pix * 8160LL * 4080LL;
    int k = 0;
    while (k < 4 && m * 2 <= 2147483647LL) { m *= 2; k++; }
    return k;
}
// N independent exact sums at once ("wide" form, any input): the partials are split into 16-bit halves (each half-sum fits
// 23 bits) after the first PRE steps; the DPP steps of the values are issued round-robin so every DPP instruction has
// independent work between it and its predecessor (DPP needs 2 wait states after a VALU write of its source).
template <int G, int N, int PRE>
__device__ __forceinline__ void group_sums_to_float(const int (&partial)[N], float (&out)[N]) {
    int u[N], v[2 * N];
#pragma unroll
    for (int i = 0; i < N; i++) u[i] = partial[i];
    if (PRE > 0) {
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = dpp_row_step<0>(u[i]);
    }
    if (PRE > 1) {
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = dpp_row_step<1>(u[i]);
    }
    if (PRE > 2) {
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = dpp_row_step<2>(u[i]);
    }
    if (PRE > 3) {
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = dpp_row_step<3>(u[i]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) { v[2 * i] = u[i] & 0xFFFF; v[2 * i + 1] = u[i] >> 16; }
    if (PRE < 1) {
#pragma unroll
        for (int i = 0; i < 2 * N; i++) v[i] = dpp_row_step<0>(v[i]);
    }
    if (PRE < 2) {
#pragma unroll
        for (int i = 0; i < 2 * N; i++) v[i] = dpp_row_step<1>(v[i]);
    }
    if (PRE < 3) {
#pragma unroll
        for (int i = 0; i < 2 * N; i++) v[i] = dpp_row_step<2>(v[i]);
    }
    if (PRE < 4) {
#pragma unroll
        for (int i = 0; i < 2 * N; i++) v[i] = dpp_row_step<3>(v[i]);
    }
#pragma unroll
    for (int i = 0; i < 2 * N; i++) v[i] = dpp_cross_rows<G>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = (float)v[2 * i + 1] * 65536.f + (float)v[2 * i];
}
// "narrow" form: the caller guarantees that the sum of |terms| stays below 2^31, so every intermediate fits int32 and
// v_cvt_f32_i32 rounds the exact total once (nearest-even) — the same float as the wide form.
template <int G, int N>
__device__ __forceinline__ void group_sums_to_float_narrow(const int (&partial)[N], float (&out)[N]) {
    int u[N];
#pragma unroll
    for (int i = 0; i < N; i++) u[i] = dpp_row_step<0>(partial[i]);
#pragma unroll
    for (int i = 0; i < N; i++) u[i] = dpp_row_step<1>(u[i]);
#pragma unroll
    for (int i = 0; i < N; i++) u[i] = dpp_row_step<2>(u[i]);
#pragma unroll
    for (int i = 0; i < N; i++) u[i] = dpp_row_step<3>(u[i]);
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = (float)dpp_cross_rows<G>(u[i]);
}

// ---- wave-wide (G == 64) sums of TWO values at once.  v_permlane32_swap exchanges the upper 32 lanes of one register with
// the lower 32 of another: after it, (r0 + r1) holds a's 32 pairwise partials in lanes 0..31 and b's in lanes 32..63, so every
// following DPP step reduces both values in one instruction (rows 0-1 belong to a, rows 2-3 to b; row_bcast:15 joins each
// pair of rows).  7 VALU instructions + 2 v_readlane instead of 12 + 2.
__device__ __forceinline__ int wave_fold2(int a, int b) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    return (int)(r[0] + r[1]);
}
__device__ __forceinline__ int wave_join_rows2(int u) {                 // after the four row steps: lane 31 = a's total, lane 63 = b's
    return u + __builtin_amdgcn_update_dpp(0, u, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
}
__device__ __forceinline__ void wave_sums2_narrow(int a, int b, float& fa, float& fb) {
    int u = wave_fold2(a, b);
    u = dpp_row_step<0>(u); u = dpp_row_step<1>(u); u = dpp_row_step<2>(u); u = dpp_row_step<3>(u);
    u = wave_join_rows2(u);
    fa = (float)__builtin_amdgcn_readlane(u, 31); fb = (float)__builtin_amdgcn_readlane(u, 63);
}
// wide form (any input): the fold and PRE - 1 row steps happen on the int32 partials (PRE doublings cannot overflow), then the
// register is split into 16-bit halves whose sums fit 23 bits
template <int PRE>
__device__ __forceinline__ void wave_sums2_wide(int a, int b, float& fa, float& fb) {
    static_assert(PRE >= 1, "the fold is one doubling");
    int u = wave_fold2(a, b);
    if (PRE > 1) u = dpp_row_step<0>(u);
    if (PRE > 2) u = dpp_row_step<1>(u);
    if (PRE > 3) u = dpp_row_step<2>(u);
    int lo = u & 0xFFFF, hi = u >> 16;
    if (PRE < 2) { lo = dpp_row_step<0>(lo); hi = dpp_row_step<0>(hi); }
    if (PRE < 3) { lo = dpp_row_step<1>(lo); hi = dpp_row_step<1>(hi); }
    if (PRE < 4) { lo = dpp_row_step<2>(lo); hi = dpp_row_step<2>(hi); }
    lo = dpp_row_step<3>(lo); hi = dpp_row_step<3>(hi);
    lo = wave_join_rows2(lo); hi = wave_join_rows2(hi);
    fa = (float)__builtin_amdgcn_readlane(hi, 31) * 65536.f + (float)__builtin_amdgcn_readlane(lo, 31);
    fb = (float)__builtin_amdgcn_readlane(hi, 63) * 65536.f + (float)__builtin_amdgcn_readlane(lo, 63);
}

// (upper half of a, lower half of b) as one packed pair
__device__ __forceinline__ unsigned hi_lo16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040302u); }
// upper halves of two registers -> one packed pair
__device__ __forceinline__ unsigned pack_hi16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x07060302u); }
// low halves of two registers -> one packed pair (one v_perm_b32)
__device__ __forceinline__ unsigned pack_lo16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }

// The four 14-bit bilinear weights of LKTrackerInvoker (lkpyramid.cpp: iw00 = cvRound((1-a)*(1-b)*(1 << W_BITS)), iw01,
// iw10 likewise, iw11 = (1 << W_BITS) - iw00 - iw01 - iw10) as the two packed operands of the dot products:
// w0 = (iw00, iw01), w1 = (iw10, iw11).  Bit-identical to the scalar recipe, in half the instructions:
//  * the factor 2^14 is applied to (1-a, a) first — scaling by a power of two commutes with the rounding of the product;
//  * round-half-even to integer is done by adding 1.5 * 2^23: the sum's low mantissa bits ARE the integer
//    (0 <= x <= 2^14), so the 16-bit halves are taken straight from the float bits with v_perm;
//  * packed f32 math (v_pk_mul_f32 / v_pk_add_f32) for the pairs.
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lk_weights(float a, float b, unsigned& w0, unsigned& w1) {
    const float MAGIC = 12582912.f;                                   // 1.5 * 2^23, bits 0x4B400000
    const float SC = (float)(1 << LK_WBITS);
    const float2v ab = {a, b};
    const float2v om = 1.f - ab;                                      // (1-a, 1-b)
    // (a (1-b), b (1-a)) = iw01, iw10 before scaling: ONE packed multiply of (a, b) with the swapped (1-b, 1-a).  The products
    // are the same two floats whichever factor carries the 2^14 (a power of two moves through the rounding), and x * 2^14 is
    // exact, so the fused multiply-add below rounds exactly where "x * 2^14 + MAGIC" in two steps does.
    const float2v om_sw = {om.y, om.x};
    const float2v cross = ab * om_sw;
    const float p00 = om.x * om.y;
    const float2v mc = __builtin_elementwise_fma(cross, (float2v){SC, SC}, (float2v){MAGIC, MAGIC});   // iw01, iw10 as magic floats
    const float m0 = __builtin_fmaf(p00, SC, MAGIC);                                                   // iw00
    const unsigned m00 = __float_as_uint(m0), m01 = __float_as_uint(mc.x), m10 = __float_as_uint(mc.y);
    const unsigned iw11 = ((1u << LK_WBITS) + 3u * 0x4B400000u) - (m00 + m01 + m10);
    w0 = pack_lo16((int)m00, (int)m01);
    w1 = pack_lo16((int)m10, (int)iw11);
}

// eps_hi / eps_lo bracket eps2 for the f32 screening of the convergence test (see newton_step)
struct LkCrit { int max_count; double eps2; float mineig_cut; float eps_hi, eps_lo; };
// With G == 64 every per-feature quantity is identical in all lanes of the wave, so every branch on one is wave-uniform.
// The compiler cannot prove that (the values live in VGPRs) and would guard each branch with exec-mask bookkeeping and keep
// loop counters in VGPRs; a ballot of the condition IS uniform by construction and costs nothing extra (v_cmp writes an
// SGPR pair either way): branches become s_cbranch, counters become SALU.  Groups smaller than a wave keep plain SIMT.
template <int G> __device__ __forceinline__ bool uni(bool c) {
    if constexpr (G == 64) return __builtin_amdgcn_ballot_w64(c) != 0ull; else return c;
}
template <int G> __device__ __forceinline__ int uni_i(int v) {
    if constexpr (G == 64) return __builtin_amdgcn_readfirstlane(v); else return v;
}
template <int SPL> struct LkSegs { int row[SPL]; int xs[SPL]; bool on[SPL]; };   // the window segments a lane owns

typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
// a0*b0 + a1*b1 + acc on packed signed 16-bit pairs (v_dot2c_i32_i16)
__device__ __forceinline__ int dot2(unsigned a, unsigned b, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), acc, false);
}
// Same product with the clamp bit set: the operands used here cannot overflow int32, so the result is identical; the bit
// only makes the compiler pick the three-address encoding (v_dot2_i32_i16 d, a, b, c) instead of copying c for the
// accumulate-in-place v_dot2c form — used where the addend must stay live.
__device__ __forceinline__ int dot2_keep(unsigned a, unsigned b, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), acc, true);
}
__device__ __forceinline__ unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }

// N bytes at p (any alignment) -> N-1 packed pairs: pair[x] = byte[x] | byte[x+1] << 16 (one v_perm_b32 each)
template <int N>
__device__ __forceinline__ void load_pairs(const uint8_t* __restrict__ p, unsigned (&pair)[N - 1]) {
    constexpr int ND = (N + 3) / 4;
    struct __attribute__((packed, aligned(1))) UD { unsigned v[ND]; };
    const UD u = *reinterpret_cast<const UD*>(p);
#pragma unroll
    for (int x = 0; x < N - 1; x++) {
        const int k = x >> 2, o = x & 3;
        const unsigned lo = u.v[k], hi = (k + 1 < ND) ? u.v[k + 1] : 0u;
        pair[x] = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (unsigned)o | ((unsigned)(o + 1) << 16));
    }
}


// ---- "float sums" mode (svo_config.lk_float_sums; deviation D1 of the oracle reverted) -----------------------------------------
// OpenCV accumulates the LK normal equations in FLOAT, in the lane order of its SIMD128 code (lkpyramid.cpp, `#if CV_SIMD128 &&
// !CV_NEON`, acctype = float).  Float addition is not associative, so reproducing its bits means reproducing its ORDER: a window
// row is E = W * CN interleaved elements (element e = pixel e / CN, channel e % CN); the vector loop takes 8 elements per step
// (NSIMD = 8 * (E / 8) of them), the scalar tail the rest.
//   A:  element e < NSIMD goes to float lane e & 3 of qA11 / qA12 / qA22 (q = fx * fy + q, unfused), rows in order; tail elements
//       go, in order, to the scalar iA; at the end iA += (q0 + q2) + (q1 + q3).
//   b:  v_dotprod pairs element 8g + k with 8g + k + 4 (k = 0..3) as an exact int32, v_cvt_f32, and adds it to one of eight float
//       lanes (k, x|y); the tail adds (float)(diff * I) to ib; at the end ib1 += (k0x + k2x) + (k1x + k3x).
// Measured against the reference's own recording (tools/deviation_ablation.py): this order reproduces run1/result.csv digit for
// digit, the exact-integer sums of the default mode flip borderline tracks at frames 14, 15, 22, 23.
// Mapping (three phases; the block is one wave): the owner lanes write their per-element integers to LDS ([row][e]); a CONVERT
// phase, one lane per (sum, window row), turns them into the floats OpenCV adds — fx * fy for A, (float)(p[k] + p[k + 4]) or
// (float)p for b, every one an independent rounding — and stores them in CHAIN ORDER, one contiguous array per chain; five chain
// lanes per sum (the four SIMD float lanes + the scalar tail) then walk their arrays with 16-byte LDS reads and ONE dependent
// float add per element.  The chain is serial by construction (105 adds for the tail at W = 21), which is why this mode costs a
// multiple of the default's Newton step and is opt-in; everything that is not the chain runs on parallel lanes.  (First version:
// the chain lanes converted, multiplied and selected per element themselves — about six instructions and one or two LDS reads per
// element on a wave with 10-15 live lanes; DESIGN.md has the A/B.)
template <int W, int CN> struct LkFs {
    static constexpr int up4(int v) { return (v + 3) & ~3; }
    static constexpr int E = W * CN, NSIMD = (E / 8) * 8;
    static constexpr int CNT_S = NSIMD / 4, CNT_T = E - NSIMD;          // A: elements per row of a SIMD chain / of the tail chain
    static constexpr int NG = NSIMD / 8;                                // b: items per row of a SIMD chain (tail: CNT_T)
    // chain arrays, per sum: four SIMD chains of SZ_S floats, then the tail chain of SZ_T (multiples of four: 16-byte reads).  Every
    // chain lane walks up4(max length) elements from its base: a shorter chain's walker runs on into the arrays behind it (stale
    // data, never used), so only the very last array needs SLACK behind it.  A's arrays start behind A's integers ([row][e]), b's
    // behind b's ([x | y][row][e]); the two phases never live together and share the block (7.1 KB at W = 21: five blocks per SIMD).
    static constexpr int LS_A = W * CNT_S, LS_B = W * NG, LT = W * CNT_T;
    static constexpr int SZS_A = up4(LS_A), SZS_B = up4(LS_B), SZ_T = up4(LT);
    static constexpr int SUM_A = 4 * SZS_A + SZ_T, SUM_B = 4 * SZS_B + SZ_T;
    static constexpr int INTS_A = up4(W * E), INTS_B = up4(2 * W * E);
    static constexpr int TOT_A = INTS_A + 3 * SUM_A + (SZS_A > SZ_T ? SZS_A - SZ_T : 0), TOT_B = INTS_B + 2 * SUM_B + (SZS_B > SZ_T ? SZS_B - SZ_T : 0);
    static constexpr int LDS_INTS = TOT_A > TOT_B ? TOT_A : TOT_B;
};
__device__ __forceinline__ float lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// One chain lane: acc = 0 + arr[0] + arr[1] + ..., one rounding per addition, in that order.  The SIMD chains hold LS elements, the
// tail chain LT; every lane walks max(LS, LT) elements (what lies behind a shorter chain's end is stale but inside the chain's
// array, whose stride is a multiple of four) and keeps a snapshot at min(LS, LT): no per-element select.  arr is 16-byte aligned
// and read with ds_read_b128, software-pipelined in blocks of FS_BLOCK reads: the next block is requested before the current one
// is added.  A rolled loop had every eight additions wait for an LDS round trip (LK 4.82 ms per 32-sequence launch against 4.17
// pipelined, same box); completely unrolled without the fences the compiler hoists every read (155 registers).  Blocks of two:
// blocks of four are 1.4 % faster at equal occupancy but cost 16 registers, and at W = 21 the fifth wave per SIMD (below) is worth more.
#ifndef FS_BLOCK
#define FS_BLOCK 2
#endif
template <int LS, int LT>
__device__ __forceinline__ float fs_chain(const float* __restrict__ arr, bool is_tail) {
    constexpr int MINL = LS < LT ? LS : LT, MAXL = LS < LT ? LT : LS;
    constexpr int NQ = (MAXL + 3) / 4, BQ = FS_BLOCK, NB = (NQ + BQ - 1) / BQ;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(arr);
    float acc = 0.f, snap = 0.f;
    float4 buf[2][BQ];
#pragma unroll
    for (int k = 0; k < BQ; k++) if (k < NQ) buf[0][k] = p[k];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (b + 1 < NB) {
#pragma unroll
            for (int k = 0; k < BQ; k++) if ((b + 1) * BQ + k < NQ) buf[(b + 1) & 1][k] = p[(b + 1) * BQ + k];
        }
        // the next block's reads are issued BEFORE this block's additions: the fence keeps the reads in front of it, the additions
        // depend on its output, and the scheduling barrier keeps the machine scheduler from undoing either
        asm volatile("" : "+v"(acc) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < BQ; k++) {
            const float4 v = buf[b & 1][k];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = (b * BQ + k) * 4 + r;
                if (i < MAXL) acc = acc + e[r];
                if (i + 1 == MINL) snap = acc;
            }
        }
    }
    return (is_tail == (LT >= LS)) ? acc : snap;
}

// lds[row * E + e] = (ix & 0xFFFF) | (iy << 16).  Chain lanes: lane = which * 8 + c, which = 0 (A11), 1 (A12), 2 (A22), c = 0..3
// the SIMD float lanes, c = 4 the scalar tail.  Returns the three sums (identical in every lane).
template <int W, int CN>
__device__ __forceinline__ void fs_sum_A(int* __restrict__ lds, float (&As)[3]) {
    using F = LkFs<W, CN>;
    float* __restrict__ fl = reinterpret_cast<float*>(lds + F::INTS_A);
    const int lane = threadIdx.x & 63;
    // convert: lane -> (sum, row); element e < NSIMD is item e / 4 of its row in SIMD chain e & 3, the others go to the tail chain
    for (int it = lane; it < 3 * W; it += 64) {
        const int which = it / W, y = it - which * W;
        const int* __restrict__ rp = lds + y * F::E;
        float* __restrict__ dS = fl + which * F::SUM_A + y * F::CNT_S;
        float* __restrict__ dT = fl + which * F::SUM_A + 4 * F::SZS_A + y * F::CNT_T;
#pragma unroll
        for (int e = 0; e < F::E; e++) {
            const int v = rp[e];
            const float fx = (float)(short)(v & 0xFFFF), fy = (float)(v >> 16);
            const float a = which == 2 ? fy : fx, b = which == 0 ? fx : fy;
            const float p = a * b;                                       // rounded here, added in the chain: v_muladd unfused (this file is built with -ffp-contract=off)
            if (e < F::NSIMD) dS[(e & 3) * F::SZS_A + (e >> 2)] = p; else dT[e - F::NSIMD] = p;
        }
    }
    __syncthreads();
    const int c = lane & 7, which = lane >> 3;
    const float acc = fs_chain<F::LS_A, F::LT>(fl + (which < 3 ? which : 2) * F::SUM_A + (c < 4 ? c : 4) * F::SZS_A, c >= 4);
#pragma unroll
    for (int sidx = 0; sidx < 3; sidx++) {
        const float q0 = lane_f(acc, sidx * 8), q1 = lane_f(acc, sidx * 8 + 1), q2 = lane_f(acc, sidx * 8 + 2), q3 = lane_f(acc, sidx * 8 + 3);
        const float t = lane_f(acc, sidx * 8 + 4);
        As[sidx] = t + ((q0 + q2) + (q1 + q3));
    }
}
// lds[xy * W * E + row * E + e] = diff * Ix (xy = 0) or diff * Iy (xy = 1), exact int32.  Chain lanes: lane = xy * 8 + c.
template <int W, int CN>
__device__ __forceinline__ void fs_sum_b(int* __restrict__ lds, float& b1, float& b2) {
    using F = LkFs<W, CN>;
    float* __restrict__ fl = reinterpret_cast<float*>(lds + F::INTS_B);
    const int lane = threadIdx.x & 63;
    // convert: lane -> (x | y, row).  v_dotprod pairs element 8g + k with 8g + k + 4 as an exact int32: item g of SIMD chain k
    if (lane < 2 * W) {
        const int xy = lane >= W ? 1 : 0, y = lane - xy * W;
        const int* __restrict__ rp = lds + xy * (W * F::E) + y * F::E;
        float* __restrict__ dS = fl + xy * F::SUM_B + y * F::NG;
        float* __restrict__ dT = fl + xy * F::SUM_B + 4 * F::SZS_B + y * F::CNT_T;
#pragma unroll
        for (int g = 0; g < F::NG; g++) {
#pragma unroll
            for (int k = 0; k < 4; k++) dS[k * F::SZS_B + g] = (float)(rp[8 * g + k] + rp[8 * g + k + 4]);
        }
#pragma unroll
        for (int t = 0; t < F::CNT_T; t++) dT[t] = (float)rp[F::NSIMD + t];
    }
    __syncthreads();
    const int c = lane & 7, xy = lane >> 3;
    const float acc = fs_chain<F::LS_B, F::LT>(fl + (xy < 2 ? xy : 1) * F::SUM_B + (c < 4 ? c : 4) * F::SZS_B, c >= 4);
    const float x0 = lane_f(acc, 0), x1 = lane_f(acc, 1), x2 = lane_f(acc, 2), x3 = lane_f(acc, 3), xt = lane_f(acc, 4);
    const float y0 = lane_f(acc, 8), y1 = lane_f(acc, 9), y2 = lane_f(acc, 10), y3 = lane_f(acc, 11), yt = lane_f(acc, 12);
    b1 = xt + ((x0 + x2) + (x1 + x3));
    b2 = yt + ((y0 + y2) + (y1 + y3));
}

// One cv::calcOpticalFlowPyrLK track of a single point across all pyramid levels (LKTrackerInvoker semantics,
// SURVEY.md Appendix A.3).  (px,py) -> (outx,outy), status.  Written as plain SIMT code: every "per feature" quantity
// lives in a VGPR and is identical across the G lanes of the feature's group; control flow diverges between groups and
// is handled by the exec mask.  segrow / segxs / segon describe the SPL window segments this lane owns.
// CN = image channels: the window sums of LKTrackerInvoker run over every channel of every pixel (x < winSize.width*cn).
// Each colour plane is its own single-channel pyramid (plane k at pyr + k * pstride); the (plane, segment) pairs a lane owns
// are flattened into one index kk = plane * SPL + segment, so CN = 3 simply triples the per-lane pixel arrays.
// FS = float-sums mode (above): fs_lds is the block's LkFs<W, CN>::LDS_INTS ints of LDS (unused otherwise).
template <int W, int G, int CN, bool FS = false>
__device__ void lk_pass(const Geometry& g, const uint8_t* __restrict__ pyrA, const uint8_t* __restrict__ pyrB, size_t pstride,
                        float px, float py, float& outx, float& outy, int& status, const LkCrit& crit,
                        const LkSegs<LkLayout<W, G>::SPL>& sg, int& n_visits, int& n_steps, int* fs_lds = nullptr) {
    static_assert(!FS || G == 64, "the float-sums mode runs one feature per wave");
    constexpr int FE = W * CN;                                          // interleaved elements per window row (float-sums mode)
    using LL = LkLayout<W, G>;
    constexpr int PPL = LL::PPL, EXT = LL::EXT, NS = LL::NS, NB = LL::NB, SPL = LL::SPL;
    constexpr int KS = SPL * CN;                                        // (plane, segment) pairs per lane
    constexpr int PRE = lk_presplit_steps(PPL * KS);
    constexpr float NARROW_LIMIT = (float)(0.95 * 4611686018427387904.0 / (8160.0 * 8160.0 * W * W * CN));
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
        const float scale = __int_as_float((127 - level) << 23);            // 2^-level, exact (= 1.f / (1 << level))
        float ppx = px * scale, ppy = py * scale;
        if (level == top) { nx = ppx; ny = ppy; } else { nx = outx * 2.f; ny = outy * 2.f; }
        outx = nx; outy = ny;
        ppx -= half; ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        // -W <= ip < size  <=>  (unsigned)(ip + W) < (unsigned)(size + W): one compare per axis
        if (uni<G>(ipx < -W || ipx >= L.w || ipy < -W || ipy >= L.h)) {
            if (level == 0) status = 0;
            continue;
        }
        unsigned w0, w1;
        lk_weights(ppx - (float)ipx, ppy - (float)ipy, w0, w1);
        // ---- template: per segment 4 rows x NS source bytes, origin (ipx-1+xs, ipy-1+row); REFLECT_101 = the pyramid border.
        // Everything is kept as packed 16-bit PAIRS (value[c], value[c+1]) — the operand layout of v_dot2c_i32_i16 —
        // and the Scharr derivatives are computed with packed 16-bit math directly on those pairs:
        //   t0(c) = 3*(s[y-1][c] + s[y+1][c]) + 10*s[y][c]      t1(c) = s[y+1][c] - s[y-1][c]
        //   dx(c) = t0(c+1) - t0(c-1)                            dy(c) = 3*(t1(c-1) + t1(c+1)) + 10*t1(c)
        // carried at 4 x their value (coefficients 12 / 40; everything still fits 16 bits: 4 |t0|, 4 |dx|, 4 |dy| <= 16320).
        // Kept for the Newton loop, per owned pixel: Kr = 2^(WBITS-6) - (I << (WBITS-5)), the seed of the bilinear dot product
        // of the search image, so that (seed + J-dot) >> (WBITS-5) IS the mismatch J - I (the subtrahend is a multiple of the
        // shift unit, so folding it in is exact); and the derivatives as packed signed 16-bit pairs of adjacent pixels, the
        // operand layout of v_dot2_i32_i16: one instruction multiplies two mismatches with two derivatives and accumulates.
        constexpr int NPR = (PPL + 1) / 2;
        int Kr[KS][PPL];
        unsigned Ixp[KS][NPR], Iyp[KS][NPR];
        int pA11 = 0, pA12 = 0, pA22 = 0;
        // 1 <= ipx && ipx + EXT + 1 < w  <=>  (unsigned)(ipx - 1) < (unsigned)(w - EXT - 2)   (a non-positive bound never holds: the image is larger than the window)
        const bool interior = uni<G>(ipx >= 1 && ipx + EXT + 1 < L.w && ipy >= 1 && ipy + W + 1 < L.h);
        // interior windows: one (wave-uniform for G == 64) base address per level visit, 32-bit lane offsets
        const uint8_t* __restrict__ Abase = A + (ptrdiff_t)(uni_i<G>(ipy) - 1) * L.stride + (uni_i<G>(ipx) - 1);
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            const int k = kk % SPL;
            const int row = sg.row[k], xs = sg.xs[k];
            unsigned Ip[2][PPL], DXp[2][PPL], DYp[2][PPL];             // [row 0/1 of the bilinear][pixel]: packed pairs
            // source pairs of the four rows: unaligned dword loads.  A window over the image border reads the level's REFLECT_101
            // border, which is stored with it (Geometry::pad, k_pad_pyramid) — the same bytes the per-byte path used to gather
            unsigned Q[4][NS - 1];
            {
                const uint8_t* p = Abase + (size_t)(kk / SPL) * pstride + (unsigned)(row * L.stride + xs);
#pragma unroll
                for (int r = 0; r < 4; r++) load_pairs<NS>(p + (unsigned)(r * L.stride), Q[r]);
            }
#pragma unroll
            for (int yy = 0; yy < 2; yy++) {
                // the vertical passes are evaluated on the EVEN pairs (columns 2i, 2i+1); an odd pair is the upper half of
                // its left neighbour next to the lower half of its right one — one v_perm instead of recomputing both columns
                ushort2v T0[NS - 1]; short2v T1[NS - 1];
#pragma unroll
                for (int c = 0; c < NS - 1; c++) {
                    if ((c & 1) && c + 1 < NS - 1) continue;
                    const ushort2v q0 = __builtin_bit_cast(ushort2v, Q[yy][c]), q1 = __builtin_bit_cast(ushort2v, Q[yy + 1][c]), q2 = __builtin_bit_cast(ushort2v, Q[yy + 2][c]);
                    T0[c] = (q0 + q2) * (unsigned short)12 + q1 * (unsigned short)40;           // 4 x Scharr smoothing (<= 16320)
                    T1[c] = __builtin_bit_cast(short2v, (ushort2v)(q2 - q0));
                }
#pragma unroll
                for (int c = 1; c + 1 < NS - 1; c += 2) {
                    T0[c] = __builtin_bit_cast(ushort2v, hi_lo16(__builtin_bit_cast(unsigned, T0[c - 1]), __builtin_bit_cast(unsigned, T0[c + 1])));
                    T1[c] = __builtin_bit_cast(short2v, hi_lo16(__builtin_bit_cast(unsigned, T1[c - 1]), __builtin_bit_cast(unsigned, T1[c + 1])));
                }
                // horizontal passes: evaluated for the even pixels; an odd pixel's pair is again (upper half of its left
                // neighbour's pair, lower half of its right neighbour's) — unless it is the last pixel, which is computed directly
#pragma unroll
                for (int x = 0; x < PPL; x++) {
                    if ((x & 1) && x + 1 < PPL) continue;
                    DXp[yy][x] = __builtin_bit_cast(unsigned, (ushort2v)(T0[x + 2] - T0[x]));
                    DYp[yy][x] = __builtin_bit_cast(unsigned, (short2v)((T1[x] + T1[x + 2]) * (short)12 + T1[x + 1] * (short)40));
                }
#pragma unroll
                for (int x = 1; x + 1 < PPL; x += 2) {
                    DXp[yy][x] = hi_lo16(DXp[yy][x - 1], DXp[yy][x + 1]);
                    DYp[yy][x] = hi_lo16(DYp[yy][x - 1], DYp[yy][x + 1]);
                }
#pragma unroll
                for (int x = 0; x < PPL; x++) Ip[yy][x] = Q[yy + 1][x + 1];
                if (!interior) {
                    // the derivative image has a CONSTANT 0 border (buildOpticalFlowPyramid: derivBorder): samples outside the
                    // level are zero, not derivatives of the reflected image
                    const int gy = ipy + row + yy;
                    const bool rowin = gy >= 0 && gy < L.h;
#pragma unroll
                    for (int x = 0; x < PPL; x++) {
                        const int gx = ipx + xs + x;
                        const unsigned m = ((rowin && gx >= 0 && gx < L.w) ? 0x0000FFFFu : 0u) | ((rowin && gx + 1 >= 0 && gx + 1 < L.w) ? 0xFFFF0000u : 0u);
                        DXp[yy][x] &= m; DYp[yy][x] &= m;
                    }
                }
            }
            // patch samples (kept in registers for the Newton loop) + covariance partials.  Segments / pixels outside
            // the window get zero derivative weights: their Ix = Iy = 0, so they contribute exact zeros everywhere.
            const unsigned wd0 = sg.on[k] ? w0 : 0u, wd1 = sg.on[k] ? w1 : 0u;
            int ixv[PPL], iyv[PPL];
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                const bool on = (EXT == W) || (xs + j < W);
                const int iacc = dot2(Ip[1][j], w1, dot2_keep(Ip[0][j], w0, 1 << (LK_WBITS - 6)));          // I = iacc >> (WBITS-5); the rounding term rides in as the (scalar) addend of the three-address form: no v_mov per pixel
                Kr[kk][j] = (1 << (LK_WBITS - 6)) - (iacc & ~((1 << (LK_WBITS - 5)) - 1));
                // the derivative pairs carry 4 x the Scharr value (coefficients 12 / 40 above, still inside 16 bits), so the
                // 14-bit descale of the bilinear sum becomes "take the upper half": (4 S + 2^15) >> 16 == (S + 2^13) >> 14
                const int ixacc = dot2(DXp[1][j], wd1, dot2_keep(DXp[0][j], wd0, 1 << (LK_WBITS + 1)));
                const int iyacc = dot2(DYp[1][j], wd1, dot2_keep(DYp[0][j], wd0, 1 << (LK_WBITS + 1)));
                ixv[j] = on ? ixacc : 0; iyv[j] = on ? iyacc : 0;
                if constexpr (FS) {
                    if (sg.on[k] && on) fs_lds[row * FE + (xs + j) * CN + kk / SPL] = (int)pack_hi16(ixacc, iyacc);
                }
            }
#pragma unroll
            for (int q = 0; q < NPR; q++) {
                Ixp[kk][q] = (2 * q + 1 < PPL) ? pack_hi16(ixv[2 * q], ixv[2 * q + 1]) : ((unsigned)ixv[2 * q] >> 16);
                Iyp[kk][q] = (2 * q + 1 < PPL) ? pack_hi16(iyv[2 * q], iyv[2 * q + 1]) : ((unsigned)iyv[2 * q] >> 16);
                if (kk == 0 && q == 0) { pA11 = dot2_keep(Ixp[0][0], Ixp[0][0], 0); pA12 = dot2_keep(Ixp[0][0], Iyp[0][0], 0); pA22 = dot2_keep(Iyp[0][0], Iyp[0][0], 0); }
                else { pA11 = dot2(Ixp[kk][q], Ixp[kk][q], pA11); pA12 = dot2(Ixp[kk][q], Iyp[kk][q], pA12); pA22 = dot2(Iyp[kk][q], Iyp[kk][q], pA22); }
            }
        }
        float As[3];
        if constexpr (FS) {
            __syncthreads();
            fs_sum_A<W, CN>(fs_lds, As);
            __syncthreads();
        } else {
            const int pa[3] = {pA11, pA12, pA22};
            // pA11, pA22 >= 0 and |pA12| <= (pA11 + pA22) / 2 per lane (|ab| <= (a^2 + b^2) / 2 term by term), so one unsigned
            // compare bounds all three partials: below 2^25 per lane the group sums stay inside int32 -> narrow reduction
            const bool big = ((unsigned)pA11 | (unsigned)pA22) >= (1u << 25);
            const unsigned long long bigs = __builtin_amdgcn_ballot_w64(big);
            const bool nar = G == 64 ? bigs == 0ull : ((bigs >> ((threadIdx.x / G) * G)) & ((G == 64 ? 0ull : (1ull << (G & 63))) - 1ull)) == 0ull;
            if constexpr (G == 64 && PRE >= 1) {
                const int p1[1] = {pA12}; float a1[1];
                if (nar) { wave_sums2_narrow(pA11, pA22, As[0], As[2]); group_sums_to_float_narrow<G, 1>(p1, a1); }
                else { wave_sums2_wide<PRE>(pA11, pA22, As[0], As[2]); group_sums_to_float<G, 1, PRE>(p1, a1); }
                As[1] = a1[0];
            } else {
                if (nar) group_sums_to_float_narrow<G, 3>(pa, As);
                else group_sums_to_float<G, 3, PRE>(pa, As);
            }
        }
        // |sum diff*Ix| <= 8160 * sqrt(CN W^2 * sum Ix^2) (Cauchy-Schwarz): below 2^31 when sum Ix^2 < 2^62 / (8160^2 CN W^2); 5 % margin
        // covers the float rounding of As.  Then the mismatch sums never leave int32 and take the narrow reduction.
        const bool narrow = As[0] < NARROW_LIMIT && As[2] < NARROW_LIMIT;
        const float A11 = As[0] * FLT_SCALE, A12 = As[1] * FLT_SCALE, A22 = As[2] * FLT_SCALE;
        float Dt = A11 * A22 - A12 * A12;
        // lkpyramid.cpp: minEig = (A22 + A11 - sqrt(...)) / (2 * winSize.area());  if (minEig < minEigThreshold || D < FLT_EPSILON) skip.
        // The f32 division and the f64 comparison are folded into one f32 comparison of the numerator against a cut-off the
        // host found by bisection over the floats (lk_mineig_cut): division by a positive constant is monotone, so
        // "(double)fl(num / den) < threshold"  <=>  "num < cut" exactly.
        const float eig_num = A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12);
        if (uni<G>(eig_num < crit.mineig_cut || Dt < 1.1920928955078125e-07f)) {
            if (level == 0) status = 0;
            continue;
        }
        // the 2^-20 scale of the mismatch sums is folded into 1/D once per level: scaling by a power of two commutes with
        // every rounding below (no overflow: |A12 b| < 2^52; no underflow: 1/D >= 2^-41), so the steps skip that multiply
        const float Dts = (1.f / Dt) * FLT_SCALE;
        n_visits++;
        nx -= half; ny -= half;
        // Newton iterations.  The search window (two rows of packed byte pairs per segment) lives in registers and is
        // re-loaded only when the INTEGER window origin floor(n) changes; most iterations move the window by a fraction of
        // a pixel and touch no memory.
        int j = 0;
        const int max_count = uni_i<G>(crit.max_count);
        float pdx = 0.f, pdy = 0.f, ldx = 0.f, ldy = 0.f;
        bool moved = false, osc = false, conv = false;
        unsigned P0[KS][PPL], P1[KS][PPL];
        auto load_window = [&](int inx, int iny) __attribute__((always_inline)) {
            // any origin in reach ([-W, size) per axis) lies inside the stored border
            const uint8_t* base = Bm + (ptrdiff_t)uni_i<G>(iny) * L.stride + uni_i<G>(inx);
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                const int k = kk % SPL;
                const uint8_t* p = base + (size_t)(kk / SPL) * pstride + (unsigned)(sg.row[k] * L.stride + sg.xs[k]);
                load_pairs<NB>(p, P0[kk]);
                load_pairs<NB>(p + L.stride, P1[kk]);
            }
        };
        // one Newton step with the window loaded at the integer origin the fractions (fa, fb) = n - origin refer to;
        // returns true when the track is finished at this level
        auto newton_step = [&](float fa, float fb) __attribute__((always_inline)) -> bool {
            lk_weights(fa, fb, w0, w1);
            if constexpr (G != 64) n_steps++;                               // one feature per wave: counted from j after the loop (scalar)
            int pb1 = 0, pb2 = 0;
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                int dv[PPL];
                // J - I, |.| <= 8160; three sweeps so that dependent dot instructions are PPL instructions apart
#pragma unroll
                for (int jj = 0; jj < PPL; jj++) dv[jj] = dot2_keep(P0[kk][jj], w0, Kr[kk][jj]);
#pragma unroll
                for (int jj = 0; jj < PPL; jj++) dv[jj] = dot2(P1[kk][jj], w1, dv[jj]);
#pragma unroll
                for (int jj = 0; jj < PPL; jj++) dv[jj] >>= (LK_WBITS - 5);
#pragma unroll
                for (int q = 0; q < NPR; q++) {
                    // masked pixels (and the unpaired upper half) have Ix = Iy = 0: whatever mismatch they see contributes an exact zero
                    const unsigned dp = (2 * q + 1 < PPL) ? pack_lo16(dv[2 * q], dv[2 * q + 1]) : (unsigned)dv[2 * q];
                    if constexpr (FS) {
                        // per-element products diff * Ix, diff * Iy (exact int32) to LDS; the chain lanes sum them in OpenCV's order
                        const int k = kk % SPL;
                        const int xs = sg.xs[k];
                        int* dst = fs_lds + sg.row[k] * FE + (xs + 2 * q) * CN + kk / SPL;
                        const unsigned dlo = dp & 0xFFFFu, dhi = dp & 0xFFFF0000u;
                        if (sg.on[k] && ((EXT == W) || (xs + 2 * q < W))) {
                            dst[0] = dot2_keep(dlo, Ixp[kk][q], 0); dst[W * FE] = dot2_keep(dlo, Iyp[kk][q], 0);
                        }
                        if (2 * q + 1 < PPL && sg.on[k] && ((EXT == W) || (xs + 2 * q + 1 < W))) {
                            dst[CN] = dot2_keep(dhi, Ixp[kk][q], 0); dst[W * FE + CN] = dot2_keep(dhi, Iyp[kk][q], 0);
                        }
                    } else {
                    if (kk == 0 && q == 0) { pb1 = dot2_keep(dp, Ixp[0][0], 0); pb2 = dot2_keep(dp, Iyp[0][0], 0); }   // no accumulator to clear first
                    else { pb1 = dot2(dp, Ixp[kk][q], pb1); pb2 = dot2(dp, Iyp[kk][q], pb2); }
                    }
                }
            }
            float bs[2];
            if constexpr (FS) {
                __syncthreads();
                fs_sum_b<W, CN>(fs_lds, bs[0], bs[1]);
                __syncthreads();
            } else {
                const int pb[2] = {pb1, pb2};
                // narrow (int32 end to end) when the level's Cauchy-Schwarz bound allows it, or when this iteration's own
                // partials are small: every lane |partial| < 2^25  =>  any sum over the <= 64 lanes of a group stays below 2^31
                bool nar = narrow;
                if (!nar) {
                    const bool big = ((unsigned)(pb1 + (1 << 25)) | (unsigned)(pb2 + (1 << 25))) >= (1u << 26);
                    const unsigned long long bigs = __builtin_amdgcn_ballot_w64(big);
                    nar = G == 64 ? bigs == 0ull : ((bigs >> ((threadIdx.x / G) * G)) & ((G == 64 ? 0ull : (1ull << (G & 63))) - 1ull)) == 0ull;
                }
                if constexpr (G == 64 && PRE >= 1) {
                    if (nar) wave_sums2_narrow(pb1, pb2, bs[0], bs[1]);
                    else wave_sums2_wide<PRE>(pb1, pb2, bs[0], bs[1]);
                } else {
                    if (nar) group_sums_to_float_narrow<G, 2>(pb, bs);
                    else group_sums_to_float<G, 2, PRE>(pb, bs);
                }
            }
            const float dx = (A12 * bs[1] - A22 * bs[0]) * Dts, dy = (A12 * bs[0] - A11 * bs[1]) * Dts;
            nx += dx; ny += dy;
            ldx = dx; ldy = dy; moved = true;
            // termination on |delta|^2 <= eps^2, an f64 comparison in lkpyramid.cpp.  The f32 sum of squares is within 2e-7
            // (relative) of the exact one, so away from the threshold it decides; the f64 form only runs in the gap.
            const float s32 = dx * dx + dy * dy;
            if (!uni<G>(s32 > crit.eps_hi)) {
                if (uni<G>(s32 < crit.eps_lo)) { conv = true; return true; }
                if (uni<G>((double)dx * (double)dx + (double)dy * (double)dy <= crit.eps2)) { conv = true; return true; }
            }
            // "(double)|v| < 0.01" for a float v is exactly "|v| < nextafterf((float)0.01)": 0.01 lies between the floats
            // 0x3C23D70A and 0x3C23D70B, so v < 0.01 (as doubles) <=> v <= 0x3C23D70A <=> v < 0x3C23D70B
            if (j > 0 && uni<G>(fabsf(dx + pdx) < 0.010000000707805157f) && uni<G>(fabsf(dy + pdy) < 0.010000000707805157f)) {
                osc = true;                                                  // nextPts -= delta * 0.5 (applied after the loop)
                return true;
            }
            pdx = dx; pdy = dy;
            return ++j >= max_count;
        };
        if (LK_LOOP_FORM(G) == 0) {
            // one or two features per wave: epochs of constant integer origin, the inner loop is pure register arithmetic
            // (measured on MI355X, LK chain ms for 32 sequences, epoch / flat: W=21 G=64 2.12 / 2.42, W=15 G=32 1.48 / 1.55)
            bool stop = max_count <= 0;
            while (!stop) {
                const float fx0 = floorf(nx), fy0 = floorf(ny);             // (float)(int)floorf(n) == floorf(n) wherever the origin is in reach
                const int inx = (int)fx0, iny = (int)fy0;
                // -W <= in < size  <=>  (unsigned)(in + W) < (unsigned)(size + W): one compare per axis
                if (uni<G>((unsigned)(inx + W) >= (unsigned)(L.w + W) || (unsigned)(iny + W) >= (unsigned)(L.h + W))) {
                    if (level == 0) status = 0;
                    break;
                }
                load_window(inx, iny);
                float fa = nx - fx0, fb = ny - fy0;
                for (;;) {
                    if (newton_step(fa, fb)) { stop = true; break; }
                    // same epoch <=> floor(n) == origin <=> 0 <= n - origin < 1.  The difference is exact there (|n| < 2^23, so it is
                    // a multiple of ulp(n) below 1: at most 24 bits) and rounding is monotone elsewhere; a float in [0, 1) is exactly
                    // a bit pattern below 0x3F800000 (negatives have the sign bit set)
                    fa = nx - fx0; fb = ny - fy0;
                    if (uni<G>(__float_as_uint(fa) >= 0x3F800000u || __float_as_uint(fb) >= 0x3F800000u)) break;
                }
            }
        } else {
            // four features per wave: one flat loop, the groups reload independently under the exec mask while the
            // arithmetic of the iteration stays converged (W=10 G=16: 1.05 flat / 1.10 epoch, W=7: 0.77 / 0.83).  A third form,
            // wave-wide epochs left by a ballot as soon as any group's origin moves, cost registers (W=21 G=64: 131 vs 112
            // VGPRs, 4 -> 3 waves per SIMD) and was slower everywhere (W=21 G=32: 2.29 ms vs 2.03 for the epoch form).
            bool run = crit.max_count > 0;
            int inx = -0x40000000, iny = -0x40000000;
            while (run) {
                const int cx = (int)floorf(nx), cy = (int)floorf(ny);
                if (cx != inx || cy != iny) {
                    inx = cx; iny = cy;
                    if (inx < -W || inx >= L.w || iny < -W || iny >= L.h) {
                        if (level == 0) status = 0;
                        break;
                    }
                    load_window(inx, iny);
                }
                if (newton_step(nx - (float)inx, ny - (float)iny)) break;
            }
        }
        if constexpr (G == 64) n_steps += j + ((conv || osc) ? 1 : 0);       // a step that ends the track returns before ++j
        if (moved) {                                                         // nextPts[i] = nextPt + halfWin (every iteration in lkpyramid.cpp)
            outx = nx + half; outy = ny + half;
            if (osc) { outx -= ldx * 0.5f; outy -= ldy * 0.5f; }
        }
        // flags = 0 with err != NULL (vo.cpp:182,203): the level-0 error block re-checks the final window origin
        if (status && level == 0) {
            int ix = (int)floorf(outx - half), iy = (int)floorf(outy - half);
            if (ix < -W || ix >= L.w || iy < -W || iy >= L.h) status = 0;
        }
    }
}

// per-lane description of the window segments it owns
template <int W, int G>
__device__ __forceinline__ void lk_segments(LkSegs<LkLayout<W, G>::SPL>& sg) {
    using LL = LkLayout<W, G>;
    const int li = threadIdx.x % G;
#pragma unroll
    for (int k = 0; k < LL::SPL; k++) {
        const int sidx = li * LL::SPL + k;
        sg.on[k] = sidx < LL::NSEG;
        const int sc = sg.on[k] ? sidx : 0;          // idle slots shadow segment 0 so their loads stay in bounds
        sg.row[k] = sc / LL::LPR;
        sg.xs[k] = (sc % LL::LPR) * LL::PPL;
    }
}

__device__ __forceinline__ LkCrit make_crit(const svo_config& c, float mineig_cut) {
    LkCrit k;
    int mc = c.lk_max_count; mc = mc < 0 ? 0 : (mc > 100 ? 100 : mc);       // TermCriteria normalisation (lkpyramid.cpp)
    double e = c.lk_epsilon; e = e < 0. ? 0. : (e > 10. ? 10. : e);
    k.max_count = mc; k.eps2 = e * e; k.mineig_cut = mineig_cut;
    // f32 screen of "dx^2 + dy^2 <= eps2" (the f32 sum is within 2e-7 relative of the exact one): above eps_hi certainly not
    // converged, below eps_lo certainly converged, in between (practically never) the exact f64 comparison decides
    if (k.eps2 > 1e-30) { k.eps_hi = (float)(k.eps2 * 1.00001); k.eps_lo = (float)(k.eps2 * 0.99999); }
    else { k.eps_hi = __builtin_inff(); k.eps_lo = -1.f; }                        // tiny epsilon: always the exact path
    return k;
}

// ---- fused circular matching: L0 -> L1 -> R1 -> R0 -> L0 + masks (vo.cpp:203-230, 341-359) ----
// Block -> (sequence, feature group) mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b lands on XCD
// b % 8, MI355X_MICROARCH.md; used for speed only, never for correctness) and every XCD has a private 4 MiB L2; one
// sequence's four pyramids are 2.5 MB and 2-3 sequences are in flight at any time.  Three forms (launch argument `mode`):
//  * LK_MAP_STRIPE (default): sequence-major, `slots` blocks per sequence; XCD x takes runs of `chunk` consecutive
//    feature groups, cyclically (run q of XCD x = groups (8q + x) * chunk ...).  Features are in bucket raster order, so
//    a run is a short horizontal strip of the image whose windows share cache lines, and every XCD touches a fraction of
//    each pyramid instead of all of it.  The runs must stay short: work per feature is spatially correlated and the
//    dispatcher is in-order, so long runs unbalance the XCDs.
//  * LK_MAP_INTERLEAVED (= chunk 1): every XCD sees every 8th feature, i.e. the whole image.
//  * LK_MAP_AFFINE (SVO_LK_XCD=1): XCD x is given the sequences x, x+8, ... and walks through them one at a time.
// Measured on MI355X, 32 sequences per launch (rocprofv3 FETCH_SIZE per launch / HIP-event time):
//   chunk 1: 340 MB / 1.90 ms    4: 214 / 1.91    16: 174 / 1.94    64: 118 / 2.18    whole eighths: 61 / 2.79
//   affine: 39 MB / 2.02 ms.  The kernel is VALU-bound and the fetches are served by the Infinity Cache, so the default
//   takes a traffic reduction that is free; the others stay selectable.  Round 1 chose 8; re-measured at the end of round 2 at
//   the default bench configuration (two contexts of 128 sequences, A/B in one call): 4 is 0.5-0.7 % faster than 8 on both the
//   mover and the static scene (1, 2 and 4 tie), 16 is 1.8 % slower: default 4.
#define LK_MAP_STRIPE 0
#define LK_MAP_AFFINE 1
#define LK_MAP_INTERLEAVED 2
// The four passes of ONE feature (or FPW features side by side) and its masks: one inlined copy inside k_lk_chain (experiments/lk_queue_fed_persistent.patch fed the same function from a work queue).
struct LkSeqCtx { const uint8_t *L0, *R0, *L1, *R1; int seq, buf; };
template <int W, int G, int CN, bool FS>
__device__ __forceinline__ void lk_chain_feature(const DevBuffers& d, const LkSeqCtx& q, int idx, const LkCrit& crit, float thr, float Wf, float Hf,
                                                 const LkSegs<LkLayout<W, G>::SPL>& sg, int early_out, int* fs_lds) {
    constexpr int FPW = 64 / G;
    const int seq = q.seq;
    const uint8_t *L0 = q.L0, *R0 = q.R0, *L1 = q.L1, *R1 = q.R1;
    const size_t o = (size_t)seq * d.CAP + idx;
    const float2 p0 = d.feat_xy[q.buf][o];                  // pointsLeftT0 = currentVOFeatures.points (vo.cpp:338)
    // the four passes share ONE inlined copy of lk_pass (a loop, not four copies): 4x less code in the instruction cache.
    // Every pass's point is written out and folded into the masks as soon as it exists, so only the running point,
    // the start point and two flags stay live across the passes (fewer registers held through lk_pass).
    const bool writer = threadIdx.x % G == 0;
    // per-feature state across the passes, packed so that it holds ONE register through lk_pass (the kernel sits at the edge
    // of its register budget: 104 VGPRs leave room for the other context's f64 kernels, svo_api.hip LkGate): bit 0 = every
    // status so far is 1, bit 1 = every point so far lies inside the image, bits 2-3 = 1 + the first pass (0..2) that
    // returned status 0 (0 = none)
    int flags = 1 | ((!((p0.x < 0) || (p0.y < 0) || (p0.y >= Hf) || (p0.x >= Wf))) << 1);         // vo.cpp:344-359, pointsLeftT0
    float2 cur = p0;
    int n_visits = 0, n_steps = 0;                                 // svo_frame_stats.lk_level_visits / lk_newton_steps
    if (writer) d.pl0[o] = p0;
#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
        const uint8_t* A = pass == 0 ? L0 : pass == 1 ? L1 : pass == 2 ? R1 : R0;      // vo.cpp:203, 206, 209, 213
        const uint8_t* Bq = pass == 0 ? L1 : pass == 1 ? R1 : pass == 2 ? R0 : L0;
        float2* out = pass == 0 ? d.pl1 : pass == 1 ? d.pr1 : pass == 2 ? d.pr0 : d.plc;
        float2 q; int st;
        lk_pass<W, G, CN, FS>(d.geom, A, Bq, (size_t)d.geom.pyr_bytes, cur.x, cur.y, q.x, q.y, st, crit, sg, n_visits, n_steps, fs_lds);
        if (pass < 3 && ((q.x < 0) || (q.y < 0) || (q.y >= Hf) || (q.x >= Wf))) flags &= ~2;     // pl1, pr1, pr0 (not the returned point)
        if (writer) out[o] = q;
        cur = q;
        // A feature whose status is 0 is deleted at vo.cpp:233-238 whatever the remaining passes return (the mask is the AND of
        // the four statuses, vo.cpp:227-230): the frame pipeline stops here — nothing it reports can tell.  The member-call form
        // (svo_circular_matching) hands the raw points of every pass to the caller and runs all four (early_out = 0).
        if (st == 0) {
            if ((flags & 1) && pass < 3) flags |= (pass + 1) << 2;
            flags &= ~1;
            if (early_out && pass < 3) { if constexpr (G == 64) { if (uni<G>(true)) break; } else if (FPW == 1) break; }
        }
    }
    if (writer) {
        float ex = fabsf(p0.x - cur.x), ey = fabsf(p0.y - cur.y);
        float off = (ex < ey) ? ey : ex;
        int circ = (flags & 1) && !(off > thr);                                         // vo.cpp:227-230
        d.okmask[o] = (uint8_t)(circ | (flags & 2));
        // work counters of svo_frame_stats: one plain store per feature, summed by k_compact (two atomicAdd per feature on
        // one address per sequence kept every wave's slot occupied until they drained: +14 % LK time, measured)
        d.lk_work[o] = ((unsigned)n_steps << 8) | ((unsigned)(flags >> 2) << 6) | (unsigned)n_visits;
    }

}

// Waves per SIMD asked of the compiler (its register budget is 512 / waves, allocated in eights).  Since the pyramid levels carry
// their border the kernel needs 82 registers at w = 21 (99 with the per-byte border paths), and more resident waves pay: measured
// on one box (tools/ab_bench.sh: LK ms per 32-sequence launch / whole job at two contexts of 128 sequences; rocprofv3 FETCH_SIZE +
// WRITE_SIZE per 32-sequence launch against 163 MB algorithmic):
//   5 waves (81 registers, the compiler's own choice)   1.76 ms / 17 240 frame-pairs/s / 302 MB
//   6 waves (80 registers, no scratch)                  1.72 ms / 17 650            / 341 MB
//   7 waves (72 registers, 10 dwords of scratch)        1.72 ms / 17 820            / 842 MB
// Six: the seventh wave's 1 % is paid with 0.5 GB of scratch traffic per launch (every single-feature wave spills at entry).  Other
// windows keep the compiler's choice (not measured).  Beside 6 x 80 registers none of the other context's f64 kernels fits, so the
// LkGate chaining (svo_api.hip) switches itself off and the two contexts' LK launches follow each other, each filling the other's tail.
// Float-sums build at W = 21: FIVE waves per SIMD — its LDS (7.1 KB per one-wave block) allows it and the register cap (96) costs
// 8 dwords of scratch; measured against four waves without scratch: LK 4.13 vs 4.22 ms per 32-sequence launch, 8 190 vs 7 910
// frame-pairs/s at the default configuration in float-sums mode (same box; -DFS_WAVES21=1 for the A/B).
#ifndef FS_WAVES21
#define FS_WAVES21 5
#endif
template <int W, int G, int CN, bool FS> constexpr int lk_min_waves() { return (W == 21 && G == 64 && CN == 1) ? (FS ? FS_WAVES21 : 6) : 1; }
#ifdef LK_EXP_MINWAVES                      // experiments: -DLK_EXP_MINWAVES=<n> overrides the table
#define LK_MIN_WAVES(W, G, CN, FS) LK_EXP_MINWAVES
#else
#define LK_MIN_WAVES(W, G, CN, FS) (lk_min_waves<W, G, CN, FS>())
#endif
template <int W, int G, int CN, bool FS = false>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 64), amdgpu_waves_per_eu(LK_MIN_WAVES(W, G, CN, FS)))) void k_lk_chain(DevBuffers d, int slots, int mode, int chunk, int early_out) {
    constexpr int FPW = 64 / G;                                       // features per wave (= per block)
    __shared__ __attribute__((aligned(16))) int fs_lds[FS ? LkFs<W, CN>::LDS_INTS : 1];            // float-sums mode only (the default build uses no LDS)
    int seq, fb;
    if (mode == LK_MAP_AFFINE) {
        const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
        const int jr = t / slots;                                     // round of sequences on this XCD
        fb = t - jr * slots;                                          // feature block
        seq = jr * 8 + xcd;
    } else {
        seq = blockIdx.x / slots; fb = blockIdx.x - seq * slots;
    }
    if (seq >= d.B) return;
    SeqState& s = d.st[seq];
    if (!s.active) return;
    int n = s.n_feat;
    if (d.cfg.max_features > 0 && n > d.cfg.max_features) n = d.cfg.max_features;
    if (fb == 0 && threadIdx.x == 0) s.n_lk = n;
    const uint8_t* L0 = d.pyr + pyr_index(d, seq, s.slot_pyr_t0, 0);
    const uint8_t* R0 = d.pyr + pyr_index(d, seq, s.slot_pyr_t0, 1);
    const uint8_t* L1 = d.pyr + pyr_index(d, seq, s.slot_t1, 0);
    const uint8_t* R1 = d.pyr + pyr_index(d, seq, s.slot_t1, 1);
    const LkSeqCtx sc = {L0, R0, L1, R1, seq, s.feat_buf};
    const LkCrit crit = make_crit(d.cfg, d.lk_mineig_cut);
    const float thr = (float)d.cfg.circular_matching_success_threshold;                // findClosePoints takes a float32 (vo.h:432)
    const float Wf = (float)d.geom.W, Hf = (float)d.geom.H;
    LkSegs<LkLayout<W, G>::SPL> sg;
    lk_segments<W, G>(sg);
    const int slot = threadIdx.x / G;
    const int ngroups = (n + FPW - 1) / FPW;                          // feature groups (one block's worth) in this sequence
    for (int gbase = 0; gbase < ngroups; gbase += slots) {
        int grp;
        if (mode == LK_MAP_STRIPE) {
            const int ng = min(slots, ngroups - gbase);              // groups handled in this round by the sequence's blocks
            const int x = fb & 7, r = fb >> 3;                       // slots % 8 == 0, so fb & 7 == blockIdx.x & 7 == the XCD
            const int q = r / chunk;                                 // the r-th block of XCD x works in that XCD's q-th run
            grp = (q * 8 + x) * chunk + (r - q * chunk);
            if (grp >= ng) break;
            grp += gbase;
        } else {
            grp = gbase + fb;
            if (grp >= ngroups) break;
        }
        const int idx = grp * FPW + slot;
        if (idx >= n) continue;                                       // whole group idle (group-uniform)
        lk_chain_feature<W, G, CN, FS>(d, sc, idx, crit, thr, Wf, Hf, sg, early_out, fs_lds);
    }
}

// ---- single pass, for the cv::calcOpticalFlowPyrLK-shaped stage API ----
template <int W, int G>
__global__ __launch_bounds__(64) void k_lk_single(DevBuffers d, int slotA, int camA, int slotB, int camB, int n,
                                                  const float2* prev, float2* next, uint8_t* status) {
    constexpr int FPW = 64 / G;
    const uint8_t* A = d.pyr + pyr_index(d, 0, slotA, camA);
    const uint8_t* Bp = d.pyr + pyr_index(d, 0, slotB, camB);
    const LkCrit crit = make_crit(d.cfg, d.lk_mineig_cut);
    LkSegs<LkLayout<W, G>::SPL> sg;
    lk_segments<W, G>(sg);
    const int slot = threadIdx.x / G;
    for (int base = blockIdx.x * FPW; base < n; base += gridDim.x * FPW) {
        const int idx = base + slot;
        if (idx >= n) continue;
        float2 p = prev[idx], q; int st, nv = 0, ns = 0;
        lk_pass<W, G, 1>(d.geom, A, Bp, 0, p.x, p.y, q.x, q.y, st, crit, sg, nv, ns);
        if (threadIdx.x % G == 0) { next[idx] = q; status[idx] = (uint8_t)st; }
    }
}

#define LK_MAX_GRID 16384
static int lk_chunk() { static int v = -1; if (v < 0) { const char* e = getenv("SVO_LK_CHUNK"); v = e ? atoi(e) : 4; if (v < 1) v = 1; if (v > 2048) v = 2048; } return v; }
static int lk_xcd_mapping() { static int v = -1; if (v < 0) { const char* e = getenv("SVO_LK_XCD"); v = e ? atoi(e) : 0; } return v; }
// (window, lanes per feature) instantiations; the FIRST entry of a window is its default, the others are selectable with
// SVO_LK_G=<lanes> for measurement.
// winSize is a mutable member in the reference (vo.h:251), so every square window from 5 to 31 is built: the tuned entries
// first (lanes per feature chosen by measurement), then the generic one-wave-per-feature form for all other sizes — the same
// code (LkLayout derives the segment shape from W), just not tuned.
#ifdef SVO_LK_DEV_W21   // developer build (-DSVO_LK_DEV_W21): only the w = 21 grey default-mode kernel, compiles in seconds; never shipped
#define LK_FOR_EACH_WINDOW(X) X(21, 64)
#define LK_FOR_EACH_WINDOW_CN3(X)
#define LK_FOR_EACH_WINDOW_FS(X)
#define LK_FOR_EACH_WINDOW_FS_CN3(X)
#else
#define LK_FOR_EACH_WINDOW_TUNED(X) X(7, 16) X(7, 64) X(10, 16) X(10, 64) X(15, 32) X(15, 64) X(21, 64) X(21, 32) X(31, 64)
#define LK_FOR_EACH_WINDOW_GENERIC(X) X(5, 64) X(6, 64) X(8, 64) X(9, 64) X(11, 64) X(12, 64) X(13, 64) X(14, 64) X(16, 64) X(17, 64) X(18, 64) \
    X(19, 64) X(20, 64) X(22, 64) X(23, 64) X(24, 64) X(25, 64) X(26, 64) X(27, 64) X(28, 64) X(29, 64) X(30, 64)
#define LK_FOR_EACH_WINDOW(X) LK_FOR_EACH_WINDOW_TUNED(X) LK_FOR_EACH_WINDOW_GENERIC(X)

// 3-channel (BGR) instantiations: one per window up to 21, at the window's default lanes-per-feature (beyond that a lane would
// hold three planes of >= 11 pixels of template and search window: more registers than a wave has)
#define LK_FOR_EACH_WINDOW_CN3(X) X(7, 16) X(7, 64) X(10, 16) X(10, 64) X(15, 32) X(15, 64) X(21, 64) X(5, 64) X(6, 64) X(8, 64) X(9, 64) X(11, 64) X(12, 64) X(13, 64) \
    X(14, 64) X(16, 64) X(17, 64) X(18, 64) X(19, 64) X(20, 64)

// float-sums builds (svo_config.lk_float_sums): one feature per wave at every window
#define LK_FOR_EACH_WINDOW_FS(X) X(5, 64) X(6, 64) X(7, 64) X(8, 64) X(9, 64) X(10, 64) X(11, 64) X(12, 64) X(13, 64) X(14, 64) X(15, 64) X(16, 64) X(17, 64) \
    X(18, 64) X(19, 64) X(20, 64) X(21, 64) X(22, 64) X(23, 64) X(24, 64) X(25, 64) X(26, 64) X(27, 64) X(28, 64) X(29, 64) X(30, 64) X(31, 64)
#define LK_FOR_EACH_WINDOW_FS_CN3(X) X(5, 64) X(6, 64) X(7, 64) X(8, 64) X(9, 64) X(10, 64) X(11, 64) X(12, 64) X(13, 64) X(14, 64) X(15, 64) X(16, 64) X(17, 64) \
    X(18, 64) X(19, 64) X(20, 64) X(21, 64)
#endif

// Smallest float x with (double)(float)(x / (2 w^2)) >= threshold — found by bisection over the floats in their numeric order
// (IEEE f32 division on the host, the same operation the kernel would do).  +inf if no finite float qualifies.
float lk_mineig_cut(int win, double threshold) {
    const float den = (float)(2 * win * win);
    auto key_to_float = [](uint32_t k) { uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &b, 4); return f; };   // increasing in k
    auto ok = [&](uint32_t k) { const float q = key_to_float(k) / den; return (double)q >= threshold; };
    uint32_t lo = 0x00800000u, hi = 0xFF7FFFFFu;          // keys of -FLT_MAX .. +FLT_MAX
    if (!ok(hi)) return __builtin_inff();
    if (ok(lo)) return key_to_float(lo);
    while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (ok(mid)) hi = mid; else lo = mid; }
    return key_to_float(hi);
}

bool lk_window_supported(int win) {
#define CHK(Wn, Gn) if (win == Wn) return true;
    LK_FOR_EACH_WINDOW(CHK)
#undef CHK
    return false;
}
bool lk_window_supported_cn(int win, int cn) {
    if (cn == 1) return lk_window_supported(win);
#define CHK(Wn, Gn) if (win == Wn) return true;
    if (cn == 3) { LK_FOR_EACH_WINDOW_CN3(CHK) }
#undef CHK
    return false;
}
// One feature per wave (64 lanes) is the default for EVERY window.  Round 1 chose four features per wave for w = 7 and 10 and two
// for w = 15 by measurement; after round 2's work on the kernel (wave-uniform control flow, scalar base addresses, two-value
// reductions — all of which need the whole wave on one feature) the 64-lane build wins at every batch size, re-measured on
// KITTI-sized frames with ~2 000 features (LK time per frame, 64 lanes vs grouped): w = 10: 111 vs 239 us at 1 sequence, 285 vs
// 411 at 8, 1 809 vs 1 954 at 64; w = 15: 135 vs 213, 363 vs 458, 2 375 vs 2 676.  A grouped wave runs until its slowest feature
// is done and cannot branch per feature.  The grouped builds stay selectable (SVO_LK_G=16 / 32) for measurement.
// cn selects the list the answer must come from: a lanes-per-feature value asked for with SVO_LK_G is honoured only if that
// (window, lanes) pair is built for this channel count, else the window's default applies (a 3-channel context with
// SVO_LK_G=32 at w = 21 used to match no instantiation and launch nothing).
static int lk_group_for(int win, int cn = 1) {
    static int env = -1;
    if (env < 0) { const char* e = getenv("SVO_LK_G"); env = e ? atoi(e) : 0; }
    int def = 0; bool have_env = false, have64 = false;
#define CHK(Wn, Gn) if (win == Wn) { if (!def) def = Gn; if (env == Gn) have_env = true; if (Gn == 64) have64 = true; }
    if (cn == 3) { LK_FOR_EACH_WINDOW_CN3(CHK) } else { LK_FOR_EACH_WINDOW(CHK) }
#undef CHK
    return have_env ? env : have64 ? 64 : def;
}

// VGPRs a SIMD has left beside a full complement of this context's LK waves (512 per SIMD lane, allocated in eights, at most 8
// waves): the 96-register builds of the f64 kernels (svo_kernels_pnp.hip) can run under another context's LK grid only if this
// is >= 96 — true at w = 21 (100 registers: four waves, 96 left) and 31, not at w = 10 (74: six waves, 32 left).  -1 if unknown.
int lk_registers_left(const DevBuffers& d) {
    const bool fs = d.cfg.lk_float_sums != 0;
    const int G = fs ? 64 : lk_group_for(d.cfg.win_w, d.CN);
    const void* fn = nullptr;
#define PICK(Wn, Gn) if (!fn && !fs && d.cfg.win_w == Wn && G == Gn) fn = (const void*)k_lk_chain<Wn, Gn, CNn>;
#define PICKFS(Wn, Gn) if (!fn && fs && d.cfg.win_w == Wn) fn = (const void*)k_lk_chain<Wn, 64, CNn, true>;
#define CNn 1
    if (d.CN == 1) { LK_FOR_EACH_WINDOW(PICK) LK_FOR_EACH_WINDOW_FS(PICKFS) }
#undef CNn
#define CNn 3
    if (d.CN == 3) { LK_FOR_EACH_WINDOW_CN3(PICK) LK_FOR_EACH_WINDOW_FS_CN3(PICKFS) }
#undef CNn
#undef PICK
#undef PICKFS
    hipFuncAttributes at;
    if (!fn || hipFuncGetAttributes(&at, fn) != hipSuccess || at.numRegs <= 0) return -1;
    const int alloc = (at.numRegs + 7) / 8 * 8;
    int waves = 512 / alloc; if (waves > 8) waves = 8;
    return 512 - waves * alloc;
}

bool launch_lk_chain(const DevBuffers& d, int grid_n, hipStream_t st, int early_out) {
    if (grid_n < 1) grid_n = 1;
    if (grid_n > d.CAP) grid_n = d.CAP;
    const bool fs = d.cfg.lk_float_sums != 0;
    const int G = fs ? 64 : lk_group_for(d.cfg.win_w, d.CN);
    const int mode = lk_xcd_mapping();
    if (fs) {
#define LAUNCHFS(Wn, Gn) if (d.cfg.win_w == Wn) { int gx = grid_n; if (gx > LK_MAX_GRID) gx = LK_MAX_GRID; \
        const int chunk = lk_chunk(); \
        gx = (gx + 8 * chunk - 1) / (8 * chunk) * (8 * chunk); \
        const int rounds = (d.B + 7) / 8; \
        const unsigned blocks = mode == LK_MAP_AFFINE ? (unsigned)gx * 8u * (unsigned)rounds : (unsigned)gx * (unsigned)d.B; \
        hipLaunchKernelGGL((k_lk_chain<Wn, 64, CNn, true>), dim3(blocks), dim3(64), 0, st, d, gx, mode, chunk, early_out); \
        return true; }
#define CNn 1
        if (d.CN == 1) { LK_FOR_EACH_WINDOW_FS(LAUNCHFS) }
#undef CNn
#define CNn 3
        if (d.CN == 3) { LK_FOR_EACH_WINDOW_FS_CN3(LAUNCHFS) }
#undef CNn
#undef LAUNCHFS
        return false;
    }
#define LAUNCH(Wn, Gn) if (d.cfg.win_w == Wn && G == Gn) { int gx = (grid_n + (64 / Gn) - 1) / (64 / Gn); if (gx > LK_MAX_GRID) gx = LK_MAX_GRID; \
        const int chunk = lk_chunk(); \
        gx = (gx + 8 * chunk - 1) / (8 * chunk) * (8 * chunk);   /* blocks per sequence: whole runs on every XCD */ \
        const int rounds = (d.B + 7) / 8; \
        const unsigned blocks = mode == LK_MAP_AFFINE ? (unsigned)gx * 8u * (unsigned)rounds : (unsigned)gx * (unsigned)d.B; \
        hipLaunchKernelGGL((k_lk_chain<Wn, Gn, CNn>), dim3(blocks), dim3(64), 0, st, d, gx, mode, chunk, early_out); \
        return true; }
#define CNn 1
    if (d.CN == 1) { LK_FOR_EACH_WINDOW(LAUNCH) }
#undef CNn
#undef LAUNCH
    // 3-channel contexts
#define LAUNCH3(Wn, Gn) if (d.cfg.win_w == Wn && G == Gn) { int gx = (grid_n + (64 / Gn) - 1) / (64 / Gn); if (gx > LK_MAX_GRID) gx = LK_MAX_GRID; \
        const int chunk = lk_chunk(); \
        gx = (gx + 8 * chunk - 1) / (8 * chunk) * (8 * chunk); \
        const int rounds = (d.B + 7) / 8; \
        const unsigned blocks = mode == LK_MAP_AFFINE ? (unsigned)gx * 8u * (unsigned)rounds : (unsigned)gx * (unsigned)d.B; \
        hipLaunchKernelGGL((k_lk_chain<Wn, Gn, 3>), dim3(blocks), dim3(64), 0, st, d, gx, mode, chunk, early_out); \
        return true; }
    if (d.CN == 3) { LK_FOR_EACH_WINDOW_CN3(LAUNCH3) }
#undef LAUNCH3
    return false;                                                     // no (window, lanes, channels) instantiation: the caller reports it
}

void launch_lk_single(const DevBuffers& d, int slotA, int camA, int slotB, int camB, int n, const float2* prev, float2* next,
                      uint8_t* status, hipStream_t st) {
    if (n <= 0) return;
    const int G = lk_group_for(d.cfg.win_w);
#define LAUNCH(Wn, Gn) if (d.cfg.win_w == Wn && G == Gn) { int gx = (n + (64 / Gn) - 1) / (64 / Gn); if (gx > LK_MAX_GRID) gx = LK_MAX_GRID; \
        hipLaunchKernelGGL((k_lk_single<Wn, Gn>), dim3(gx), dim3(64), 0, st, d, slotA, camA, slotB, camB, n, prev, next, status); return; }
    LK_FOR_EACH_WINDOW(LAUNCH)
#undef LAUNCH
}
