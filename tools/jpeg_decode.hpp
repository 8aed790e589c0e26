// jpeg_decode.hpp — baseline JPEG decoder for the CLI harness (the reference's other bundled recordings,
// slam_feats/ and rand_feats/, are JPEG frames read with cv::imread, main.cpp:21-47).  Host-side file decoding, like
// cv::imread in the reference: not part of the GPU hot path.
//
// Written to reproduce what libjpeg(-turbo) — the decoder behind cv::imread and PIL — outputs with its default settings,
// byte for byte: baseline / extended-sequential Huffman, 8-bit, 1 or 3 components, any restart interval; the "slow integer"
// inverse DCT (13-bit constants, two passes), "fancy" triangle-filter upsampling for 2:1 horizontal (4:2:2) and 2x2 (4:2:0)
// chroma, JFIF YCbCr -> RGB with 16-bit fixed-point tables.  tests/test_jpeg_decode.py checks it against PIL on every
// sampling layout.  Progressive and arithmetic-coded files are rejected (ok = false).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace svo_jpeg {

struct Image { int w = 0, h = 0, channels = 0; std::vector<uint8_t> px; bool ok = false; };   // px: interleaved R,G,B (or gray)

namespace detail {
struct Huff { int mincode[17], maxcode[18], valptr[17]; uint8_t vals[256]; bool present = false; };
struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, bw = 0, bh = 0, dw = 0, dh = 0; std::vector<uint8_t> plane; };
static const uint8_t ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct BitReader {
    const uint8_t* p; const uint8_t* end; uint32_t acc = 0; int n = 0; bool hit_marker = false;
    void fill() {
        while (n <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;             // stuffed byte
                    else { hit_marker = true; b = 0; }                   // a marker: feed zeros, leave p on it
                } else p++;
            }
            acc |= (uint32_t)b << (24 - n); n += 8;
        }
    }
    int bit() { if (n == 0) fill(); int b = acc >> 31; acc <<= 1; n--; return b; }
    int bits(int k) { int v = 0; for (int i = 0; i < k; i++) v = (v << 1) | bit(); return v; }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};
inline int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
inline int decode(BitReader& br, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}
inline uint8_t range_limit(int x) {                                     // libjpeg's masked range table incl. the +128 level shift
    int i = x & 1023;
    return (uint8_t)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
}
// jpeg_idct_islow (jidctint.c): coefficients already dequantised, natural order; out: 8 rows of `stride`
inline void idct_islow(const int* in, uint8_t* out, int stride) {
    const int CB = 13, P1 = 2;
    const int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
              F2053 = 16819, F2562 = 20995, F3072 = 25172;
    int ws[64];
    auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
    for (int c = 0; c < 8; c++) {
        const int* ip = in + c; int* wp = ws + c;
        if (!ip[8] && !ip[16] && !ip[24] && !ip[32] && !ip[40] && !ip[48] && !ip[56]) {
            int dc = ip[0] * (1 << P1);
            for (int r = 0; r < 8; r++) wp[8 * r] = dc;
            continue;
        }
        long z2 = ip[16], z3 = ip[48];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = ip[0]; z3 = ip[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = ip[56]; tmp1 = ip[40]; tmp2 = ip[24]; tmp3 = ip[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
        wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
        wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
        wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; r++) {
        const int* wp = ws + 8 * r; uint8_t* op = out + (size_t)r * stride;
        if (!wp[1] && !wp[2] && !wp[3] && !wp[4] && !wp[5] && !wp[6] && !wp[7]) {
            uint8_t dc = range_limit(descale(wp[0], P1 + 3));
            for (int c = 0; c < 8; c++) op[c] = dc;
            continue;
        }
        long z2 = wp[2], z3 = wp[6];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        long tmp0 = ((long)wp[0] + wp[4]) * (1L << CB), tmp1 = ((long)wp[0] - wp[4]) * (1L << CB);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int S = CB + P1 + 3;
        op[0] = range_limit(descale(tmp10 + tmp3, S)); op[7] = range_limit(descale(tmp10 - tmp3, S));
        op[1] = range_limit(descale(tmp11 + tmp2, S)); op[6] = range_limit(descale(tmp11 - tmp2, S));
        op[2] = range_limit(descale(tmp12 + tmp1, S)); op[5] = range_limit(descale(tmp12 - tmp1, S));
        op[3] = range_limit(descale(tmp13 + tmp0, S)); op[4] = range_limit(descale(tmp13 - tmp0, S));
    }
}
}   // namespace detail

inline Image decode(const uint8_t* data, size_t size) {
    using namespace detail;
    Image img;
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return img;
    uint16_t qt[4][64] = {}; bool have_q[4] = {};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    int W = 0, H = 0, restart = 0, adobe_transform = -1;
    bool jfif = false;
    size_t o = 2;
    auto be16 = [&](size_t p) { return (int)data[p] << 8 | data[p + 1]; };
    while (o + 4 <= size) {
        if (data[o] != 0xFF) return img;
        int m = data[o + 1];
        if (m == 0xFF) { o++; continue; }
        o += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return img;
        int len = be16(o);
        if (len < 2 || o + len > size) return img;
        const size_t seg = o + 2, end = o + len;
        if (m == 0xDB) {
            size_t p = seg;
            while (p < end) {
                int pq = data[p] >> 4, tq = data[p] & 15; p++;
                if (tq > 3 || p + (pq ? 128 : 64) > end) return img;
                for (int i = 0; i < 64; i++) { qt[tq][ZZ[i]] = pq ? (uint16_t)be16(p) : data[p]; p += pq ? 2 : 1; }
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t p = seg;
            while (p + 17 <= end) {
                int tc = data[p] >> 4, th = data[p] & 15; p++;
                if (tc > 1 || th > 3) return img;
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0, code = 0, k = 0;
                const uint8_t* bitsp = data + p;
                for (int l = 1; l <= 16; l++) total += bitsp[l - 1];
                if (total > 256 || p + 16 + total > end) return img;
                memcpy(h.vals, data + p + 16, (size_t)total);
                for (int l = 1; l <= 16; l++) {
                    h.valptr[l] = k; h.mincode[l] = code;
                    k += bitsp[l - 1]; code += bitsp[l - 1];
                    h.maxcode[l] = bitsp[l - 1] ? code - 1 : -1;
                    code <<= 1;
                }
                h.present = true;
                p += 16 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (seg + 6 > end) return img;                              // precision, height, width, component count must lie inside the segment
            if (data[seg] != 8) return img;
            H = be16(seg + 1); W = be16(seg + 3);
            int nc = data[seg + 5];
            if (!W || !H || (nc != 1 && nc != 3) || seg + 6 + 3 * nc > end) return img;
            comps.resize(nc);
            for (int i = 0; i < nc; i++) {
                comps[i].id = data[seg + 6 + 3 * i]; comps[i].h = data[seg + 7 + 3 * i] >> 4; comps[i].v = data[seg + 7 + 3 * i] & 15;
                comps[i].tq = data[seg + 8 + 3 * i];
                if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3) return img;
            }
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8)) {
            return img;                                                 // progressive / lossless / arithmetic: not handled
        } else if (m == 0xDD) {
            if (seg + 2 > end) return img;
            restart = be16(seg);
        } else if (m == 0xE0) {
            if (len >= 7 && !memcmp(data + seg, "JFIF", 5)) jfif = true;
        } else if (m == 0xEE) {
            if (len >= 14 && !memcmp(data + seg, "Adobe", 5)) adobe_transform = data[seg + 11];
        } else if (m == 0xDA) {
            if (comps.empty() || seg + 1 > end) return img;
            int ns = data[seg];
            if (ns != (int)comps.size()) return img;                    // baseline files are one interleaved scan
            if (seg + 1 + 2 * ns + 3 > end) return img;                 // component selectors + Ss, Se, Ah/Al must lie inside the segment
            for (int i = 0; i < ns; i++) {
                int cid = data[seg + 1 + 2 * i], tt = data[seg + 2 + 2 * i];
                if ((tt >> 4) > 3 || (tt & 15) > 3) return img;         // table selectors index dc[4] / ac[4]
                bool found = false;
                for (auto& c : comps) if (c.id == cid) { c.td = tt >> 4; c.ta = tt & 15; found = true; }
                if (!found) return img;
            }
            o = end;
            break;
        }
        o = end;
    }
    if (comps.empty() || o >= size) return img;
    int hmax = 1, vmax = 1;
    for (auto& c : comps) { if (c.h > hmax) hmax = c.h; if (c.v > vmax) vmax = c.v; }
    if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }          // a single-component scan is never interleaved
    const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    for (auto& c : comps) {
        if (!have_q[c.tq] || !dc[c.td].present || !ac[c.ta].present) return img;
        c.bw = mcux * c.h * 8; c.bh = mcuy * c.v * 8;
        c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;       // downsampled_width / height
        c.plane.assign((size_t)c.bw * c.bh, 0);
    }
    // ---- entropy decoding + dequantisation + inverse DCT
    BitReader br; br.p = data + o; br.end = data + size;
    int until_restart = restart;
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            if (restart && until_restart == 0) {
                br.reset();
                while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
                if (br.p + 1 < br.end) br.p += 2;
                for (auto& c : comps) c.pred = 0;
                until_restart = restart;
            }
            for (auto& c : comps)
                for (int by = 0; by < c.v; by++)
                    for (int bx = 0; bx < c.h; bx++) {
                        int coef[64] = {0};
                        int t = decode(br, dc[c.td]);
                        if (t < 0 || t > 15) return img;
                        c.pred += extend(br.bits(t), t);
                        coef[0] = c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            int rs = decode(br, ac[c.ta]);
                            if (rs < 0) return img;
                            int r = rs >> 4, s = rs & 15;
                            if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                            k += r;
                            if (k > 63) return img;
                            coef[ZZ[k]] = extend(br.bits(s), s) * qt[c.tq][ZZ[k]];
                            k++;
                        }
                        idct_islow(coef, &c.plane[(size_t)((my * c.v + by) * 8) * c.bw + (mx * c.h + bx) * 8], c.bw);
                    }
            if (restart) until_restart--;
        }
    // ---- upsampling to full resolution (jdsample.c: fancy triangle filters for 2h1v and 2h2v, replication otherwise)
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        Comp& c = comps[ci];
        const int fh = hmax / c.h, fv = vmax / c.v;
        std::vector<uint8_t>& out = full[ci];
        out.assign((size_t)W * H, 0);
        auto row = [&](int y) { if (y < 0) y = 0; if (y >= c.dh) y = c.dh - 1; return &c.plane[(size_t)y * c.bw]; };   // edge rows are duplicated
        if (hmax % c.h || vmax % c.v) return img;
        if (fh == 1 && fv == 1) {
            for (int y = 0; y < H; y++) memcpy(&out[(size_t)y * W], row(y), (size_t)W);
        } else if (fh == 2 && (fv == 1 || fv == 2) && c.dw >= 2) {
            std::vector<int> sum(c.dw);
            std::vector<uint8_t> line(2 * (size_t)c.dw);
            for (int y = 0; y < H; y++) {
                const int iy = fv == 2 ? y / 2 : y;
                const uint8_t* r0 = row(iy);
                if (fv == 2) {
                    const uint8_t* r1 = row((y & 1) ? iy + 1 : iy - 1);    // the nearer neighbouring row
                    for (int x = 0; x < c.dw; x++) sum[x] = r0[x] * 3 + r1[x];
                    line[0] = (uint8_t)((sum[0] * 4 + 8) >> 4); line[1] = (uint8_t)((sum[0] * 3 + sum[1] + 7) >> 4);
                    for (int x = 1; x < c.dw - 1; x++) {
                        line[2 * x] = (uint8_t)((sum[x] * 3 + sum[x - 1] + 8) >> 4);
                        line[2 * x + 1] = (uint8_t)((sum[x] * 3 + sum[x + 1] + 7) >> 4);
                    }
                    const int x = c.dw - 1;
                    line[2 * x] = (uint8_t)((sum[x] * 3 + sum[x - 1] + 8) >> 4); line[2 * x + 1] = (uint8_t)((sum[x] * 4 + 7) >> 4);
                } else {
                    line[0] = r0[0]; line[1] = (uint8_t)((r0[0] * 3 + r0[1] + 2) >> 2);
                    for (int x = 1; x < c.dw - 1; x++) {
                        line[2 * x] = (uint8_t)((r0[x] * 3 + r0[x - 1] + 1) >> 2);
                        line[2 * x + 1] = (uint8_t)((r0[x] * 3 + r0[x + 1] + 2) >> 2);
                    }
                    const int x = c.dw - 1;
                    line[2 * x] = (uint8_t)((r0[x] * 3 + r0[x - 1] + 1) >> 2); line[2 * x + 1] = r0[x];
                }
                memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
            }
        } else {
            for (int y = 0; y < H; y++) { const uint8_t* r0 = row(y / fv); for (int x = 0; x < W; x++) out[(size_t)y * W + x] = r0[x / fh]; }
        }
    }
    // ---- colour conversion (jdcolor.c, JFIF YCbCr -> RGB) or gray
    img.w = W; img.h = H; img.channels = (int)comps.size();
    img.px.resize((size_t)W * H * img.channels);
    if (comps.size() == 1) img.px = full[0];
    else {
        const bool rgb = adobe_transform == 0 || (!jfif && adobe_transform < 0 && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
        auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        const int ONE_HALF = 1 << 15;
        auto FIX = [](double x) { return (int)(x * 65536 + 0.5); };
        const int c_r = FIX(1.40200), c_b = FIX(1.77200), c_gr = FIX(0.71414), c_gb = FIX(0.34414);
        for (size_t i = 0; i < (size_t)W * H; i++) {
            int y = full[0][i], cb = full[1][i], cr = full[2][i];
            if (rgb) { img.px[3 * i] = (uint8_t)y; img.px[3 * i + 1] = (uint8_t)cb; img.px[3 * i + 2] = (uint8_t)cr; continue; }
            const int xb = cb - 128, xr = cr - 128;
            const int r = y + ((c_r * xr + ONE_HALF) >> 16);
            const int b = y + ((c_b * xb + ONE_HALF) >> 16);
            const int g = y + (((-c_gb) * xb + ONE_HALF + (-c_gr) * xr) >> 16);
            img.px[3 * i] = clamp(r); img.px[3 * i + 1] = clamp(g); img.px[3 * i + 2] = clamp(b);
        }
    }
    img.ok = true;
    return img;
}

}   // namespace svo_jpeg
