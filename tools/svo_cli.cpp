// svo_cli — counterpart of the reference's CLI harness `vo <N_FRAMES> <folder>` (reference src/main.cpp:21-47,
// 315-407) on top of the C++ facade / C-ABI.  SURVEY.md §8 f-1.
//
//   svo_cli <N_FRAMES> <folder> [--calib file.yaml] [--out result.csv] [--device d] [--gray 1] [--identity-start 1]
//                               [--float-sums 1] [--ref-format 1]
//
// Reads folder/left/frameNNNNNN.{pgm,png,jpg} (6 digits, as `run1`) or frameNNNN.{jpg,png,pgm} (4 digits, as the reference's
// other two bundled sets slam_feats/ and rand_feats/, which its current CLI cannot open; tools/jpeg_decode.hpp) and
// folder/right/..., runs VisualOdometry::stereo_callback per pair, integrates frame_pose = frame_pose * T from the
// 26-degree pitched initial pose (main.cpp:368-373, 396) and writes `x,y,z,gtx,gty` rows (main.cpp:346-348, 397-400),
// reading folder/gt.csv with the reference's column quirk (main.cpp:336-344, 385-392).
// Like the reference, the frames go in as 3-channel BGR: readImages computes a gray image and then returns the colour Mats
// (main.cpp:38-46; cv::imread turns gray files into 3-channel ones too).  `--gray 1` converts with the
// cv::cvtColor BGR2GRAY integer formula and runs the single-channel path instead (what the ROS node delivers).
// `--identity-start 1` starts from the identity pose: the bundled run1/result.csv was recorded that way (it predates the
// 26-degree pitch of main.cpp:368-373), and with it this tool reproduces that file (tests/test_run1_color.py).
// `--float-sums 1` selects svo_config.lk_float_sums (LK sums in float, in the lane order of OpenCV's SIMD128 code) and
// `--ref-format 1` prints the rows like the reference's ofstream does (6 significant digits, main.cpp:397-400): together with
// `--identity-start 1` the tool then writes run1/result.csv's x,y,z columns digit for digit (tests/test_run1_color.py).
// Differences, on purpose:
//   * the run stops cleanly at the first missing image pair (the reference throws on run1's frame 128, B-12);
//   * calibration may come from a YAML file with either key style (stereo_vo.cpp:40-44 `fx:` or kitti00.yaml `Camera.fx:`);
//     without --calib the hard-coded run1 projection of main.cpp:357-364 is used.
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "svo/visual_odometry.hpp"
#include "jpeg_decode.hpp"
using namespace visual_odometry;

struct Gray { int w = 0, h = 0; std::vector<uint8_t> px, bgr; bool ok() const { return w > 0; } };   // px: gray, bgr: interleaved B,G,R

static bool read_file(const std::string& p, std::vector<uint8_t>& out) {
    std::ifstream f(p, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

static Gray read_pgm(const std::vector<uint8_t>& d) {
    Gray g; size_t i = 0; int vals[3], n = 0;
    if (d.size() < 2 || d[0] != 'P' || d[1] != '5') return g;
    i = 2;
    while (n < 3 && i < d.size()) {
        while (i < d.size() && isspace(d[i])) i++;
        if (i < d.size() && d[i] == '#') { while (i < d.size() && d[i] != '\n') i++; continue; }
        int v = 0; bool any = false;
        while (i < d.size() && isdigit(d[i])) { v = v * 10 + (d[i] - '0'); i++; any = true; }
        if (!any) return g;
        vals[n++] = v;
    }
    i++;                                            // single whitespace after maxval
    if (n < 3 || vals[2] != 255 || d.size() < i + (size_t)vals[0] * vals[1]) return g;
    g.w = vals[0]; g.h = vals[1]; g.px.assign(d.begin() + i, d.begin() + i + (size_t)g.w * g.h);
    g.bgr.resize(g.px.size() * 3);
    for (size_t k = 0; k < g.px.size(); k++) g.bgr[3 * k] = g.bgr[3 * k + 1] = g.bgr[3 * k + 2] = g.px[k];   // cv::imread replicates gray files
    return g;
}

// minimal PNG: 8-bit, colour type 0 (gray) / 2 (RGB) / 4 (gray+alpha) / 6 (RGBA), non-interlaced
static Gray read_png(const std::vector<uint8_t>& d) {
    Gray g;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 33 || memcmp(d.data(), sig, 8)) return g;
    auto be32 = [&](size_t o) { return (uint32_t)d[o] << 24 | (uint32_t)d[o + 1] << 16 | (uint32_t)d[o + 2] << 8 | d[o + 3]; };
    size_t o = 8; uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (o + 12 <= d.size()) {
        uint32_t len = be32(o); std::string type((const char*)&d[o + 4], 4);
        if (o + 12 + len > d.size()) return g;
        if (type == "IHDR") { w = be32(o + 8); h = be32(o + 12); depth = d[o + 16]; ctype = d[o + 17]; interlace = d[o + 20]; }
        else if (type == "IDAT") idat.insert(idat.end(), d.begin() + o + 8, d.begin() + o + 8 + len);
        else if (type == "IEND") break;
        o += 12 + len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!w || !h || depth != 8 || !ch || interlace) return g;
    size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) return g;
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* in = &raw[(stride + 1) * y + 1]; int ft = raw[(stride + 1) * y];
        uint8_t* cur = &img[stride * y]; const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0, v = in[x];
            if (ft == 1) v += a; else if (ft == 2) v += b; else if (ft == 3) v += (a + b) >> 1;
            else if (ft == 4) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[x] = (uint8_t)v;
        }
    }
    g.w = w; g.h = h; g.px.resize((size_t)w * h); g.bgr.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        int R, G, B;
        if (ch <= 2) { R = G = B = img[i * ch]; g.px[i] = (uint8_t)R; }
        else { R = img[i * ch]; G = img[i * ch + 1]; B = img[i * ch + 2]; g.px[i] = (uint8_t)((B * 1868 + G * 9617 + R * 4899 + 8192) >> 14); }   // BGR2GRAY
        g.bgr[3 * i] = (uint8_t)B; g.bgr[3 * i + 1] = (uint8_t)G; g.bgr[3 * i + 2] = (uint8_t)R;        // cv::imread: BGR, gray files replicated
    }
    return g;
}

static Gray read_jpeg(const std::vector<uint8_t>& d) {
    Gray g;
    svo_jpeg::Image im = svo_jpeg::decode(d.data(), d.size());
    if (!im.ok) return g;
    g.w = im.w; g.h = im.h; g.px.resize((size_t)im.w * im.h); g.bgr.resize((size_t)im.w * im.h * 3);
    for (size_t i = 0; i < (size_t)im.w * im.h; i++) {
        const int R = im.px[i * im.channels], G = im.channels == 3 ? im.px[i * 3 + 1] : R, B = im.channels == 3 ? im.px[i * 3 + 2] : R;
        g.px[i] = im.channels == 3 ? (uint8_t)((B * 1868 + G * 9617 + R * 4899 + 8192) >> 14) : (uint8_t)R;    // BGR2GRAY
        g.bgr[3 * i] = (uint8_t)B; g.bgr[3 * i + 1] = (uint8_t)G; g.bgr[3 * i + 2] = (uint8_t)R;                // cv::imread: BGR
    }
    return g;
}

// folder/side/frame%06d.* (run1's naming, main.cpp:21-36) or frame%04d.* (slam_feats / rand_feats)
static Gray read_image(const std::string& dir, int index) {
    std::vector<uint8_t> d;
    for (const char* fmt : {"/frame%06d", "/frame%04d"}) {
        char name[64]; std::snprintf(name, sizeof(name), fmt, index);
        const std::string base = dir + name;
        if (read_file(base + ".pgm", d)) return read_pgm(d);
        if (read_file(base + ".png", d)) return read_png(d);
        if (read_file(base + ".jpg", d)) return read_jpeg(d);
    }
    return Gray();
}

// OpenCV-YAML calibration: `%YAML:1.0` header, `key: value`, `#` comments; keys fx fy cx cy bf, optionally `Camera.`-prefixed
static bool read_calibration(const std::string& path, Mat34f& Pl, Mat34f& Pr) {
    std::ifstream f(path);
    if (!f) return false;
    std::map<std::string, double> kv; std::string line;
    while (std::getline(f, line)) {
        size_t h = line.find('#'); if (h != std::string::npos) line.resize(h);
        if (line.empty() || line[0] == '%') continue;
        size_t c = line.find(':'); if (c == std::string::npos) continue;
        std::string k = line.substr(0, c), v = line.substr(c + 1);
        k.erase(0, k.find_first_not_of(" \t")); k.erase(k.find_last_not_of(" \t") + 1);
        if (k.rfind("Camera.", 0) == 0) k = k.substr(7);
        char* end = nullptr; double x = strtod(v.c_str(), &end);
        if (end != v.c_str()) kv[k] = x;
    }
    for (const char* k : {"fx", "fy", "cx", "cy", "bf"}) if (!kv.count(k)) return false;
    Pl = {(float)kv["fx"], 0, (float)kv["cx"], 0, 0, (float)kv["fy"], (float)kv["cy"], 0, 0, 0, 1, 0};    // stereo_vo.cpp:46-47
    Pr = Pl; Pr[3] = (float)kv["bf"];
    return true;
}

static void matmul4(const double* A, const double* B, double* C) {
    double t[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j]; t[4 * i + j] = s; }
    memcpy(C, t, sizeof(t));
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <N_FRAMES> <folder> [--calib file.yaml] [--out result.csv] [--device d] [--gray 1] [--identity-start 1] [--float-sums 1] [--ref-format 1]\n", argv[0]); return 2; }
    const int N_FRAMES = std::atoi(argv[1]);
    const std::string folder = argv[2];
    std::string calib, out = folder + "/result.csv";
    bool gray = false, identity_start = false, float_sums = false, ref_format = false;
    for (int i = 3; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--gray")) gray = std::atoi(argv[i + 1]) != 0;
        else if (!strcmp(argv[i], "--identity-start")) identity_start = std::atoi(argv[i + 1]) != 0;
        else if (!strcmp(argv[i], "--float-sums")) float_sums = std::atoi(argv[i + 1]) != 0;
        else if (!strcmp(argv[i], "--ref-format")) ref_format = std::atoi(argv[i + 1]) != 0;
        else if (!strcmp(argv[i], "--calib")) calib = argv[i + 1];
        else if (!strcmp(argv[i], "--out")) out = argv[i + 1];
        else if (!strcmp(argv[i], "--device")) default_device() = std::atoi(argv[i + 1]);
    }
    Mat34f Pl = {322.11376f, 0, 327.47336f, 0, 0, 322.11376f, 176.33722f, 0, 0, 0, 1, 0};               // main.cpp:357-362
    Mat34f Pr = Pl; Pr[3] = -22.5428f;
    if (!calib.empty() && !read_calibration(calib, Pl, Pr)) { std::fprintf(stderr, "cannot read calibration %s\n", calib.c_str()); return 2; }
    // ground truth: skip the header line, then drop the first column of every row (main.cpp:336-344, 385-392)
    std::ifstream gt(folder + "/gt.csv");
    const bool has_gt = (bool)gt;
    std::string tmp;
    if (has_gt) { std::getline(gt, tmp); std::getline(gt, tmp, ','); }
    std::ofstream res(out);
    if (!res) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 2; }
    res << "x,y,z,gtx,gty\n";
    try {
        svo_config cfg; svo_config_default(&cfg);
        cfg.lk_float_sums = float_sums ? 1 : 0;
        VisualOdometry vo(cfg);
        vo.initalize_projection_matricies(Pl, Pr);
        const double theta = (26.0 / 360) * 2 * M_PI;                                                    // main.cpp:368-373
        double pose[16] = {1, 0, 0, 0, 0, cos(theta), sin(theta), 0, 0, -sin(theta), cos(theta), 0, 0, 0, 0, 1};
        if (identity_start) { const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; memcpy(pose, I, sizeof(I)); }
        int done = 0;
        for (int i = 0; i < N_FRAMES; i++) {
            Gray l = read_image(folder + "/left", i), r = read_image(folder + "/right", i);
            if (!l.ok() || !r.ok() || l.w != r.w || l.h != r.h) break;                                   // stop at the first missing pair
            double gtx = 0, gty = 0;
            if (has_gt) {
                std::string xs, ys, dxs, dys;
                std::getline(gt, xs, ','); std::getline(gt, ys, ','); std::getline(gt, dxs, ','); std::getline(gt, dys, ',');
                gtx = atof(xs.c_str()); gty = atof(ys.c_str());
            }
            auto o = gray ? vo.stereo_callback(Image(l.px.data(), l.h, l.w), Image(r.px.data(), r.h, r.w))
                          : vo.stereo_callback(Image(l.bgr.data(), l.h, l.w, 0, 3), Image(r.bgr.data(), r.h, r.w, 0, 3));
            matmul4(pose, o.second.data(), pose);                                                         // applied even when !ok (main.cpp:394-396)
            char row[256];
            std::snprintf(row, sizeof(row), ref_format ? "%g,%g,%g,%g,%g\n" : "%.9g,%.9g,%.9g,%.9g,%.9g\n", pose[3], pose[7], pose[11], gtx, gty);
            res << row;
            std::printf("Frame %d: ok=%d tracks=%d inliers=%d\n", i, (int)o.first, vo.stats.n_after_bounds, vo.stats.n_inliers);
            done++;
        }
        std::printf("processed %d frame pairs -> %s\n", done, out.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
