// svo_latency — single-stream latency of the synchronous callback as a C / C++ host sees it (no Python in the loop): the
// reference's own call pattern, one robot, one stream, stereo_callback(left, right) per camera frame (src/stereo_vo.cpp:61-62,
// src/main.cpp:394).  Reads stereo pairs from a raw file (int32 n, h, w; then n x (left, right) gray images), plays them
// ping-pong through svo_process and prints the mean / median / p95 wall time per call, with ordinary heap buffers and with
// page-locked ones (svo_alloc_pinned).
//   svo_latency frames.bin [win=21] [calls=200] [max_translation=2.0]   (KITTI-00 intrinsics: the file comes from tools/latency_cpp.py)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>
#include "svo.h"

static double run(const std::vector<const uint8_t*>& L, const std::vector<const uint8_t*>& R, int w, int h, int win, int calls, double max_t, const char* label) {
    svo_config cfg; svo_config_default(&cfg);
    cfg.win_w = cfg.win_h = win; cfg.max_translation_norm = max_t;
    svo_context* ctx = nullptr;
    if (svo_create(&cfg, 0, 1, w, h, &ctx) != SVO_OK) { std::fprintf(stderr, "svo_create: %s\n", svo_last_error()); std::exit(1); }
    const float Pl[12] = {718.856f, 0, 607.1928f, 0, 0, 718.856f, 185.2157f, 0, 0, 0, 1, 0};
    float Pr[12]; std::memcpy(Pr, Pl, sizeof(Pl)); Pr[3] = -386.1448f;
    svo_set_projection(ctx, -1, Pl, Pr);
    const int n = (int)L.size();
    auto pp = [&](int i) { const int p = i % (2 * n - 2); return p < n ? p : 2 * n - 2 - p; };
    double T[16]; svo_frame_stats st; int n_ok = 0;
    for (int i = 0; i < 8; i++) svo_process(ctx, L[pp(i)], R[pp(i)], w, T, &st);
    std::vector<double> us(calls);
    for (int i = 0; i < calls; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = svo_process(ctx, L[pp(8 + i)], R[pp(8 + i)], w, T, &st);
        us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rc < 0) { std::fprintf(stderr, "svo_process: %s\n", svo_last_error()); std::exit(1); }
        n_ok += rc == 1;
    }
    svo_destroy(ctx);
    double mean = 0; for (double v : us) mean += v; mean /= calls;
    std::sort(us.begin(), us.end());
    std::printf("%-28s mean %.1f us   median %.1f   p95 %.1f   min %.1f   (%d calls, %d poses ok, %d features into LK)\n", label, mean, us[calls / 2], us[calls * 95 / 100], us[0], calls, n_ok, st.n_into_lk);
    return us[calls / 2];
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: svo_latency frames.bin [win] [calls] [max_translation]\n"); return 2; }
    const int win = argc > 2 ? std::atoi(argv[2]) : 21, calls = argc > 3 ? std::atoi(argv[3]) : 200;
    const double max_t = argc > 4 ? std::atof(argv[4]) : 2.0;
    std::ifstream f(argv[1], std::ios::binary);
    int hdr[3];
    f.read((char*)hdr, sizeof(hdr));
    const int n = hdr[0], h = hdr[1], w = hdr[2];
    const size_t img = (size_t)w * h;
    std::vector<uint8_t> heap(img * 2 * n);
    f.read((char*)heap.data(), (std::streamsize)heap.size());
    if (!f || n < 2) { std::fprintf(stderr, "bad frame file\n"); return 2; }
    std::vector<const uint8_t*> L(n), R(n);
    for (int k = 0; k < n; k++) { L[k] = heap.data() + img * (2 * k); R[k] = heap.data() + img * (2 * k + 1); }
    run(L, R, w, h, win, calls, max_t, "heap buffers:");
    uint8_t* pin = (uint8_t*)svo_alloc_pinned(heap.size());
    if (pin) {
        std::memcpy(pin, heap.data(), heap.size());
        for (int k = 0; k < n; k++) { L[k] = pin + img * (2 * k); R[k] = pin + img * (2 * k + 1); }
        run(L, R, w, h, win, calls, max_t, "page-locked buffers:");
        svo_free_pinned(pin);
    }
    return 0;
}
