// svo_multi_gpu — BASELINE configs[3] from a C++ host, no Python: S independent stereo sequences sharded over the GPUs of the
// node (sequence s -> device s mod N), one host thread per GPU driving its sequences in lock-step through the
// C-ABI — the reference's driver loop (src/main.cpp:365-396: read pair, stereo_callback, frame_pose = frame_pose * T, write row)
// once per GPU — then ONE gather of the pose streams to device 0 over RCCL (include/svo_gather.h) and one result_seqNN.csv per
// sequence with the reference's columns (x,y,z,gtx,gty; main.cpp:346-348, 397-400).
//
//   svo_multi_gpu frames.bin n_devices [out_dir] [win=21] [max_translation=2.0]
//     frames.bin: int32 S, F, h, w; float fx, cx, cy, bf; then S x F x (left, right) gray images (tools/run_multi_gpu_cpp.py writes it)
//     prints "MULTI OK S sequences F frames N devices" and writes out_dir/result_seqNN.csv
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "svo.h"
#include "svo_gather.h"

static void matmul4(const double* a, const double* b, double* out) {
    double r[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += a[4 * i + k] * b[4 * k + j]; r[4 * i + j] = s; }
    std::memcpy(out, r, sizeof(r));
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: svo_multi_gpu frames.bin n_devices [out_dir] [win] [max_translation]\n"); return 2; }
    const int n_dev = std::atoi(argv[2]);
    const std::string out_dir = argc > 3 ? argv[3] : ".";
    const int win = argc > 4 ? std::atoi(argv[4]) : 21;
    const double max_t = argc > 5 ? std::atof(argv[5]) : 2.0;
    std::ifstream f(argv[1], std::ios::binary);
    int hdr[4]; float cal[4];
    f.read((char*)hdr, sizeof(hdr)); f.read((char*)cal, sizeof(cal));
    const int S = hdr[0], F = hdr[1], h = hdr[2], w = hdr[3];
    if (!f || S < 1 || F < 2 || n_dev < 1 || n_dev > svo_device_count()) { std::fprintf(stderr, "bad input (S=%d F=%d devices=%d of %d)\n", S, F, n_dev, svo_device_count()); return 2; }
    const size_t img = (size_t)w * h;
    std::vector<uint8_t> frames((size_t)S * F * 2 * img);
    f.read((char*)frames.data(), (std::streamsize)frames.size());
    if (!f) { std::fprintf(stderr, "short frame file\n"); return 2; }
    const float Pl[12] = {cal[0], 0, cal[1], 0, 0, cal[0], cal[2], 0, 0, 0, 1, 0};
    float Pr[12]; std::memcpy(Pr, Pl, sizeof(Pl)); Pr[3] = cal[3];

    // every device gets the same number of sequence slots (the gather's blocks are equal-sized); unused slots stay zero
    const int per_dev = (S + n_dev - 1) / n_dev;
    std::vector<double> local((size_t)n_dev * per_dev * F * SVO_POSE_STRIDE, 0.0), all(local.size(), 0.0);
    std::vector<int> status(n_dev, 0);
    std::vector<std::thread> workers;
    for (int d = 0; d < n_dev; d++) workers.emplace_back([&, d]() {
        std::vector<int> mine;
        for (int s = d; s < S; s += n_dev) mine.push_back(s);                          // sequence s -> device s mod N
        if (mine.empty()) return;
        const int B = (int)mine.size();
        svo_config cfg; svo_config_default(&cfg);
        cfg.win_w = cfg.win_h = win; cfg.max_translation_norm = max_t;
        svo_context* ctx = nullptr;
        if (svo_create(&cfg, d, B, w, h, &ctx) != SVO_OK || svo_set_projection(ctx, -1, Pl, Pr) != SVO_OK) { std::fprintf(stderr, "device %d: %s\n", d, svo_last_error()); status[d] = 1; return; }
        std::vector<const uint8_t*> L(B), R(B);
        std::vector<double> T((size_t)B * 16); std::vector<int> ok(B);
        for (int k = 0; k < F && !status[d]; k++) {
            for (int i = 0; i < B; i++) { const uint8_t* p = frames.data() + ((size_t)mine[i] * F + k) * 2 * img; L[i] = p; R[i] = p + img; }
            if (svo_process_batch(ctx, L.data(), R.data(), w, 0, T.data(), ok.data(), nullptr) != SVO_OK) { std::fprintf(stderr, "device %d: %s\n", d, svo_last_error()); status[d] = 1; break; }
            for (int i = 0; i < B; i++) {
                double* row = local.data() + (((size_t)d * per_dev + i) * F + k) * SVO_POSE_STRIDE;
                std::memcpy(row, T.data() + 16 * i, sizeof(double) * 16); row[16] = ok[i];
            }
        }
        svo_destroy(ctx);
    });
    for (auto& t : workers) t.join();
    for (int d = 0; d < n_dev; d++) if (status[d]) return 1;
    if (svo_gather_pose_streams(local.data(), per_dev, F, n_dev, all.data()) != 0) { std::fprintf(stderr, "gather: %s\n", svo_gather_last_error()); return 1; }
    // rank-0 work: integrate and write, sequence by sequence (device d, slot i holds sequence d + i * n_dev)
    int n_ok = 0;
    for (int s = 0; s < S; s++) {
        const int d = s % n_dev, i = s / n_dev;
        char name[64]; std::snprintf(name, sizeof(name), "/result_seq%02d.csv", s);
        std::ofstream res(out_dir + name);
        res << "x,y,z,gtx,gty\n";
        double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int k = 0; k < F; k++) {
            const double* row = all.data() + (((size_t)d * per_dev + i) * F + k) * SVO_POSE_STRIDE;
            matmul4(pose, row, pose);                                                   // applied even when !ok (main.cpp:394-396)
            char line[200]; std::snprintf(line, sizeof(line), "%.9g,%.9g,%.9g,0,0\n", pose[3], pose[7], pose[11]);
            res << line;
            n_ok += row[16] != 0.0;
        }
    }
    std::printf("MULTI OK %d sequences %d frames %d devices, %d poses ok\n", S, F, n_dev, n_ok);
    return 0;
}
