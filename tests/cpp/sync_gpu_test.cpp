// SURVEY.md §8 f-2 on the GPU: the binding of the reference's ROS node — ApproximateTime synchroniser (queue 10) ->
// VisualOdometry::stereo_callback (src/stereo_vo.cpp:53-62) — driven end to end through make_stereo_vo_callback and the HIP
// path.  Jittered, stamped mono8 frames (the right camera lags by 3 ms, arrives out of phase, and loses one frame) go
// through the synchroniser; the poses it produces must equal, bit for bit, direct stereo_callback calls on the pairs it formed.
//   usage: sync_gpu_test frames.bin     (frames.bin: int32 n, rows, cols, then n x [left rows*cols bytes, right rows*cols bytes])
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "svo/stereo_sync.hpp"
using namespace visual_odometry;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

int main(int argc, char** argv) {
    CHECK(argc == 2);
    FILE* f = std::fopen(argv[1], "rb");
    CHECK(f);
    int hdr[3];
    CHECK(std::fread(hdr, sizeof(int), 3, f) == 3);
    const int n = hdr[0], rows = hdr[1], cols = hdr[2];
    std::vector<StampedImage> L(n), R(n);
    for (int i = 0; i < n; i++) {
        for (StampedImage* m : {&L[i], &R[i]}) {
            m->rows = rows; m->cols = cols; m->mono8.resize((size_t)rows * cols);
            CHECK(std::fread(m->mono8.data(), 1, m->mono8.size(), f) == m->mono8.size());
        }
        L[i].stamp = 0.1 * i; R[i].stamp = 0.1 * i + 0.003;           // 10 Hz, right lagging by 3 ms
    }
    std::fclose(f);
    const Mat34f Pl = {718.856f, 0, cols / 2.f, 0, 0, 718.856f, rows / 2.f, 0, 0, 0, 1, 0};
    Mat34f Pr = Pl; Pr[3] = -386.1448f;
    svo_config cfg; svo_config_default(&cfg); cfg.max_translation_norm = 2.0;

    // ---- through the synchroniser
    VisualOdometry vo(cfg); vo.initalize_projection_matricies(Pl, Pr);
    struct Out { bool ok; Mat44 T; };
    std::vector<Out> via_sync;
    std::vector<std::pair<int, int>> pairs;
    StereoSynchronizer sync([&](const StampedImage& l, const StampedImage& r) {
        pairs.push_back({(int)std::lround(l.stamp / 0.1), (int)std::lround((r.stamp - 0.003) / 0.1)});
        auto out = vo.stereo_callback(l.view(), r.view());
        via_sync.push_back({out.first, out.second});
    }, 10);
    const int lost = 2;                                               // the right camera drops this frame
    for (int i = 0; i < n; i++) {                                     // right frame i arrives after left frame i + 1 (out of phase)
        sync.push_left(L[i]);
        if (i > 0 && i - 1 != lost) sync.push_right(R[i - 1]);
    }
    if (n - 1 != lost) sync.push_right(R[n - 1]);
    CHECK((int)pairs.size() >= n - 2);
    for (auto& p : pairs) CHECK(p.first == p.second && p.first != lost);   // only true stereo pairs, never across frames
    // the same through make_stereo_vo_callback (the stereo_vo.cpp:61-62 binding itself)
    VisualOdometry vo_b(cfg); vo_b.initalize_projection_matricies(Pl, Pr);
    std::vector<Out> via_bind;
    StereoSynchronizer bound = make_stereo_vo_callback(vo_b, [&](bool ok, const Mat44& T) { via_bind.push_back({ok, T}); }, 10);
    for (int i = 0; i < n; i++) {
        bound.push_left(L[i]);
        if (i > 0 && i - 1 != lost) bound.push_right(R[i - 1]);
    }
    if (n - 1 != lost) bound.push_right(R[n - 1]);
    // ---- direct calls on the pairs the synchroniser formed
    VisualOdometry direct(cfg); direct.initalize_projection_matricies(Pl, Pr);
    int n_ok = 0;
    CHECK(via_bind.size() == pairs.size() && via_sync.size() == pairs.size());
    for (size_t k = 0; k < pairs.size(); k++) {
        auto out = direct.stereo_callback(L[pairs[k].first].view(), R[pairs[k].second].view());
        CHECK(out.first == via_sync[k].ok && out.first == via_bind[k].ok);
        CHECK(std::memcmp(out.second.data(), via_sync[k].T.data(), sizeof(double) * 16) == 0);
        CHECK(std::memcmp(out.second.data(), via_bind[k].T.data(), sizeof(double) * 16) == 0);
        n_ok += out.first;
    }
    CHECK(n_ok >= (int)pairs.size() - 2);                             // frame 0 primes; the pair after the lost frame may fail a gate
    std::printf("SYNC GPU OK: %zu pairs, %d poses\n", pairs.size(), n_ok);
    return 0;
}
