// host-only test of the ROS-free stereo synchroniser (include/svo/stereo_sync.hpp); no GPU needed.
#include <cstdio>
#include <cstdlib>
#include "svo/stereo_sync.hpp"
using namespace visual_odometry;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)
static StampedImage msg(double t) { StampedImage m; m.stamp = t; m.rows = 1; m.cols = 1; m.mono8 = {0}; return m; }
int main() {
    std::vector<std::pair<double, double>> got;
    StereoSynchronizer s([&](const StampedImage& l, const StampedImage& r) { got.push_back({l.stamp, r.stamp}); }, 10);
    // jittered 10 Hz streams, right lagging by 3 ms; one right frame lost
    for (int i = 0; i < 6; i++) {
        s.push_left(msg(0.1 * i));
        if (i != 3) s.push_right(msg(0.1 * i + 0.003));
    }
    s.push_left(msg(0.6)); s.push_right(msg(0.603));
    CHECK(got.size() >= 5);
    for (auto& p : got) CHECK(std::abs(p.first - p.second) < 0.05);           // never pairs across frames when a partner exists
    for (size_t i = 1; i < got.size(); i++) CHECK(got[i].first > got[i - 1].first);   // oldest first
    // bounded queues: a dead right camera must not grow the left queue past 10
    StereoSynchronizer d([&](const StampedImage&, const StampedImage&) {}, 10);
    for (int i = 0; i < 100; i++) d.push_left(msg(i));
    CHECK(d.dropped() == 90 && d.pairs_emitted() == 0);
    std::puts("SYNC OK");
    return 0;
}
