// The C entry of the multi-GPU exchange (include/svo_gather.h, libsvo_rccl.so) on real hardware: a communicator over the devices
// this process sees (ONE on the test box — RCCL accepts a 1-rank communicator, and the call runs the same staging, group and
// copy-back code) gathers packed pose streams; equal-length and ragged forms; argument errors are reported, not crashes.
//   gather_gpu_test [n_devices]      prints "GATHER OK <n_devices>"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "svo_gather.h"

static double value(int dev, int seq, int frame, int k) { return dev * 1e6 + seq * 1e3 + frame + k / 32.0; }

int main(int argc, char** argv) {
    const int nd = argc > 1 ? std::atoi(argv[1]) : 1;
    const int n_seq = 3, frames = 5;
    std::vector<double> local((size_t)nd * n_seq * frames * SVO_POSE_STRIDE), out(local.size(), -1.0);
    size_t i = 0;
    for (int d = 0; d < nd; d++) for (int s = 0; s < n_seq; s++) for (int f = 0; f < frames; f++) for (int k = 0; k < SVO_POSE_STRIDE; k++) local[i++] = value(d, s, f, k);
    if (svo_gather_pose_streams(local.data(), n_seq, frames, nd, out.data()) != 0) { std::fprintf(stderr, "gather failed: %s\n", svo_gather_last_error()); return 1; }
    for (size_t j = 0; j < local.size(); j++) if (out[j] != local[j]) { std::fprintf(stderr, "mismatch at %zu\n", j); return 1; }
    // ragged: device d holds d + 2 frames per sequence (and one device with none when there are several)
    std::vector<int> fpd(nd);
    size_t total = 0;
    for (int d = 0; d < nd; d++) { fpd[d] = (nd > 2 && d == 1) ? 0 : d + 2; total += (size_t)n_seq * fpd[d] * SVO_POSE_STRIDE; }
    std::vector<double> rl(total), ro(total, -1.0);
    i = 0;
    for (int d = 0; d < nd; d++) for (int s = 0; s < n_seq; s++) for (int f = 0; f < fpd[d]; f++) for (int k = 0; k < SVO_POSE_STRIDE; k++) rl[i++] = value(d, s, f, k) + 0.5;
    if (svo_gather_pose_streams_ragged(rl.data(), n_seq, fpd.data(), nd, ro.data()) != 0) { std::fprintf(stderr, "ragged gather failed: %s\n", svo_gather_last_error()); return 1; }
    for (size_t j = 0; j < total; j++) if (ro[j] != rl[j]) { std::fprintf(stderr, "ragged mismatch at %zu\n", j); return 1; }
    // errors
    if (svo_gather_pose_streams(nullptr, n_seq, frames, nd, out.data()) != -1) return 1;
    if (svo_gather_pose_streams(local.data(), 0, frames, nd, out.data()) != -1) return 1;
    if (svo_gather_pose_streams(local.data(), n_seq, frames, 1000, out.data()) != -1) return 1;
    if (!svo_gather_last_error()[0]) return 1;
    std::printf("GATHER OK %d\n", nd);
    return 0;
}
