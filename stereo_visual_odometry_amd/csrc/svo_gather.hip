// svo_gather.hip — libsvo_rccl.so: the pose-stream gather of include/svo_gather.h over RCCL (single process, all GPUs of a
// node; SURVEY.md §8e "one host thread per GPU under a single process (ncclCommInitAll) is enough" — here one thread issues the
// whole group, which RCCL allows between ncclGroupStart / ncclGroupEnd).  Not on the per-frame path; libsvo_hip.so does not
// link RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/svo_gather.h"

static thread_local std::string g_gerr;
extern "C" const char* svo_gather_last_error(void) { return g_gerr.c_str(); }

namespace {
struct Dev {
    int id = -1; hipStream_t st = nullptr; double* send = nullptr; ncclComm_t comm = nullptr;
};
struct Cleanup {
    std::vector<Dev>& devs; double*& recv0;
    ~Cleanup() {
        for (Dev& d : devs) {
            if (d.id < 0) continue;
            (void)hipSetDevice(d.id);
            if (d.st) (void)hipStreamSynchronize(d.st);
            if (d.comm) (void)ncclCommDestroy(d.comm);
            if (d.send) (void)hipFree(d.send);
            if (d.st) (void)hipStreamDestroy(d.st);
        }
        if (recv0) { (void)hipSetDevice(devs.empty() ? 0 : devs[0].id); (void)hipFree(recv0); }
    }
};
int fail(int code, const char* what, const char* detail) { char b[400]; snprintf(b, sizeof(b), "%s: %s", what, detail); g_gerr = b; return code; }
}  // namespace

#define GHIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) return fail(-2, #e, hipGetErrorString(_e)); } while (0)
#define GNCCL(e) do { ncclResult_t _r = (e); if (_r != ncclSuccess) return fail(-3, #e, ncclGetErrorString(_r)); } while (0)

extern "C" int svo_gather_pose_streams_ragged(const double* local, int n_seq, const int* frames_per_device, int n_devices, double* out) {
    if (!local || !out || !frames_per_device || n_seq < 1 || n_devices < 1) return fail(-1, "svo_gather_pose_streams", "bad arguments");
    int ndev = 0;
    GHIP(hipGetDeviceCount(&ndev));
    if (n_devices > ndev) return fail(-1, "svo_gather_pose_streams", "more devices asked for than this process sees");
    std::vector<size_t> count(n_devices), off(n_devices + 1, 0);
    for (int d = 0; d < n_devices; d++) {
        if (frames_per_device[d] < 0) return fail(-1, "svo_gather_pose_streams", "negative frame count");
        count[d] = (size_t)n_seq * (size_t)frames_per_device[d] * SVO_POSE_STRIDE;
        off[d + 1] = off[d] + count[d];
    }
    const size_t total = off[n_devices];
    if (total == 0) return 0;
    std::vector<Dev> devs(n_devices);
    double* recv0 = nullptr;
    Cleanup cleanup{devs, recv0};
    std::vector<int> ids(n_devices);
    std::vector<ncclComm_t> comms(n_devices);
    for (int d = 0; d < n_devices; d++) ids[d] = d;
    GNCCL(ncclCommInitAll(comms.data(), n_devices, ids.data()));
    for (int d = 0; d < n_devices; d++) { devs[d].id = d; devs[d].comm = comms[d]; }
    for (int d = 0; d < n_devices; d++) {
        GHIP(hipSetDevice(d));
        GHIP(hipStreamCreateWithFlags(&devs[d].st, hipStreamNonBlocking));
        GHIP(hipMalloc((void**)&devs[d].send, (count[d] ? count[d] : 1) * sizeof(double)));
        if (count[d]) GHIP(hipMemcpyAsync(devs[d].send, local + off[d], count[d] * sizeof(double), hipMemcpyHostToDevice, devs[d].st));
    }
    GHIP(hipSetDevice(0));
    GHIP(hipMalloc((void**)&recv0, total * sizeof(double)));
    // device 0's own block moves on the device; everyone else's crosses xGMI in ONE group (point-to-point: every sender has
    // its own link to device 0, so the transfers proceed side by side)
    if (count[0]) GHIP(hipMemcpyAsync(recv0, devs[0].send, count[0] * sizeof(double), hipMemcpyDeviceToDevice, devs[0].st));
    GNCCL(ncclGroupStart());
    for (int d = 1; d < n_devices; d++) {
        if (!count[d]) continue;
        GNCCL(ncclSend(devs[d].send, count[d], ncclDouble, 0, devs[d].comm, devs[d].st));
        GNCCL(ncclRecv(recv0 + off[d], count[d], ncclDouble, d, devs[0].comm, devs[0].st));
    }
    GNCCL(ncclGroupEnd());
    for (int d = 1; d < n_devices; d++) { GHIP(hipSetDevice(d)); GHIP(hipStreamSynchronize(devs[d].st)); }
    GHIP(hipSetDevice(0));
    GHIP(hipMemcpyAsync(out, recv0, total * sizeof(double), hipMemcpyDeviceToHost, devs[0].st));
    GHIP(hipStreamSynchronize(devs[0].st));
    return 0;
}

extern "C" int svo_gather_pose_streams(const double* local, int n_seq, int frames, int n_devices, double* out) {
    if (n_devices < 1 || frames < 0) return fail(-1, "svo_gather_pose_streams", "bad arguments");
    std::vector<int> f(n_devices, frames);
    return svo_gather_pose_streams_ragged(local, n_seq, f.data(), n_devices, out);
}
