/*
 * svo_gather.h — the multi-GPU exchange of the path as a C entry point (libsvo_rccl.so).
 *
 * The path shards across SEQUENCES (each VisualOdometry instance, reference include/vo.h:231-380, is an
 * independent state machine; frames inside one are serially dependent), one (batch of) sequence(s) per GPU, with no data-path
 * collective.  The ONLY exchange is the gather of the finished pose streams to one place.  bench.py / sharding.py do it with
 * torch.distributed (one process per GPU); this header is the same exchange for a C / C++ host that drives all GPUs of a node
 * from ONE process — the shape of the reference's own driver loop (src/main.cpp:365-396: one process, one loop, poses integrated
 * and written by the caller) extended to n_devices loops — over RCCL: ncclCommInitAll + one group of ncclSend / ncclRecv to
 * device 0 across xGMI.  Payload: 17 doubles per frame (the 4x4 of stereo_callback's pair.second + the pair.first flag);
 * KITTI-00's 4541 frames are 617 KB per sequence: latency-bound, the link rate is irrelevant.
 *
 * The reference has no counterpart (it has no multi-GPU code at all); nothing here is on the per-frame hot path, and
 * libsvo_hip.so does not depend on RCCL — this is a separate library so that a single-GPU integration never loads it.
 */
#ifndef SVO_GATHER_H
#define SVO_GATHER_H

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_POSE_STRIDE 17   /* doubles per frame: T (16, row-major 4x4) + ok flag */

/* Gathers the pose streams of n_devices GPUs to the caller.
 *   local : host array [n_devices][n_seq][frames][17] — block d is what device d produced (svo_collect writes T_out / ok_out to
 *           host memory; the caller packs them).  Block d is staged on device d, devices 1.. send to device 0 over RCCL, device 0
 *           receives into one buffer that is copied back.
 *   out   : host array [n_devices * n_seq][frames][17], sequence-major in device order.
 * Devices 0 .. n_devices-1 of this process are used.  n_devices = 1 is valid (a 1-rank communicator: no transfer, the same
 * code path otherwise).  Returns 0, or a negative value with svo_gather_last_error() describing the failure (-1 bad argument,
 * -2 HIP error, -3 RCCL error). */
int svo_gather_pose_streams(const double* local, int n_seq, int frames, int n_devices, double* out);

/* The ragged form (sequences of different lengths, e.g. KITTI 00-07): frames_per_device[d] frames for every sequence of
 * device d; local is the concatenation of the device blocks [n_seq][frames_per_device[d]][17], out likewise. */
int svo_gather_pose_streams_ragged(const double* local, int n_seq, const int* frames_per_device, int n_devices, double* out);

const char* svo_gather_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
