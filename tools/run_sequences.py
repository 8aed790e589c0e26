#!/usr/bin/env python3
"""BASELINE configs[3]: S independent stereo sequences of different lengths, sharded one-(or more)-per-GPU, pose
streams gathered to rank 0 over RCCL, one result_seqNN.csv per sequence (replicas-only fallback: every rank writes
its own files when --no-gather is given).

  python tools/run_sequences.py --lengths 40,12,45,9 --out /tmp/poses            # 1 GPU, 4 batched sequences
  python -m torch.distributed.run --nproc-per-node 8 tools/run_sequences.py --lengths 4541,1101,4661,801,271,2761,1101,1101

Sequences are synthetic KITTI-00-shaped renders (seed 0x5EED0040 + id); a sequence that has ended is fed its last
frame again (its outputs are discarded) so the batch keeps advancing in lock-step."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lengths", default="24,10,30,8")
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--out", default="")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from stereo_visual_odometry_amd import api, sharding, synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
    lengths = [int(x) for x in args.lengths.split(",")]
    mine = sharding.shard_sequences(len(lengths), rank, world)
    cal = dict(syn.KITTI00, width=args.width, height=args.height)
    if (args.width, args.height) != (1241, 376):
        cal["cx"], cal["cy"] = args.width / 2.0, args.height / 2.0
    Pl, Pr = syn.projection_matrices(cal)
    seqs = [syn.StereoSequence(cal=cal, n_frames=lengths[s], seed=0x5EED0040 + s, step=0.5, cell_px=17.6) for s in mine]
    streams = [np.zeros((lengths[s], 17)) for s in mine]
    if mine:
        vo = api.BatchVisualOdometry(args.width, args.height, len(mine), api.default_config(win_w=21, win_h=21, max_translation_norm=2.0), device=local_rank)
        vo.initalize_projection_matricies(Pl, Pr)
        for k in range(max(lengths[s] for s in mine)):
            L = [q.left[min(k, q.n_frames - 1)] for q in seqs]
            R = [q.right[min(k, q.n_frames - 1)] for q in seqs]
            ok, T = vo.stereo_callback_batch(L, R)
            for i, s in enumerate(mine):
                if k < lengths[s]:
                    streams[i][k, :16] = T[i].reshape(16); streams[i][k, 16] = ok[i]
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if args.no_gather or world == 1:
        result = {s: streams[i] for i, s in enumerate(mine)}
    else:
        result = sharding.gather_ragged_pose_streams(streams, dst=0, device=comm_dev)
    if result is not None:
        for s, rows in sorted(result.items()):
            T, ok = sharding.unpack_pose_stream(rows)
            poses = syn.integrate(T)
            print("sequence %d: %d frames, %d poses ok, end position %s" % (s, len(rows), int(ok.sum()), np.round(poses[-1][:3, 3], 3)))
            if args.out:
                os.makedirs(args.out, exist_ok=True)
                with open(os.path.join(args.out, "result_seq%02d.csv" % s), "w") as f:
                    f.write("x,y,z,ok\n")
                    for p, o in zip(poses, ok):
                        f.write("%.9g,%.9g,%.9g,%d\n" % (p[0, 3], p[1, 3], p[2, 3], int(o)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
