"""Multi-GPU layout of the path: independent stereo sequences are sharded across ranks (one process per
GPU); nothing is exchanged while frames are processed.  The only collective is the gather of the
per-sequence pose streams to rank 0 at the end (RCCL over xGMI when the backend is "nccl", gloo on CPU).

The reference has no counterpart (single process, SURVEY.md §2 "no collectives"); this is new design
mandated by BASELINE.json north_star / SURVEY.md §8e.
"""
import numpy as np


def shard_sequences(n_sequences, rank, world_size):
    """Sequence ids owned by `rank`: sequence s lives on GPU s mod world_size."""
    return [s for s in range(n_sequences) if s % world_size == rank]


def pack_pose_stream(transforms, oks):
    """[(frames, 4, 4) f64, (frames,) bool] -> (frames, 17) f64 rows = 16 pose entries + ok flag (136 B / frame)."""
    T = np.asarray(transforms, np.float64).reshape(len(oks), 16)
    return np.concatenate([T, np.asarray(oks, np.float64).reshape(-1, 1)], 1)


def unpack_pose_stream(rows):
    rows = np.asarray(rows, np.float64)
    return rows[:, :16].reshape(-1, 4, 4), rows[:, 16] > 0.5


def gather_pose_streams(local, dst=0):
    """local: torch tensor (n_local_seq, frames, 17) f64 on this rank's device (cuda for nccl, cpu for gloo).
    Returns on rank dst a list (len world_size) of tensors, None elsewhere.  Equal shapes on all ranks
    (weak scaling: every rank owns the same number of sequences and frames)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return [local]                       # single process, no process group: nothing to exchange
    world = dist.get_world_size()            # a 1-rank group still goes through the collective (RCCL rehearsal on one GPU)
    rank = dist.get_rank()
    out = [local.new_empty(local.shape) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out


def gather_ragged_pose_streams(streams, dst=0, device=None):
    """Sequences of different lengths (BASELINE configs[3]: KITTI 00-07 have 271 .. 4661 frames).
    streams: list of (frames_i, 17) float64 numpy arrays owned by this rank (may be empty).
    Every rank pads its streams to the global maximum length, one gather moves the padded block plus the true
    lengths to rank dst, which strips the padding.  Returns on dst {sequence_id: (frames, 17) array} using the
    s = rank + k * world_size ownership of shard_sequences; None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    dev = device if device is not None else torch.device("cpu")
    n_local = torch.tensor([len(streams)], dtype=torch.int64, device=dev)
    max_len = torch.tensor([max([len(x) for x in streams], default=0)], dtype=torch.int64, device=dev)
    grouped = dist.is_initialized()                              # a 1-rank group still runs every collective
    if grouped:
        dist.all_reduce(n_local, op=dist.ReduceOp.MAX)           # ranks may own different numbers of sequences
        dist.all_reduce(max_len, op=dist.ReduceOp.MAX)
    S, F = int(n_local.item()), int(max_len.item())
    block = np.zeros((S, F, 17)); lens = np.full(S, -1, np.int64)
    for k, x in enumerate(streams):
        block[k, :len(x)] = x
        lens[k] = len(x)
    tb = torch.from_numpy(block).to(dev); tl = torch.from_numpy(lens).to(dev)
    if grouped:
        ob = [torch.empty_like(tb) for _ in range(world)] if rank == dst else None
        ol = [torch.empty_like(tl) for _ in range(world)] if rank == dst else None
        dist.gather(tb, ob, dst=dst)
        dist.gather(tl, ol, dst=dst)
    else:
        ob, ol = [tb], [tl]
    if rank != dst:
        return None
    out = {}
    for r in range(world):
        b, l = ob[r].cpu().numpy(), ol[r].cpu().numpy()
        for k in range(S):
            if l[k] >= 0:
                out[r + k * world] = b[k, :l[k]].copy()
    return out
