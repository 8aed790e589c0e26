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
    world = dist.get_world_size()
    if world == 1:
        return [local]
    rank = dist.get_rank()
    out = [local.new_empty(local.shape) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out
