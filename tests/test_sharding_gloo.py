"""The N > 1 path on CPU: sequences sharded one-per-rank, pose streams gathered to rank 0 (gloo, world_size 2)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stereo_visual_odometry_amd import sharding


def test_shard_sequences_partition():
    for world in (1, 2, 4, 8):
        owned = [sharding.shard_sequences(8, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(8))
        assert all(len(o) == 8 // world for o in owned)
    assert sharding.shard_sequences(8, 3, 8) == [3]


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    T = rng.normal(size=(5, 4, 4)); ok = np.array([1, 0, 1, 1, 0], bool)
    rows = sharding.pack_pose_stream(T, ok)
    assert rows.shape == (5, 17)
    T2, ok2 = sharding.unpack_pose_stream(rows)
    assert np.array_equal(T, T2) and np.array_equal(ok, ok2)


def _pose_stream_for(seq_id, frames):
    rng = np.random.default_rng(1000 + seq_id)
    return sharding.pack_pose_stream(rng.normal(size=(frames, 4, 4)), rng.random(frames) > 0.3)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = sharding.shard_sequences(4, rank, world)               # 4 sequences over 2 ranks
    local = torch.from_numpy(np.stack([_pose_stream_for(s, 6) for s in seqs]))
    out = sharding.gather_pose_streams(local, dst=0)
    if rank == 0:
        q.put([o.numpy() for o in out])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pose_streams_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(got) == 2
    for r in range(2):
        want = np.stack([_pose_stream_for(s, 6) for s in sharding.shard_sequences(4, r, 2)])
        assert np.array_equal(got[r], want)                       # every sequence's stream arrives intact, in rank order


def _ragged_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = [27, 11, 46, 8, 3]                                  # 5 sequences of different lengths over 2 ranks (3 + 2)
    mine = [_pose_stream_for(s, lengths[s]) for s in sharding.shard_sequences(5, rank, world)]
    out = sharding.gather_ragged_pose_streams(mine, dst=0)
    if rank == 0:
        q.put({k: v for k, v in out.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_ragged_pose_streams_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lengths = [27, 11, 46, 8, 3]
    assert sorted(got) == [0, 1, 2, 3, 4]
    for sid in range(5):
        assert got[sid].shape == (lengths[sid], 17)
        assert np.array_equal(got[sid], _pose_stream_for(sid, lengths[sid]))


def test_gather_ragged_single_process():
    out = sharding.gather_ragged_pose_streams([_pose_stream_for(0, 4), _pose_stream_for(1, 9)])
    assert out[0].shape == (4, 17) and out[1].shape == (9, 17)
