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


def _world1_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    local = torch.from_numpy(np.stack([_pose_stream_for(s, 5) for s in range(3)]))
    out = sharding.gather_pose_streams(local, dst=0)                # a 1-rank group still runs the collective
    rag = sharding.gather_ragged_pose_streams([_pose_stream_for(0, 4), _pose_stream_for(1, 9)], dst=0)
    q.put((out[0].numpy(), {k: v for k, v in rag.items()}))
    dist.destroy_process_group()


def test_gather_world1_group_runs_collectives():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(port, q))
    p.start()
    out, rag = q.get(timeout=120)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert np.array_equal(out, np.stack([_pose_stream_for(s, 5) for s in range(3)]))
    assert np.array_equal(rag[0], _pose_stream_for(0, 4)) and np.array_equal(rag[1], _pose_stream_for(1, 9))


def _bench(args, env_extra):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_launch_contract_world_mismatch():
    """`--gpus N` is never silently a 1-GPU run: a WORLD_SIZE that disagrees is an error before anything touches a GPU."""
    r = _bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in r.stderr
    r = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_self_launch_starts_n_ranks():
    """Bare `python bench.py --gpus 2` starts 2 ranks as a child torch.distributed.run; without 2 GPUs the ranks fail
    loudly (non-zero exit relayed), they never fall back to one GPU."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 GPUs: the real run is bench.py's own business")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0
    assert ("needs a GPU" in r.stderr) or ("needs 2 GPUs" in r.stderr)
    assert '"n_gpus"' not in r.stdout
