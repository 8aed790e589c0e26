#!/usr/bin/env python3
"""bench.py — stereo frame-pairs/s of the HIP front end on KITTI-00-shaped synthetic input.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  A step = one frame pair for each of the --seqs
sequences resident on a GPU (inputs already in HBM).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)


def host_cores():
    """CPU cores this process may really use: the cgroup CPU quota if one is set (a GPU box exposes all of the host's
    logical CPUs but grants a share of them), else the affinity mask; capped at 64."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]             # cgroup v2
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def opencv_probe(seq, cal, over, ping_pong, frames):
    """BASELINE.md §3.2 / SURVEY §8d secondary CPU baseline: if THIS box has OpenCV 4 (never assumed, never installed), build
    tools/opencv_baseline.cpp — own code calling the same seven cv:: functions with the reference's parameters — time it with
    default threads and with one, and report it; otherwise say what was probed."""
    import glob
    import shutil
    import subprocess
    import tempfile
    probed, cflags, libs = [], None, None
    if shutil.which("pkg-config"):
        probed.append("pkg-config opencv4")
        if subprocess.run(["pkg-config", "--exists", "opencv4"]).returncode == 0:
            cflags = subprocess.check_output(["pkg-config", "--cflags", "opencv4"], text=True).split()
            libs = subprocess.check_output(["pkg-config", "--libs", "opencv4"], text=True).split()
    if cflags is None:
        probed.append("ldconfig -p libopencv_video")
        try:
            have = "libopencv_video" in subprocess.run(["ldconfig", "-p"], capture_output=True, text=True).stdout
        except Exception:
            have = False
        probed.append("/usr/include/opencv4, /usr/local/include/opencv4")
        inc = [d for d in ("/usr/include/opencv4", "/usr/local/include/opencv4") if os.path.isdir(os.path.join(d, "opencv2"))]
        if not have:
            have = bool(glob.glob("/usr/lib/*/libopencv_video.so*") + glob.glob("/usr/local/lib/libopencv_video.so*"))
        if have and inc:
            cflags = ["-I" + inc[0]]
            libs = ["-lopencv_calib3d", "-lopencv_video", "-lopencv_features2d", "-lopencv_imgproc", "-lopencv_core"]
    if cflags is None:
        return "not present (probed: %s)" % ", ".join(probed)
    try:
        tmp = tempfile.mkdtemp(prefix="svo_ocv_")
        exe = os.path.join(tmp, "opencv_baseline")
        subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", os.path.join(ROOT, "tools", "opencv_baseline.cpp"), "-o", exe] + cflags + libs,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        raw = os.path.join(tmp, "frames.raw")
        order = [0] + [ping_pong(i) for i in range(1, frames + 1)]
        with open(raw, "wb") as f:
            for k in order:
                f.write(np.ascontiguousarray(seq.left[k]).tobytes()); f.write(np.ascontiguousarray(seq.right[k]).tobytes())
        out = {}
        for label, threads in (("default_threads", 0), ("one_thread", 1)):
            r = subprocess.run([exe, raw, str(cal["width"]), str(cal["height"]), str(len(order)), str(cal["fx"]), str(cal["cx"]), str(cal["cy"]), str(cal["bf"]),
                                str(over["win_w"]), str(over["max_level"]), str(over["ransac_iterations"]), str(over["max_translation_norm"]), str(threads),
                                os.path.join(tmp, "poses_%s.txt" % label)], capture_output=True, text=True, timeout=600)
            out[label] = r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "failed: " + r.stderr[-200:]
        return {"present": True, "found_with": probed[-1], **out}
    except Exception as e:                                     # an OpenCV that does not build / link is reported, not fatal
        return "present but tools/opencv_baseline.cpp did not build or run: %r (probed: %s)" % (e, ", ".join(probed))


def level_sizes(W, H, win, max_level):
    out = []
    w, h = W, H
    for l in range(max_level + 1):
        out.append(w * h)
        w, h = (w + 1) // 2, (h + 1) // 2
        if w <= win or h <= win:
            break
    return out


def algorithmic_bytes(W, H, N, win, max_level, K, r=4):
    """SURVEY.md §8(d) / BASELINE.md §4 byte model per frame pair -> (total, lk_chain_only)."""
    px = level_sizes(W, H, win, max_level)
    L = len(px) - 1
    ingest = 2 * px[0]
    pyr = 2 * (sum(px[:L]) + sum(px[1:]))
    fast = px[0] + 16 * N
    lk = 4 * (sum(min(N * ((win + 3) ** 2 + (win + 1 + 2 * r) ** 2), 2 * p) for p in px) + 17 * N)
    geo = 49 * N + 48 * K
    return ingest + pyr + fast + lk + geo, lk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--seqs", type=int, default=512, help="independent stereo sequences batched per GPU (two contexts of 256: the per-frame chain of small kernels between two LK launches is latency-bound and costs the same for 128 or 256 sequences, so larger batches dilute it — 256 / 512 / 1024 sequences per GPU: 17.8k / 18.2k / 18.3k frame-pairs/s)")
    ap.add_argument("--depth", type=int, default=4, help="frames kept in flight per context (<= 8)")
    ap.add_argument("--contexts", type=int, default=2, help="sequence groups per GPU, each on its own HIP stream (their kernels overlap)")
    ap.add_argument("--pool", type=int, default=16, help="distinct synthetic sequences rendered per rank (own seed each)")
    ap.add_argument("--movers", type=float, default=None, help="fraction of the pixels covered by an independently moving foreground layer "
                    "(RANSAC-PnP outliers; 0 = static scene, the best case for PnP).  Default: 0.3 for cfg2 (the metric's workload), 0 for the others")
    ap.add_argument("--frames", type=int, default=10, help="frames rendered per pool sequence (ping-pong replay)")
    ap.add_argument("--cpu-frames", type=int, default=120, help="frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--ate-frames", type=int, default=40, help="frame transitions of the ATE leg: one freshly rendered sequence (not a ping-pong "
                    "replay) through a one-sequence context and through the CPU oracle, outside the timed region (0 = only the timed steps of slot 0)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg5"],
                    help="BASELINE.json configs[1] (the metric's configuration, default) / configs[2] / configs[4]; the others are extra measurements")
    ap.add_argument("--float-sums", type=int, default=0, help="1 = svo_config.lk_float_sums (LK sums in float in OpenCV's SIMD128 lane order: the mode that "
                    "reproduces the reference's recording digit for digit; several times slower in LK).  An extra measurement, not the bench line")
    ap.add_argument("--float-sums-steps", type=int, default=10, help="steps of the extra leg that re-times the SAME workload with lk_float_sums = 1 (the mode whose "
                    "output is the reference's own rounding) on rank 0 after the timed region -> float_sums_mode in the line; skipped when 0, when the line itself "
                    "is a float-sums run, and when --cpu-frames is 0 (the profiling runs: their counter passes must only see the default kernels)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    # ---- launch contract.  N > 1 is one process per GPU.  The driver normally starts the ranks itself
    # (python -m torch.distributed.run ... bench.py --gpus N): then WORLD_SIZE is set and must equal --gpus.  When
    # `python bench.py --gpus N` is run bare, this process starts the N ranks as a FRESH CHILD (never an exec, and before
    # torch or HIP has been touched here), relays the child's output (rank 0's JSON line) and exits with its status.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU: python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d, or run `python bench.py --gpus %d` bare)"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))

    import torch
    import torch.distributed as dist
    from stereo_visual_odometry_amd import api, sharding, synthetic as syn

    if args.same_device:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if not args.same_device and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d needs %d GPUs on this node, %d visible" % (world, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # a process group exists for N > 1, and for a 1-rank rehearsal of the exchange on one GPU (SVO_BENCH_FORCE_GROUP=1:
    # init, warm-up gather, timed gather and the all-reduces all go through RCCL with a single rank)
    grouped = world > 1 or os.environ.get("SVO_BENCH_FORCE_GROUP", "") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("WORLD_SIZE", str(world)); os.environ.setdefault("RANK", str(rank))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")     # where collective payloads live

    # ---- workload: BASELINE.json configs[1] — KITTI-00 shaped 1241x376, ~2000 FAST features, LK 21x21, maxLevel 3
    win = int(os.environ.get("SVO_BENCH_WIN", "21"))
    WL = {   # calibration, scene parameters, config overrides, description
        "cfg2": (syn.KITTI00, dict(seed=0x5EED0002, step=0.5, cell_px=float(os.environ.get("SVO_BENCH_CELL", "16.6"))), dict(max_level=3, ransac_iterations=100),
                 "BASELINE configs[1]: KITTI-00 calibration, 1241x376, LK 21x21 win, maxLevel 3, 100 RANSAC-PnP iterations"),
        "cfg3": (syn.KITTI00, dict(seed=0x5EED0003, step=0.5, cell_px=12.0), dict(max_level=4, ransac_iterations=1000),
                 "BASELINE configs[2]: KITTI-00 calibration, 1241x376, ~4000 features, LK 21x21 win, maxLevel 4, 1000 RANSAC-PnP iterations"),
        "cfg5": (syn.ZED, dict(seed=0x5EED0005, step=0.2, cell_px=14.0, depth=(6.0, 40.0)), dict(max_level=3, ransac_iterations=100),
                 "BASELINE configs[4]: ZED calibration, 1920x1080, ~8000 features, LK 21x21 win, maxLevel 3, 100 RANSAC-PnP iterations"),
    }
    cal, scene, cfg_over, workload_name = WL[args.workload]
    if args.movers is None:
        args.movers = 0.3 if args.workload == "cfg2" else 0.0
    W, H = cal["width"], cal["height"]
    over = dict(win_w=win, win_h=win, max_translation_norm=2.0, **cfg_over)
    if args.float_sums:
        over["lk_float_sums"] = 1
    B, F = args.seqs, args.frames
    seed0 = scene.pop("seed")
    pool = [syn.StereoSequence(cal=cal, n_frames=F, seed=seed0 + 97 * rank + g, movers=args.movers, **scene)
            for g in range(args.pool)]
    left = torch.stack([torch.from_numpy(np.stack(s.left)) for s in pool]).to(dev)      # (G, F, H, W) u8, resident in HBM
    right = torch.stack([torch.from_numpy(np.stack(s.right)) for s in pool]).to(dev)
    frame_bytes = W * H

    def ping_pong(i):
        p = i % (2 * F - 2)
        return p if p < F else 2 * F - 2 - p

    C = max(1, min(args.contexts, B))
    while B % C:
        C -= 1
    Bc = B // C

    def ptrs(step, c):
        lp, rp = [], []
        for b in range(c * Bc, (c + 1) * Bc):
            g = b % args.pool
            f = ping_pong(step + (b // args.pool) * 3)       # phase offsets: every slot sees a different frame stream
            lp.append(left.data_ptr() + (g * F + f) * frame_bytes)
            rp.append(right.data_ptr() + (g * F + f) * frame_bytes)
        return lp, rp

    Pl, Pr = syn.projection_matrices(cal)
    os.environ.setdefault("SVO_GRAPH", "0")      # the roofline needs the LK kernel's own HIP events: launch-list mode even for tiny --seqs
    vos = []
    for c in range(C):
        v = api.BatchVisualOdometry(W, H, Bc, api.default_config(**over), device=local_rank)
        v.initalize_projection_matricies(Pl, Pr)
        v.set_stage_timing(True)                 # the roofline needs the LK kernel's own HIP events
        vos.append(v)

    total = args.warmup + args.steps
    depth = max(1, min(args.depth, 8))
    poses = np.zeros((B, args.steps, 17))
    n_lk, n_ok, lk_ms, fr_ms = [], 0, [], []
    n_bounds, n_inl, n_iters, n_vis, n_stp, n_dead = [], [], [], [], [], []

    def run(first, count, record):
        nonlocal n_ok
        sub = col = 0
        while col < count:
            while sub < count and sub - col < depth:
                for c, vo in enumerate(vos):
                    lp, rp = ptrs(first + sub, c)
                    vo.submit_device(lp, rp, W)
                sub += 1
            for c, vo in enumerate(vos):
                ok, T = vo.collect()
                if record:
                    a, b = vo.last_timing()                  # HIP events on the context's own stream
                    lk_ms.append(a); fr_ms.append(b)
                    n_lk.append(np.mean([s.n_into_lk for s in vo.stats]))
                    n_bounds.append(np.mean([s.n_after_bounds for s in vo.stats]))
                    n_inl.append(np.mean([s.n_inliers for s in vo.stats]))
                    n_iters.append(np.mean([s.ransac_iters for s in vo.stats]))
                    n_vis.append(np.mean([s.lk_level_visits for s in vo.stats]))
                    n_stp.append(np.mean([s.lk_newton_steps for s in vo.stats]))
                    n_dead.append(np.mean([[s.lk_dead_after_pass0, s.lk_dead_after_pass1, s.lk_dead_after_pass2] for s in vo.stats], axis=0))
                    n_ok += int(ok.sum())
                    poses[c * Bc:(c + 1) * Bc, col, :16] = T.reshape(Bc, 16); poses[c * Bc:(c + 1) * Bc, col, 16] = ok
            col += 1

    run(0, args.warmup + 1, False)                           # frame 0 only primes the pipeline (vo.cpp:47-56), then W warm-up steps
    if grouped:
        # warm-up of the exchange as well: the first gather / all-reduce set up RCCL's point-to-point channels
        sharding.gather_pose_streams(torch.zeros(poses.shape, dtype=torch.float64, device=comm_dev), dst=0)
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=comm_dev), op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.warmup + 1, args.steps, True)
    if grouped:                                              # the path's only exchange: pose streams -> rank 0 (RCCL over xGMI)
        gathered = sharding.gather_pose_streams(torch.from_numpy(poses).to(comm_dev), dst=0)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    dt = dt_local = time.perf_counter() - t0
    if grouped:
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        okt = torch.tensor([n_ok], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(okt)
        n_ok_all = int(okt.item())
    else:
        n_ok_all = n_ok
    # the dominant kernel's duration and every rank's own rate, so that an N > 1 line shows its slowest rank (a straggler GPU is
    # what a scaling run is there to expose): max over ranks of the mean LK launch duration, and each rank's frame-pairs/s
    lk_avg_ms = float(np.mean(lk_ms))
    per_rank_value = [B * args.steps / dt_local]
    lk_avg_ms_ranks = [lk_avg_ms]
    if grouped:
        mine = torch.tensor([B * args.steps / dt_local, lk_avg_ms], dtype=torch.float64, device=comm_dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank_value = [float(v[0].item()) for v in allv]
        lk_avg_ms_ranks = [float(v[1].item()) for v in allv]
        lk_avg_ms = max(lk_avg_ms_ranks)
        if rank == 0:                                         # rank 0 now holds every sequence's pose stream
            assert len(gathered) == world and all(tuple(g.shape) == (B, args.steps, 17) for g in gathered)
            assert np.array_equal(gathered[0].cpu().numpy(), poses)
        dist.barrier()
        dist.destroy_process_group()                          # everything below is rank 0's own work (CPU baseline, ATE): no rank waits for it

    if rank == 0:
        N = float(np.mean(n_lk))
        bytes_total, bytes_lk = algorithmic_bytes(W, H, N, win, over["max_level"], over["ransac_iterations"])
        achieved = bytes_lk * Bc / (lk_avg_ms * 1e-3) / 1e9    # algorithmic GB/s of the dominant kernel: one launch covers Bc sequences
        value = world * B * args.steps / dt
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
        # separate runs of this same script, profiles/summarize.py), rescaled to this run's sequences per launch
        traffic = None; traffic_src = None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_lk_chain_pmc.json")))
        profiled = args.workload == "cfg2" and win == 21 and abs(args.movers - 0.3) < 1e-9   # the committed counter passes are of this workload's kernel on this scene
        if pmcs and profiled:
            try:
                j = json.load(open(pmcs[-1]))
                traffic = j["hbm_bytes_per_launch"] * Bc / float(j.get("sequences_per_launch", 32))
                ps = int(j.get("sequences_per_launch", 32))
                traffic_src = "committed rocprofv3 --pmc passes of this same command (FETCH_SIZE, WRITE_SIZE in separate passes), profiles/%s, %s" % (
                    os.path.basename(pmcs[-1]), "taken at this launch size (%d sequences per launch)" % ps if ps == Bc else
                    "rescaled from %d to %d sequences per launch" % (ps, Bc))
            except Exception:
                traffic = None
        # secondary (SURVEY.md 8d asks for the VALU view too, the kernel being instruction-bound): share of the GPU's VALU
        # issue slots (1024 SIMDs, one wave64 instruction per 4 cycles, 2.4 GHz nominal) the LK instructions of this run take,
        # from the committed SQ_INSTS_VALU pass
        valu = None
        sqs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_lk_chain_sq.json")))
        if sqs and profiled:
            try:
                ipf = float(json.load(open(sqs[-1]))["valu_instructions_per_feature"])
                valu = {"lk_valu_instructions_per_feature": ipf,
                        "issue_slot_frac_at_2.4GHz": value / world * N * ipf / (1024 * 2.4e9 / 4)}
            except Exception:
                valu = None
        # the LK flop model of SURVEY.md 8d with the MEASURED work terms: 24 w^2 per level visit that builds a template (bilinear
        # I / Ix / Iy patches, normal matrix) + 10 w^2 per Newton step, against the f32 VALU peak (157.3 TFLOP/s)
        vis, stp = float(np.mean(n_vis)), float(np.mean(n_stp))
        lk_flops = win * win * (24.0 * vis + 10.0 * stp)
        valu_flop = {"lk_flops_per_frame_pair": lk_flops, "level_visits_per_feature": vis / max(N, 1.0), "newton_steps_per_feature": stp / max(N, 1.0),
                     "peak_TFLOPs": 157.3, "frac_at_job_rate": lk_flops * value / world / 157.3e12,
                     "frac_in_kernel": lk_flops * Bc / (lk_avg_ms * 1e-3) / 157.3e12}
        cpu = None
        if args.cpu_frames > 0:                               # rank 0, any N: the timed region and the last barrier are behind us
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            # the CPU number is taken with -O3 -march=native, so that build is made HERE, on the machine that
            # times it (never shipped: a -march=native object from another host may not even run); same IEEE-strict flags
            # as the checker build, hence the same results
            import subprocess
            orc_build = "-O3 -march=native"
            try:
                subprocess.check_call(["make", "-B", "-C", os.path.join(ROOT, "oracle"), "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                os.environ["SVO_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "libsvo_oracle_native.so")
            except Exception:
                orc_build = "-O2 (the portable checker build: no compiler on this box for the native one)"
            import oracle_lib as orc

            cpu_T = {}                                        # step -> 4x4 the oracle returned (slot 0 sees the same frame stream)

            def cpu_rate(threads, frames):
                used = orc.set_threads(threads)
                o = orc.VisualOdometry(orc.default_config(**over)); o.initalize_projection_matricies(Pl, Pr)
                sq = pool[0]
                o.stereo_callback(sq.left[0], sq.right[0])
                c0 = time.perf_counter()
                for i in range(1, frames + 1):
                    f = ping_pong(i)
                    cpu_T[i] = o.stereo_callback(sq.left[f], sq.right[f])[1]
                return frames / (time.perf_counter() - c0), used

            single, _ = cpu_rate(1, max(4, args.cpu_frames // 3))
            allc, cores = cpu_rate(host_cores(), args.cpu_frames)   # every core this process is granted (OpenMP over LK points / image rows)
            orc.set_threads(1)
            cpu = {"value": allc, "unit": "frame-pairs/s", "cores": cores, "kind": "port", "opencv": opencv_probe(pool[0], cal, over, ping_pong, args.cpu_frames),
                   "sample": "%d frame pairs of pool sequence 0 of the same workload through oracle/ — this repo's plain-C RESTATEMENT of the "
                             "reference's pipeline and of the OpenCV 4.5 calls it makes, NOT OpenCV itself (absent from the image) — built %s, "
                             "OpenMP over the points of each LK pass and over image rows (the LK inner loops are auto-vectorised by gcc), %d threads; single thread: %.2f frame-pairs/s"
                             % (args.cpu_frames, orc_build, cores, single)}
        # the same workload in the mode that reproduces the reference's recording digit for digit (svo_config.lk_float_sums: LK
        # sums in float in OpenCV's SIMD128 lane order) — an extra figure next to `value`, never `value`: rank 0,
        # its own GPU, fresh contexts, same sequences / contexts / frames in flight, after everything timed above
        fs_leg = None
        if args.float_sums_steps > 0 and args.cpu_frames > 0 and not args.float_sums:
            for v in vos:
                v.close()
            fvos = []
            for c in range(C):
                v = api.BatchVisualOdometry(W, H, Bc, api.default_config(**dict(over, lk_float_sums=1)), device=local_rank)
                v.initalize_projection_matricies(Pl, Pr); v.set_stage_timing(True); fvos.append(v)
            fs_lk, fs_ok = [], []

            def frun(first, count, record):
                sub = col = 0
                while col < count:
                    while sub < count and sub - col < depth:
                        for c, vo in enumerate(fvos):
                            lp, rp = ptrs(first + sub, c)
                            vo.submit_device(lp, rp, W)
                        sub += 1
                    for vo in fvos:
                        ok, _ = vo.collect()
                        if record:
                            fs_lk.append(vo.last_timing()[0]); fs_ok.append(float(ok.mean()))
                    col += 1

            frun(0, 3, False)
            torch.cuda.synchronize()
            f0 = time.perf_counter()
            frun(3, args.float_sums_steps, True)
            torch.cuda.synchronize()
            fdt = time.perf_counter() - f0
            fs_leg = {"value": B * args.float_sums_steps / fdt, "unit": "frame-pairs/s", "n_gpus": 1, "steps": args.float_sums_steps, "warmup": 2,
                      "ms_per_step": fdt / args.float_sums_steps * 1e3, "lk_kernel_avg_ms": float(np.mean(fs_lk)), "pose_ok_fraction": float(np.mean(fs_ok)),
                      "what": "the same workload, sequences per GPU and contexts with svo_config.lk_float_sums = 1 (LK normal equations summed in float in "
                              "the lane order of OpenCV's SIMD128 code: the mode in which the oracle prints 127 of the 128 rows of the reference's run1/result.csv "
                              "digit for digit and the HIP path equals the oracle bit for bit); rank 0's GPU, outside the timed region; `value` above is the "
                              "default exact-integer mode"}
            for v in fvos:
                v.close()
        # ATE (the second half of BASELINE.json's metric), outside the timed region: slot 0's pose stream over the timed steps,
        # integrated as frame_pose = frame_pose * T (main.cpp:396), against the renderer's ground truth for the same frame
        # transitions and against the CPU oracle on the steps its bounded sample covers
        first = args.warmup + 1
        est = [poses[0, c, :16].reshape(4, 4) for c in range(args.steps)]
        gt = [np.linalg.inv(pool[0].poses[ping_pong(first + c - 1)]) @ pool[0].poses[ping_pong(first + c)] for c in range(args.steps)]
        ate = {"vs_ground_truth_m": syn.ate_rmse(syn.integrate(est), syn.integrate(gt)), "frames": args.steps,
               "path_length_m": float(sum(np.linalg.norm(g[:3, 3]) for g in gt)), "vs_cpu_oracle_m": None, "oracle_frames": 0}
        if cpu is not None:
            common = [c for c in range(args.steps) if (first + c) in cpu_T]
            if common:
                ate["vs_cpu_oracle_m"] = syn.ate_rmse(syn.integrate([est[c] for c in common]), syn.integrate([cpu_T[first + c] for c in common]))
                ate["oracle_frames"] = len(common)
        if args.ate_frames > 0:
            # the metric's "ATE vs ref" on a real path: one freshly rendered sequence of ate_frames + 1 frames (forward motion, no
            # ping-pong), through a one-sequence HIP context and, frame for frame, through the CPU oracle
            sq = syn.StereoSequence(cal=cal, n_frames=args.ate_frames + 1, seed=seed0 + 7919, movers=args.movers, **scene)
            g1 = api.BatchVisualOdometry(W, H, 1, api.default_config(**over), device=local_rank); g1.initalize_projection_matricies(Pl, Pr)
            o1 = None
            if cpu is not None:
                orc.set_threads(host_cores())
                o1 = orc.VisualOdometry(orc.default_config(**over)); o1.initalize_projection_matricies(Pl, Pr)
            est_l, orc_l, n_same = [], [], 0
            for k in range(args.ate_frames + 1):
                okg, Tg = g1.stereo_callback_batch([sq.left[k]], [sq.right[k]])
                if o1 is not None:
                    oko, To = o1.stereo_callback(sq.left[k], sq.right[k])
                    n_same += int(bool(okg[0]) == oko and {f[0]: getattr(o1.stats, f[0]) for f in o1.stats._fields_} == g1.stats[0].as_dict())
                    if k: orc_l.append(To)
                if k: est_l.append(Tg[0])
            if o1 is not None:
                orc.set_threads(1)
            gt_l = [np.linalg.inv(sq.poses[k - 1]) @ sq.poses[k] for k in range(1, args.ate_frames + 1)]
            ate["long_run"] = {"frames": args.ate_frames, "path_length_m": float(sum(np.linalg.norm(g[:3, 3]) for g in gt_l)),
                               "vs_ground_truth_m": syn.ate_rmse(syn.integrate(est_l), syn.integrate(gt_l)),
                               "vs_cpu_oracle_m": syn.ate_rmse(syn.integrate(est_l), syn.integrate(orc_l)) if orc_l else None,
                               "frames_with_identical_flags_and_counters": n_same if o1 is not None else None}
        dead = np.mean(n_dead, axis=0) if n_dead else np.zeros(3)
        out = {
            "metric": "stereo frame-pairs/sec on KITTI-00 1241x376 @2k feats; ATE vs ref", "value": value, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int64/f32 (LK), f64 (PnP)",
            "data": "synthetic",
            "config": {"workload": workload_name + ("; lk_float_sums = 1 (OpenCV-order float sums)" if args.float_sums else "") + "; max_translation_norm 2.0 (reference 0.1: its gate is tuned for a 3 cm/frame rover, "
                                   "the scene moves 0.5 m/frame); %.0f %% of the pixels on an independently moving layer" % (100 * args.movers),
                       "sequences_per_gpu": B, "contexts_per_gpu": C, "frames_in_flight": depth, "mean_features_into_lk": N,
                       "mean_tracks_after_bounds": float(np.mean(n_bounds)), "mean_inliers": float(np.mean(n_inl)),
                       "mean_ransac_iters": float(np.mean(n_iters)),
                       "mean_features_dead_after_lk_pass_0_1_2": [float(v) for v in dead],
                       "ate_frames": {"timed_steps_of_slot_0": args.steps, "long_run": args.ate_frames},
                       "distinct_streams": min(B, args.pool * max(1, (2 * F - 2) // 3)),
                       "distinct_rendered_sequences": args.pool, "frames_per_sequence": F,
                       "pose_ok_fraction": n_ok_all / float(world * B * args.steps)},
            "roofline": {"bound": "hbm", "kernel": "k_lk_chain<%d>" % win, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src, "valu_issue": valu, "valu_flop_frac": valu_flop["frac_in_kernel"], "valu_flop": valu_flop,
                         "algorithmic_bytes_per_frame_pair": {"lk_chain": bytes_lk, "whole_frame": bytes_total},
                         "kernel_avg_ms": lk_avg_ms, "frame_avg_ms": float(np.mean(fr_ms)),
                         "whole_frame_frac": bytes_total * value / world / 1e9 / PEAK_HBM_GBS},
            "cpu_baseline": cpu, "ate": ate, "float_sums_mode": fs_leg,
            "per_rank_value": per_rank_value, "per_rank_lk_kernel_avg_ms": lk_avg_ms_ranks,
            # the line's own proof that the device worked through the timed region (an smi sampler misses a region this short):
            # the LK launches' HIP-event durations of rank 0, summed, against rank 0's wall clock; launches of different contexts
            # run on different streams, so their tails may overlap and the share is an upper estimate of LK's part
            "device_busy": {"lk_kernel_ms_in_timed_region": float(np.sum(lk_ms)), "lk_launches": len(lk_ms),
                            "timed_region_ms": dt_local * 1e3, "lk_share_of_timed_region": float(np.sum(lk_ms)) / (dt_local * 1e3)},
        }
        print(json.dumps(out))


if __name__ == "__main__":
    main()
