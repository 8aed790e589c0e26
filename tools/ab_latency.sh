#!/usr/bin/env bash
# Same-box A/B of the single-stream latency of builds of libsvo_hip.so (the latency counterpart of tools/ab_bench.sh):
#   gpurun -- 'bash tools/ab_latency.sh scratch/libsvo_A.so scratch/libsvo_B.so'
# prints, twice per build, ms per synchronous svo_process call (host images) and the per-stage HIP-event times, static and mover scene.
set -e
cd "$(dirname "$0")/.."
for rep in 1 2; do for v in "$@"; do cp "$v" stereo_visual_odometry_amd/libsvo_hip.so; echo "== $v"; timeout -k 10 200 python tools/stage_latency.py; done; done
