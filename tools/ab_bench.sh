#!/usr/bin/env bash
# Same-box A/B(/C...) of builds of libsvo_hip.so.  MI355X devices of a pool differ by a few per cent on the LK kernel, and
# every `gpurun` call may land on another one, so builds are only comparable when timed alternately inside ONE call:
#
#   cp stereo_visual_odometry_amd/libsvo_hip.so scratch/libsvo_A.so          # build A
#   ... change, rebuild ...; cp stereo_visual_odometry_amd/libsvo_hip.so scratch/libsvo_B.so   # build B
#   gpurun -- 'bash tools/ab_bench.sh scratch/libsvo_A.so scratch/libsvo_B.so [more.so ...]'
#
# Prints, three times per build, the HIP-event LK time of a 32-sequence launch and the default whole-job rate.
# (The installed library is overwritten by the last build timed; gpurun boxes are scratch copies.)  AB_ARGS = extra bench.py args.
set -e
cd "$(dirname "$0")/.."
[ $# -ge 2 ] || { echo "usage: ab_bench.sh A.so B.so [C.so ...]"; exit 2; }
q() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($1)"; }
for rep in 1 2 3; do for v in "$@"; do
  cp "$v" stereo_visual_odometry_amd/libsvo_hip.so
  lk=$(timeout -k 10 300 python bench.py --seqs 32 --contexts 1 --cpu-frames 0 $AB_ARGS 2>/dev/null | q "round(j['roofline']['kernel_avg_ms'],3)")
  fps=$(timeout -k 10 300 python bench.py --cpu-frames 0 $AB_ARGS 2>/dev/null | q "round(j['value'])")
  echo "$(basename "$v")  lk_ms(32 seq) $lk   frame-pairs/s(default) $fps"
done; done
