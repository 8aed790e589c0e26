#!/usr/bin/env bash
# Same-box A/B of the LK kernel's INSTRUCTION COUNT (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU), beside tools/ab_bench.sh's times:
#
#   gpurun -- 'bash tools/ab_valu.sh scratch/libsvo_A.so scratch/libsvo_B.so [more.so ...]'
#
# Every build runs the same frames (32 sequences, one context, 8 timed steps), so the totals over all k_lk_chain dispatches are
# directly comparable; the per-feature figure divides by the features that entered LK (the bench line's mean count x sequences x
# steps).  The installed library is overwritten by the last build; gpurun boxes are scratch copies.
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
[ $# -ge 1 ] || { echo "usage: ab_valu.sh A.so [B.so ...]"; exit 2; }
mkdir -p gpurun_out/abv
for v in "$@"; do
  n=$(basename "$v" .so)
  cp "$v" stereo_visual_odometry_amd/libsvo_hip.so
  rm -rf gpurun_out/abv/$n
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/abv/$n -o c -- python3 bench.py --seqs 32 --contexts 1 --steps 8 --warmup 2 --cpu-frames 0 --ate-frames 0 $AB_ARGS > gpurun_out/abv/$n.log 2> gpurun_out/abv/$n.err
  python3 - "$n" gpurun_out/abv/$n gpurun_out/abv/$n.log <<'EOF'
import csv, glob, json, sys
name, d, log = sys.argv[1:4]
tot = {}; launches = 0
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lk_chain" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            launches += r["Counter_Name"] == "SQ_INSTS_VALU"
line = json.loads([l for l in open(log).read().splitlines() if l.startswith('{"metric"')][-1])
feat = line["config"].get("mean_features_into_lk") or line["config"].get("features_per_frame")
print("%-14s k_lk_chain launches %d  VALU %.4e  SALU %.4e  per launch: VALU %.4e%s" % (
    name, launches, tot.get("SQ_INSTS_VALU", 0), tot.get("SQ_INSTS_SALU", 0), tot.get("SQ_INSTS_VALU", 0) / max(launches, 1),
    ("  per feature %.0f" % (tot["SQ_INSTS_VALU"] / launches / (32 * feat))) if feat else ""))
EOF
  find gpurun_out/abv/$n -name '*kernel_trace.csv' -delete || true
done
