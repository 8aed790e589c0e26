#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the LK kernel per build (32 sequences, one context), separate --pmc passes.
#   gpurun -- 'bash tools/lk_traffic.sh scratch/libsvo_A.so scratch/libsvo_B.so'   (the installed library is overwritten by the last build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --seqs 32 --contexts 1 --steps 8 --warmup 2 --cpu-frames 0 --ate-frames 0"
for v in "$@"; do
cp $v stereo_visual_odometry_amd/libsvo_hip.so; n=$(basename $v .so)
for c in FETCH_SIZE WRITE_SIZE; do
  O=gpurun_out/tr_${n}_$c; rm -rf $O
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o c -- $B > $O.log 2>&1 || { echo "$n $c failed"; continue; }
  python3 - $O $n $c <<'PY'
import csv, glob, sys
tot=0; n=0
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lk_chain" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[3]:
            tot+=float(r["Counter_Value"]); n+=1
print(sys.argv[2], sys.argv[3], "per launch %.1f MB over %d launches"%(tot*1024/n/1e6, n), flush=True)
PY
  rm -rf $O
done; done
