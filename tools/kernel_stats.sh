#!/bin/bash
# rocprofv3 per-kernel average durations at one context of --seqs sequences (bench default) for each library given, one line per build.
#   gpurun -- 'bash tools/kernel_stats.sh scratch/libsvo_A.so scratch/libsvo_B.so'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
cp $v stereo_visual_odometry_amd/libsvo_hip.so; n=$(basename $v .so)
rm -rf gpurun_out/ks_$n
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$n -o k -- python3 bench.py --steps 12 --warmup 3 --cpu-frames 0 --ate-frames 0 --contexts 1 > gpurun_out/ks_$n.log 2>&1
find gpurun_out/ks_$n -name "*kernel_trace.csv" -delete
python3 - gpurun_out/ks_$n $n <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(sys.argv[2], " | ".join("%s %.0f"%(r['Name'].split('(')[0].replace('void ','')[:16], float(r['AverageNs'])/1e3) for r in rows[:12]), flush=True)
PY
done
