# Whole-job rate against the number of contexts per GPU and the LK chaining (one gpurun call, one box):  gpurun -- bash tools/contexts_sweep.sh
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ctx
q() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), round(j['roofline']['kernel_avg_ms'],2), round(j['device_busy']['lk_share_of_timed_region'],3))"; }
run() { # gate seqs contexts depth
  r=$(SVO_LK_GATE=$1 timeout -k 10 200 python bench.py --seqs $2 --contexts $3 --depth $4 --cpu-frames 0 --ate-frames 0 2>/dev/null | q)
  echo "gate=$1 seqs=$2 contexts=$3 depth=$4 : frame-pairs/s, LK ms per launch, LK share = $r" | tee -a gpurun_out/ctx/sweep.txt
}
run 1 512 2 4
run 0 512 2 4
run 1 768 3 4
run 0 768 3 4
run 1 1024 4 4
run 0 1024 4 4
run 1 512 4 4
run 0 512 4 4
run 1 512 2 4
