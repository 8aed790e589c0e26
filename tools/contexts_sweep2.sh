# One context per GPU against two (the chains of small kernels cannot run under a six-wave LK grid anyway):  gpurun -- bash tools/contexts_sweep2.sh
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ctx
q() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), round(j['roofline']['kernel_avg_ms'],2), round(j['device_busy']['lk_share_of_timed_region'],3))"; }
run() { # seqs contexts depth
  r=$(timeout -k 10 200 python bench.py --seqs $1 --contexts $2 --depth $3 --cpu-frames 0 --ate-frames 0 2>/dev/null | q)
  echo "seqs=$1 contexts=$2 depth=$3 : frame-pairs/s, LK ms per launch, LK share = $r" | tee -a gpurun_out/ctx/sweep2.txt
}
for rep in 1 2; do
run 512 2 4
run 512 1 4
run 1024 1 4
run 1024 2 4
run 256 1 4
done
