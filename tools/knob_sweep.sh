# Whole-job rate against the run-time knobs of the default configuration (LK run length per XCD, frames in flight), one box:  gpurun -- bash tools/knob_sweep.sh
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/knobs
q() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), round(j['roofline']['kernel_avg_ms'],2))"; }
for rep in 1 2; do
for chunk in 2 4 8; do for depth in 4 8; do
  r=$(SVO_LK_CHUNK=$chunk timeout -k 10 200 python bench.py --depth $depth --cpu-frames 0 --ate-frames 0 2>/dev/null | q)
  echo "chunk=$chunk depth=$depth : frame-pairs/s, LK ms per launch = $r" | tee -a gpurun_out/knobs/sweep.txt
done; done; done
