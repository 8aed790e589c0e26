# Same-box A/B of builds of the float-sums LK:  gpurun -- 'bash tools/fs_ab.sh scratch/libsvo_A.so scratch/libsvo_B.so [...]'
# (LK ms per 32-sequence launch and the whole-job rate at the default configuration, both with --float-sums 1; the last build stays installed)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fs
q() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($1)"; }
for rep in 1 2; do for v in "$@"; do
  cp "$v" stereo_visual_odometry_amd/libsvo_hip.so
  lk=$(timeout -k 10 300 python bench.py --seqs 32 --contexts 1 --cpu-frames 0 --ate-frames 0 --float-sums 1 2>/dev/null | q "round(j['roofline']['kernel_avg_ms'],3)")
  fps=$(timeout -k 10 300 python bench.py --cpu-frames 0 --ate-frames 0 --float-sums 1 2>/dev/null | q "round(j['value'])")
  echo "$(basename "$v")  FS lk_ms(32 seq) $lk   frame-pairs/s(default cfg, float sums) $fps" | tee -a gpurun_out/fs/ab.txt
done; done
