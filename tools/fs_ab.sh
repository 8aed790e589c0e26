# Same-box A/B of two builds of the float-sums LK (scratch/libsvo_fsold.so, scratch/libsvo_fsnew.so) + the float-sums parity runs:  gpurun -- bash tools/fs_ab.sh
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fs
q() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($1)"; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_run1_color.py tests/test_run1_cli.py -m gpu -x -q -k "float_sums or recorded or float or cli" > gpurun_out/fs/t1.log 2>&1 || { tail -30 gpurun_out/fs/t1.log; exit 1; }
tail -2 gpurun_out/fs/t1.log
SVO_FUZZ_CASES=140 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_scenes" > gpurun_out/fs/t2.log 2>&1 || { tail -30 gpurun_out/fs/t2.log; exit 1; }
tail -2 gpurun_out/fs/t2.log
for rep in 1 2; do for v in scratch/libsvo_fsold.so scratch/libsvo_fsnew.so; do
  cp "$v" stereo_visual_odometry_amd/libsvo_hip.so
  lk=$(timeout -k 10 300 python bench.py --seqs 32 --contexts 1 --cpu-frames 0 --ate-frames 0 --float-sums 1 2>/dev/null | q "round(j['roofline']['kernel_avg_ms'],3)")
  fps=$(timeout -k 10 300 python bench.py --cpu-frames 0 --ate-frames 0 --float-sums 1 2>/dev/null | q "round(j['value'])")
  echo "$(basename "$v")  FS lk_ms(32 seq) $lk   frame-pairs/s(default cfg, float sums) $fps" | tee -a gpurun_out/fs/ab.txt
done; done
