# One-call collection of a round's rocprofv3 evidence (MI355X box):  bash tools/collect_profiles.sh r03
#   -> gpurun_out/<tag>p/ ; then, in the build container:  bash tools/collect_profiles.sh r03 --summarize   copies / reduces into profiles/
# Kernel stats at 32 / 1 / 512 (default) sequences, the counter passes (FETCH_SIZE, WRITE_SIZE, two SQ passes, GRBM_GUI_ACTIVE — one --pmc
# pass each, with --kernel-trace only) at the DEFAULT bench configuration (two contexts of 256 sequences: the launch the bench
# line's roofline is about), the bench lines and the latency figures.
set -e
TAG=${1:-r03}
if [ "$2" = "--summarize" ]; then
  O=gpurun_out/${TAG}p
  python3 profiles/summarize.py ${TAG} $O/s32 $O/fetch $O/write --seqs 256
  python3 profiles/summarize.py ${TAG} --sq $O/sq1 --bench $O/bench.json --extra $O/sq2 --gui $O/gui
  cp $(find $O/s256 -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats_default_2ctx.csv
  cp $(find $O/s1 -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats_1seq.csv
  cp $(find $O/s1s -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats_1seq_static.csv
  [ -d $O/one512 ] && cp $(find $O/one512 -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats_512seq_1ctx.csv
  [ -d $O/fs32 ] && cp $(find $O/fs32 -name '*kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats_float_sums.csv && grep '^{"metric"' $O/fs32.log | tail -1 > profiles/${TAG}_bench_32seq_float_sums.json
  cp $O/bench.json profiles/${TAG}_bench.json; cp $O/bench_static.json profiles/${TAG}_bench_static.json
  grep '^{"metric"' $O/s32.log | tail -1 > profiles/${TAG}_bench_32seq_1ctx.json; grep '^{"metric"' $O/s256.log | tail -1 > profiles/${TAG}_bench_default_under_rocprof.json
  cp $O/latency.txt profiles/${TAG}_latency.txt
  cp $O/latency_cpp.txt profiles/${TAG}_latency_cpp.txt
  exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}p; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 4 --cpu-frames 0 --ate-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s32 -o s32 -- $B --seqs 32 --contexts 1 > $O/s32.log 2>&1
echo s32 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o s1 -- python3 bench.py --seqs 1 --contexts 1 --depth 1 --steps 40 --cpu-frames 0 --ate-frames 0 > $O/s1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1s -o s1s -- python3 bench.py --seqs 1 --contexts 1 --depth 1 --steps 40 --cpu-frames 0 --ate-frames 0 --movers 0 > $O/s1s.log 2>&1
echo s1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s256 -o s256 -- $B > $O/s256.log 2>&1
echo s256 done
# every kernel's OWN duration at the bench's batch size: one context of 512 sequences, image stream off, nothing overlaps
SVO_INGEST_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one512 -o one512 -- python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 --ate-frames 0 --seqs 512 --contexts 1 > $O/one512.log 2>&1
# float-sums mode (the reference's own LK rounding), 32 sequences in one context
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fs32 -o fs32 -- $B --seqs 32 --contexts 1 --float-sums 1 > $O/fs32.log 2>&1
echo one512 + fs32 done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $B > $O/write.log 2>&1
echo hbm counters done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1 -o sq1 -- $B > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/sq2 -o sq2 -- $B > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui -o gui -- $B > $O/gui.log 2>&1
echo sq counters done
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --movers 0 --cpu-frames 0 > $O/bench_static.json 2> $O/bench_static.err
python3 tools/measure_pcie.py > $O/latency.txt 2>&1
python3 tools/stage_latency.py >> $O/latency.txt 2>&1
python3 tools/latency_cpp.py 21 300 > $O/latency_cpp.txt 2>/dev/null
python3 tools/latency_cpp.py 10 300 >> $O/latency_cpp.txt 2>/dev/null
python3 tools/cfg1_latency.py >> $O/latency_cpp.txt 2>/dev/null
# keep what travels back small: the per-dispatch traces are large, the summaries and counter CSVs are what profiles/ needs
find $O -name '*kernel_trace.csv' -size +8M -delete || true
du -sh $O
