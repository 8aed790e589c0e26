# The round-2 collection: kernel stats at 32 / 1 / 256 sequences under rocprofv3, the bench lines, the latency figures.
# usage (MI355X box): bash tools/collect_profiles.sh   -> gpurun_out/r02f/ ; the files are then copied into profiles/ (profiles/README.md)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s32 -o s32 -- python3 bench.py --steps 20 --warmup 4 --cpu-frames 0 --seqs 32 --contexts 1 > $O/s32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o s1 -- python3 bench.py --seqs 1 --contexts 1 --depth 1 --steps 40 --cpu-frames 0 > $O/s1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1s -o s1s -- python3 bench.py --seqs 1 --contexts 1 --depth 1 --steps 40 --cpu-frames 0 --movers 0 > $O/s1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s256 -o s256 -- python3 bench.py --steps 20 --warmup 4 --cpu-frames 0 > $O/s256.log 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --movers 0 --cpu-frames 0 > $O/bench_static.json 2> $O/bench_static.err
python3 tools/measure_pcie.py > $O/latency.txt 2>&1
python3 tools/stage_latency.py >> $O/latency.txt 2>&1
