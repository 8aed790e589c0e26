#!/bin/bash
# Memory-path counters of the LK kernel (TA / TCP / TCC; 32 sequences, one context) -> gpurun_out/memctr, one line per pass on stdout.
# Two counters of a hardware block per pass (more: "Request exceeds the capabilities of the hardware to collect", and the run hangs),
# every pass under its own timeout.   gpurun -- 'bash tools/lk_memory_counters.sh'   (profiles/r03_memory_path_counters.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --seqs 32 --contexts 1 --steps 8 --warmup 2 --cpu-frames 0 --ate-frames 0"
O=gpurun_out/memctr; rm -rf $O; mkdir -p $O
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o c -- $B > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; grep -m1 -i "exceed\|error" $O/p$i.log; continue; }
  find $O/p$i -name '*kernel_trace.csv' -delete
  python3 - $O/p$i <<'PY'
import csv, glob, sys
tot={}
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lk_chain" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
print({k: "%.5g"%v for k,v in tot.items()}, flush=True)
PY
done
