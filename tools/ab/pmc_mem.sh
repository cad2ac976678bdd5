#!/bin/bash
# memory-pipeline counters of the hot kernels (TA / TCP / UTCL1 / TCC), three rocprofv3 passes over one bench step
# usage (gpurun, repo root): bash tools/ab/pmc_mem.sh [gops] ; output gpurun_out/clk/pmc_mem_*.csv
GOPS=${1:-64}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/clk; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PB="python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops $GOPS --prof-kernel none --no-extras"
i=0
for C in "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE SQ_WAVES" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TD_TD_BUSY_sum MemUnitStalled"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm$i -- $PB > $OUT/pm$i.log 2>&1 < /dev/null
  python3 $REPO/tools/pmc_summary.py "/tmp/pm$i/**/*counter_collection.csv" > $OUT/pmc_mem_$i.csv 2>>$OUT/pm$i.log
  head -4 $OUT/pmc_mem_$i.csv | cut -c1-400
done
