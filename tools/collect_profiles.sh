#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   1. the default bench line, 2. rocprofv3 --kernel-trace --stats of the same command,
#   3. two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass) + one SQ pass.
# Everything lands in gpurun_out/$TAG/ ; copy what should be judged into profiles/.
TAG=${1:-r01}
GOPS=${2:-32}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $REPO/bench.py --cpu-gops 0 > "$OUT/bench_under_rocprof.json" 2>/dev/null
f=$(ls /tmp/kt/*/*kernel_stats.csv | head -1); cp "$f" "$OUT/rocprofv3_kernel_stats.csv"
t=$(ls /tmp/kt/*/*kernel_trace.csv | head -1); python3 $REPO/tools/trace_summary.py "$t" > "$OUT/kernel_trace_summary.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops $GOPS --prof-kernel none > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops $GOPS --prof-kernel none > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/ps -- python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops $GOPS --prof-kernel none > /dev/null 2>&1
python3 $REPO/tools/pmc_summary.py "/tmp/pf/**/*counter_collection.csv" "/tmp/pw/**/*counter_collection.csv" > "$OUT/pmc_hbm_per_kernel.csv"
python3 $REPO/tools/pmc_summary.py "/tmp/ps/**/*counter_collection.csv" > "$OUT/pmc_sq_per_kernel.csv"
cat "$OUT/bench.json"
