#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   1. PMC passes over `bench.py --steps 1`: FETCH_SIZE and WRITE_SIZE in separate runs (they do not fit one
#      pass), one SQ pass -> per-kernel summaries -> pmc_traffic.json (what bench.py reads for `traffic`);
#   2. the default bench line; 3. rocprofv3 --kernel-trace --stats of the same command.
# Everything lands in gpurun_out/$TAG/ ; copy what should be judged into profiles/.
TAG=${1:-r01}
GOPS=${2:-320}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf /tmp/pw /tmp/ps /tmp/kt /tmp/k1
PB="python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops $GOPS --prof-kernel none --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- $PB > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- $PB > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/ps -- $PB > /dev/null 2>&1
python3 $REPO/tools/pmc_summary.py "/tmp/pf/**/*counter_collection.csv" "/tmp/pw/**/*counter_collection.csv" > "$OUT/pmc_hbm_per_kernel.csv"
python3 $REPO/tools/pmc_summary.py "/tmp/ps/**/*counter_collection.csv" > "$OUT/pmc_sq_per_kernel.csv"
# which roof: cycles with a VALU / scalar instruction in flight against the cycles the chip was busy
rm -rf /tmp/pr
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pr -- $PB > /dev/null 2>&1
python3 $REPO/tools/pmc_summary.py "/tmp/pr/**/*counter_collection.csv" > "$OUT/pmc_roof_per_kernel.csv"
python3 $REPO/tools/make_pmc_traffic.py "$OUT/pmc_hbm_per_kernel.csv" $GOPS "$OUT/pmc_traffic.json" "$OUT/pmc_sq_per_kernel.csv" "$OUT/pmc_roof_per_kernel.csv"
cp "$OUT/pmc_traffic.json" $REPO/profiles/pmc_traffic.json
cd $REPO && python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $REPO/bench.py --cpu-gops 0 --no-extras > "$OUT/bench_under_rocprof.json" 2>/dev/null
f=$(ls /tmp/kt/*/*kernel_stats.csv | head -1); cp "$f" "$OUT/rocprofv3_kernel_stats.csv"
t=$(ls /tmp/kt/*/*kernel_trace.csv | head -1); python3 $REPO/tools/trace_summary.py "$t" > "$OUT/kernel_trace_summary.txt"
# the same with ONE coding stream: every kernel alone on the chip, one launch = all pictures of a frame step -- the averages
# that compare with the bench line's `exclusive` / `others_exclusive` figures
DSV1_CODE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1 -- python3 $REPO/bench.py --cpu-gops 0 --no-extras --prof-kernel none > /dev/null 2>&1
f=$(ls /tmp/k1/*/*kernel_stats.csv | head -1); cp "$f" "$OUT/rocprofv3_kernel_stats_one_coding_stream.csv"
cat "$OUT/bench.json"
