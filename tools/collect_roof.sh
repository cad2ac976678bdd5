#!/bin/bash
# Evidence for the VALU roof (run through gpurun from the repo root):
#   1. tools/ubench/valu_rates at 1/2/3/4/8 waves per SIMD -> $OUT/valu_rates.txt
#   2. one rocprofv3 pass with the SQ busy / active-instruction counters over bench.py --steps 1 -> per-kernel summary
# Copy what should be judged into profiles/.
TAG=${1:-r03}
GOPS=${2:-320}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
timeout 300 $REPO/tools/ubench/valu_rates > "$OUT/valu_rates.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
# what FETCH_SIZE / WRITE_SIZE report for known access shapes (1 GiB streams): GB/s per shape, then the counters per kernel
if [ -x $REPO/tools/ubench/fetch_calib ]; then
  timeout 120 $REPO/tools/ubench/fetch_calib > "$OUT/fetch_calib.txt" 2>&1
  rm -rf /tmp/fc1 /tmp/fc2 /tmp/fc3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc1 -- $REPO/tools/ubench/fetch_calib > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/fc2 -- $REPO/tools/ubench/fetch_calib > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d /tmp/fc3 -- $REPO/tools/ubench/fetch_calib > /dev/null 2>&1
  python3 $REPO/tools/pmc_summary.py "/tmp/fc1/**/*counter_collection.csv" "/tmp/fc2/**/*counter_collection.csv" "/tmp/fc3/**/*counter_collection.csv" > "$OUT/fetch_calib_counters.csv"
  cat "$OUT/fetch_calib.txt" "$OUT/fetch_calib_counters.csv"
fi
PB="python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops $GOPS --prof-kernel none --no-extras"
rm -rf /tmp/pa /tmp/pb
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pa -- $PB > "$OUT/pa.log" 2>&1
python3 $REPO/tools/pmc_summary.py "/tmp/pa/**/*counter_collection.csv" > "$OUT/pmc_roof_per_kernel.csv"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/pb -- $PB > "$OUT/pb.log" 2>&1
python3 $REPO/tools/pmc_summary.py "/tmp/pb/**/*counter_collection.csv" > "$OUT/pmc_roof2_per_kernel.csv"
cat "$OUT/valu_rates.txt"
head -12 "$OUT/pmc_roof_per_kernel.csv"
