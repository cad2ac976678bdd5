#!/bin/bash
# round 6: per-kernel LANE utilisation of the VALU work (thread-cycles / (instruction-cycles x 64)) over one headline step -- which kernels run
# their instructions with most lanes off (k_inv_b4t's column pass had 136 of 256 threads busy before its tile change)
REPO=$PWD; OUT=$REPO/gpurun_out/${1:-r06_lanes}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pl
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/pl -- python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops 320 --prof-kernel none --no-extras > /dev/null 2> $OUT/err.txt
python3 $REPO/tools/pmc_summary.py "/tmp/pl/**/*counter_collection.csv" > $OUT/pmc_lanes_per_kernel.csv
head -40 $OUT/pmc_lanes_per_kernel.csv
