#!/bin/bash
# one SQ counter pass over bench.py --steps 1 and the per-kernel summary (run through gpurun from the repo root)
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
PB="python3 $REPO/bench.py --cpu-gops 0 --steps 1 --warmup 1 --gops 64 --prof-kernel none --no-extras"
rm -rf /tmp/ps
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/ps -- $PB > /dev/null 2>&1
python3 $REPO/tools/pmc_summary.py "/tmp/ps/**/*counter_collection.csv"
