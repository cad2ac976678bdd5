#!/bin/bash
# one counter pass (--pmc only, with the kernel trace) over a shape of tools/bench_shape.py, per-kernel summary
# usage (through gpurun): tools/ab/pmc_shape.sh <tag> "<counters>" <bench_shape.py arguments...>
REPO=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; CNT=$2; shift 2
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm_$TAG
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d /tmp/pm_$TAG -- python3 $REPO/tools/bench_shape.py "$@" > /dev/null 2>$OUT/${TAG}_pmc.err
python3 $REPO/tools/pmc_summary.py "/tmp/pm_$TAG/**/*counter_collection.csv" > $OUT/${TAG}_pmc.csv
head -14 $OUT/${TAG}_pmc.csv
