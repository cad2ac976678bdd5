#!/bin/bash
# per-queue timeline of the headline loop: bench line first (exclusive kernel times), then a kernel trace of the same loop
# usage (through gpurun): tools/ab/qtrace.sh <tag> [env settings...]
REPO=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO && env "$@" python3 bench.py --cpu-gops 0 --steps 6 --no-extras > $OUT/${TAG}_bench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/qt_$TAG
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/qt_$TAG -- python3 $REPO/bench.py --cpu-gops 0 --steps 12 --warmup 2 --no-extras --prof-kernel none > $OUT/${TAG}_bench_traced.json 2>/dev/null
k=$(ls /tmp/qt_$TAG/*/*kernel_trace.csv | head -1)
python3 $REPO/tools/trace_queues.py "$k" $OUT/${TAG}_bench.json --dump $OUT/${TAG}_window.csv.gz > $OUT/${TAG}_queues.txt
cat $OUT/${TAG}_queues.txt
