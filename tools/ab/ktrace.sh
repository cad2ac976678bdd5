#!/bin/bash
# per-kernel means of one bench run under rocprofv3 --kernel-trace (headline workload only); usage: ktrace.sh [pattern]
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktx
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktx -- python3 $REPO/bench.py --cpu-gops 0 --steps 4 --warmup 2 --no-extras --prof-kernel none > /dev/null 2>&1
t=$(ls /tmp/ktx/*/*kernel_trace.csv | head -1)
python3 $REPO/tools/trace_summary.py "$t" "$@" | head -60
