#!/bin/bash
# kernel-level picture of the worst-case shape (a third of the P blocks intra): rocprofv3 --kernel-trace --stats over tools/worst_bench.py
REPO=$PWD; OUT=$REPO/gpurun_out/worst; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/wk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wk -- python3 $REPO/tools/worst_bench.py ${1:-160} ${2:-4} > $OUT/worst.json 2>/dev/null
t=$(ls /tmp/wk/*/*kernel_trace.csv | head -1); python3 $REPO/tools/trace_summary.py "$t" > $OUT/worst_kernel_trace_summary.txt
cat $OUT/worst.json; head -36 $OUT/worst_kernel_trace_summary.txt
