#!/bin/bash
# kernel-level picture of the intra-only shape: rocprofv3 --kernel-trace --stats over tools/intra_bench.py
REPO=$PWD; OUT=$REPO/gpurun_out/intra; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ik
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ik -- python3 $REPO/tools/intra_bench.py ${1:-64} > $OUT/intra.json 2>/dev/null
t=$(ls /tmp/ik/*/*kernel_trace.csv | head -1); python3 $REPO/tools/trace_summary.py "$t" > $OUT/intra_kernel_trace_summary.txt
cat $OUT/intra.json; head -24 $OUT/intra_kernel_trace_summary.txt
