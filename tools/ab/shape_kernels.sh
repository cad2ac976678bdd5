#!/bin/bash
# per-kernel launch counts and mean durations of one shape of tools/bench_shape.py (durations are valid under the tracer, gaps are not)
# usage (through gpurun): tools/ab/shape_kernels.sh <tag> <bench_shape.py arguments...>
REPO=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO && python3 tools/bench_shape.py "$@" 2>/dev/null | tail -1 | tee $OUT/${TAG}_shape.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sk_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/sk_$TAG -- python3 $REPO/tools/bench_shape.py "$@" > /dev/null 2>&1
t=$(ls /tmp/sk_$TAG/*/*kernel_trace.csv | head -1)
python3 $REPO/tools/trace_summary.py "$t" | head -40 | tee $OUT/${TAG}_kernels.txt
