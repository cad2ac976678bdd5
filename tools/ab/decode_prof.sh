#!/bin/bash
# kernel-level picture of the batched decoder: rocprofv3 --kernel-trace --stats over tools/decode_bench.py
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-dec}
mkdir -p $OUT
python3 tools/decode_bench.py 64 4 > $OUT/decode.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk -- python3 $REPO/tools/decode_bench.py 64 4 > /dev/null 2>&1
t=$(ls /tmp/dk/*/*kernel_trace.csv | head -1); python3 $REPO/tools/trace_summary.py "$t" > $OUT/decode_kernel_trace_summary.txt
cat $OUT/decode.json; head -24 $OUT/decode_kernel_trace_summary.txt
