#!/bin/bash
# GPU busy fraction and idle gaps of the headline loop (rocprofv3 --kernel-trace, tools/trace_busy.py)
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktb
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktb -- python3 $REPO/bench.py --cpu-gops 0 --steps 8 --warmup 2 --no-extras --prof-kernel none > /tmp/ktb.json 2>/dev/null
t=$(ls /tmp/ktb/*/*kernel_trace.csv | head -1)
python3 $REPO/tools/trace_busy.py "$t"
python3 -c "import json; d=json.loads(open('/tmp/ktb.json').read().strip().splitlines()[-1]); print('under rocprof:', d['value'], d['ms_per_step'])"
