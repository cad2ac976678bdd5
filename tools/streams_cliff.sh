#!/bin/bash
# Why do three coding streams run slower than two?  The same bench step under rocprofv3 --kernel-trace with 1, 2, 3 and 4
# coding streams (DSV1_CODE_STREAMS): step time, how many kernels are in flight, queue shares, and the mean duration of
# the large kernels in each configuration.  Run through gpurun from the repo root; results in gpurun_out/cliff/.
REPO=$PWD
OUT=$REPO/gpurun_out/cliff
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in 1 2 3 4; do
  export DSV1_CODE_STREAMS=$n
  rm -rf /tmp/kt$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$n -- python3 $REPO/bench.py --cpu-gops 0 --steps 4 --warmup 2 --no-extras --prof-kernel none > "$OUT/bench_$n.json" 2>/dev/null
  t=$(ls /tmp/kt$n/*/*kernel_trace.csv | head -1)
  echo "== $n coding stream(s): $(python3 -c "import json;d=json.load(open('$OUT/bench_$n.json'));print(d['ms_per_step'],'ms/step',d['value'],'Mpix/s')")"
  python3 $REPO/tools/trace_overlap.py "$t"
  python3 $REPO/tools/trace_kernel_means.py "$t"
done 2>&1 | tee "$OUT/summary.txt"
