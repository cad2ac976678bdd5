#!/bin/bash
# kernel trace of one shape of tools/bench_shape.py: the chain of ONE frame step (kernels between two consecutive launches of the
# lead kernel) with durations and gaps -- what a latency-bound shape (config 5: two 4K 4:4:4 pictures per frame step) waits for
# usage (through gpurun, from the repo root): tools/ab/shape_trace.sh <tag> <lead kernel substring> <bench_shape.py arguments...>
REPO=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; LEAD=$2; shift 2
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$TAG -- python3 $REPO/tools/bench_shape.py "$@" > $OUT/${TAG}_shape.txt 2>/dev/null
t=$(ls /tmp/st_$TAG/*/*kernel_trace.csv | head -1)
python3 - "$t" "$LEAD" > $OUT/${TAG}_step_chain.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r["Queue_Id"]) for r in rows)
lead = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
i0, i1 = lead[len(lead) // 2], lead[len(lead) // 2 + 1]
t0 = ev[i0][0]; last = None; busy = 0; n = 0
for s, e, k, qq in ev[i0:i1]:
    gap = (s - last) / 1e3 if last else 0.0
    print("%8.1f  dur %7.1f  gap %6.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, qq, k[:60]))
    busy += e - s; last = max(last or 0, e); n += 1
print("step %.1f us, %d kernels, kernel time %.1f us" % ((ev[i1][0] - t0) / 1e3, n, busy / 1e3))
P
cat $OUT/${TAG}_shape.txt | tail -1; cat $OUT/${TAG}_step_chain.txt
