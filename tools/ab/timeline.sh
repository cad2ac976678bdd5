#!/bin/bash
# where the wall time of one pipelined step of a small shape goes: kernel trace of tools/bench_shape.py, then for a window in the
# middle of the run the union of the kernel intervals (GPU busy), the idle gaps longer than 20 us with the kernels around them,
# and the time per queue.  usage (through gpurun): tools/ab/timeline.sh <tag> <bench_shape.py arguments...>
REPO=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$TAG
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$TAG -- python3 $REPO/tools/bench_shape.py "$@" > $OUT/${TAG}_shape.txt 2>/dev/null
k=$(ls /tmp/tl_$TAG/*/*kernel_trace.csv | head -1); m=$(ls /tmp/tl_$TAG/*/*memory_copy_trace.csv 2>/dev/null | head -1)
python3 - "$k" "$m" > $OUT/${TAG}_timeline.txt <<'P'
import csv, sys
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], "q" + r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1]))]
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        for r in csv.DictReader(open(sys.argv[2])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20] + " %s B" % r.get("Size", "?"), "copy"))
    except Exception as e:
        print("no copy trace:", e)
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
a, b = t0 + (t1 - t0) * 0.55, t0 + (t1 - t0) * 0.85          # a window late in the run (the timed loop)
w = [e for e in ev if e[0] >= a and e[1] <= b]
busy = 0; cur_s, cur_e = w[0][0], w[0][1]; gaps = []; prev = w[0]
for e in w[1:]:
    if e[0] > cur_e:
        busy += cur_e - cur_s
        if e[0] - cur_e > 20000: gaps.append((e[0] - cur_e, prev, e))
        cur_s, cur_e = e[0], e[1]
    else:
        cur_e = max(cur_e, e[1])
    if e[1] >= prev[1]: prev = e
busy += cur_e - cur_s
span = w[-1][1] - w[0][0]
print("window %.2f ms, %d events, GPU busy (union) %.2f ms = %.0f %%" % (span / 1e6, len(w), busy / 1e6, 100.0 * busy / span))
per = {}
for e in w: per[e[3]] = per.get(e[3], 0) + e[1] - e[0]
print("time per queue (ms):", {k: round(v / 1e6, 2) for k, v in sorted(per.items())})
pk = {}
for e in w: pk[e[2]] = pk.get(e[2], 0) + e[1] - e[0]
print("top kernels (ms):", [(k, round(v / 1e6, 2)) for k, v in sorted(pk.items(), key=lambda kv: -kv[1])[:14]])
print("idle gaps > 20 us: %d, total %.2f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for g, p, n in sorted(gaps, key=lambda x: -x[0])[:25]:
    print("  %7.1f us idle after %-46s (%s) before %-46s (%s)" % (g / 1e3, p[2], p[3], n[2], n[3]))
P
tail -1 $OUT/${TAG}_shape.txt; cat $OUT/${TAG}_timeline.txt
