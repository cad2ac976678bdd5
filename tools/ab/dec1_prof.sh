#!/bin/bash
# the one-picture-per-call decoder (dsv_dec): frames/s unprofiled, then the kernels of a call under rocprofv3 --kernel-trace
REPO=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-dec1}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO && DEC_STREAMS=1 python3 tools/decode_time.py 2>/dev/null | head -1 | tee $OUT/${TAG}_decode_time.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/d1
DEC_STREAMS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/d1 -- python3 $REPO/tools/decode_time.py > /dev/null 2>&1
t=$(ls /tmp/d1/*/*kernel_trace.csv | head -1); m=$(ls /tmp/d1/*/*memory_copy_trace.csv | head -1)
python3 - "$t" "$m" <<'P' | tee $OUT/${TAG}_kernels.txt
import csv, sys, collections
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(sys.argv[1]))]
try:
    for r in csv.DictReader(open(sys.argv[2])): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %sB" % (r.get("Direction","")[:14], r.get("Size","?"))))
except Exception as e: print("no copy trace", e)
ev.sort()
# the decode calls of the second pass: find launches of k_hz_parse, take the sequence between two consecutive ones late in the run
idx = [i for i, e in enumerate(ev) if e[2].startswith("k_hz_parse")]
a, b = idx[-6], idx[-5]
print("one P picture's call (between two k_hz_parse launches), times relative to the first, us:")
t0 = ev[a][0]
for e in ev[a:b]:
    print("  %8.1f  +%6.1f  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
print("span %.1f us, kernel+copy time %.1f us, %d items" % ((ev[b][0] - t0) / 1e3, sum(e[1] - e[0] for e in ev[a:b]) / 1e3, b - a))
P
