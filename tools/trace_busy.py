#!/usr/bin/env python3
"""GPU busy fraction from a rocprofv3 --kernel-trace CSV: union of the kernel intervals over the span of the last
N dispatches (steady state), and the histogram of idle gaps between consecutive intervals of that union."""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
tail = rows[len(rows) // 2:]                 # second half: past warm-up
t0, t1 = tail[0][0], max(e for _, e in tail)
busy, gaps, cur_s, cur_e = 0, [], tail[0][0], tail[0][1]
for s, e in tail[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.2f ms, busy %.2f ms (%.1f %%), %d gaps: total %.2f ms, median %.1f us, >20us: %d (%.2f ms)" % (
    (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(gaps), sum(gaps) / 1e6,
    sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
