#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel launches / total / avg, and the per-launch durations of
the kernels named on the command line (substring match)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
tot = sum(x[1] for v in d.values() for x in v)
print(f"{'kernel':58s} {'n':>6s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    t = sum(x[1] for x in v)
    print(f"{k[:58]:58s} {len(v):6d} {t / 1e6:10.3f} {t / len(v) / 1e3:9.1f} {100.0 * t / tot:6.2f}")
for pat in sys.argv[2:]:
    for k, v in d.items():
        if pat in k:
            v.sort()
            print(k[:58], [round(x[1] / 1000, 1) for x in v[-int(26):]])
