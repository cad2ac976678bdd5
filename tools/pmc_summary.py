#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: sum and per-launch mean of every counter."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(set)
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nl[k].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches"] + [n + "_per_launch" for n in names])
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = max(1, len(nl[k]))
    w.writerow([k[:60], n] + [f"{acc[k].get(c, 0.0) / n:.4g}" for c in names])
