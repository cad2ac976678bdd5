#!/usr/bin/env python3
"""mean duration and launch count of the largest kernels of a rocprofv3 --kernel-trace CSV (second half of the run), and
the queues each of them ran on"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
acc = collections.defaultdict(lambda: [0, 0, collections.Counter()])
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    a = acc[k]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1; a[2][r.get("Queue_Id", "?")] += 1
for k, (t, n, q) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:10]:
    print("  %-44s %5d launches  mean %8.1f us  total %7.2f ms  queues %s" % (k[:44], n, t / n / 1e3, t / 1e6, dict(q)))
