#!/usr/bin/env python3
"""Concurrency picture from a rocprofv3 --kernel-trace CSV (second half of the run): how much of the time 0, 1, 2, 3+
kernels are in flight, the busy share of every hardware queue, and which kernels run ALONE most (time during which a
kernel is the only one in flight: the first place to look for idle compute units)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?")) for r in rows)
ev = ev[len(ev) // 2:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for i, (s, e, k, q) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
hist = collections.Counter(); alone = collections.Counter(); active = set(); last = t0
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        hist[min(len(active), 4)] += dt
        if len(active) == 1:
            alone[ev[next(iter(active))][2]] += dt
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
span = t1 - t0
print("span %.2f ms; kernels in flight: " % (span / 1e6) + ", ".join("%s: %.1f %%" % (("%d" % n if n < 4 else "4+"), 100.0 * hist[n] / span) for n in range(5)))
qb = collections.Counter()
for s, e, k, q in ev: qb[q] += e - s
print("queue busy: " + ", ".join("q%s %.1f %%" % (q, 100.0 * b / span) for q, b in sorted(qb.items())))
print("alone on the chip (ms): " + ", ".join("%s %.2f" % (k[:36], v / 1e6) for k, v in alone.most_common(8)))
