#!/usr/bin/env python3
"""Per-queue picture of the steady state from a rocprofv3 --kernel-trace CSV (the last third of the run): for every hardware
queue its busy share, the kernels it ran and the gaps between consecutive kernels (count, total, the contexts of the longest
ones); for every kernel the slowdown against a table of exclusive times (optional second argument: a bench JSON line whose
roofline.all_kernels_ms_one_step holds them) weighted by how many other kernels were in flight beside it.
usage: trace_queues.py <kernel_trace.csv> [bench.json] [--dump <out.csv.gz>: the window's rows, for later digging]"""
import collections, csv, gzip, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?")) for r in rows)
# the steady state of the timed loop: every batch starts with one large k_unpack launch -- the window runs from the start of the
# batch a third of the way in to the start of the batch two thirds of the way in (whole steps, away from warm-up and the checks
# after the loop)
U = [e for e in ev if e[2] == "k_unpack"]
big = max(e[1] - e[0] for e in U) if U else 0
U = [e for e in U if e[1] - e[0] >= big * 0.5]
if len(U) >= 6:
    a, b_ = U[len(U) // 3][0], U[(2 * len(U)) // 3][0]
    nsteps_w = (2 * len(U)) // 3 - len(U) // 3
else:
    T0, T1 = ev[0][0], max(e[1] for e in ev)
    a, b_, nsteps_w = T0 + (T1 - T0) * 2 // 3, T1, 0
w = [e for e in ev if e[0] >= a and e[0] < b_]
span = b_ - a
print("window %.2f ms = %d steps of %.2f ms, %d launches" % (span / 1e6, nsteps_w, span / 1e6 / max(nsteps_w, 1), len(w)))
if "--dump" in sys.argv:
    with gzip.open(sys.argv[sys.argv.index("--dump") + 1], "wt") as f:
        for s, e, k, q in w: f.write("%d,%d,%s,%s\n" % (s - w[0][0], e - w[0][0], k, q))

# concurrency: at every moment the number of kernels in flight
pts = []
for i, (s, e, k, q) in enumerate(w): pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
active = set(); last = w[0][0]; hist = collections.Counter()
share = collections.Counter()      # kernel -> sum over its life of dt / (kernels in flight)
life = collections.Counter(); cnt = collections.Counter()
for t, d, i in pts:
    dt = t - last
    if dt > 0 and active:
        n = len(active)
        hist[min(n, 5)] += dt
        for j in active: share[w[j][2]] += dt / n
    elif dt > 0: hist[0] += dt
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
for s, e, k, q in w: life[k] += e - s; cnt[k] += 1
print("kernels in flight: " + ", ".join("%s: %.1f %%" % (("%d" % n if n < 5 else "5+"), 100.0 * hist[n] / span) for n in range(6)))

# per queue
byq = collections.defaultdict(list)
for e in w: byq[e[3]].append(e)
for q, L in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in L)
    gaps = [(L[i + 1][0] - L[i][1], L[i][2], L[i + 1][2]) for i in range(len(L) - 1)]
    g_pos = [g for g in gaps if g[0] > 0]
    tot = sum(g[0] for g in g_pos)
    big = [g for g in g_pos if g[0] > 30000]
    print("queue %s: %d launches, busy %.2f ms (%.0f %%), gaps %.2f ms: <=10us %d (%.2f ms), 10-30us %d (%.2f ms), >30us %d (%.2f ms)" % (
        q, len(L), busy / 1e6, 100.0 * busy / span, tot / 1e6,
        sum(1 for g in g_pos if g[0] <= 10000), sum(g[0] for g in g_pos if g[0] <= 10000) / 1e6,
        sum(1 for g in g_pos if 10000 < g[0] <= 30000), sum(g[0] for g in g_pos if 10000 < g[0] <= 30000) / 1e6,
        len(big), sum(g[0] for g in big) / 1e6))
    ctx = collections.Counter(); ctn = collections.Counter()
    for g, p, n in g_pos: ctx[(p, n)] += g; ctn[(p, n)] += 1
    for (p, n), g in ctx.most_common(6):
        print("      %8.2f ms in %4d gaps (avg %6.1f us)  %s -> %s" % (g / 1e6, ctn[(p, n)], g / ctn[(p, n)] / 1e3, p, n))
    top = collections.Counter()
    for e in L: top[e[2]] += e[1] - e[0]
    print("      kernels: " + ", ".join("%s %.1f" % (k, v / 1e6) for k, v in top.most_common(8)))

# slowdown against the exclusive times
if len(sys.argv) > 2 and not sys.argv[2].startswith("--"):
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    ex = d["roofline"]["all_kernels_ms_one_step"]
    steps = nsteps_w if nsteps_w else span / 1e6 / d["ms_per_step"]
    print("window = %.2f steps (untraced step: %.2f ms); per kernel: life in flight per step, its share of the chip (life / kernels in flight), exclusive ms per step" % (steps, d["ms_per_step"]))
    for k, v in sorted(life.items(), key=lambda kv: -kv[1])[:24]:
        x = [t for kk, t in ex.items() if kk.split("(")[0][:40] == k]
        print("  %-42s life %6.2f  share %6.2f  exclusive %s" % (k, v / 1e6 / steps, share[k] / 1e6 / steps, ("%6.2f" % x[0]) if x else "     ?"))
    print("  sum of shares per step %.2f ms (= busy time), sum of lives %.2f" % (sum(share.values()) / 1e6 / steps, sum(life.values()) / 1e6 / steps))
