#!/usr/bin/env python3
"""The LAST dump of a DSV1_TIMELINE=1 run (the timed loop of bench.py: the marks between its two syncs) as one row per mark, sorted by
device time, with the phase lengths per batch: load (load0..load1), motion search (hme0..hme1), coding on both streams (code0..code1),
fetch (fetch0..fetch1) -- and for every coding phase how much of it overlapped a load / motion-search phase."""
import re, sys
dumps, cur = [], None
for ln in open(sys.argv[1]):
    if "[dsvg timeline]" not in ln: continue
    if " marks;" in ln:
        cur = []; dumps.append(cur); continue
    m = re.search(r"\[dsvg timeline\] (\S+)\s+dev\s+(-?[\d.]+)\s+host\s+(-?[\d.]+)", ln)
    if m and cur is not None: cur.append((m.group(1), float(m.group(2)), float(m.group(3))))
if not dumps: sys.exit("no timeline in " + sys.argv[1])
d = max(dumps, key=len)
print("%d marks in the largest dump" % len(d))
def phases(a, b):
    out, start = [], None
    for w, dev, host in d:
        if w == a: start = (dev, host)
        elif w == b and start is not None: out.append((start[0], dev, start[1], host)); start = None
    return out
P = {"upload": phases("up0", "up1"), "load": phases("load0", "load1"), "hme": phases("hme0", "hme1"), "code": phases("code0", "code1"), "fetch": phases("fetch0", "fetch1")}
for k, v in P.items():
    print("%-6s" % k, " ".join("[%.1f-%.1f | host enq %.1f-%.1f]" % x for x in v))
def ov(a, b): return max(0.0, min(a[1], b[1]) - max(a[0], b[0]))
ana = P["load"] + P["hme"]
for c in P["code"]:
    print("coding %.1f-%.1f (%.1f ms): %.1f ms of it with a load / motion-search phase in flight" % (c[0], c[1], c[1] - c[0], sum(ov(c, a) for a in ana)))
cs = P["code"]
if len(cs) > 2:
    per = (cs[-1][0] - cs[1][0]) / (len(cs) - 2)
    print("period %.2f ms; coding phases %.2f ms on average; idle between coding phases %.2f ms" % (per, sum(c[1] - c[0] for c in cs[1:]) / (len(cs) - 1), sum(cs[i + 1][0] - cs[i][1] for i in range(1, len(cs) - 1)) / max(1, len(cs) - 2)))
