#!/usr/bin/env python3
"""print the headline and the per-kernel exclusive times of bench.py JSON lines (files given as arguments)"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    r = d.get("roofline") or {}
    print("%s: %.0f Mpix/s  %.3f ms/step  bit_exact=%s" % (f, d["value"], d["ms_per_step"], d.get("bit_exact_vs_cpu")))
    t = r.get("all_kernels_ms_one_step", {})
    print("   sum of exclusive kernel ms: %.2f" % sum(t.values()))
    for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:14]:
        print("   %7.3f  %s" % (v, k))
    if r.get("sparse_inverse_tiles_one_step"): print("   tiles:", r["sparse_inverse_tiles_one_step"])
    for k in ("value_host_pinned", "shapes"):
        if k in d: print("  ", k, json.dumps(d[k]))
