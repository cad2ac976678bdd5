#!/usr/bin/env python3
"""One frame step of the coding chain from a rocprofv3 --kernel-trace CSV: the kernels between two consecutive
k_fwd_mc_pix<0> launches of one queue, with start offset, duration and the gap before each (microseconds)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r["Queue_Id"]) for r in rows)
lead = [i for i, e in enumerate(ev) if e[2].startswith("void k_fwd_mc_pix<0>")]
i0, i1 = lead[len(lead) // 2], lead[len(lead) // 2 + 1]
q = ev[i0][3]
t0 = ev[i0][0]; last = None; busy = 0
for s, e, k, qq in ev[i0:i1]:
    if qq != q: continue
    gap = (s - last) / 1e3 if last else 0.0
    print("%8.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, k[:50]))
    busy += e - s; last = e
print("step %.1f us, kernels %.1f us" % ((ev[i1][0] - t0) / 1e3, busy / 1e3))
