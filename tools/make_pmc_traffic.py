#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel PMC summary of tools/collect_profiles.sh.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide coalesced read at 64 bytes (MI355X_MICROARCH.md, HBM), so
the read side is doubled.  That calibration is for 16 B/lane streams; narrower accesses may be over-corrected,
which the file says in `source`.
If the SQ summary is given too, the VALU instruction count per launch is added (bench.py turns it into an issue
utilisation: one wave64 VALU instruction occupies its SIMD for 4 cycles).
usage: make_pmc_traffic.py <pmc_hbm_per_kernel.csv> <gops> <out.json> [pmc_sq_per_kernel.csv]"""
import csv
import json
import re
import sys


def kid(name):
    """rocprofv3 prints k_hme_level<true, 12, 1> (second argument: rows per lane of a full block, picked per geometry by the
    launcher; third: 1 = the launch over the full blocks, 2 = the partial blocks at the frame's edge); the profiling API of
    the library and bench.py name the kernel by its first argument only"""
    name = re.sub(r"k_hme_level<(true|false), \d+, [013]>", r"k_hme_level<\1>", name)          # the full blocks (or every block)
    return re.sub(r"k_hme_level<(true|false), \d+, 2>", r"k_hme_level<\1> (partial blocks)", name)


rows = list(csv.DictReader(open(sys.argv[1])))
out = {"gops": int(sys.argv[2]),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --gops %s --steps 1`; "
                 "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, read side doubled per the gfx950 correction "
                 "(calibrated for 16 B/lane streams)" % sys.argv[2],
       "kernels": {}}
for r in rows:
    f = float(r.get("FETCH_SIZE_per_launch", 0) or 0)
    w = float(r.get("WRITE_SIZE_per_launch", 0) or 0)
    out["kernels"][kid(r["kernel"])] = {"launches": int(r["launches"]), "fetch_kib_per_launch": f, "write_kib_per_launch": w,
                                   "hbm_bytes_per_launch": round((2 * f + w) * 1024),
                                   "hbm_bytes_per_launch_raw": round((f + w) * 1024)}
if len(sys.argv) > 4:
    for r in csv.DictReader(open(sys.argv[4])):
        e = out["kernels"].setdefault(kid(r["kernel"]), {"launches": int(r["launches"])})
        e["valu_insts_per_launch"] = float(r.get("SQ_INSTS_VALU_per_launch", 0) or 0)
        e["salu_insts_per_launch"] = float(r.get("SQ_INSTS_SALU_per_launch", 0) or 0)
        e["waves_per_launch"] = float(r.get("SQ_WAVES_per_launch", 0) or 0)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
