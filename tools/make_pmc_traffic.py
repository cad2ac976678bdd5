#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel PMC summary of tools/collect_profiles.sh.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on gfx950
FETCH_SIZE tallies every 128-byte read request at 64 bytes (MI355X_MICROARCH.md, HBM), so the read side is
doubled.  profiles/r03_fetch_calib.txt (tools/ubench/fetch_calib.hip) checked that on this chip for 4 / 8 / 16
bytes per lane and for reads that use only half or a quarter of each line: every touched line is ONE 128-byte
request whatever part of it is used, always tallied at 64 bytes; WRITE_SIZE is exact.
`step`: all kernels of one frame step of the batch summed (launch counts divided by the steps the profile ran:
k_unpack runs once per step) -- what bench.py's `pipeline` object prices against the step time.
If the SQ summaries are given too, per-launch instruction counts and (second file: the pass with
SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE) the VALU busy fraction by the counters are added.
usage: make_pmc_traffic.py <pmc_hbm_per_kernel.csv> <gops> <out.json> [pmc_sq_per_kernel.csv [pmc_roof_per_kernel.csv]]"""
import csv
import hashlib
import json
import os
import re
import sys


def csrc_sha16(root):
    """what the counters were taken on: sha256 over the kernel / shim sources (the GPU box has no .git).  bench.py computes the same
    over its own tree and says `traffic_stale` when they differ."""
    d = os.path.join(root, "digital-subband-video-1_amd", "csrc")
    names = sorted(n for n in os.listdir(d) if n.endswith((".hip", ".hpp")) or n == "Makefile")
    h = hashlib.sha256()
    for n in names + [os.path.join("..", "..", "include", "dsvg_rc.h")]:
        h.update(os.path.basename(n).encode() + b"\0" + open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]



def kid(name):
    """rocprofv3 prints k_hme_level<true, 12, 1> (second argument: rows per lane of a full block, picked per geometry by the
    launcher; third: 1 = the launch over the full blocks, 2 = the partial blocks at the frame's edge); the profiling API of
    the library and bench.py name the kernel by its first argument only"""
    name = re.sub(r"^void (k_tail_q|k_hz_scan|k_hz_collect_list|k_hz_emit_list)<\d+>$", r"\1", name)                             # (round 4: templated on the workgroup size the launcher picks)
    name = re.sub(r"k_hme_level<(true|false), \d+, [013]>", r"k_hme_level<\1>", name)          # the full blocks (or every block)
    return re.sub(r"k_hme_level<(true|false), \d+, 2>", r"k_hme_level<\1> (partial blocks)", name)


def main_variant(rows):
    """kernels templated on a size the launcher picks per batch (k_tail_q, k_hz_scan, the list kernels) show up once per variant;
    the timed batch's variant is the one with the most launches (the bench's small check batches take the other): keep that row"""
    best = {}
    for r in rows:
        k = kid(r["kernel"])
        if k not in best or int(r["launches"]) > int(best[k]["launches"]):
            best[k] = r
    return list(best.values())


rows = main_variant(list(csv.DictReader(open(sys.argv[1]))))
out = {"gops": int(sys.argv[2]),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --gops %s --steps 1`; "
                 "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, read side doubled: FETCH_SIZE tallies each 128-byte request at 64 bytes "
                 "on gfx950 (checked for every access shape used here: profiles/r03_fetch_calib.txt)" % sys.argv[2],
       "kernels": {}}
for r in rows:
    f = float(r.get("FETCH_SIZE_per_launch", 0) or 0)
    w = float(r.get("WRITE_SIZE_per_launch", 0) or 0)
    out["kernels"][kid(r["kernel"])] = {"launches": int(r["launches"]), "fetch_kib_per_launch": f, "write_kib_per_launch": w,
                                   "hbm_bytes_per_launch": round((2 * f + w) * 1024),
                                   "hbm_bytes_per_launch_raw": round((f + w) * 1024)}
if len(sys.argv) > 4:
    for r in main_variant(list(csv.DictReader(open(sys.argv[4])))):
        e = out["kernels"].setdefault(kid(r["kernel"]), {"launches": int(r["launches"])})
        e["valu_insts_per_launch"] = float(r.get("SQ_INSTS_VALU_per_launch", 0) or 0)
        e["salu_insts_per_launch"] = float(r.get("SQ_INSTS_SALU_per_launch", 0) or 0)
        e["waves_per_launch"] = float(r.get("SQ_WAVES_per_launch", 0) or 0)
if len(sys.argv) > 5:
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs:
    # busy fraction of the 1024 SIMDs = 4 * ACTIVE_INST_VALU / (128 * GUI_ACTIVE)
    for r in main_variant(list(csv.DictReader(open(sys.argv[5])))):
        e = out["kernels"].setdefault(kid(r["kernel"]), {"launches": int(r["launches"])})
        g = float(r.get("GRBM_GUI_ACTIVE_per_launch", 0) or 0)
        if g > 0:
            e["valu_busy_by_counters"] = round(4.0 * float(r.get("SQ_ACTIVE_INST_VALU_per_launch", 0) or 0) / (128.0 * g), 4)
            e["salu_busy_by_counters"] = round(4.0 * float(r.get("SQ_ACTIVE_INST_SCA_per_launch", 0) or 0) / (128.0 * g), 4)
            e["gui_active_cycles_per_launch"] = g / 8.0
steps = max(1, out["kernels"].get("k_unpack", {}).get("launches", 1))
tot = sum(e.get("hbm_bytes_per_launch", 0) * e["launches"] for k, e in out["kernels"].items() if not k.startswith("__amd") and k != "k_spin")
raw = sum(e.get("hbm_bytes_per_launch_raw", 0) * e["launches"] for k, e in out["kernels"].items() if not k.startswith("__amd") and k != "k_spin")
out["step"] = {"steps_profiled": steps, "hbm_bytes": round(tot / steps), "hbm_bytes_raw": round(raw / steps)}
# the step against the VALU issue roof (bench.py `pipeline.valu`): wave64 instructions of every kernel of a step
vi = sum(e.get("valu_insts_per_launch", 0.0) * e["launches"] for k, e in out["kernels"].items() if not k.startswith("__amd") and k != "k_spin")
if vi:
    out["step"]["valu_insts"] = round(vi / steps)
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out["csrc_sha16"] = csrc_sha16(_root)
out["commit"] = os.environ.get("DSV1_COMMIT", "unknown (set DSV1_COMMIT=$(git rev-parse --short HEAD) in the gpurun command)")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
