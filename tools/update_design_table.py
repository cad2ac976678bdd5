#!/usr/bin/env python3
"""copy the evidence of tools/collect_profiles.sh <tag> (gpurun_out/<tag>/...) into profiles/ and rewrite DESIGN.md's round-3 table and headline
numbers from it (so that the table is the committed run's, not typed).  usage: update_design_table.py [tag=r03f]"""
import json, os, re, shutil, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03f"
g = os.path.join(ROOT, "gpurun_out")
for f in ("bench.json", "kernel_trace_summary.txt", "pmc_hbm_per_kernel.csv", "pmc_roof_per_kernel.csv", "pmc_sq_per_kernel.csv", "rocprofv3_kernel_stats.csv",
          "rocprofv3_kernel_stats_one_coding_stream.csv"):
    shutil.copy(os.path.join(g, tag, f), os.path.join(ROOT, "profiles", "%s_%s" % (tag, f)))
shutil.copy(os.path.join(g, tag, "pmc_traffic.json"), os.path.join(ROOT, "profiles", "pmc_traffic.json"))
for src, dst in (("%s_dropin_fps.txt" % tag,) * 2, ("%s_dropin_abr_fps.txt" % tag,) * 2, ("%s_decode_time.txt" % tag,) * 2,
                 (os.path.join("%s_dec" % tag, "decode_kernel_trace_summary.txt"), "%s_decode_kernel_trace_summary.txt" % tag),
                 (os.path.join("%s_dec" % tag, "decode.json"), "%s_decode_result.json" % tag)):
    if os.path.exists(os.path.join(g, src)):
        shutil.copy(os.path.join(g, src), os.path.join(ROOT, "profiles", dst))
T = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
d = json.load(open(os.path.join(ROOT, "profiles", "%s_bench.json" % tag)))
ms = d["roofline"]["all_kernels_ms_one_step"]


def f(name):
    v = T["kernels"][name]; m = ms[name]
    gb = v["hbm_bytes_per_launch"] * (v["launches"] / T["step"]["steps_profiled"]) / 1e9
    return m, gb, v.get("valu_busy_by_counters"), v.get("salu_busy_by_counters"), round(v["valu_insts_per_launch"] / v["waves_per_launch"])


out = []
m, gb, va, sa, iw = f("void k_hme_level<true>")
out.append(f"| `k_hme_level<true>` | {m:.2f} ({m/2:.2f}; 3.27 on the fastest box seen; `r03z` 4.41) | {gb:.1f} GB → {gb/m:.1f} (14.6 GB algorithmic at 2 B/luma-px) | {va:.2f} (scalar unit {sa:.2f}) | {iw} (1 227) |")
for name, lab in (("void k_inv_p_tile<true>", "`k_inv_p_tile<true>`"), ("void k_fwd_mc_fast<0>", "`k_fwd_mc_fast<0>`"), ("k_unpack", "`k_unpack`"),
                  ("void k_fwd_mc_fast<1>", "`k_fwd_mc_fast<1>`"), ("k_inv_patch_c", "`k_inv_patch_c`"), ("void k_inv_b4t<true>", "`k_inv_b4t<true>`")):
    m, gb, va, sa, iw = f(name)
    vs = f"**{va:.2f}**" if va > 0.8 else f"{va:.2f}"
    if name == "k_unpack": vs += f" (scalar {sa:.2f})"
    out.append(f"| {lab} | {m:.2f} ({m/2:.2f}) | {gb:.1f} GB → {gb/m:.1f} | {vs} | {iw} |")
m, gb, va, sa, iw = f("k_extend16"); out.append(f"| `k_extend16` | {m:.2f} ({m/2:.2f}) | {gb:.1f} GB | | |")
m, gb, va, sa, iw = f("void k_hme_level<false>"); out.append(f"| `k_hme_level<false>` | {m:.2f} ({m/2:.2f}) | {gb:.1f} GB | {va:.2f} (scalar {sa:.2f}) | |")
m, gb, va, sa, iw = f("k_hme_csum"); out.append(f"| `k_hme_csum` (new) | {m:.2f} ({m/2:.2f}) | {gb:.1f} GB → {gb/m:.1f} | {va:.2f} | |")
m1, gb1, va1, _, _ = f("k_hz_collect_list"); m2, gb2, va2, _, _ = f("k_hz_emit_list")
out.append(f"| `k_hz_collect_list` + `k_hz_emit_list` | {m1:.2f} + {m2:.2f} | {gb1+gb2:.1f} GB | {va1:.2f} / {va2:.2f} | |")
tot = sum(ms.values()); hb = T["step"]["hbm_bytes"] / 1e9
out.append(f"| sum of all kernels | **{tot:.1f}** ({tot/2:.1f}; 40.5 on the fastest box seen; `r03z` 21.9, round 2 23.0) | **{hb:.1f} GB** ({hb/2:.1f} per 160 GOPs; `r03z` 77.3) | | |")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("| `k_hme_level<true>` | ", s.index("| Kernel, `r03f`"))
b = s.index("**The whole step against the HBM roof**")
s = s[:a] + "\n".join(out) + "\n\n" + s[b:]
s = re.sub(r"\(`profiles/r03f_bench.json`: [0-9.]+, [0-9.]+ ms per", "(`profiles/r03f_bench.json`: %.1f, %.2f ms per" % (d["value"] / 1000, d["ms_per_step"]), s)
s = re.sub(r"one core of an EPYC 9575F, [0-9.]+ Mpix/s\.  Bit-exact against the reference on 16 streams in the bench itself\.  Input in pinned",
           "one core of an EPYC 9575F, %.1f Mpix/s.  Bit-exact against the reference on 16 streams in the bench itself.  Input in pinned" % d["cpu_baseline"]["value"], s)
s = re.sub(r"[0-9.]+ GB of counter bytes over [0-9.]+ ms = [0-9.]+ TB/s", "%.1f GB of counter bytes over %.2f ms = %.2f TB/s" % (hb, d["ms_per_step"], hb / d["ms_per_step"]), s)
sh = d["shapes"]
s = re.sub(r"`r03f`: config 2 [0-9.]+, config 4 [0-9.]+, config 5 [0-9.]+, worst case \*\*[0-9.]+\*\* at",
           "`r03f`: config 2 %.1f, config 4 %.1f, config 5 %.1f, worst case **%.1f** at" % (sh["cfg2_1080p_intra"]["Mpix_s"] / 1e3, sh["cfg4_4k_gop12"]["Mpix_s"] / 1e3,
                                                                                         sh["cfg5_4k_444_abr"]["Mpix_s"] / 1e3, sh["cfg3_worstcase"]["Mpix_s"] / 1e3), s)
s = re.sub(r"batched decoder [0-9.]+ Gpix/s;", "batched decoder %.1f Gpix/s;" % (sh["decode_1080p_batched"]["Mpix_s"] / 1e3), s)
open(p, "w").write(s)
print("value %.1f, kernel sum %.1f ms, %.1f GB/step, shapes %s" % (d["value"] / 1000, tot, hb, {k: round(v.get("Mpix_s", 0) / 1e3, 1) for k, v in sh.items()}))
