#!/usr/bin/env python3
"""Static instruction counts per stage of the level-0 motion search kernel (k_hme_level<true,12,1>), from a listing compiled with
-DAB_HME_ASMMARK (HME_MARK(i) leaves '; HMEMARK i' in the assembly).  The kernel is issue-bound on the SUM of its scalar and vector
instructions (tools/ab probes AB_HME_DUMMY_SALU / _VALU: 200 more of either cost the same 0.55-0.6 ms per 320-GOP step), so this is
the map of where the scalar half goes.  usage: hme_asm_regions.py [NKB]   (cross-compiles here, no GPU)"""
import collections, os, re, subprocess, sys
repo = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
nkb = sys.argv[1] if len(sys.argv) > 1 else "12"
lvl0, part = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("1", "1")      # [NKB [LEVEL0 PART]]: 12 0 3 = the upper levels' kernel
s_path = "/tmp/k_hme_mark.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-DAB_HME_ASMMARK",
                       os.path.join(repo, "digital-subband-video-1_amd/csrc/k_hme.hip"), "-o", s_path], stderr=subprocess.DEVNULL, cwd="/tmp")
name = "_Z11k_hme_levelILb%sELi%sELi%sEEv7HmeArgsiiiijj" % (lvl0, nkb, part)
lines = open(s_path).read().split("\n")
i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
region, cnt = "entry", collections.OrderedDict()
for l in lines[i0:i1]:
    m = re.search(r"; HMEMARK (\d+)", l)
    if m:
        region = "after mark " + m.group(1)
        continue
    m = re.match(r"\s+([a-z][a-z0-9_]+)", l)
    if not m:
        continue
    op = m.group(1)
    kind = ("wait/nop" if op in ("s_waitcnt", "s_nop") else "branch" if op.startswith("s_cbranch") or op == "s_branch" else
            "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem")
    cnt.setdefault(region, collections.Counter())[kind] += 1
print("%-16s %6s %6s %8s %6s %5s %5s" % ("region", "salu", "branch", "wait/nop", "valu", "lds", "vmem"))
tot = collections.Counter()
for r, c in cnt.items():
    print("%-16s %6d %6d %8d %6d %5d %5d" % (r, c["salu"], c["branch"], c["wait/nop"], c["valu"], c["lds"], c["vmem"]))
    tot.update(c)
print("%-16s %6d %6d %8d %6d %5d %5d" % ("total", tot["salu"], tot["branch"], tot["wait/nop"], tot["valu"], tot["lds"], tot["vmem"]))
