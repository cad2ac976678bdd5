#!/usr/bin/env python3
"""Where a kernel waits for memory between its loads: for every kernel of an assembly listing (hipcc -S --cuda-device-only), the
sequence of vector-memory load groups and s_waitcnt vmcnt(...) up to the first store / barrier -- a 'L3 w2 L4 w0 L8' line shows
loads that were meant to be in flight together but are serialised by an early use of a loaded value (round 4: k_inv_p_tile's flag
classes, k_fwd_mc_fast's short-circuit tests).  usage: asm_waits.py file.s [kernel-name-substring]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
name, seq, done = None, [], False
def flush():
    if name and seq and pat in name:
        print(name[:70]); print("   ", " ".join(seq))
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        flush(); name, seq, done = m.group(1), [], False
        continue
    if name is None or done:
        continue
    t = l.strip().split(" ")[0] if l.strip() else ""
    if t.startswith(("global_load", "flat_load", "buffer_load")):
        if seq and seq[-1].startswith("L"): seq[-1] = "L%d" % (int(seq[-1][1:]) + 1)
        else: seq.append("L1")
    elif t.startswith("s_load"):
        if seq and seq[-1].startswith("S"): seq[-1] = "S%d" % (int(seq[-1][1:]) + 1)
        else: seq.append("S1")
    elif t == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", l)
        k = re.search(r"lgkmcnt\((\d+)\)", l)
        seq.append(("w%s" % m.group(1)) if m else ("k%s" % k.group(1) if k else "w?"))
    elif t.startswith(("s_cbranch", "s_branch")):
        if seq and seq[-1] != "|": seq.append("|")
    elif t.startswith(("global_store", "flat_store", "s_barrier", "global_atomic", "buffer_store")):
        seq.append("#" + t.split("_")[1]); done = len([x for x in seq if x.startswith("#")]) >= 2
    elif t == "s_endpgm" and not seq:
        pass
flush()
