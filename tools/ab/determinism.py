#!/usr/bin/env python3
"""encode the same four clips as S streams side by side, frame step by frame step, R times over: every stream must
equal its twin (stream % 4) and every repetition the first one.  usage: determinism.py [streams] [frames] [reps] [frames per batch]"""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
F = int(sys.argv[4]) if len(sys.argv) > 4 else 1
W, H, FMT = 1920, 1080, A.SUBSAMP_420
clips = [A.gen_clip(W, H, FMT, 0x10800003 + g, N, style=0) for g in range(4)]
first = None
nbad = 0
for r in range(R):
    b = pkg.Batch(pkg.make_encoder_cfg(W, H, FMT, qp=85, gop=12, rc_mode_cli=1), S, F)
    sig = []
    for t in range(0, N, F):
        fr = np.stack([clips[s % 4][t:t + F] for s in range(S)]).reshape(S, F, -1)
        pk = b.encode(fr)
        bad = [s for s in range(S) if pk[s] != pk[s % 4]]
        if bad:
            nbad += 1
            print("rep %d frame %d: streams %s differ from their twins (length differences %s)" % (r, t, bad[:12], [len(pk[s]) - len(pk[s % 4]) for s in bad[:12]]))
        sig.append(hashlib.md5(b"".join(pk[:4])).hexdigest()[:8])
    b.close()
    if first is None: first = sig
    elif sig != first:
        nbad += 1
        print("rep %d differs from rep 0: %s vs %s" % (r, sig, first))
print("determinism: %d streams x %d frames (batches of %d) x %d reps: %s" % (S, N, F, R, "OK" if not nbad else "%d PROBLEMS" % nbad), first)
