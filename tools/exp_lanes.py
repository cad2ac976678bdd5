#!/usr/bin/env python3
"""Experiment: split the GOPs of a step over several independent batches (own HIP streams) and interleave them."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
W, H, FMT, GOP, QP = 1920, 1080, A.SUBSAMP_420 if hasattr(A, "SUBSAMP_420") else 5, 12, 85
gops = int(sys.argv[1]); lanes = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
fb = A.frame_bytes(W, H, FMT)
clip = A.gen_clip(W, H, FMT, 0x10800003, GOP, style=0)
per = gops // lanes
batch_in = np.empty((per, GOP, fb), dtype=np.uint8)
for s in range(per):
    batch_in[s] = clip
cfg = pkg.make_encoder_cfg(W, H, FMT, qp=QP, gop=GOP, rc_mode_cli=1)
bs = [pkg.Batch(cfg, per, GOP, device=0) for _ in range(lanes)]
dp = [b.upload(batch_in) for b in bs]
for b, d in zip(bs, dp):
    b.encode(d, on_device=True); b.encode(d, on_device=True)
for b, d in zip(bs, dp):
    b.submit(d, on_device=True)
for b in bs: b.sync()
t0 = time.perf_counter()
for _ in range(steps):
    for b, d in zip(bs, dp):
        b.submit(d, on_device=True)
        outs = b.collect()
for b in bs: b.sync()
dt = time.perf_counter() - t0
print("gops %d lanes %d: %.3f ms/step  %.1f Gpix/s" % (gops, lanes, 1000 * dt / steps, gops * GOP * W * H * steps / dt / 1e9))
