#!/usr/bin/env python3
"""How long the host spends inside dsv1_batch_submit / dsv1_batch_collect per step (64 GOPs x 12 frames, 1080p):
if submit + collect approaches the GPU's step time the session layer, not the kernels, bounds the throughput."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
W, H, FMT, GOP, QP = 1920, 1080, 5, 12, 85
gops = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fb = A.frame_bytes(W, H, FMT)
clip = A.gen_clip(W, H, FMT, 0x10800003, GOP, style=0)
batch_in = np.empty((gops, GOP, fb), dtype=np.uint8)
batch_in[:] = clip
cfg = pkg.make_encoder_cfg(W, H, FMT, qp=QP, gop=GOP, rc_mode_cli=1)
b = pkg.Batch(cfg, gops, GOP, device=0)
d = b.upload(batch_in)
b.encode(d, on_device=True); b.encode(d, on_device=True)
b.submit(d, on_device=True, held=True); b.sync()
ts, tc = [], []
t0 = time.perf_counter(); c0 = time.process_time()
for _ in range(8):
    a = time.perf_counter(); b.submit(d, on_device=True, held=True); c = time.perf_counter(); b.collect(); e = time.perf_counter()
    ts.append(c - a); tc.append(e - c)
b.sync(); dt = (time.perf_counter() - t0) / 8; cpu = (time.process_time() - c0) / 8
print("step %.2f ms wall, %.2f ms CPU time of this process; inside submit %.2f ms, collect %.2f ms (both include waiting for the GPU)" % (1e3 * dt, 1e3 * cpu, 1e3 * np.mean(ts), 1e3 * np.mean(tc)))
