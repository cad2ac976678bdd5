#!/usr/bin/env python3
"""Throughput of the batched encoder on an arbitrary shape (the other BASELINE.json configs):
   bench_shape.py W H FMT(0=444,1=422,2=420,3=411) GOPS GOP QP RC(1=CRF,0=ABR) [kbps] [steps]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
W, H, fcli, gops, gop, qp, rc = [int(x) for x in sys.argv[1:8]]
kbps = int(sys.argv[8]) if len(sys.argv) > 8 else 0
steps = int(sys.argv[9]) if len(sys.argv) > 9 else 4
FMT = {0: A.SUBSAMP_444, 1: A.SUBSAMP_422, 2: A.SUBSAMP_420, 3: A.SUBSAMP_411}[fcli]
F = gop if gop > 0 else 12
fb = A.frame_bytes(W, H, FMT)
clip = A.gen_clip(W, H, FMT, 0x21600004, F, style=0)
batch_in = np.empty((gops, F, fb), dtype=np.uint8)
batch_in[:] = clip
kw = dict(qp=qp, gop=gop, rc_mode_cli=rc)
if kbps:
    kw["kbps"] = kbps
cfg = pkg.make_encoder_cfg(W, H, FMT, **kw)
b = pkg.Batch(cfg, gops, F, device=0)
d = b.upload(batch_in)
b.encode(d, on_device=True)
t0 = time.perf_counter()
if rc == 1 or os.environ.get('DSV1_ABR_SERIAL', '0') in ('', '0'):      # ABR pipelines too since round 4 (rate control on the device)
    b.submit(d, on_device=True, held=True)
    for _ in range(steps):
        b.submit(d, on_device=True, held=True); outs = b.collect(copy=False)
    b.collect(copy=False)
    n = steps + 1
else:
    for _ in range(steps):
        outs = b.encode(d, on_device=True)
    n = steps
b.sync()
dt = time.perf_counter() - t0
pix = n * gops * F * W * H
print("%dx%d fmt%d gop%d qp%d %s: %d streams x %d frames per step, %.2f ms/step, %.1f Gpix/s, %.0f frames/s, %d bytes/step" % (
    W, H, fcli, gop, qp, "CRF" if rc else "ABR", gops, F, 1e3 * dt / n, pix / dt / 1e9, n * gops * F / dt, sum(len(o) for o in outs)))
if os.environ.get("SHAPE_PROF"):
    names = b.kernel_names()
    b.prof_enable(names)
    b.encode(d, on_device=True); b.sync()
    t = {k: b.prof_get(k)[0] for k in names}
    print("kernel ms per step (sum %.2f):" % sum(t.values()))
    for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:14]:
        print("  %-44s %7.3f" % (k, v))
b.close()
