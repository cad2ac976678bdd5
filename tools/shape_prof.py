#!/usr/bin/env python3
"""Per-kernel GPU time of one shape, measured by the library's own HIP-event brackets (dsvg_prof_*: no profiler in the way -- under rocprofv3 a
4 us kernel shows as 25 us): bench_shape.py's arguments; prints launches, mean us per launch and ms per step for every kernel, and the step time
with and without the brackets."""
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
clip = A.gen_clip(W, H, FMT, 0x21600004, F, style=int(os.environ.get("SHAPE_STYLE", "0")))      # (SHAPE_STYLE=7: dense residuals)
batch_in = np.empty((gops, F, A.frame_bytes(W, H, FMT)), dtype=np.uint8)
batch_in[:] = clip
kw = dict(qp=qp, gop=gop, rc_mode_cli=rc)
if kbps:
    kw["kbps"] = kbps
b = pkg.Batch(pkg.make_encoder_cfg(W, H, FMT, **kw), gops, F, device=0)
d = b.upload(batch_in)
b.encode(d, on_device=True)


def loop(n):
    b.submit(d, on_device=True, held=True)
    b.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        b.submit(d, on_device=True, held=True); b.collect(copy=False)
    b.sync()
    dt = time.perf_counter() - t0
    b.collect(copy=False)
    return 1e3 * dt / n


plain = loop(steps)
names = b.kernel_names()
b.prof_enable(names)
brk = loop(steps)
rows = []
for k in names:
    ms, n, _ = b.prof_get(k)
    if n:
        rows.append((ms / (steps + 1), n / (steps + 1), 1e3 * ms / n, k))
b.prof_enable([])
b.close()
print("%dx%d fmt%d gop%d: %d streams x %d frames per step: %.2f ms per step plain, %.2f with every launch bracketed by events" % (W, H, fcli, gop, gops, F, plain, brk))
print("%9s %9s %9s  kernel" % ("ms/step", "launches", "us/launch"))
for r in sorted(rows, reverse=True):
    print("%9.3f %9.1f %9.1f  %s" % r)
print("sum %.3f ms per step" % sum(r[0] for r in rows))
