#!/usr/bin/env python3
"""Open / encode / close many batch encoders and watch device + host memory."""
import importlib, os, sys, resource
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _cabi as A
import torch
pkg = importlib.import_module("digital-subband-video-1_amd")
W, H, FMT = 352, 288, 5
clip = A.gen_clip(W, H, FMT, 0x1234, 6, style=2)
def mem():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for i in range(60):
    s = pkg.encode_clip(clip, W, H, FMT, qp=85, gop=12, rc_mode_cli=1)
    if i in (4, 59):
        d, h = mem()
        print("iter %2d: device used %.0f MiB, host max RSS %.0f MiB, stream %d bytes" % (i, d, h, len(s)))

# wider batches (two coding streams, intra-block lists, pinned ingest) and the batched decoder, opened and closed repeatedly
S, GOP = 16, 4
clips = np.stack([A.gen_clip(W, H, FMT, 0x2200 + s, GOP, style=s % 3) for s in range(S)])
cfg = pkg.make_encoder_cfg(W, H, FMT, qp=85, gop=GOP, rc_mode_cli=1)
for i in range(40):
    b = pkg.Batch(cfg, S, GOP)
    host = b.pinned(clips.shape)
    host[...] = clips
    b.stage(host)
    b.submit(host)
    out = b.collect()
    b.close()
    pk = [A.split_packets(bytes(o)) for o in out]
    d = pkg.DecBatch(W, H, FMT, S)
    for k in range(len(pk[0])):
        d.decode([p[k] for p in pk], on_device=(k & 1) == 0)
    d.close()
    if i in (4, 39):
        dm, h = mem()
        print("wide iter %2d: device used %.0f MiB, host max RSS %.0f MiB" % (i, dm, h))
