#!/usr/bin/env python3
"""Frames/s of the drop-in decoder (dsv_dec: one picture per call, host sync per picture) on a 1080p GOP=12 stream."""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
L = pkg.lib()
W, H, FMT, N = 1920, 1080, A.SUBSAMP_420, 48
clip = A.gen_clip(W, H, FMT, 0x10800003, 12, style=0)
clip = np.concatenate([clip] * (N // 12), axis=0)
stream = pkg.encode_clip(clip, W, H, FMT, qp=85, gop=12, rc_mode_cli=1)
pk = A.split_packets(stream)


class Decoder(C.Structure):
    _fields_ = [("vidmeta", A.Meta), ("ref", C.c_void_p), ("draw_info", C.c_int), ("got_metadata", C.c_int)]


L.dsv_alloc.restype = C.c_void_p
L.dsv_dec.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
L.dsv_frame_ref_dec.argtypes = [C.c_void_p]
L.dsv_frame_ref_dec.restype = None
for rep in range(3):
    dec = Decoder()
    # the packets as a C caller has them: in dsv_alloc'd buffers (dsv_dec frees them), prepared before the clock starts
    bufs = []
    LOOPS = 10                                         # the stream ten times over through ONE decoder (its end-of-stream packet left out): the
    for p in [q for q in pk if q[5] != 0x10] * LOOPS:  # session's set-up (context, pinned frames: ~15 ms) is not what the figure is about
        buf = pkg.Buf()
        mem = L.dsv_alloc(len(p)); C.memmove(mem, p, len(p))
        buf.data = C.cast(mem, C.POINTER(C.c_uint8)); buf.len = len(p)
        bufs.append(buf)
    frame = C.c_void_p(None); fn = C.c_uint32(0)
    dref, bref, fref, nref = C.byref(dec), [C.byref(b) for b in bufs], C.byref(frame), C.byref(fn)
    t0 = time.perf_counter(); n = 0
    for br in bref:
        frame.value = None
        rc = L.dsv_dec(dref, br, fref, nref)
        if rc == 0 and frame.value:
            n += 1; L.dsv_frame_ref_dec(frame)
    dt = time.perf_counter() - t0
    L.dsv_dec_free(C.byref(dec))
print("dsv_dec, one picture per call, host frame out: %d frames 1920x1080 decoded in %.3f s: %.0f frames/s, %.2f Gpix/s (stream %d bytes)" % (n, dt, n / dt, n * W * H / dt / 1e9, len(stream)))

# ---- batched decoder (dsv1_decbatch_*): S copies of the stream side by side, one packet per stream per call ----
S = int(os.environ.get("DEC_STREAMS", "64"))
for on_device in (1, 0):
    d = pkg.DecBatch(W, H, FMT, S)
    keep = [np.frombuffer(bytes(p) + b"\0" * 16, dtype=np.uint8).copy() for p in pk]
    calls = []
    for k, p in enumerate(pk):
        bufs = (pkg.Buf * S)()
        for s in range(S):
            bufs[s].data = keep[k].ctypes.data_as(C.POINTER(C.c_uint8)); bufs[s].len = len(p)
        calls.append(bufs)
    status = (C.c_int * S)(); fnum = (C.c_uint32 * S)()
    if on_device:
        dst = C.c_void_p(None)
        assert L.dsvg_dev_alloc(d.ctx, C.byref(dst), d.frame_bytes * S) == 0
    else:
        b = pkg.Batch(pkg.make_encoder_cfg(W, H, FMT), 1, 1)
        host = b.pinned((S, d.frame_bytes))
        dst = C.c_void_p(host.ctypes.data)
    for rep in range(2):
        d.sync()
        t0 = time.perf_counter(); n = 0
        for bufs in calls:
            rc = L.dsv1_decbatch_decode(d.h, bufs, dst, d.frame_bytes, on_device, status, fnum)
            assert rc == 0, L.dsvg_last_error()
            n += sum(1 for s in range(S) if status[s] == 0 and fnum[s] != 0xFFFFFFFF)
        d.sync()
        dt = time.perf_counter() - t0
    print("batched, %d streams, output %s: %d frames in %.3f s: %.0f frames/s, %.2f Gpix/s" %
          (S, "left in HBM" if on_device else "copied to pinned host memory", n, dt, n / dt, n * W * H / dt / 1e9))
    if on_device:
        L.dsvg_dev_free(d.ctx, dst)
    d.close()
