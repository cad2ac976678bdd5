#!/usr/bin/env python3
"""Frames/s of ONE 1080p 4:2:0 stream with the reference CLI's DEFAULT rate control (ABR, -scd1, GOP 12) through dsv_enc on the GPU
library, frames in host memory: gathered analysis (default) against one frame per call (DSV1_ENC_PIPELINE=0), both checked against
the oracle encoder's bytes.   usage: dropin_abr_fps.py [frames=768] [gathered|serial]"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import _cabi as A
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
W, H, FMT = 1920, 1080, A.SUBSAMP_420
cli = dict(qp=85, gop=12, rc_mode_cli=0, kbps=8000)
base = A.gen_clip(W, H, FMT, 0x10800444, 48, style=5)
clip = np.concatenate([base] * ((N + 47) // 48), axis=0)[:N]
pkg = importlib.import_module("digital-subband-video-1_amd")


def drive(frames, env):
    for k, v in env.items():
        os.environ[k] = v
    L = pkg.lib()
    enc = pkg.make_encoder_cfg(W, H, FMT, **cli)
    L.dsv_enc_start(C.byref(enc))
    L.dsv_load_planar_frame.restype = C.c_void_p
    L.dsv_load_planar_frame.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsv_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    bufs = (pkg.Buf * 4)()
    out = []
    t0 = time.perf_counter()
    for t in range(frames.shape[0]):
        fr = L.dsv_load_planar_frame(FMT, frames[t].ctypes.data, W, H)
        nb = L.dsv_enc(C.byref(enc), fr, bufs) & 3
        for i in range(nb):
            out.append(C.string_at(bufs[i].data, bufs[i].len))
            L.dsv_buf_free(C.byref(bufs[i]))
    L.dsv_enc_end_of_stream(C.byref(enc), bufs)
    out.append(C.string_at(bufs[0].data, bufs[0].len))
    L.dsv_buf_free(C.byref(bufs[0]))
    dt = time.perf_counter() - t0
    L.dsv_enc_free(C.byref(enc))
    for k in env:
        os.environ.pop(k, None)
    return dt, b"".join(out)


drive(clip[:40], {})
want, _ = A.orc_encode(clip[:60], A.orc_cfg(W, H, FMT, **cli))
MODES = (("gathered analysis (32 frames)", {}), ("one frame per call", {"DSV1_ENC_PIPELINE": "0"}))
if len(sys.argv) > 2:                 # "gathered" / "serial": one mode only (e.g. under DSV1_HOST_PROF=1, which reports the last session)
    MODES = MODES[:1] if sys.argv[2].startswith("g") else MODES[1:]
for name, env in MODES:
    t_half, _ = drive(clip[:N // 2], env)
    t_all, s_all = drive(clip, env)
    _, s60 = drive(clip[:60], env)
    print("dsv_enc ABR, %-30s: %4d frames in %5.2f s = %7.1f frames/s; marginal %7.1f frames/s; first 60 frames == oracle: %s" %
          (name, N, t_all, N / t_all, (N - N // 2) / (t_all - t_half), s60 == want))
