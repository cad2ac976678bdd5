#!/usr/bin/env python3
"""Frames/s of ONE 1080p 4:2:0 GOP=12 CRF stream WITH SCENE CUTS (CLI defaults otherwise: -scd1) through the reference's
frame-at-a-time API on the GPU library:
  * the reference's own CLI (dsv_main.c, unmodified) linked against the reference objects (oracle/_ref/dsv1) and against
    libdsv1_mi355x.so (oracle/_ref/dsv1_dropin): file I/O on a tmpfs, process start and context creation included;
  * the same dsv_enc calls from a driver that holds the frames in host memory ("frames pre-loaded in host RAM"):
    what a library caller sees -- GOP-parallel chain mode (default lookahead), and frame-serial (DSV1_ENC_PIPELINE=0).
The stream is checked against the reference CLI's bytes.   usage: dropin_fps.py [frames=1536] [ref_frames=24]   (marginal rates: N against N/2 frames, so that start-up -- context creation,
the first batch's buffer allocations -- cancels)"""
import ctypes as C
import importlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import _cabi as A

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
NREF = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W, H, FMT = 1920, 1080, A.SUBSAMP_420
DROPIN = os.path.join(A.ROOT, "oracle", "_ref", "dsv1_dropin")
base = A.gen_clip(W, H, FMT, 0x10800333, 48, style=5)          # scene cuts every 7 frames (and at every repetition)
clip = np.concatenate([base] * ((N + 47) // 48), axis=0)[:N]
env = dict(os.environ)
env["LD_LIBRARY_PATH"] = A.PKG_DIR + ":" + env.get("LD_LIBRARY_PATH", "")
flags = ["-w%d" % W, "-h%d" % H, "-fmt2", "-gop12", "-qp85", "-rc_mode1"]
cli = dict(qp=85, gop=12, rc_mode_cli=1)


def drive(pkg, frames, extra_env):
    """dsv_enc frame by frame from host memory (the caller's picture buffer is one frame, reused: dsv_main.c:506-520)"""
    for k, v in extra_env.items():
        os.environ[k] = v
    L = pkg.lib()
    enc = pkg.make_encoder_cfg(W, H, FMT, **cli)
    L.dsv_enc_start(C.byref(enc))
    L.dsv_load_planar_frame.restype = C.c_void_p
    L.dsv_load_planar_frame.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsv_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    bufs = (pkg.Buf * 4)()
    out = []
    per_call = []
    t0 = time.perf_counter()
    for t in range(frames.shape[0]):
        fr = L.dsv_load_planar_frame(FMT, frames[t].ctypes.data, W, H)
        tc = time.perf_counter()
        nb = L.dsv_enc(C.byref(enc), fr, bufs) & 3
        per_call.append(time.perf_counter() - tc)
        for i in range(nb):
            out.append(C.string_at(bufs[i].data, bufs[i].len))
            L.dsv_buf_free(C.byref(bufs[i]))
    L.dsv_enc_end_of_stream(C.byref(enc), bufs)
    out.append(C.string_at(bufs[0].data, bufs[0].len))
    L.dsv_buf_free(C.byref(bufs[0]))
    dt = time.perf_counter() - t0
    L.dsv_enc_free(C.byref(enc))
    for k in extra_env:
        os.environ.pop(k, None)
    pc = np.sort(np.array(per_call))
    drive.last = "dsv_enc calls: median %.0f us, p90 %.0f us, max %.1f ms, sum of the %d longest %.1f ms of %.1f ms" % (
        1e6 * pc[len(pc) // 2], 1e6 * pc[int(len(pc) * 0.9)], 1e3 * pc[-1], max(1, len(pc) // 100), 1e3 * pc[-max(1, len(pc) // 100):].sum(), 1e3 * dt)
    return dt, b"".join(out)


td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    inp = os.path.join(td, "in.yuv")
    clip.tofile(inp)

    def run(binary, out, nfr, extra_env=None):
        e = dict(env)
        e.update(extra_env or {})
        t0 = time.perf_counter()
        r = subprocess.run([binary, "e", "-y", "-inp_" + inp, "-out_" + os.path.join(td, out), "-nfr%d" % nfr] + flags, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=e)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, binary
        return dt, open(os.path.join(td, out), "rb").read()

    have_cli = os.path.exists(DROPIN) and os.path.exists(A.REF_CLI)
    if have_cli:
        run(DROPIN, "warm.dsv", 12)                          # page the library in
        t_ref, s_ref = run(A.REF_CLI, "ref.dsv", NREF)
        t_gpu, s_gpu = run(DROPIN, "gpu.dsv", N)
        t_gpu4, _ = run(DROPIN, "gpu4.dsv", N // 2)
        t_ser, s_ser = run(DROPIN, "ser.dsv", N // 4, {"DSV1_ENC_PIPELINE": "0"})
        t_ser4, _ = run(DROPIN, "ser4.dsv", N // 16, {"DSV1_ENC_PIPELINE": "0"})
        print("reference CLI                      : %4d frames in %6.2f s = %8.1f frames/s" % (NREF, t_ref, NREF / t_ref))
        print("drop-in CLI, GOP-parallel lookahead: %4d frames in %6.2f s = %8.1f frames/s (process start, context creation, file I/O included)" % (N, t_gpu, N / t_gpu))
        print("drop-in CLI, frame-serial          : %4d frames in %6.2f s = %8.1f frames/s" % (N // 4, t_ser, (N // 4) / t_ser))
        print("marginal CLI rate (start-up cancels): GOP-parallel %.1f frames/s, frame-serial %.1f frames/s" %
              ((N - N // 2) / (t_gpu - t_gpu4), (N // 4 - N // 16) / (t_ser - t_ser4)))
        print("CLI streams: first %d frames == reference CLI: %s; frame-serial prefix equal: %s" %
              (NREF, s_gpu[:len(s_ref) - 14] == s_ref[:len(s_ref) - 14], s_gpu[:len(s_ser) - 14] == s_ser[:len(s_ser) - 14]))
    pkg = importlib.import_module("digital-subband-video-1_amd")
    drive(pkg, clip[:24], {})                                # context creation paged in
    t_par4, _ = drive(pkg, clip[:N // 2], {})
    t_par, s_par = drive(pkg, clip, {})
    par_calls = drive.last
    t_one, s_one = drive(pkg, clip[:N // 4], {"DSV1_ENC_PIPELINE": "0"})
    t_one4, _ = drive(pkg, clip[:N // 16], {"DSV1_ENC_PIPELINE": "0"})
    print("    GOP-parallel", par_calls)
    print("    frame-serial", drive.last)
    print("dsv_enc from host memory, GOP-parallel: %4d frames in %6.2f s = %8.1f frames/s; marginal %.1f frames/s" % (N, t_par, N / t_par, (N - N // 2) / (t_par - t_par4)))
    print("dsv_enc from host memory, frame-serial: %4d frames in %6.2f s = %8.1f frames/s; marginal %.1f frames/s" % (N // 4, t_one, (N // 4) / t_one, (N // 4 - N // 16) / (t_one - t_one4)))
    print("library streams: GOP-parallel prefix == frame-serial: %s%s" % (s_par[:len(s_one) - 14] == s_one[:len(s_one) - 14],
          ("; == CLI stream: %s" % (s_par == s_gpu)) if have_cli else ""))
finally:
    import shutil
    shutil.rmtree(td, ignore_errors=True)
