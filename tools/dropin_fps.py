#!/usr/bin/env python3
"""Frames/s of the reference's own CLI (dsv_main.c, unmodified) on a 1080p 4:2:0 GOP=12 CRF clip: linked against the
reference objects (oracle/_ref/dsv1) and against libdsv1_mi355x.so (oracle/_ref/dsv1_dropin; the frame-at-a-time
dsv_enc behind it is pipelined in GOP batches, DSV1_ENC_PIPELINE=0 switches that off).  File I/O on a tmpfs included.
usage: dropin_fps.py [frames=96] [ref_frames=24]"""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import _cabi as A
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
NREF = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W, H, FMT = 1920, 1080, A.SUBSAMP_420
DROPIN = os.path.join(A.ROOT, "oracle", "_ref", "dsv1_dropin")
gop = A.gen_clip(W, H, FMT, 0x10800003, 12, style=0)
clip = np.concatenate([gop] * (N // 12), axis=0)
env = dict(os.environ)
env["LD_LIBRARY_PATH"] = A.PKG_DIR + ":" + env.get("LD_LIBRARY_PATH", "")
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    inp = os.path.join(td, "in.yuv")
    clip.tofile(inp)
    flags = ["-y", "-inp_" + inp, "-w%d" % W, "-h%d" % H, "-fmt2", "-gop12", "-qp85", "-rc_mode1"]

    def run(binary, out, nfr, extra_env=None):
        e = dict(env)
        e.update(extra_env or {})
        t0 = time.perf_counter()
        r = subprocess.run([binary, "e", "-out_" + os.path.join(td, out), "-nfr%d" % nfr] + flags, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=e)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, binary
        return dt, open(os.path.join(td, out), "rb").read()

    run(DROPIN, "warm.dsv", 12)                          # page the library in
    t_ref, s_ref = run(A.REF_CLI, "ref.dsv", NREF)
    t_gpu, s_gpu = run(DROPIN, "gpu.dsv", N)
    t_ser, s_ser = run(DROPIN, "ser.dsv", N, {"DSV1_ENC_PIPELINE": "0"})
    t_gpu4, _ = run(DROPIN, "gpu4.dsv", N // 4)
    t_ser4, _ = run(DROPIN, "ser4.dsv", N // 4, {"DSV1_ENC_PIPELINE": "0"})
    _, s_ref_full = (0, None)
    print("reference CLI            : %4d frames in %6.2f s = %7.1f frames/s" % (NREF, t_ref, NREF / t_ref))
    print("drop-in CLI, pipelined   : %4d frames in %6.2f s = %7.1f frames/s (process start, context creation and file I/O included)" % (N, t_gpu, N / t_gpu))
    print("drop-in CLI, frame-serial: %4d frames in %6.2f s = %7.1f frames/s" % (N, t_ser, N / t_ser))
    print("marginal rate (N vs N/4 frames, fixed start-up cost cancels): pipelined %.1f frames/s, frame-serial %.1f frames/s" %
          ((N - N // 4) / (t_gpu - t_gpu4), (N - N // 4) / (t_ser - t_ser4)))
    print("streams equal (pipelined == frame-serial): %s; first %d frames == reference CLI: %s" %
          (s_gpu == s_ser, NREF, s_gpu[:len(s_ref) - 14] == s_ref[:len(s_ref) - 14]))
finally:
    import shutil
    shutil.rmtree(td, ignore_errors=True)
