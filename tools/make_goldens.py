#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref, compiled from /root/reference).

Run in the build container only (the reference does not exist on the GPU box):
    make -C oracle ref && python tools/make_goldens.py

A fixture is data: inputs come from the committed integer generator (tools/clipgen/clipgen.c, regenerated at
test time from (w,h,fmt,seed,style)), expected outputs are produced here by the reference:
  * streams.json   -- for each stream case: CLI flags, sha256 + length of the whole .dsv, sha256 of every
                      packet, sha256 of every decoded frame (reference decoder)
  * long_streams.json -- full-length streams of BASELINE configs 4 and 5 and two 1080p streams that need the always-exact
                      scheme (scene cuts; GOP 30): length, sha256, packet count, hashes of the first / last packets
  * cif_gop12.dsv  -- one complete small stream kept verbatim (config-1 size, 6 frames)
  * ops.json       -- operator-level known answers: sha256 of dsv_fwd_sbt / dsv_inv_sbt / dsv_encode_plane
                      / dsv_sub_pred / dsv_hme outputs on seeded inputs
"""
import ctypes as C
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cabi as A  # noqa: E402
import golden_cases as G  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def ref_encode_clip(clip, w, h, fmt, kw):
    """one stream through the real reference LIBRARY (oracle/_ref/libdsv1ref.so), parameters mapped like the CLI does"""
    import importlib
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("digital-subband-video-1_amd")       # only for the DSV_ENCODER struct + CLI flag mapping (host C, no GPU)
    L = C.CDLL(A.REF_SO)
    L.dsv_load_planar_frame.restype = C.c_void_p
    L.dsv_load_planar_frame.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsv_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    enc = pkg.make_encoder_cfg(w, h, fmt, **kw)
    L.dsv_enc_start(C.byref(enc))
    out = b""
    bufs = (pkg.Buf * 4)()
    for t in range(clip.shape[0]):
        fr = L.dsv_load_planar_frame(fmt, clip[t].ctypes.data, w, h)
        nb = L.dsv_enc(C.byref(enc), fr, bufs) & 3
        for i in range(nb):
            out += C.string_at(bufs[i].data, bufs[i].len)
            L.dsv_buf_free(C.byref(bufs[i]))
    L.dsv_enc_free(C.byref(enc))
    return out


def main():
    assert A.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    os.makedirs(OUT, exist_ok=True)
    ref = A.load_ref()
    streams = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (w, h, fmt, n, style, seed, flags, kw) in G.STREAM_CASES.items():
            clip = A.gen_clip(w, h, fmt, seed, n, style=style)
            dsv = A.ref_cli_encode(clip, w, h, A.FMT_CLI[fmt], flags, td)
            dec = A.ref_cli_decode(dsv, td).reshape(n, -1)
            streams[name] = {
                "len": len(dsv), "sha256": sha(dsv),
                "packets": [sha(p) for p in A.split_packets(dsv)],
                "decoded": [sha(dec[t]) for t in range(n)],
            }
            if name == "cif_gop12_style2":
                with open(os.path.join(OUT, "cif_gop12.dsv"), "wb") as f:
                    f.write(dsv)
            print(name, len(dsv), streams[name]["sha256"][:16])
    with open(os.path.join(OUT, "streams.json"), "w") as f:
        json.dump(streams, f, indent=1, sort_keys=True)

    if "--skip-long" not in sys.argv:
        longs = {}
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            for name, (w, h, fmt, n, style, seed, flags, kw) in G.LONG_STREAM_CASES.items():
                clip = A.gen_clip(w, h, fmt, seed, n, style=style)
                dsv = A.ref_cli_encode(clip, w, h, A.FMT_CLI[fmt], flags, td)
                pk = A.split_packets(dsv)
                longs[name] = {"len": len(dsv), "sha256": sha(dsv), "packets": len(pk),
                               "intra_pictures": sum(1 for p in pk if (p[5] & 4) and not (p[5] & 1)),
                               "first_packets": [sha(p) for p in pk[:4]], "last_packets": [sha(p) for p in pk[-3:]]}
                print(name, len(dsv), longs[name]["sha256"][:16], "I pictures:", longs[name]["intra_pictures"])
        with open(os.path.join(OUT, "long_streams.json"), "w") as f:
            json.dump(longs, f, indent=1, sort_keys=True)

    ops = {}
    for name, case in G.OP_CASES.items():
        ops[name] = G.run_op_case(case, "ref", ref)
        print(name, {k: v[:12] for k, v in ops[name].items()})
    with open(os.path.join(OUT, "ops.json"), "w") as f:
        json.dump(ops, f, indent=1, sort_keys=True)

    # Inputs of the GPU sweep (tests/test_gpu_fuzz.py) on which the REFERENCE itself dies: each case is encoded by the
    # reference library in a child process; what ends on a signal is listed, and the GPU tests skip it (there is no answer
    # to be bit-exact with).  Probed here, once, instead of forking inside a process that holds the GPU.
    skips = {}

    def probe(case_id, clip, w, h, fmt, kw):
        pid = os.fork()
        if pid == 0:
            try:
                ref_encode_clip(clip, w, h, fmt, kw)
            finally:
                os._exit(0)
        _, status = os.waitpid(pid, 0)
        if os.WIFSIGNALED(status):
            skips[case_id] = "signal %d" % os.WTERMSIG(status)
            print("reference dies on", case_id, skips[case_id])

    for c in G.fuzz_cases():
        w, h, fmt, n, style, kw, seed = c
        probe("fuzz:" + G.fuzz_id(c), A.gen_clip(w, h, fmt, seed, n, style=style), w, h, fmt, kw)
    for kind in G.EXTREME_KINDS:
        for qp in G.EXTREME_QPS:
            w, h, fmt, n = G.EXTREME_GEOM
            probe("extreme:%s:%d" % (kind, qp), G.extreme_clip(kind, qp), w, h, fmt, G.EXTREME_KW(qp))
    with open(os.path.join(OUT, "ref_crash_skips.json"), "w") as f:
        json.dump(skips, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
