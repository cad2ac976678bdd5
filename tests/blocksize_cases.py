"""Streams whose block size is not the encoder's rule for the frame size (the reference DECODER takes the block size from
every picture packet, dsv_decoder.c:335-360; the reference ENCODER always applies its rule, dsv_encoder.c:557-592, so such
streams come from other encoders).  Made with the oracle encoder and ORC_BLK_OVERRIDE, frame by frame; shared by the CPU
test that pins the oracle decoder to the reference decoder on them and by the GPU tests of the product decoders."""
import ctypes as C
import os

import numpy as np

import _cabi as A

# w, h, fmt, frames, style, qp, block size of each frame ("WxH"; the first is the smallest: the oracle encoder sizes its
# per-block state once)
CASES = [
    (352, 288, A.SUBSAMP_420, 4, 2, 80, ["32x24"] * 4),                      # constant, not the rule (16x16)
    (704, 480, A.SUBSAMP_420, 3, 0, 85, ["20x44"] * 3),                      # multiples of 4 only
    (352, 288, A.SUBSAMP_444, 5, 1, 85, ["16x16", "16x16", "24x32", "24x32", "64x16"]),   # changes at P pictures
    (1920, 1080, A.SUBSAMP_420, 3, 0, 85, ["32x32", "32x32", "64x48"]),       # 1080p: to the rule's size at a P picture
]


def make_stream(case):
    w, h, fmt, n, style, qp, blks = CASES[case]
    clip = A.gen_clip(w, h, fmt, 0xB10C + case, n, style=style)
    L = A.load_orc()
    cfg = A.orc_cfg(w, h, fmt, qp=qp, gop=12, rc_mode_cli=1, scd=0)
    e = L.orc_enc_open(C.byref(cfg))
    L.orc_enc_set_next_fnum(e, 0)
    out, n_, cap = C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
    try:
        for t in range(n):
            os.environ["ORC_BLK_OVERRIDE"] = blks[t]
            L.orc_enc_frame(e, clip[t].ctypes.data, C.byref(out), C.byref(n_), C.byref(cap), None)
    finally:
        os.environ.pop("ORC_BLK_OVERRIDE", None)
    L.orc_enc_eos(e, C.byref(out), C.byref(n_), C.byref(cap))
    data = C.string_at(out.value, n_.value)
    C.CDLL(None).free(out)
    L.orc_enc_close(e)
    return w, h, fmt, n, data
