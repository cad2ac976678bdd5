"""GOP-parallel coding of ONE stream (chain mode: dsv1_stream_open, and the drop-in dsv_enc on top of it) must give the
frame-serial encoder's bytes for every CRF configuration -- the always-exact scheme of SURVEY.md 8(e): source-only analysis of
all frames, the serial state machine (GOP starts dsv_encoder.c:624-641, scene changes :538-552, forced-intra pictures :645-653,
stability accumulators :345-399 with a stable_refresh that does not line up with the GOP) replayed on the host, chains of
pictures coded side by side."""
import importlib

import numpy as np
import pytest

import _cabi as A
from test_gpu_stream import _drive_dsv_enc, explain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


CASES = [
    # w, h, fmt, frames, style, frames_per_call, chains, CLI-style flags
    (352, 288, A.SUBSAMP_420, 48, 5, 24, 3, dict(qp=85, gop=12, rc_mode_cli=1)),            # scene cuts every 7 frames, fast pan, CLI defaults otherwise (scd on)
    (352, 288, A.SUBSAMP_420, 48, 5, 16, 2, dict(qp=85, gop=12, rc_mode_cli=1)),            # more chains in a call than run side by side; calls that end in mid-GOP
    (176, 144, A.SUBSAMP_420, 60, 0, 20, 2, dict(qp=85, gop=30, rc_mode_cli=1)),            # GOP 30: stable_refresh 14, stability state crosses GOPs and calls
    (176, 144, A.SUBSAMP_420, 60, 5, 30, 4, dict(qp=70, gop=30, rc_mode_cli=1)),
    (352, 288, A.SUBSAMP_420, 24, 1, 12, 2, dict(qp=85, gop=12, rc_mode_cli=1, ipct=20)),   # forced-intra P pictures (many intra blocks)
    (352, 288, A.SUBSAMP_420, 24, 2, 8, 3, dict(qp=85, gop=12, rc_mode_cli=1)),             # luma step: a scene change inside a call
    (320, 240, A.SUBSAMP_444, 16, 3, 8, 8, dict(qp=85, gop=0, rc_mode_cli=1)),              # intra-only: every picture its own chain
    (352, 288, A.SUBSAMP_422, 36, 3, 12, 1, dict(qp=60, gop=5, rc_mode_cli=1, scd=0)),      # one chain at a time = the serial order
    (704, 480, A.SUBSAMP_420, 24, 5, 12, 4, dict(qp=85, gop=8, rc_mode_cli=1)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_chain_mode_equals_serial_encoder(pkg, orc, case):
    w, h, fmt, n, style, fpc, chains, kw = CASES[case]
    clip = A.gen_clip(w, h, fmt, 0xC4A100 + case, n, style=style)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    got = pkg.encode_stream(clip, w, h, fmt, fpc, chains, **kw)
    assert got == want, explain(got, want)
    if style in (3, 5) and kw.get("scd", 1) and kw["gop"]:
        # the clip really has scene changes: more I pictures than GOP starts
        pk = A.split_packets(want)
        pics = [p for p in pk if p[5] & 4]
        intra = sum(1 for p in pics if not (p[5] & 1))
        assert intra > (n + kw["gop"] - 1) // kw["gop"], "no scene change was detected: the case tests nothing"


def test_chain_mode_reconstruction_carried_across_calls(pkg, orc):
    """the call's last reconstruction (the reference of the next call's first P picture) is the oracle's recon_frame"""
    w, h, fmt, n = 352, 288, A.SUBSAMP_420, 18
    kw = dict(qp=85, gop=12, rc_mode_cli=1)
    clip = A.gen_clip(w, h, fmt, 0xC4A1FF, n, style=0)
    _, recs = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw), want_recon=True)
    import ctypes as C
    L = pkg.lib()
    L.dsv1_batch_recon_slot.argtypes = [C.c_void_p, C.c_int]
    L.dsvg_download_recon.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), 1, 9, chains=2)
    try:
        for i in range(2):
            b.encode(clip[9 * i:9 * i + 9].reshape(1, 9, -1))
            slot = L.dsv1_batch_recon_slot(b.h, 0)
            assert slot >= 0
            got = np.empty(clip.shape[1], dtype=np.uint8)
            A.chk(L, L.dsvg_download_recon(b.ctx, slot, got.ctypes.data))
            A.assert_same("reconstruction after call %d" % i, got, recs[9 * i + 8])
    finally:
        b.close()


@pytest.mark.parametrize("n,look,style,cli", [
    (50, "16", 5, dict(qp=85, gop=12, rc_mode_cli=1)),       # lookahead 16: three full batches + a tail of 2, scene cuts, calls ending in mid-GOP
    (45, "30", 5, dict(qp=85, gop=30, rc_mode_cli=1)),       # GOP 30 with the CLI's stable_refresh of 14
    (31, "12", 1, dict(qp=85, gop=12, rc_mode_cli=1)),
    (20, "8", 2, dict(qp=85, gop=0, rc_mode_cli=1)),
    (30, None, 3, dict(qp=85, gop=12, rc_mode_cli=1)),       # default lookahead (16 GOPs): everything comes out at end of stream
])
def test_drop_in_dsv_enc_is_gop_parallel_and_exact(pkg, orc, monkeypatch, n, look, style, cli):
    if look:
        monkeypatch.setenv("DSV1_ENC_LOOKAHEAD", look)
    else:
        monkeypatch.delenv("DSV1_ENC_LOOKAHEAD", raising=False)
    w, h, fmt = 352, 288, A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0xD3090 + n, n, style=style)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli))
    out, counts = _drive_dsv_enc(pkg, clip, w, h, fmt, **cli)
    assert out == want, explain(out, want)
    assert counts[0] == 0 and max(counts) <= 2
