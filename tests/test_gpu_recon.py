"""GPU parity of the fused encoder's RECONSTRUCTION frames (dsv_encoder.c:523-525,663-674): after every
picture the frame the next P picture will predict from -- picture area AND the 64-pixel replicated border the motion
compensation reads (dsv_extend_frame frame.c:263-295) -- must equal the oracle encoder's recon_frame byte for byte.
The stream tests only see this indirectly (through the next picture's bytes); the last picture of a stream, the border
and the sparse zero-tile path of k_inv_haar_tile (reconstruction = prediction, written in place by k_fwd_mc_pix) are
observed here directly."""
import ctypes as C
import importlib

import numpy as np
import pytest

import _cabi as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


def expected_raw(w, h, fmt, planar):
    """the reference frame allocation for a picture: planes in place, borders replicated"""
    bf = A.BorderedFrame(w, h, fmt)
    o = 0
    for i in range(3):
        pw, ph = bf.dims[i]
        pl = planar[o:o + pw * ph].reshape(ph, pw)
        o += pw * ph
        ext = np.pad(pl, A.BORDER, mode="edge")
        s = bf.strides[i]
        start = bf.offs[i] - (s * A.BORDER + A.BORDER)
        view = np.lib.stride_tricks.as_strided(bf.buf[start:], shape=(ph + 2 * A.BORDER, pw + 2 * A.BORDER), strides=(s, 1))
        view[:, :] = ext
    return bf.raw().copy()


def make_clip(w, h, fmt, seed, n, style):
    """styles 0-2: the generator's; 10: a static scene (every P tile empty: reconstruction = prediction); 11: a static
    scene with a small square moving over it (empty tiles next to tiles with a residual)"""
    if style < 10:
        return A.gen_clip(w, h, fmt, seed, n, style=style)
    base = A.gen_clip(w, h, fmt, seed, 1, style=0)[0]
    clip = np.repeat(base[None, :], n, axis=0).copy()
    if style == 11:
        for t in range(n):
            y = clip[t, :w * h].reshape(h, w)
            x0, y0 = (40 + 23 * t) % (w - 48), (24 + 9 * t) % (h - 48)
            y[y0:y0 + 40, x0:x0 + 40] = (37 * t + np.arange(40)[None, :] * 5 + np.arange(40)[:, None] * 3) % 256
    return clip


CASES = [
    # w, h, fmt, frames, style, cli
    (352, 288, A.SUBSAMP_420, 5, 2, dict(qp=85, gop=12, rc_mode_cli=1)),          # flat objects: intra blocks, residual tiles
    (352, 288, A.SUBSAMP_420, 4, 0, dict(qp=85, gop=12, rc_mode_cli=1)),          # pan + texture: mostly empty P tiles
    (352, 288, A.SUBSAMP_444, 3, 1, dict(qp=40, gop=12, rc_mode_cli=1)),
    (250, 130, A.SUBSAMP_420, 4, 2, dict(qp=70, gop=12, rc_mode_cli=1)),          # ragged tiles, odd chroma
    (704, 480, A.SUBSAMP_422, 3, 2, dict(qp=95, gop=12, rc_mode_cli=1)),
    (1920, 1080, A.SUBSAMP_420, 3, 0, dict(qp=85, gop=12, rc_mode_cli=1)),        # 960x540 chroma: overlapping scan regions
    (1920, 1080, A.SUBSAMP_420, 3, 2, dict(qp=85, gop=12, rc_mode_cli=1)),
    (352, 288, A.SUBSAMP_420, 4, 10, dict(qp=85, gop=12, rc_mode_cli=1)),         # static: the zero-tile path everywhere
    (704, 480, A.SUBSAMP_420, 5, 11, dict(qp=85, gop=12, rc_mode_cli=1)),         # static + moving square: both paths
    (1920, 1080, A.SUBSAMP_420, 3, 11, dict(qp=85, gop=12, rc_mode_cli=1)),
    (250, 130, A.SUBSAMP_444, 4, 11, dict(qp=85, gop=12, rc_mode_cli=1, scd=0)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_recon_frames_equal_oracle(pkg, case):
    w, h, fmt, n, style, cli = CASES[case]
    clip = make_clip(w, h, fmt, 0x7EC0 + case, n, style)
    want_stream, want_rec = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli), want_recon=True, eos=False)
    L = pkg.lib()
    L.dsv1_batch_recon_slot.argtypes = [C.c_void_p, C.c_int]
    L.dsvg_download_recon_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), 1, 1)
    try:
        got_stream = b""
        for t in range(n):
            got_stream += b.encode(clip[t].reshape(1, 1, -1))[0]
            slot = L.dsv1_batch_recon_slot(b.h, 0)
            assert slot >= 0
            want = expected_raw(w, h, fmt, want_rec[t])
            got = np.zeros_like(want)
            assert L.dsvg_download_recon_raw(b.ctx, slot, got.ctypes.data, got.size) == 0, L.dsvg_last_error()
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, "frame %d: %d reconstruction bytes differ, first at raw offset %d" % (t, bad.size, int(bad[0]))
        assert got_stream == want_stream
    finally:
        b.close()


def test_zero_and_general_tiles_both_taken(pkg):
    """the sparse inverse must have exercised both of its paths in the cases above: a static clip takes the zero path
    everywhere, a clip with a moving object takes the general path where the object is"""
    w, h, fmt = 704, 480, A.SUBSAMP_420
    for style, want_general in ((10, False), (11, True)):
        clip = make_clip(w, h, fmt, 0x51AB + style, 4, style)
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, qp=85, gop=12, rc_mode_cli=1), 1, 4)
        try:
            b.tile_stats()
            b.encode(clip.reshape(1, 4, -1))
            st = b.tile_stats(enable=False)
        finally:
            b.close()
        # (the chroma planes of P pictures go through the thread-per-patch kernel, which is not counted in tiles)
        assert st["zero_luma"] > 0, st
        if want_general:
            assert st["general_luma"] > 0, st
        else:
            # the I picture's quantisation error leaves a residual here and there
            assert st["zero_luma"] > st["general_luma"], st
