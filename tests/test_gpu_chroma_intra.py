"""k_hme_csum: the sums of the chroma blocks that level 0's variance test needs (c_maxvar hme.c:269-300, the test hme.c:667-681) come from a
per-frame table for full 64-wide blocks; partial blocks, and everything with DSV1_NO_CHROMA_SUMS=1, fetch the blocks inside the search.
Both ways must give the oracle's bytes, for every chroma format (chroma blocks 64, 32 and 16 wide), for clips in host memory and for
device clips whose chroma stays in place (other strides and plane addresses), on content whose chroma changes where luma does not (so that
the test decides blocks)."""
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


def chroma_event_clip(w, h, fmt, seed, n, chroma_events=True):
    """pan + texture, and a rectangle over the middle of the frame whose luma is fresh low-amplitude noise in every frame (no vector predicts
    it, yet its mean, variance and texture are the same in every frame: none of the luma conditions of hme.c:652-666 asks for intra) and whose
    chroma is textured in the even frames and flat in the odd ones: in a pair (current odd, reference even) the reference's chroma variance
    exceeds four times the source's -- intra by the chroma test alone (hme.c:667-681)"""
    clip = A.gen_clip(w, h, fmt, seed, n, style=0).copy()
    cw, ch = A.chroma_dims(w, h, fmt)
    rng = np.random.default_rng(seed)
    y0, y1, x0, x1 = h // 4, 3 * h // 4, w // 5, 4 * w // 5
    sx, sy = w // cw, h // ch
    for f in range(n):
        fr = clip[f]
        yp = fr[:w * h].reshape(h, w)
        yp[y0:y1, x0:x1] = (128 + rng.integers(-40, 41, (y1 - y0, x1 - x0))).astype(np.uint8)
        u = fr[w * h:w * h + cw * ch].reshape(ch, cw)
        v = fr[w * h + cw * ch:].reshape(ch, cw)
        cy0, cy1, cx0, cx1 = y0 // sy, y1 // sy, x0 // sx, x1 // sx
        if (f & 1) or not chroma_events:
            u[cy0:cy1, cx0:cx1] = 16         # (small: the reference squares the block's sum in 32 bits -- at 128 the wrap-around leaves a
            v[cy0:cy1, cx0:cx1] = 16         #  flat block a 'variance' of millions and the test never fires)
        else:
            u[cy0:cy1, cx0:cx1] = rng.integers(0, 256, (cy1 - cy0, cx1 - cx0), dtype=np.uint8)
            v[cy0:cy1, cx0:cx1] = rng.integers(0, 256, (cy1 - cy0, cx1 - cx0), dtype=np.uint8)
    return clip


def intra_pct(clip, w, h, fmt, **cli):
    """share of intra blocks in the P pictures, from the oracle encoder's motion fields"""
    import ctypes as C
    L = A.load_orc()
    cfg = A.orc_cfg(w, h, fmt, **cli)
    e = L.orc_enc_open(C.byref(cfg))
    out, n, cap = C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
    intra = total = 0
    for t in range(clip.shape[0]):
        L.orc_enc_frame(e, clip[t].ctypes.data, C.byref(out), C.byref(n), C.byref(cap), None)
        cnt = C.c_int(0)
        p = L.orc_enc_last_mvs(e, C.byref(cnt))
        if t == 0 or not p or cnt.value == 0:
            continue
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cnt.value * 12,)).reshape(cnt.value, 12)
        intra += int((a[:, 4] != 0).sum())
        total += cnt.value
    C.CDLL(None).free(out)
    L.orc_enc_close(e)
    return 100.0 * intra / max(total, 1)


@pytest.mark.parametrize("w,h,fmt", [(1920, 1080, A.SUBSAMP_420), (1600, 900, A.SUBSAMP_444), (1920, 1080, A.SUBSAMP_422), (1920, 1088, A.SUBSAMP_411),
                                     (1928, 1084, A.SUBSAMP_420)])      # (a partial block column and row beside the full blocks)
def test_table_and_in_kernel_sums_equal_oracle(pkg, orc, monkeypatch, w, h, fmt):
    n = 5
    kw = dict(qp=85, gop=12, rc_mode_cli=1, scd=0)
    clip = chroma_event_clip(w, h, fmt, 0xC5A0 + w, n)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw), eos=False)
    outs = {}
    for mode in ("table", "in_kernel", "table_device_clip"):
        if mode == "in_kernel":
            monkeypatch.setenv("DSV1_NO_CHROMA_SUMS", "1")
        else:
            monkeypatch.delenv("DSV1_NO_CHROMA_SUMS", raising=False)
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), 1, n)
        try:
            if mode == "table_device_clip":
                outs[mode] = b.encode(b.upload(clip[None]), on_device=True)[0]
            else:
                outs[mode] = b.encode(clip[None])[0]
        finally:
            b.close()
    for mode, got in outs.items():
        assert got == want, "%s differs from the oracle" % mode
    # the content does what it is for: the chroma test alone turns blocks intra (the same clip with flat chroma everywhere: next to none)
    assert intra_pct(clip, w, h, fmt, **kw) > 10.0
    assert intra_pct(chroma_event_clip(w, h, fmt, 0xC5A0 + w, n, chroma_events=False), w, h, fmt, **kw) < 1.0


@pytest.mark.parametrize("w,h,fmt", [(1920, 1080, A.SUBSAMP_420), (1600, 900, A.SUBSAMP_444), (1920, 1080, A.SUBSAMP_422), (1920, 1088, A.SUBSAMP_411),
                                     (1928, 1084, A.SUBSAMP_420), (3840, 2160, A.SUBSAMP_420)])
def test_intra_heavy_content_on_64_wide_blocks(pkg, orc, w, h, fmt):
    """a third of the P blocks intra (clip style 4) at the sizes whose blocks are 64 wide: k_mc's complete intra blocks take the path
    without the LDS window (quadrant means by v_sad_u8, bmc.c:176-189,254-283), their patches the lean forward kernel; the partial
    blocks of 1928x1084 keep the staged path, side by side"""
    n = 4
    kw = dict(qp=85, gop=12, rc_mode_cli=1, scd=0)
    clip = A.gen_clip(w, h, fmt, 0x1A7A + w, n, style=4)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw), eos=False)
    assert intra_pct(clip, w, h, fmt, **kw) > 15.0
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), 2, n)
    try:
        got = b.encode(np.stack([clip, clip]))
    finally:
        b.close()
    assert got[0] == want and got[1] == want
    # ... and the decoder's intra blocks (k_mc without the residual) give the oracle's frames
    from test_gpu_stream import product_decode
    frames = product_decode(pkg, want)
    ref = A.orc_decode(want, w, h, fmt)
    assert len(frames) == len(ref) == n and all((f == r).all() for f, r in zip(frames, ref))
