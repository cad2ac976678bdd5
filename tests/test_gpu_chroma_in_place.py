"""A clip that lives in the caller's device memory keeps its chroma planes where they are (dsvg_load_frames_map_ex): the
forward transform and the motion search's chroma test read the packed planes, only luma is copied into the bordered layout
(frame.c:122-164, hme.c:667-681, sbt.c:576-592 need no chroma border).  Same bytes as the oracle; the clip may change as soon as
its batch is collected -- each stream's last frame is copied whole for the next batch's motion search."""
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


@pytest.mark.parametrize("w,h,fmt,expect", [(352, 288, A.SUBSAMP_420, True), (704, 480, A.SUBSAMP_422, True), (320, 240, A.SUBSAMP_444, True),
                                            (360, 200, A.SUBSAMP_420, False),      # chroma 180 wide: not whole 8-byte patch rows -> plain copy
                                            (250, 130, A.SUBSAMP_420, False)])     # odd chroma height
def test_device_clip_chroma_in_place_equals_oracle(pkg, orc, w, h, fmt, expect):
    L = pkg.lib()
    L.dsvg_ctx_chroma_in_place_frames.restype = C.c_long
    L.dsvg_ctx_chroma_in_place_frames.argtypes = [C.c_void_p]
    S, F, gop = 2, 6, 12                                # two calls per GOP: the second call's first frame is a P picture
    kw = dict(qp=85, gop=gop, rc_mode_cli=1)
    clips = [A.gen_clip(w, h, fmt, 0xC1AC0 + s, 2 * F, style=1 + s) for s in range(S)]
    want = [A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, **kw), eos=False)[0] for s in range(S)]
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), S, F)
    try:
        got = [b""] * S
        dev = [b.upload(np.stack([clips[s][k * F:(k + 1) * F] for s in range(S)])) for k in range(2)]
        garbage = np.full(S * F * clips[0].shape[1], 0xA5, dtype=np.uint8)
        for k in range(2):
            part = b.encode(dev[k], on_device=True)
            got = [g + p for g, p in zip(got, part)]
            # the batch is collected: its clip may change now (the next call still searches against its last frames)
            A.chk(L, L.dsvg_dev_upload(b.ctx, dev[k], garbage.ctypes.data, garbage.nbytes))
        n_in_place = L.dsvg_ctx_chroma_in_place_frames(b.ctx)
    finally:
        b.close()
    for s in range(S):
        assert got[s] == want[s], "stream %d differs" % s
    assert (n_in_place == 2 * S * (F - 1)) if expect else (n_in_place == 0), n_in_place


def test_switch_gives_the_same_bytes(pkg, orc, monkeypatch):
    w, h, fmt, S, F = 352, 288, A.SUBSAMP_420, 3, 12
    kw = dict(qp=70, gop=12, rc_mode_cli=1)
    clip = np.stack([A.gen_clip(w, h, fmt, 0xC1AD0 + s, F, style=s) for s in range(S)])
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("DSV1_NO_CHROMA_IN_PLACE", "1")
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), S, F)
        try:
            outs.append(b.encode(b.upload(clip), on_device=True))
        finally:
            b.close()
    assert outs[0] == outs[1]


def test_plain_device_clip_may_change_right_after_submit(pkg, orc):
    """dsv1_batch_submit(.., yuv_on_device = 1): the call has finished with the clip when it returns (advisor, round 3: the in-place
    chroma of a HELD clip is opt-in) -- the clip is overwritten between submit and collect and the streams still equal the oracle's;
    with DSV1_CLIP_HELD the same misuse is the caller's error and is not exercised here"""
    L = pkg.lib()
    L.dsvg_ctx_chroma_in_place_frames.restype = C.c_long
    L.dsvg_ctx_chroma_in_place_frames.argtypes = [C.c_void_p]
    w, h, fmt, S, F = 352, 288, A.SUBSAMP_420, 3, 6
    kw = dict(qp=85, gop=12, rc_mode_cli=1)
    clips = [A.gen_clip(w, h, fmt, 0xC1AE0 + s, 2 * F, style=s) for s in range(S)]
    want = [A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, **kw), eos=False)[0] for s in range(S)]
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), S, F)
    try:
        got = [b""] * S
        dev = b.upload(np.stack([clips[s][:F] for s in range(S)]))
        garbage = np.full(S * F * clips[0].shape[1], 0x3C, dtype=np.uint8)
        b.submit(dev, on_device=True, held=False)
        A.chk(L, L.dsvg_dev_upload(b.ctx, dev, garbage.ctypes.data, garbage.nbytes))          # the clip is gone before the batch is collected
        nxt = np.ascontiguousarray(np.stack([clips[s][F:] for s in range(S)]))
        A.chk(L, L.dsvg_dev_upload(b.ctx, dev, nxt.ctypes.data, nxt.nbytes))                  # ... and reused for the next batch at once
        b.submit(dev, on_device=True, held=False)
        A.chk(L, L.dsvg_dev_upload(b.ctx, dev, garbage.ctypes.data, garbage.nbytes))
        for _ in range(2):
            got = [g + p for g, p in zip(got, b.collect())]
        assert L.dsvg_ctx_chroma_in_place_frames(b.ctx) == 0
    finally:
        b.close()
    for s in range(S):
        assert got[s] == want[s], "stream %d differs" % s


@pytest.mark.parametrize("w,h,fmt,S,F", [(704, 480, A.SUBSAMP_420, 2, 6),      # 24-pixel blocks: the last column is 8 wide, its statistics window leaves the picture
                                         (704, 480, A.SUBSAMP_422, 2, 6),
                                         (1280, 720, A.SUBSAMP_420, 1, 4),
                                         (1920, 1080, A.SUBSAMP_420, 1, 3),    # full 64x48 blocks: the motion search's register body
                                         (1920, 1088, A.SUBSAMP_444, 1, 3)])
def test_luma_in_place_equals_oracle_and_the_switch(pkg, orc, monkeypatch, w, h, fmt, S, F):
    """round 5: a held device clip keeps its LUMA where it is too (include/dsvg.h, dsvg_load_frames_map_ex): the forward transforms and the
    level-0 motion search's interior blocks read the clip, the bordered copy is a ring.  Same bytes as the oracle and as with the
    switch off; the clip is overwritten as soon as its batch is collected."""
    L = pkg.lib()
    L.dsvg_ctx_luma_in_place_frames.restype = C.c_long
    L.dsvg_ctx_luma_in_place_frames.argtypes = [C.c_void_p]
    kw = dict(qp=80, gop=2 * F, rc_mode_cli=1)
    clips = [A.gen_clip(w, h, fmt, 0x1A7A0 + s, 2 * F, style=1 + s) for s in range(S)]
    want = [A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, **kw), eos=False)[0] for s in range(S)]
    garbage = np.full(S * F * clips[0].shape[1], 0x5A, dtype=np.uint8)
    for off in (False, True):
        if off:
            monkeypatch.setenv("DSV1_NO_LUMA_IN_PLACE", "1")
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), S, F)
        try:
            got = [b""] * S
            dev = [b.upload(np.stack([clips[s][k * F:(k + 1) * F] for s in range(S)])) for k in range(2)]
            for k in range(2):
                part = b.encode(dev[k], on_device=True)
                got = [g + p for g, p in zip(got, part)]
                A.chk(L, L.dsvg_dev_upload(b.ctx, dev[k], garbage.ctypes.data, garbage.nbytes))
            n = L.dsvg_ctx_luma_in_place_frames(b.ctx)
        finally:
            b.close()
        for s in range(S):
            assert got[s] == want[s], "stream %d differs (luma in place %s)" % (s, "off" if off else "on")
        assert n == (0 if off else 2 * S * (F - 1)), n
