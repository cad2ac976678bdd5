"""The device pipeline's C ABI used directly (include/dsvg.h), the way a host other than our session layer would:
  * dsvg_code_batch with the jobs of a step in ANOTHER order than the step before -- with two coding streams the
    reference picture of a job was then written on the other stream; the call must notice and stay correct
    (dsv_encoder.c:657-674: a P picture predicts from the reconstruction of the picture before it);
  * truncated / hostile packets into the decoder: refused, never read past (dsv_decoder.c:286-472 trusts the packet)."""
import ctypes as C
import importlib

import numpy as np
import pytest

import _cabi as A

pytestmark = pytest.mark.gpu


class PicJob(C.Structure):
    _fields_ = [("src_slot", C.c_int), ("ref_recon_slot", C.c_int), ("recon_slot", C.c_int), ("quant", C.c_int),
                ("mvs", C.c_void_p), ("stable_blocks", C.c_void_p), ("out_slot", C.c_int), ("no_intra_blocks", C.c_int), ("has_reach", C.c_int), ("mv_reach", C.c_short * 4), ("border_hint", C.c_int)]


class PicOut(C.Structure):
    _fields_ = [("dc", C.c_int32 * 3), ("nruns", C.c_uint32 * 3), ("nbytes", C.c_uint32 * 3), ("payload", C.c_void_p * 3),
                ("rc_quant", C.c_int32), ("rc_pkt_len", C.c_uint32)]


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


def _code(pkg, clips, order1, code_streams):
    """16 streams x 2 frames: step 0 = I pictures in stream order, step 1 = P pictures in `order1`; returns per
    (frame, stream) the coded planes"""
    L = pkg.lib()
    S, w, h, fmt = clips.shape[0], 352, 288, A.SUBSAMP_420
    L.dsvg_ctx_create.argtypes = [C.POINTER(C.c_void_p)] + [C.c_int] * 9
    L.dsvg_load_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsvg_code_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(PicJob)]
    L.dsvg_fetch_pictures.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(PicOut)]
    L.dsvg_ctx_destroy.argtypes = [C.c_void_p]
    ctx = C.c_void_p(None)
    assert L.dsvg_ctx_create(C.byref(ctx), 0, w, h, fmt, 0, 2 * S, 2 * S, S, 2 * S) == 0, L.dsvg_last_error()
    try:
        L.dsvg_ctx_code_streams(ctx, code_streams)
        frames = np.ascontiguousarray(clips.transpose(1, 0, 2))               # slot = t * S + s
        assert L.dsvg_load_frames(ctx, 0, 2 * S, frames.ctypes.data, 0, 1) == 0, L.dsvg_last_error()
        bw, bh, nbh, nbv = A.block_dims(w, h)
        nblk = nbh * nbv
        mvs = np.zeros((S, nblk), dtype=A.MV_DTYPE)
        for s in range(S):
            mvs[s]["x"] = 2 * (s % 5) - 4                                     # different vectors per stream, half-pel for odd s
            mvs[s]["y"] = s % 3
        stable = np.zeros((S, nblk), dtype=np.uint8)
        jobs = (PicJob * (2 * S))()
        for i in range(S):
            jobs[i] = PicJob(i, -1, i, 313, None, stable[i].ctypes.data, i, 0)
        for i, s in enumerate(order1):
            # ping-pong slots: the P picture of stream s goes to slot S + s
            jobs[S + i] = PicJob(S + s, s, S + s, 313, mvs[s].ctypes.data, stable[s].ctypes.data, S + s, 1)
        assert L.dsvg_code_batch(ctx, 2, S, jobs) == 0, L.dsvg_last_error()
        slots = (C.c_int * (2 * S))(*range(2 * S))
        outs = (PicOut * (2 * S))()
        assert L.dsvg_fetch_pictures(ctx, 2 * S, slots, outs) == 0, L.dsvg_last_error()
        res = []
        for o in outs:
            res.append(tuple((o.dc[p], o.nruns[p], C.string_at(o.payload[p], o.nbytes[p])) for p in range(3)))
        return res
    finally:
        L.dsvg_ctx_destroy(ctx)


def test_code_batch_is_independent_of_the_job_order_between_steps(pkg):
    S = 16
    clips = np.stack([A.gen_clip(352, 288, A.SUBSAMP_420, 0xAB10 + s, 2, style=s % 3) for s in range(S)])
    base = _code(pkg, clips, list(range(S)), 1)                 # one coding stream, natural order: the trusted arrangement
    for order in (list(range(S)), list(reversed(range(S))), [(5 * i + 3) % S for i in range(S)]):
        got = _code(pkg, clips, order, 2)
        assert got == base, "job order %s with two coding streams changes the coded pictures" % order


def test_truncated_packets_are_refused(pkg):
    w, h, fmt = 352, 288, A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0xAB40, 3, style=2)
    stream = pkg.encode_clip(clip, w, h, fmt, qp=85, gop=12, rc_mode_cli=1)
    pk = A.split_packets(stream)
    pics = [p for p in pk if p[5] & 4]
    d = pkg.DecBatch(w, h, fmt, 1)
    try:
        _, st, _ = d.decode([pics[0]])
        assert st[0] == 0
        p = pics[1]                                             # a P picture: stability + motion + three planes
        whole, st, _ = d.decode([p])
        assert st[0] == 0
        whole = whole.copy()
        for cut in [0, 5, 13, 14, 18, 19, 22, 30, 60, len(p) // 3, len(p) // 2, len(p) - 40, len(p) - 1]:
            _, st, _ = d.decode([p[:cut]])
            assert st[0] != 0, "a packet cut to %d of %d bytes was accepted" % (cut, len(p))
        # sub-lengths that point past the packet: every length byte of the side information blown up
        for off in range(18, 40):
            bad = bytearray(p)
            bad[off] = 0x01                                     # a long exp-Golomb prefix: a huge length
            bad[off + 1] = 0x00
            _, st, _ = d.decode([bytes(bad)])                   # any status is fine; it must come back
        _, st, _ = d.decode([pics[0]])                          # (a corrupted packet that parsed has replaced the reference)
        assert st[0] == 0
        again, st, _ = d.decode([p])                            # and the decoder still works
        assert st[0] == 0 and (again == whole).all()
    finally:
        d.close()
