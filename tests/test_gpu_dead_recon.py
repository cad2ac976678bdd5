"""Round 5: a reference picture nobody predicts from is coded without its reconstruction (include/dsv1_api.h, dsv1_batch_dropped_recons):
the last picture of a closed GOP, the picture in front of a scene cut or of a picture that turned intra.  The reference encoder builds
that reconstruction and never reads it (dsv_encoder.c:665-708): the streams must not change by a bit -- and a stream renumbered so that
the promised GOP start becomes a P picture gets the dropped picture coded again first."""
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


def _pics(stream):
    return [p for p in A.split_packets(stream) if p[5] & 4]


@pytest.mark.parametrize("F,gop,ncalls,abr", [(6, 6, 2, False), (3, 6, 4, False), (12, 6, 1, False), (4, 6, 3, False), (6, 6, 2, True), (12, 6, 1, True)])
def test_unread_reconstructions_are_dropped_and_the_streams_stay(pkg, orc, monkeypatch, F, gop, ncalls, abr):
    w, h, fmt, S = 352, 288, A.SUBSAMP_420, 3
    cli = dict(qp=80, gop=gop, rc_mode_cli=0 if abr else 1, scd=1)
    if abr:
        cli["kbps"] = 400
    n = F * ncalls
    clips = [A.gen_clip(w, h, fmt, 0xDEAD0 + s, n, style=(0, 2, 5)[s]) for s in range(S)]      # pan / scene cuts / mixed
    want = [A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, **cli), eos=False)[0] for s in range(S)]
    # what can be known: a picture whose successor in the same call has no reference; a call's last picture when the next frame number
    # starts a GOP (CRF only: an ABR stream keeps it)
    expect = 0
    for s in range(S):
        hr = [p[5] & 1 for p in _pics(want[s])]
        assert len(hr) == n
        for t in range(n):
            if (t + 1) % F:
                expect += not hr[t + 1]
            else:
                expect += (not abr) and (t + 1) % gop == 0
    for keep_all in (False, True):
        if keep_all:
            monkeypatch.setenv("DSV1_RECON_ALL", "1")
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), S, F)
        try:
            got = [b""] * S
            for k in range(ncalls):
                part = b.encode(np.stack([clips[s][k * F:(k + 1) * F] for s in range(S)]))
                got = [g + p for g, p in zip(got, part)]
            dropped = b.dropped_recons()
        finally:
            b.close()
        for s in range(S):
            assert got[s] == want[s], "stream %d differs (keep_all %s)" % (s, keep_all)
        assert dropped == ((0, 0) if keep_all else (expect, 0)), (dropped, expect)
    assert expect > 0


@pytest.mark.parametrize("pipelined", [False, True])
def test_renumbered_stream_gets_its_dropped_reconstruction_back(pkg, orc, pipelined):
    """frames 0..5 are a closed GOP (the sixth picture's reconstruction is dropped: frame number 6 would start a GOP), then the caller
    renumbers the streams to 3: the seventh picture is a P picture and predicts from the sixth -- which is coded once more, kept"""
    w, h, fmt, gop, S = 352, 288, A.SUBSAMP_420, 6, 2
    cli = dict(qp=85, gop=gop, rc_mode_cli=1, scd=0)
    clips = [A.gen_clip(w, h, fmt, 0x5E80 + s, 3 * gop, style=s) for s in range(S)]
    Lo = A.load_orc()
    want = []
    for s in range(S):
        cfg = A.orc_cfg(w, h, fmt, **cli)
        e = Lo.orc_enc_open(C.byref(cfg))
        out, n_, cap = C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
        Lo.orc_enc_set_next_fnum(e, 0)
        for t in range(3 * gop):
            if t == gop and s == 0:
                Lo.orc_enc_set_next_fnum(e, 3)              # stream 0 only: stream 1 keeps its numbering (and its GOP start)
            Lo.orc_enc_frame(e, clips[s][t].ctypes.data, C.byref(out), C.byref(n_), C.byref(cap), None)
        want.append(C.string_at(out.value, n_.value))
        C.CDLL(None).free(out)
        Lo.orc_enc_close(e)
    assert _pics(want[0])[gop][5] & 1, "the test wants a P picture right after the renumbering"
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), S, gop)
    try:
        calls = [np.stack([clips[s][k * gop:(k + 1) * gop] for s in range(S)]) for k in range(3)]
        got = [b""] * S
        if pipelined:
            b.submit(calls[0])
            b.set_fnum(0, 3)                                # (the first batch is still in flight)
            b.submit(calls[1])
            parts = [b.collect()]
            b.submit(calls[2])
            parts += [b.collect(), b.collect()]
            for part in parts:
                got = [g + p for g, p in zip(got, part)]
        else:
            for k in range(3):
                if k == 1:
                    b.set_fnum(0, 3)
                part = b.encode(calls[k])
                got = [g + p for g, p in zip(got, part)]
        dropped, remedied = b.dropped_recons()
    finally:
        b.close()
    for s in range(S):
        assert got[s] == want[s], "stream %d differs" % s
    assert remedied == 1 and dropped >= 3, (dropped, remedied)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18])
def test_random_renumbering_against_the_oracle(pkg, orc, seed):
    """random GOP lengths, frames per call and renumberings between calls (forwards, backwards, onto and off GOP boundaries), two
    streams renumbered independently, plain and pipelined calls: whatever was dropped and whatever had to be made again, the
    streams are the oracle's, renumbered the same way"""
    import random
    rng = random.Random(seed)
    w, h, fmt, S = 352, 288, A.SUBSAMP_420, 2
    gop, F = rng.choice([2, 3, 4, 5, 6, 7]), rng.choice([1, 2, 3, 4, 6])
    ncalls = rng.choice([5, 6, 8])
    cli = dict(qp=rng.choice([60, 85]), gop=gop, rc_mode_cli=1, scd=rng.choice([0, 1]))
    n = F * ncalls
    clips = [A.gen_clip(w, h, fmt, 0xF00D0 + 97 * seed + s, n, style=rng.choice([0, 1, 2, 5])) for s in range(S)]
    # renumberings: before call k stream s continues at frame number renum[(k, s)]
    renum = {}
    for k in range(1, ncalls):
        for s in range(S):
            if rng.random() < 0.4:
                renum[(k, s)] = rng.choice([0, 1, gop - 1, gop, gop + 1, 2 * gop, k * F, k * F - 1, rng.randrange(0, 40)])
    Lo = A.load_orc()
    want = []
    for s in range(S):
        cfg = A.orc_cfg(w, h, fmt, **cli)
        e = Lo.orc_enc_open(C.byref(cfg))
        out, n_, cap = C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
        Lo.orc_enc_set_next_fnum(e, 0)
        for t in range(n):
            if t % F == 0 and (t // F, s) in renum:
                Lo.orc_enc_set_next_fnum(e, max(renum[(t // F, s)], 0))
            Lo.orc_enc_frame(e, clips[s][t].ctypes.data, C.byref(out), C.byref(n_), C.byref(cap), None)
        want.append(C.string_at(out.value, n_.value))
        C.CDLL(None).free(out)
        Lo.orc_enc_close(e)
    pipelined = rng.random() < 0.5
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), S, F)
    try:
        calls = [np.stack([clips[s][k * F:(k + 1) * F] for s in range(S)]) for k in range(ncalls)]
        parts = []
        for k in range(ncalls):
            for s in range(S):
                if (k, s) in renum:
                    b.set_fnum(s, max(renum[(k, s)], 0))
            if pipelined:
                b.submit(calls[k])
                if k > 0:
                    parts.append(b.collect())
            else:
                parts.append(b.encode(calls[k]))
        if pipelined:
            parts.append(b.collect())
        dropped, remedied = b.dropped_recons()
    finally:
        b.close()
    got = [b"".join(p[s] for p in parts) for s in range(S)]
    for s in range(S):
        assert got[s] == want[s], "stream %d differs (gop %d, F %d, %d calls, pipelined %s, renumberings %s; dropped %d, made again %d)" % (
            s, gop, F, ncalls, pipelined, renum, dropped, remedied)
