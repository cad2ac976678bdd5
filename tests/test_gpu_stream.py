"""GPU parity at the stream level: whole .dsv streams produced by the MI355X path (C session layer +
HIP kernels, through the C ABI) must equal, byte for byte, the streams of the oracle session (pinned
to the real reference CLI by test_oracle_vs_ref.py); the GPU decoder must reproduce the oracle's
decoded frames; GOP-sharded batches must equal the serial stream."""
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


def first_diff(a, b):
    n = min(len(a), len(b))
    aa = np.frombuffer(a[:n], np.uint8)
    bb = np.frombuffer(b[:n], np.uint8)
    d = np.nonzero(aa != bb)[0]
    return int(d[0]) if d.size else n


def explain(got, want):
    pg, pw = A.split_packets(got), A.split_packets(want)
    msg = ["len %d vs %d, packets %d vs %d, first differing byte %d" % (len(got), len(want), len(pg), len(pw), first_diff(got, want))]
    for i, (x, y) in enumerate(zip(pg, pw)):
        if x != y:
            msg.append("packet %d type %#x/%#x len %d/%d first diff at %d" % (i, x[5], y[5], len(x), len(y), first_diff(x, y)))
            break
    return "; ".join(msg)


CASES = [
    # w, h, fmt, frames, style, kwargs (CLI-style)
    (352, 288, A.SUBSAMP_420, 6, 0, dict(qp=85, gop=0, rc_mode_cli=1)),
    (352, 288, A.SUBSAMP_420, 14, 0, dict(qp=85, gop=12, rc_mode_cli=1)),
    (352, 288, A.SUBSAMP_420, 14, 1, dict(qp=85, gop=12, rc_mode_cli=1)),
    (352, 288, A.SUBSAMP_420, 9, 2, dict(qp=85, gop=12, rc_mode_cli=1)),
    (352, 288, A.SUBSAMP_420, 6, 1, dict(qp=85, gop=12, rc_mode_cli=1, ipct=20)),
    (352, 288, A.SUBSAMP_420, 10, 1, dict(qp=30, gop=5, rc_mode_cli=1, scd=0)),
    (352, 288, A.SUBSAMP_420, 14, 1, dict(qp=60, gop=12, rc_mode_cli=0)),                 # ABR auto bitrate
    (352, 288, A.SUBSAMP_420, 6, 0, dict(qp=85, gop=0, rc_mode_cli=0)),                  # intra-only ABR (cfg 1)
    (320, 240, A.SUBSAMP_444, 8, 1, dict(qp=95, gop=12, rc_mode_cli=1)),
    (320, 240, A.SUBSAMP_422, 8, 1, dict(qp=85, gop=12, rc_mode_cli=0, kbps=800)),
    (352, 288, A.SUBSAMP_411, 6, 0, dict(qp=85, gop=12, rc_mode_cli=1)),
    (704, 480, A.SUBSAMP_420, 7, 2, dict(qp=70, gop=12, rc_mode_cli=1)),
    (176, 144, A.SUBSAMP_420, 34, 2, dict(qp=85, gop=30, rc_mode_cli=0)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_stream_bit_exact(pkg, orc, case):
    w, h, fmt, n, style, kw = CASES[case]
    clip = A.gen_clip(w, h, fmt, 0xABC000 + case, n, style=style)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    got = pkg.encode_clip(clip, w, h, fmt, **kw)
    assert got == want, explain(got, want)


def test_1080p_gop12_bit_exact(pkg, orc):
    """BASELINE config 3 shape (1920x1080 4:2:0 GOP=12 CRF), two GOPs"""
    w, h, fmt = 1920, 1080, A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0x10800003, 24, style=0)
    kw = dict(qp=85, gop=12, rc_mode_cli=1)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    got = pkg.encode_clip(clip, w, h, fmt, **kw)
    assert got == want, explain(got, want)
    # GOP-sharded batch == serial stream
    sharded = pkg.encode_gops(clip, w, h, fmt, 12, qp=85, rc_mode_cli=1)
    assert sharded == want, explain(sharded, want)


def test_1080p_intra_bit_exact(pkg, orc):
    """BASELINE config 2 shape (1920x1080 4:2:0 intra only)"""
    w, h, fmt = 1920, 1080, A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0x10800001, 4, style=1)
    kw = dict(qp=85, gop=0, rc_mode_cli=1)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    got = pkg.encode_clip(clip, w, h, fmt, **kw)
    assert got == want, explain(got, want)


@pytest.mark.parametrize("gop,qp", [(0, 100), (0, 97), (12, 100)])
def test_dense_chunks_with_long_runs_and_large_symbols(pkg, orc, gop, qp):
    """hzcc.c:137-293 / bs.c:129-206 on the chunks the packed emit path has a detour for: a 1280-wide picture whose left half is
    full-range noise and whose right half is flat -- every level-1 subband row is 320 symbols of several hundred (quantiser 1 at
    -qp 100) followed by a run of 320 zeros, so a chunk of 2048 scan cells holds far more than 128 entries (rounds of 256 entries)
    AND codes past the 31 bits those rounds join in registers (a round of 256 then falls back to four rounds of 64)"""
    w, h, fmt, n = 1280, 256, A.SUBSAMP_420, 3
    rng = np.random.default_rng(0xDE45E + gop + qp)
    cw, ch = A.chroma_dims(w, h, fmt)
    clip = np.empty((n, A.frame_bytes(w, h, fmt)), dtype=np.uint8)
    for t in range(n):
        y = np.full((h, w), 128, dtype=np.uint8)
        y[:, : w // 2] = rng.integers(0, 256, size=(h, w // 2), dtype=np.uint8)
        if t:
            y[:, w // 2:] = 128 + 8 * t                                  # (P pictures: a flat residual on the right, noise on the left)
        c = np.full((2, ch, cw), 128, dtype=np.uint8)
        c[:, :, : cw // 2] = rng.integers(0, 256, size=(2, ch, cw // 2), dtype=np.uint8)
        clip[t] = np.concatenate([y.ravel(), c.ravel()])
    kw = dict(qp=qp, gop=gop, rc_mode_cli=1, scd=0)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    got = pkg.encode_clip(clip, w, h, fmt, **kw)
    assert got == want, explain(got, want)


def test_gop_sharding_matches_serial_cif(pkg, orc):
    w, h, fmt = 352, 288, A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0x5A4D, 36, style=0)
    kw = dict(qp=85, rc_mode_cli=1)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, gop=12, **kw))
    got = pkg.encode_gops(clip, w, h, fmt, 12, **kw)
    assert got == want, explain(got, want)


def test_pipelined_submit_collect_equals_plain_encode(pkg, orc):
    """software-pipelined batches (two HIP streams, two batches in flight) give the same bytes"""
    w, h, fmt, gop, S, nb = 352, 288, A.SUBSAMP_420, 6, 3, 4
    clips = [A.gen_clip(w, h, fmt, 0x7100 + s, gop * nb, style=s % 3) for s in range(S)]
    cfg = pkg.make_encoder_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1)
    want = []
    for s in range(S):
        st, _ = A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1), eos=False)
        want.append(st)
    b = pkg.Batch(cfg, S, gop)
    got = [b""] * S
    batches = [np.stack([clips[s][i * gop:(i + 1) * gop] for s in range(S)]) for i in range(nb)]
    b.submit(batches[0])
    for i in range(1, nb):
        b.submit(batches[i])
        part = b.collect()
        got = [g + p for g, p in zip(got, part)]
    part = b.collect()
    got = [g + p for g, p in zip(got, part)]
    b.close()
    for s in range(S):
        assert got[s] == want[s], "stream %d: %s" % (s, explain(got[s], want[s]))


@pytest.mark.parametrize("pinned", [True, False])
def test_staged_host_ingest_equals_plain_encode(pkg, orc, pinned):
    """host-resident clips through the double-buffered ingest (dsv1_batch_stage: the upload of batch i+1 is queued
    before batch i is submitted and batch i-1 collected): every batch has different content, three host buffers rotate"""
    w, h, fmt, gop, S, nb = 352, 288, A.SUBSAMP_420, 4, 3, 6
    clips = [A.gen_clip(w, h, fmt, 0x7300 + s, gop * nb, style=s % 3) for s in range(S)]
    cfg = pkg.make_encoder_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1)
    want = [A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1), eos=False)[0] for s in range(S)]
    b = pkg.Batch(cfg, S, gop)
    shape = (S, gop, clips[0].shape[1])
    hostbuf = [b.pinned(shape) if pinned else np.empty(shape, np.uint8) for _ in range(3)]   # two staged + one being coded

    def fill(i):
        hb = hostbuf[i % 3]
        for s in range(S):
            hb[s] = clips[s][i * gop:(i + 1) * gop]
        return hb

    got = [b""] * S
    b.stage(fill(0))
    b.stage(fill(1))
    b.submit(hostbuf[0])
    for i in range(1, nb):
        if i + 1 < nb:
            b.stage(fill(i + 1))             # buffer of batch i-2: collected in the previous iteration
        b.submit(hostbuf[i % 3])
        part = b.collect()                   # batch i-1
        got = [g + p for g, p in zip(got, part)]
    part = b.collect()
    got = [g + p for g, p in zip(got, part)]
    b.close()
    for s in range(S):
        assert got[s] == want[s], "stream %d: %s" % (s, explain(got[s], want[s]))


def _drive_dsv_enc(pkg, clip, w, h, fmt, **cli):
    """the frame-at-a-time API driven the way dsv_main.c:506-537 does: whatever count comes back is written out"""
    L = pkg.lib()
    enc = pkg.make_encoder_cfg(w, h, fmt, **cli)
    L.dsv_enc_start(C.byref(enc))
    L.dsv_load_planar_frame.restype = C.c_void_p
    L.dsv_load_planar_frame.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsv_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    out = b""
    counts = []
    bufs = (pkg.Buf * 4)()
    scratch = np.empty_like(clip[0])
    for t in range(clip.shape[0]):
        scratch[...] = clip[t]                           # the caller reuses ONE picture buffer (dsv_main.c:506-520)
        frame = L.dsv_load_planar_frame(fmt, scratch.ctypes.data, w, h)
        nb = L.dsv_enc(C.byref(enc), frame, bufs) & 3
        counts.append(nb)
        for i in range(nb):
            out += C.string_at(bufs[i].data, bufs[i].len)
            L.dsv_buf_free(C.byref(bufs[i]))
        scratch[...] = 0xA5                              # ... and overwrites it right after the call
    L.dsv_enc_end_of_stream(C.byref(enc), bufs)
    out += C.string_at(bufs[0].data, bufs[0].len)
    L.dsv_buf_free(C.byref(bufs[0]))
    L.dsv_enc_free(C.byref(enc))
    return out, counts


@pytest.mark.parametrize("n,cli", [
    (7, dict(qp=85, gop=12, rc_mode_cli=1)),                 # shorter than one batch: everything comes out at end of stream
    (31, dict(qp=85, gop=12, rc_mode_cli=1)),                # two full batches + a tail of 7, a forced-intra style clip
    (40, dict(qp=70, gop=9, rc_mode_cli=1, scd=0)),
    (21, dict(qp=85, gop=0, rc_mode_cli=1)),                 # intra-only CRF: batches of 16
    (9, dict(qp=60, gop=12, rc_mode_cli=0)),                 # ABR: shorter than one analysis group -- everything at end of stream
    (31, dict(qp=60, gop=12, rc_mode_cli=0, kbps=900)),      # ABR: two groups of 12 + a tail of 7, the rate control working against a tight budget
])
def test_drop_in_dsv_enc_api(pkg, orc, monkeypatch, n, cli):
    """frame-at-a-time dsv_enc_* API (dsv_encoder.h:112-121): CRF streams are pipelined in batches behind it (deferred
    output, flushed by dsv_enc_end_of_stream); ABR streams gather a group of frames for a common analysis pass and code them
    one after the other (every packet's size feeds the next quantiser) -- the bytes are the serial encoder's"""
    monkeypatch.setenv("DSV1_ENC_LOOKAHEAD", "12" if cli["gop"] else "16")     # batches of a GOP, as the comments of the cases say
    w, h, fmt = 352, 288, A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0xD209 + n, n, style=2)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli))
    out, counts = _drive_dsv_enc(pkg, clip, w, h, fmt, **cli)
    assert out == want, explain(out, want)
    assert counts[0] == 0 and max(counts) <= 2              # deferred, never more than the reference's two buffers


@pytest.mark.parametrize("n,cli", [
    (31, dict(qp=85, gop=12, rc_mode_cli=1)),                # CRF: two full batches + a tail
    (31, dict(qp=60, gop=12, rc_mode_cli=0, kbps=900)),      # ABR
    (5, dict(qp=85, gop=12, rc_mode_cli=1)),                 # shorter than one batch: everything is owed when the input ends
])
def test_drop_in_reference_packet_contract_with_flush_calls(pkg, orc, monkeypatch, n, cli):
    """dsv_encoder.c:766-810 for a library caller: one packet per DSV_BUF, at most two per call, exactly ONE EOS packet from
    dsv_enc_end_of_stream -- reached by draining the session with flush calls dsv_enc(enc, NULL, bufs) first (dsv1_api.h)"""
    monkeypatch.setenv("DSV1_ENC_LOOKAHEAD", "12")
    w, h, fmt = 352, 288, A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0xD209 + n, n, style=2)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli))
    L = pkg.lib()
    enc = pkg.make_encoder_cfg(w, h, fmt, **cli)
    L.dsv_enc_start(C.byref(enc))
    L.dsv_load_planar_frame.restype = C.c_void_p
    L.dsv_load_planar_frame.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsv_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    bufs = (pkg.Buf * 4)()
    pk = []

    def take(nb):
        assert 0 <= nb <= 2
        for i in range(nb):
            b_ = C.string_at(bufs[i].data, bufs[i].len)
            assert len(A.split_packets(b_)) == 1 and len(A.split_packets(b_)[0]) == len(b_)     # ONE whole packet per buffer
            pk.append(b_)
            L.dsv_buf_free(C.byref(bufs[i]))

    assert L.dsv_enc(C.byref(enc), None, bufs) == 0          # a flush call before the first frame owes nothing
    for t in range(n):
        take(L.dsv_enc(C.byref(enc), L.dsv_load_planar_frame(fmt, clip[t].ctypes.data, w, h), bufs) & 3)
    flushes = 0
    while True:
        nb = L.dsv_enc(C.byref(enc), None, bufs) & 3
        if nb == 0:
            break
        take(nb)
        flushes += 1
    assert flushes >= 1
    L.dsv_enc_end_of_stream(C.byref(enc), bufs)
    eos = C.string_at(bufs[0].data, bufs[0].len)
    L.dsv_buf_free(C.byref(bufs[0]))
    L.dsv_enc_free(C.byref(enc))
    assert len(eos) == 14 and eos[5] == 0x10                 # the EOS packet alone
    out = b"".join(pk) + eos
    assert out == want, explain(out, want)
    # a metadata packet never travels without the picture that follows it (dsv_encoder.c:804-810)
    assert all(not (p[5] == 0 and i + 1 < len(pk) and not (pk[i + 1][5] & 4)) for i, p in enumerate(pk))


@pytest.mark.parametrize("rc", [1, 0])
def test_drop_in_dsv_enc_unpipelined_switch(pkg, orc, monkeypatch, rc):
    """DSV1_ENC_PIPELINE=0: one picture per call, CRF and ABR"""
    monkeypatch.setenv("DSV1_ENC_PIPELINE", "0")
    w, h, fmt, n = 352, 288, A.SUBSAMP_420, 7
    clip = A.gen_clip(w, h, fmt, 0xD209, n, style=2)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, qp=85, gop=12, rc_mode_cli=rc))
    out, counts = _drive_dsv_enc(pkg, clip, w, h, fmt, qp=85, gop=12, rc_mode_cli=rc)
    assert out == want and all(c >= 1 for c in counts)


@pytest.mark.parametrize("cli", [dict(qp=85, gop=6, rc_mode_cli=1), dict(qp=60, gop=12, rc_mode_cli=0, kbps=900)])
def test_drop_in_strict_packet_contract_by_api_call(pkg, orc, cli):
    """dsv1_enc_set_strict_packets(enc, 1) (round 6; no environment variable): the session is the reference's own contract, dsv_encoder.c:766-810 --
    EVERY dsv_enc call returns its frame's packets (metadata + picture at a GOP start, the picture otherwise), one whole packet per DSV_BUF, and
    dsv_enc_end_of_stream returns the 14-byte EOS packet alone.  After the first frame the switch is refused."""
    w, h, fmt, n = 352, 288, A.SUBSAMP_420, 14
    clip = A.gen_clip(w, h, fmt, 0xD2A9, n, style=2)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli))
    wantpk = A.split_packets(want)
    L = pkg.lib()
    L.dsv1_enc_set_strict_packets.argtypes = [C.c_void_p, C.c_int]
    enc = pkg.make_encoder_cfg(w, h, fmt, **cli)
    L.dsv_enc_start(C.byref(enc))
    assert L.dsv1_enc_set_strict_packets(C.byref(enc), 1) == 0
    L.dsv_load_planar_frame.restype = C.c_void_p
    L.dsv_load_planar_frame.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsv_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    bufs = (pkg.Buf * 4)()
    pk, k = [], 0
    for t in range(n):
        nb = L.dsv_enc(C.byref(enc), L.dsv_load_planar_frame(fmt, clip[t].ctypes.data, w, h), bufs) & 3
        expect = 2 if wantpk[k][5] == 0 else 1               # a metadata packet in front of the picture where the reference sends one
        assert nb == expect, (t, nb, expect)
        for i in range(nb):
            b_ = C.string_at(bufs[i].data, bufs[i].len)
            assert b_ == wantpk[k], "frame %d buffer %d" % (t, i)
            pk.append(b_)
            k += 1
            L.dsv_buf_free(C.byref(bufs[i]))
        if t == 0:
            assert L.dsv1_enc_set_strict_packets(C.byref(enc), 0) != 0      # the session exists: refused
    L.dsv_enc_end_of_stream(C.byref(enc), bufs)
    eos = C.string_at(bufs[0].data, bufs[0].len)
    L.dsv_buf_free(C.byref(bufs[0]))
    L.dsv_enc_free(C.byref(enc))
    assert len(eos) == 14 and eos[5] == 0x10
    assert b"".join(pk) + eos == want


class Decoder(C.Structure):
    _fields_ = [("vidmeta", A.Meta), ("ref", C.c_void_p), ("draw_info", C.c_int), ("got_metadata", C.c_int)]


@pytest.mark.parametrize("case", [1, 3, 6, 8, 9])
def test_gpu_decoder_matches_oracle(pkg, orc, case):
    w, h, fmt, n, style, kw = CASES[case]
    _decode_and_compare(pkg, w, h, fmt, n, style, kw, 0xABC000 + case)


@pytest.mark.parametrize("kw,n,style", [(dict(qp=95, gop=0, rc_mode_cli=1), 2, 1), (dict(qp=85, gop=12, rc_mode_cli=1), 3, 2)])
def test_gpu_decoder_1080p(pkg, orc, kw, n, style):
    """plane payloads far beyond one pass of the device-side entropy parser (16 KB), intra and inter"""
    _decode_and_compare(pkg, 1920, 1080, A.SUBSAMP_420, n, style, kw, 0xABD001 + n)


def product_decode(pkg, stream):
    """the drop-in dsv_dec, packet by packet: list of decoded frames (packed planar)"""
    L = pkg.lib()
    L.dsv_alloc.restype = C.c_void_p
    L.dsv_dec.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    L.dsv_frame_ref_dec.argtypes = [C.c_void_p]
    dec = Decoder()
    got = []
    try:
        for p in A.split_packets(stream):
            buf = pkg.Buf()
            mem = L.dsv_alloc(len(p))
            C.memmove(mem, p, len(p))
            buf.data = C.cast(mem, C.POINTER(C.c_uint8))
            buf.len = len(p)
            frame = C.c_void_p(None)
            fn = C.c_uint32(0)
            rc = L.dsv_dec(C.byref(dec), C.byref(buf), C.byref(frame), C.byref(fn))
            if rc == 0 and frame.value:
                f = C.cast(frame, C.POINTER(A.Frame)).contents
                planes = []
                for c in range(3):
                    pl = f.planes[c]
                    arr = np.ctypeslib.as_array(pl.data, shape=(pl.h * pl.stride,))
                    planes.append(np.lib.stride_tricks.as_strided(arr, shape=(pl.h, pl.w), strides=(pl.stride, 1)).copy().reshape(-1))
                got.append(np.concatenate(planes))
                L.dsv_frame_ref_dec(frame)
            elif rc == 1:
                raise AssertionError("dsv_dec error: %s" % L.dsvg_last_error())
    finally:
        L.dsv_dec_free(C.byref(dec))
    return got


def _decode_and_compare(pkg, w, h, fmt, n, style, kw, seed, check_recon=True):
    clip = A.gen_clip(w, h, fmt, seed, n, style=style)
    stream, recs = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw), want_recon=True)
    want = A.orc_decode(stream, w, h, fmt)
    got = product_decode(pkg, stream)
    assert len(got) == len(want) == n
    for t in range(n):
        A.assert_same("decoded frame %d" % t, got[t], want[t])
        if check_recon:     # holds unless HZCC scan regions overlap (a shared cell can re-quantise to 0 in the encoder's
            # second pass while the decoder keeps the first symbol: encoder/decoder drift of the reference itself)
            A.assert_same("decode == encoder recon %d" % t, got[t], recs[t])


@pytest.mark.parametrize("geom", [(352, 288, A.SUBSAMP_420), (320, 240, A.SUBSAMP_444), (1920, 1080, A.SUBSAMP_420)])
@pytest.mark.parametrize("on_device", [False, True])
def test_batched_decoder_matches_oracle(pkg, orc, geom, on_device):
    """dsv1_decbatch_*: one packet per stream per call, all picture packets of a call decoded as one device batch.
    Streams differ in content, GOP length (so I and P pictures meet in one call) and length (EOS comes at different
    calls); stream 3 is intra-only (non-reference pictures)."""
    w, h, fmt = geom
    big = w >= 1920
    S = 4
    gops = [3, 5, 4, 0]
    nfr = [3, 4, 3, 2] if big else [7, 9, 6, 4]
    streams, want = [], []
    for s in range(S):
        clip = A.gen_clip(w, h, fmt, 0xDEC0 + 16 * s + w, nfr[s], style=s % 3)
        st, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, qp=85 if s != 1 else 60, gop=gops[s], rc_mode_cli=1))
        streams.append(A.split_packets(st))
        want.append(A.orc_decode(st, w, h, fmt))
        assert len(want[s]) == nfr[s]
    d = pkg.DecBatch(w, h, fmt, S)
    got = [[] for _ in range(S)]
    fnums = [[] for _ in range(S)]
    eos = bytes(streams[0][-1])
    ncalls = max(len(p) for p in streams)
    for k in range(ncalls):
        pk = [streams[s][k] if k < len(streams[s]) else eos for s in range(S)]
        if on_device:
            _, status, fnum = d.decode(pk, on_device=True)
            out = d.download()
        else:
            out, status, fnum = d.decode(pk)
        for s in range(S):
            if k >= len(streams[s]):
                assert status[s] == 2          # DSV_DEC_EOS
            elif streams[s][k][5] & 4:         # picture packet
                assert status[s] == 0, "stream %d call %d: status %d" % (s, k, status[s])
                got[s].append(out[s].copy())
                fnums[s].append(fnum[s])
            else:
                assert status[s] in (2, 3)     # EOS / GOT_META
    d.close()
    for s in range(S):
        assert len(got[s]) == nfr[s]
        assert fnums[s] == list(range(nfr[s]))
        for t in range(nfr[s]):
            A.assert_same("stream %d frame %d" % (s, t), got[s][t], want[s][t])


@pytest.mark.parametrize("nstreams_code", [1, 2, 3])
def test_several_coding_streams_bit_exact(pkg, orc, nstreams_code, monkeypatch):
    """DSV1_CODE_STREAMS: the pictures of every frame step are shared out over several HIP streams (each group runs
    the whole chain on its own).  24 GOP streams with different content -- flat objects force intra blocks, so every
    group has its own intra-block list; style 0 streams have none -- must give the bytes of the one-stream path."""
    monkeypatch.setenv("DSV1_CODE_STREAMS", str(nstreams_code))
    w, h, fmt, gop, S = 352, 288, A.SUBSAMP_420, 5, 24
    clips = np.stack([A.gen_clip(w, h, fmt, 0x5C00 + s, gop, style=(s % 3)) for s in range(S)])
    cfg = pkg.make_encoder_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1)
    b = pkg.Batch(cfg, S, gop)          # the context reads the variable when it is created
    got = b.encode(clips)
    b.close()
    for s in range(S):
        want, _ = A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1), eos=False)
        assert got[s] == want, "stream %d: %s" % (s, explain(got[s], want))


def test_1080p_two_coding_streams_bit_exact(pkg, orc):
    """the default two coding streams at the bench shape: 16 streams (two clips, eight times each) of 1080p, one I and
    three P pictures: both halves must give the oracle's bytes"""
    w, h, fmt, gop, S = 1920, 1080, A.SUBSAMP_420, 4, 16
    two = [A.gen_clip(w, h, fmt, 0x10800003 + 31 * k, gop, style=k) for k in range(2)]
    want = [A.orc_encode(two[k], A.orc_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1), eos=False)[0] for k in range(2)]
    cfg = pkg.make_encoder_cfg(w, h, fmt, qp=85, gop=gop, rc_mode_cli=1)
    b = pkg.Batch(cfg, S, gop)
    assert b.code_streams(0) == 2
    got = b.encode(np.stack([two[s & 1] for s in range(S)]))
    b.close()
    for s in range(S):
        assert got[s] == want[s & 1], "stream %d: %s" % (s, explain(got[s], want[s & 1]))


@pytest.mark.parametrize("shift", [4, 8, 1])
def test_device_clip_at_an_odd_address(pkg, shift):
    """A caller's device-resident clip need not be 16-byte aligned: the frame load then takes its scalar paths and the
    pyramid its separate passes (k_unpack fuses the first two levels only for aligned frames) -- same stream."""
    w, h, fmt, n = 352, 288, A.SUBSAMP_420, 5
    cli = dict(qp=85, gop=12, rc_mode_cli=1)
    clip = A.gen_clip(w, h, fmt, 0x0DD0 + shift, n, style=0)
    want = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli), eos=False)[0]
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), 1, n)
    try:
        base = b.upload(np.concatenate([np.zeros(shift, np.uint8), clip.reshape(-1)]))
        got = b.encode(C.c_void_p(base.value + shift), on_device=True)[0]
    finally:
        b.close()
    assert got == want


@pytest.mark.parametrize("rep", [0, 1, 2])
def test_batched_decoder_back_to_back_calls_on_two_streams(pkg, orc, rep):
    """round 5: the batched decoder runs a call's motion compensation and the packing pass of device output on the second coding stream,
    beside the entropy decoding of the same / the next call (dsvg_decode_pictures, dsvg_pack_recons).  Sixteen streams, GOPs of
    different lengths, every call enqueued without waiting for the one before and packed into a device buffer of its own; all of
    them read back after ONE synchronisation at the end -- an ordering mistake between the streams shows as a wrong frame."""
    w, h, fmt, S, n = 352, 288, A.SUBSAMP_420, 16, 12
    streams, want = [], []
    for s in range(4):                                  # four distinct streams, repeated
        clip = A.gen_clip(w, h, fmt, 0xB2B0 + 7 * s + rep, n, style=(0, 1, 2, 5)[s])
        st, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, qp=(85, 60, 90, 75)[s], gop=(12, 5, 4, 7)[s], rc_mode_cli=1))
        pk_ = A.split_packets(st)
        streams.append(pk_[:1] + [p for p in pk_ if p[5] & 4])      # the first metadata packet, then the pictures (a GOP's repeated metadata left out: the streams stay in step)
        want.append(A.orc_decode(st, w, h, fmt))
    d = pkg.DecBatch(w, h, fmt, S)
    try:
        ncalls = len(streams[0])
        assert ncalls == n + 1 and all(len(p) == ncalls for p in streams)
        bufs, stat = [], []
        for k in range(ncalls):
            pk = [streams[s % 4][k] for s in range(S)]
            if not (pk[0][5] & 4):
                d.decode(pk)                            # metadata: nothing to pack
                continue
            buf = d.dev_alloc()
            _, status, fnum = d.decode(pk, out=buf, on_device=True)
            assert all(x == 0 for x in status), status
            bufs.append(buf)
        frames = [d.download(b) for b in bufs]          # (the first download synchronises)
    finally:
        d.close()
    assert len(frames) == n
    for t in range(n):
        for s in range(S):
            A.assert_same("stream %d frame %d" % (s, t), frames[t][s], want[s % 4][t])


@pytest.mark.parametrize("how", ["handle_after_pack", "join_own_stream"])
def test_packed_device_output_is_consumed_in_stream_order(pkg, orc, how):
    """advisor round 5: the packing pass of device output runs on the SECOND coding stream (n >= 4); include/dsvg.h promises that a consumer
    ordered on a dsvg_ctx_stream() handle fetched after the pass -- or on its own stream after dsvg_ctx_join -- reads finished frames.
    No dsvg_ctx_sync anywhere between the decode call and the read: an asynchronous copy on that stream, then a wait for THAT stream only."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    w, h, fmt, S, n = 1280, 720, A.SUBSAMP_420, 16, 6
    clip = A.gen_clip(w, h, fmt, 0x0CDE5, n, style=2)
    st, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, qp=85, gop=12, rc_mode_cli=1))
    pks = [p for p in A.split_packets(st) if (p[5] & 4) or p[5] == 0]
    want = A.orc_decode(st, w, h, fmt)
    d = pkg.DecBatch(w, h, fmt, S)
    L = pkg.lib()
    own = C.c_void_p(None)
    try:
        if how == "join_own_stream":
            assert hip.hipStreamCreateWithFlags(C.byref(own), 1) == 0          # hipStreamNonBlocking
        k = 0
        for p in pks:
            if not (p[5] & 4):
                d.decode([p] * S)
                continue
            buf = d.dev_alloc()
            _, status, _ = d.decode([p] * S, out=buf, on_device=True)
            assert all(x == 0 for x in status), status
            if how == "handle_after_pack":
                s_ = L.dsvg_ctx_stream(d.ctx)
                assert s_
            else:
                assert L.dsvg_ctx_join(d.ctx, own) == 0, L.dsvg_last_error()
                s_ = own.value
            got = np.empty((S, d.frame_bytes), dtype=np.uint8)
            assert hip.hipMemcpyAsync(got.ctypes.data, buf, got.nbytes, 2, s_) == 0
            assert hip.hipStreamSynchronize(s_) == 0
            for s in (0, 7, S - 1):
                A.assert_same("picture %d stream %d" % (k, s), got[s], want[k])
            k += 1
        assert k == len(want)
    finally:
        d.close()
        if own:
            hip.hipStreamDestroy(own)
