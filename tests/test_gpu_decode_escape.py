"""What the decoder's int16 symbol path cannot hold exactly must not be decoded differently from the reference:
  * a symbol beyond int16 -- not producible from 8-bit video, but a parsable stream may hold one and hzcc_dec
    (hzcc.c:295-435) has no such limit (round 2 saturated it silently);
  * a cell shared by two scan regions (250x130, 960x540) whose LATER symbol is absent keeps the EARLIER region's
    dequantised value in the reference decoder.
Hand-built plane payloads (Python restatement of hzcc_enc's emission order, hzcc.c:137-293, and of the plane framing
:449-476) are spliced into a real P-picture packet; the oracle decoder (pinned to the reference) gives the expected frames;
the drop-in dsv_dec and the batched decoder (host and device output) must equal them, and the device must have taken the
int32 second pass exactly where one is needed."""
import ctypes as C
import importlib

import numpy as np
import pytest

import _cabi as A
from test_gpu_stream import product_decode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


class BW:
    """MSB-first bit writer with the interleaved exp-Golomb codes (bs.c:129-206)"""

    def __init__(self):
        self.bits = []

    def put(self, n, v):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def ueg(self, v):
        m = v + 1
        k = m.bit_length() - 1
        for i in range(k - 1, -1, -1):
            self.bits.append(0)
            self.bits.append((m >> i) & 1)
        self.bits.append(1)

    def seg(self, v):
        self.ueg(abs(v))
        if v:
            self.bits.append(1 if v < 0 else 0)

    def neg(self, v):
        self.ueg(abs(v) - 1)
        self.bits.append(1 if v < 0 else 0)

    def bytes(self):
        self.align()
        a = np.packbits(np.array(self.bits, dtype=np.uint8))
        return a.tobytes()


def plane_payload(dc, entries):
    """bytes that follow a plane's 32-bit length: SEG(DC), run count, UEG(run) / NEG(value) chain, end-of-plane symbol.
    entries: (scan position, symbol) pairs, positions increasing, position 0 is the DC's cell and never coded"""
    w = BW()
    w.seg(dc)
    w.align()
    w.put(32, len(entries))
    w.align()
    prev, stored = 0, 0
    for pos, v in entries:
        assert pos > prev - 1 and v != 0
        w.ueg(pos - prev)          # zeros skipped since the cell after the previous non-zero (run restarts at 0 there)
        if stored:
            w.neg(stored)
        stored = v
        prev = pos + 1
    if stored:
        w.neg(stored)
    w.align()
    w.put(8, 0x55)
    w.align()
    return w.bytes()


class BR:
    def __init__(self, data):
        self.d, self.pos = data, 0

    def bit(self):
        b = (self.d[self.pos >> 3] >> (7 - (self.pos & 7))) & 1
        self.pos += 1
        return b

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v

    def align(self):
        self.pos = (self.pos + 7) & ~7

    def ueg(self):
        m = 1
        while not self.bit():
            m = (m << 1) | self.bit()
        return m - 1


def plane_offsets(pkt):
    """byte offsets of the three planes' 32-bit length fields in a picture packet (dsv_decoder.c:335-400)"""
    r = BR(pkt)
    r.pos = 14 * 8
    r.align(); r.bits(32); r.align(); r.ueg(); r.ueg(); r.align()
    r.align(); n = r.ueg(); r.align(); r.pos += 8 * n
    if pkt[5] & 1:
        r.align()
        for _ in range(4):
            n = r.ueg(); r.align(); r.pos += 8 * n
    r.align()
    r.bits(11)
    offs = []
    for _ in range(3):
        r.align()
        offs.append(r.pos >> 3)
        plen = r.bits(32)
        r.align()
        r.pos += 8 * plen
    assert (r.pos >> 3) == len(pkt)
    return offs


def splice(pkt, planes):
    """the packet with planes {index: payload bytes} replaced; next-link word updated"""
    offs = plane_offsets(pkt)
    out = bytearray(pkt[:offs[0]])
    for p in range(3):
        if p in planes:
            pay = planes[p]
        else:
            ln = int.from_bytes(pkt[offs[p]:offs[p] + 4], "big")
            pay = pkt[offs[p] + 4:offs[p] + 4 + ln]
        out += len(pay).to_bytes(4, "big") + pay
    out[10:14] = len(out).to_bytes(4, "big")
    return bytes(out)


def region_base(w, h, l, s):
    """first scan position of sub-band s (1 LH, 2 HL, 3 HH) of scan level l (hzcc.c:30-48: round-up dimensions per level)"""
    dim = lambda v, lv: (v + (1 << (3 - lv)) - 1) >> (3 - lv)
    base = dim(w, 0) * dim(h, 0)
    for lv in range(l):
        base += 3 * dim(w, lv) * dim(h, lv)
    return base + (s - 1) * dim(w, l) * dim(h, l), dim(w, l)


def _expect_and_decode(pkg, stream, w, h, fmt, want_redone):
    want = A.orc_decode(stream, w, h, fmt)
    assert len(want) == 2
    # 1. the drop-in dsv_dec (one picture per call, host frame)
    got = product_decode(pkg, stream)
    assert len(got) == 2
    for t in range(2):
        A.assert_same("dsv_dec frame %d" % t, got[t], want[t])
    # 2. the batched decoder, host output and device output (flags settled lazily: at the next call / the sync)
    L = pkg.lib()
    L.dsvg_ctx_decoder_redone.restype = C.c_long
    L.dsvg_ctx_decoder_redone.argtypes = [C.c_void_p]
    pk = A.split_packets(stream)
    for on_device in (False, True):
        d = pkg.DecBatch(w, h, fmt, 2)
        try:
            k = 0
            for p in pk:
                out, status, fnum = d.decode([p, p], on_device=on_device)
                if status[0] == 0 and (p[5] & 4):
                    frames = d.download() if on_device else out
                    A.assert_same("batched decoder (device output %s) frame %d" % (on_device, k), frames[0], want[k])
                    A.assert_same("batched decoder stream 1 frame %d" % k, frames[1], want[k])
                    k += 1
            assert k == 2
            redone = L.dsvg_ctx_decoder_redone(d.ctx)
            assert (redone >= 1) == want_redone, "calls decoded again from int32 coefficients: %d" % redone
        finally:
            d.close()


def _two_picture_stream(w, h, fmt, seed):
    clip = A.gen_clip(w, h, fmt, seed, 2, style=0)
    stream, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, qp=85, gop=12, rc_mode_cli=1))
    pk = A.split_packets(stream)
    pics = [i for i, p in enumerate(pk) if p[5] & 4]
    assert len(pics) == 2 and (pk[pics[1]][5] & 1)
    return pk, pics[1]


@pytest.mark.parametrize("plane", [0, 2])
@pytest.mark.parametrize("sym", [40000, -33000, 32767, 9])
def test_intra_picture_symbols_beyond_int16(pkg, orc, plane, sym):
    """I pictures take the symbol planes too; their inverse dequantises in 32 bits, so only a symbol beyond int16 needs the int32 pass"""
    w, h, fmt = 352, 288, A.SUBSAMP_444
    pk, ip = _two_picture_stream(w, h, fmt, 0xE5CA9F)
    ii = [i for i, p in enumerate(pk) if (p[5] & 4) and not (p[5] & 1)][0]
    b2, sw2 = region_base(w, h, 2, 3)                  # level-2 HH: shift quantiser
    b0, sw0 = region_base(w, h, 0, 1)                  # level-0 LH
    entries = sorted([(3, -4), (b0 + 2 * sw0 + 5, 6), (b2 + 17 * sw2 + 40, sym), (b2 + 60 * sw2 + 3, -2)])
    pk[ii] = splice(pk[ii], {plane: plane_payload(-11, entries)})
    _expect_and_decode(pkg, b"".join(pk), w, h, fmt, want_redone=abs(sym) > 32767)


@pytest.mark.parametrize("plane", [0, 1])
@pytest.mark.parametrize("sym", [40000, -70000, 32767, 500, 1])
def test_symbol_beyond_the_encoders_range_is_decoded_like_the_reference(pkg, orc, plane, sym):
    """beyond int16 (40000, -70000), inside int16 but far beyond what level 1 of an 8-bit residual can hold (32767, 500: the
    dequantised value leaves the range the packed int16 inverse is exact for), and an ordinary symbol (1: no second pass)"""
    w, h, fmt = 352, 288, A.SUBSAMP_444
    pk, ip = _two_picture_stream(w, h, fmt, 0xE5CA9E)
    b2, sw2 = region_base(w, h, 2, 1)                  # level-2 LH: shift quantiser
    b1, sw1 = region_base(w, h, 1, 2)                  # level-1 HL
    entries = sorted([(5, 3), (b1 + 4 * sw1 + 9, -2), (b2 + 10 * sw2 + 10, sym), (b2 + 30 * sw2 + 77, 1)])
    pk[ip] = splice(pk[ip], {plane: plane_payload(7, entries)})
    _expect_and_decode(pkg, b"".join(pk), w, h, fmt, want_redone=abs(sym) > 1)


@pytest.mark.parametrize("plane", [0, 2])
@pytest.mark.parametrize("later_present", [False, True])
def test_shared_scan_cell_keeps_the_earlier_value_when_the_later_symbol_is_absent(pkg, orc, plane, later_present):
    """250x130: column 63 of the coefficient plane belongs to level-0 LH (its 32nd column) AND to level-1 LH (its first)"""
    w, h, fmt = 250, 130, A.SUBSAMP_444
    pk, ip = _two_picture_stream(w, h, fmt, 0xE5CA11)
    e0, esw = region_base(w, h, 0, 1)                  # level-0 LH: 32 x 17, origin column 32
    l1, lsw = region_base(w, h, 1, 1)                  # level-1 LH: 63 x 33, origin column 63
    assert esw == 32 and lsw == 63
    entries = [(e0 + 5 * esw + 31, 3), (e0 + 7 * esw + 31, 2), (l1 + 7 * lsw + 0, -1), (l1 + 9 * lsw + 0, 2)]
    if later_present:
        entries.append((l1 + 5 * lsw + 0, 1))          # the later region's symbol of the same cell: it wins, nothing special
    pk[ip] = splice(pk[ip], {plane: plane_payload(-3, sorted(entries))})
    _expect_and_decode(pkg, b"".join(pk), w, h, fmt, want_redone=not later_present)


def test_natural_1080p_stream_takes_the_symbol_path_for_chroma_too(pkg, orc):
    """960x540 chroma planes share scan cells: P pictures now go through the symbol planes there as well -- and a natural stream
    (every later symbol the encoder wrote, hzcc.c:172-184) needs no second pass"""
    w, h, fmt, n = 1920, 1080, A.SUBSAMP_420, 4
    clip = A.gen_clip(w, h, fmt, 0xE5CA20, n, style=1)
    stream, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, qp=85, gop=12, rc_mode_cli=1))
    want = A.orc_decode(stream, w, h, fmt)
    L = pkg.lib()
    L.dsvg_ctx_decoder_redone.restype = C.c_long
    L.dsvg_ctx_decoder_redone.argtypes = [C.c_void_p]
    d = pkg.DecBatch(w, h, fmt, 1)
    try:
        k = 0
        for p in A.split_packets(stream):
            out, status, fnum = d.decode([p])
            if status[0] == 0 and (p[5] & 4):
                A.assert_same("frame %d" % k, out[0], want[k])
                k += 1
        assert k == n
        # (the encoder's own second quantiser pass may zero a shared cell the first pass kept: then the decoder must keep the
        # earlier value and takes the second pass -- legal, but it must stay the exception)
        assert L.dsvg_ctx_decoder_redone(d.ctx) <= 1
    finally:
        d.close()
