"""GPU parity tests proper: the HIP path, driven through the C ABI of libdsv1_mi355x.so at the
operator seam (dsvg_op_* = twins of dsv_internal.h:94-109), must be BIT-EXACT against the oracle
(oracle/liborc.so, pinned to the real reference by test_oracle_vs_ref.py) and, where the compiled
reference travelled with the snapshot (oracle/_ref), against the reference itself."""
import ctypes as C

import numpy as np
import pytest

import _cabi as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prod():
    L = A.load_prod()
    assert L.dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return L


def rnd_plane(rng, w, h, smooth=True):
    if smooth:
        base = rng.integers(0, 256, size=(h // 8 + 2, w // 8 + 2)).astype(np.float64)
        img = np.kron(base, np.ones((8, 8)))[:h, :w] + rng.integers(-6, 7, size=(h, w))
        return np.clip(img, 0, 255).astype(np.uint8)
    return rng.integers(0, 256, size=(h, w), dtype=np.uint8)


def mk_frame(rng, w, h, fmt, orc, smooth=True):
    f = A.BorderedFrame(w, h, fmt)
    for i in range(3):
        pw, ph = f.dims[i]
        f.plane(i)[:, :] = rnd_plane(rng, pw, ph, smooth)
    orc.orc_frame_extend(f.ptr())
    return f


SBT_CASES = [(40, 36), (32, 48), (176, 144), (352, 288), (250, 130), (960, 540), (1920, 1080), (100, 36),
             (20, 30), (16, 16), (8, 32), (16, 10), (32, 32), (8, 8)]      # planes with 5, 4 and 3 transform levels


@pytest.mark.parametrize("w,h", SBT_CASES)
@pytest.mark.parametrize("isP", [1, 0])
def test_fwd_inv_sbt(prod, orc, w, h, isP):
    if not isP and ((w | h) & 1):
        pytest.skip("odd dims are undefined for the intra B4T in the reference")
    rng = np.random.default_rng(w * 131 + h * 7 + isP)
    f = mk_frame(rng, w, h, A.SUBSAMP_444, orc, smooth=(w * h) % 3 != 0)
    want = np.zeros(w * h, dtype=np.int32)
    got = np.zeros(w * h, dtype=np.int32)
    orc.orc_fwd_sbt(C.byref(f.c.planes[0]), C.byref(A.Coefs(A.i32p(want), w, h)), isP)
    A.chk(prod, prod.dsvg_op_fwd_sbt(C.byref(f.c.planes[0]), C.byref(A.Coefs(A.i32p(got), w, h)), isP))
    A.assert_same("fwd_sbt %dx%d isP=%d" % (w, h, isP), got, want, (h, w))
    for c in (0, 1):
        for q in (40, 313, 1500):
            co = want.copy()
            co[1:] = (co[1:] // 24) * 24          # coarse quantisation so the smoothing filter engages
            a, b = co.copy(), co.copy()
            fa, fb = A.BorderedFrame(w, h, A.SUBSAMP_444), A.BorderedFrame(w, h, A.SUBSAMP_444)
            orc.orc_inv_sbt(C.byref(fa.c.planes[c]), C.byref(A.Coefs(A.i32p(a), w, h)), q, isP, c)
            A.chk(prod, prod.dsvg_op_inv_sbt(C.byref(fb.c.planes[c]), C.byref(A.Coefs(A.i32p(b), w, h)), q, isP, c))
            A.assert_same("inv_sbt %dx%d isP=%d c=%d q=%d" % (w, h, isP, c, q), fb.plane(c), fa.plane(c), (h, w))


def test_fwd_sbt_odd_chroma_reads_border(prod, orc):
    """coefficient plane one column/row larger than the pixel plane (frame.c:41-42, sbt.c:584-590)"""
    rng = np.random.default_rng(3)
    w, h = 101, 77                       # 4:2:0 chroma 51x39 -> coefs 52x40
    f = mk_frame(rng, w, h, A.SUBSAMP_420, orc)
    cw, ch = A.coef_dims(w, h, A.SUBSAMP_420, 1)
    for isP in (1, 0):
        want = np.zeros(cw * ch, dtype=np.int32)
        got = np.zeros(cw * ch, dtype=np.int32)
        orc.orc_fwd_sbt(C.byref(f.c.planes[1]), C.byref(A.Coefs(A.i32p(want), cw, ch)), isP)
        A.chk(prod, prod.dsvg_op_fwd_sbt(C.byref(f.c.planes[1]), C.byref(A.Coefs(A.i32p(got), cw, ch)), isP))
        A.assert_same("fwd_sbt odd chroma isP=%d" % isP, got, want, (ch, cw))
        fa, fb = A.BorderedFrame(w, h, A.SUBSAMP_420), A.BorderedFrame(w, h, A.SUBSAMP_420)
        a, b = want.copy(), want.copy()
        orc.orc_inv_sbt(C.byref(fa.c.planes[1]), C.byref(A.Coefs(A.i32p(a), cw, ch)), 200, isP, 1)
        A.chk(prod, prod.dsvg_op_inv_sbt(C.byref(fb.c.planes[1]), C.byref(A.Coefs(A.i32p(b), cw, ch)), 200, isP, 1))
        A.assert_same("inv_sbt odd chroma isP=%d" % isP, fb.raw(), fa.raw())


def stab_for(rng, fw, fh, isP, cur_plane, flags=None):
    bw, bh, nbh, nbv = A.block_dims(fw, fh)
    meta = A.Meta(fw, fh, A.SUBSAMP_420, 30, 1, 1, 1)
    prm = A.Params(C.pointer(meta), 1, isP, bw, bh, nbh, nbv)
    sb = rng.integers(0, 4, size=nbh * nbv).astype(np.uint8) if flags is None else np.full(nbh * nbv, flags, np.uint8)
    return A.Stability(C.pointer(prm), A.u8p(sb), cur_plane, isP), (meta, prm, sb)


def parse_plane(buf):
    """(dc, nruns, [(scan position, value)...]) of one packed plane (hzcc.c:295-435 framing)"""
    bits = np.unpackbits(np.asarray(buf, dtype=np.uint8))
    pos = [32]

    def bit():
        b = int(bits[pos[0]]); pos[0] += 1; return b

    def ueg():
        m = 1
        while not bit():
            m = (m << 1) | bit()
        return m - 1

    def align():
        pos[0] = (pos[0] + 7) & ~7
    dc = ueg()
    if dc and bit():
        dc = -dc
    align()
    nruns = 0
    for _ in range(32):
        nruns = (nruns << 1) | bit()
    align()
    out = []
    if nruns > 0:
        p = ueg()
        for k in range(nruns):
            nxt = ueg() if k + 1 < nruns else None
            v = ueg() + 1
            if bit():
                v = -v
            out.append((p, v))
            if nxt is None:
                break
            p += 1 + nxt
    return dc, nruns, out


def explain_plane_diff(got, want):
    try:
        a, b = parse_plane(got), parse_plane(want)
    except Exception as e:      # noqa
        return "unparseable (%s)" % e
    msg = ["dc %d/%d nruns %d/%d pairs %d/%d" % (a[0], b[0], a[1], b[1], len(a[2]), len(b[2]))]
    for i, (x, y) in enumerate(zip(a[2], b[2])):
        if x != y:
            msg.append("first differing pair #%d: got %s want %s (prev %s)" % (i, x, y, a[2][i - 1] if i else None))
            break
    return "; ".join(msg)


HZ_CASES = [(352, 288, 0), (176, 144, 1), (960, 540, 1), (250, 130, 0), (1920, 1080, 0), (64, 64, 0), (16, 16, 1), (20, 30, 0), (8, 8, 1)]


@pytest.mark.parametrize("w,h,cur_plane", HZ_CASES)
@pytest.mark.parametrize("isP", [0, 1])
@pytest.mark.parametrize("q", [16, 313, 3000])
def test_encode_decode_plane(prod, orc, w, h, cur_plane, isP, q):
    rng = np.random.default_rng(w + h * 3 + isP * 5 + q + cur_plane)
    fw, fh = (w * 2, h * 2) if cur_plane else (w, h)
    st, keep = stab_for(rng, fw, fh, isP, cur_plane)
    scale = 40 if q < 1000 else 400
    co = rng.laplace(0, scale, size=(h, w)).astype(np.int32)
    co[: h // 8, : w // 8] *= 16
    co = co.reshape(-1)
    a, b = co.copy(), co.copy()
    bufa = np.zeros(w * h * 8 + 64, dtype=np.uint8)
    bufb = np.zeros_like(bufa)
    bsa, bsb = A.BS(A.u8p(bufa), 0), A.BS(A.u8p(bufb), 0)
    orc.orc_encode_plane(C.byref(bsa), C.byref(A.Coefs(A.i32p(a), w, h)), q, C.byref(st))
    A.chk(prod, prod.dsvg_op_encode_plane(C.byref(bsb), C.byref(A.Coefs(A.i32p(b), w, h)), q, C.byref(st)))
    n = max(bsa.pos, bsb.pos) // 8
    if bsa.pos != bsb.pos or not np.array_equal(bufa[:n], bufb[:n]):
        bad = np.nonzero(a != b)[0]
        raise AssertionError("packed plane differs: bits %d vs %d; %s; coef mismatches %d %s" % (
            bsb.pos, bsa.pos, explain_plane_diff(bufb[:n + 8], bufa[:n + 8]), bad.size,
            [(int(i) // w, int(i) % w, int(b[i]), int(a[i]), int(co[i])) for i in bad[:6]]))
    A.assert_same("dequantised coefficients", b, a, (h, w))
    plen = int.from_bytes(bufa[:4].tobytes(), "big")
    da = np.zeros(w * h, dtype=np.int32)
    db = np.zeros(w * h, dtype=np.int32)
    pa = bufa[4:4 + plen + 8].copy()
    pb = pa.copy()
    orc.orc_decode_plane(A.u8p(pa), plen, C.byref(A.Coefs(A.i32p(da), w, h)), q, C.byref(st))
    A.chk(prod, prod.dsvg_op_decode_plane(A.u8p(pb), plen, C.byref(A.Coefs(A.i32p(db), w, h)), q, C.byref(st)))
    A.assert_same("decoded coefficients", db, da, (h, w))


@pytest.mark.parametrize("w,h,q", [(352, 288, 16), (960, 540, 313)])
def test_decode_plane_truncated(prod, orc, w, h, q):
    """plane data cut short: the decoder stops at the first value whose bits reach the end (hzcc.c:337-339);
    the device-side parser must stop at the same entry for every cut, down to a few bytes"""
    rng = np.random.default_rng(w + q)
    st, keep = stab_for(rng, w, h, 1, 0)
    co = rng.laplace(0, 60, size=w * h).astype(np.int32)
    buf = np.zeros(w * h * 8 + 64, dtype=np.uint8)
    bs = A.BS(A.u8p(buf), 0)
    work = co.copy()
    orc.orc_encode_plane(C.byref(bs), C.byref(A.Coefs(A.i32p(work), w, h)), q, C.byref(st))
    plen = int.from_bytes(buf[:4].tobytes(), "big")
    for cut in (plen, plen - 1, plen - 2, plen // 2, plen // 3 + 1, 4097, 64, 12, 9):
        if cut > plen:
            continue
        pa = buf[4:4 + plen + 8].copy()
        pb = pa.copy()
        da = np.zeros(w * h, dtype=np.int32)
        db = np.zeros(w * h, dtype=np.int32)
        orc.orc_decode_plane(A.u8p(pa), cut, C.byref(A.Coefs(A.i32p(da), w, h)), q, C.byref(st))
        A.chk(prod, prod.dsvg_op_decode_plane(A.u8p(pb), cut, C.byref(A.Coefs(A.i32p(db), w, h)), q, C.byref(st)))
        A.assert_same("decoded coefficients, %d of %d bytes" % (cut, plen), db, da, (h, w))


def test_encode_plane_empty_and_dense(prod, orc):
    """nruns == 0 planes and very long zero runs (UEG > 16 bits), every coefficient non-zero"""
    rng = np.random.default_rng(9)
    w, h = 352, 288
    st, keep = stab_for(rng, w, h, 1, 0, flags=1)
    for kind in ("empty", "single_far", "dense"):
        co = np.zeros(w * h, dtype=np.int32)
        if kind == "single_far":
            co[w * h - 1] = -777
            co[0] = 1234
        elif kind == "dense":
            co[:] = rng.integers(200, 5000, size=w * h) * rng.choice([-1, 1], size=w * h)
        a, b = co.copy(), co.copy()
        bufa = np.zeros(w * h * 8 + 64, dtype=np.uint8)
        bufb = np.zeros_like(bufa)
        bsa, bsb = A.BS(A.u8p(bufa), 0), A.BS(A.u8p(bufb), 0)
        orc.orc_encode_plane(C.byref(bsa), C.byref(A.Coefs(A.i32p(a), w, h)), 200, C.byref(st))
        rc = prod.dsvg_op_encode_plane(C.byref(bsb), C.byref(A.Coefs(A.i32p(b), w, h)), 200, C.byref(st))
        A.chk(prod, rc)
        assert bsa.pos == bsb.pos, kind
        A.assert_same("packed plane " + kind, bufb[: bsa.pos // 8], bufa[: bsa.pos // 8])
        A.assert_same("coefs " + kind, b, a, (h, w))


def rnd_mvs(rng, nbh, nbv, span):
    mv = np.zeros(nbh * nbv, dtype=A.MV_DTYPE)
    mv["x"] = rng.integers(-span, span + 1, size=nbh * nbv)
    mv["y"] = rng.integers(-span, span + 1, size=nbh * nbv)
    intra = rng.random(nbh * nbv) < 0.3
    mv["mode"] = intra
    mv["submask"] = np.where(intra, rng.integers(1, 16, size=nbh * nbv), 0)
    return mv


@pytest.mark.parametrize("w,h,fmt", [(352, 288, A.SUBSAMP_420), (176, 144, A.SUBSAMP_444), (360, 200, A.SUBSAMP_422),
                                      (352, 288, A.SUBSAMP_411), (1920, 1080, A.SUBSAMP_420)])
def test_sub_add_pred(prod, orc, w, h, fmt):
    rng = np.random.default_rng(w * 3 + h + fmt)
    bw, bh, nbh, nbv = A.block_dims(w, h)
    meta = A.Meta(w, h, fmt, 30, 1, 1, 1)
    prm = A.Params(C.pointer(meta), 1, 1, bw, bh, nbh, nbv)
    reff = mk_frame(rng, w, h, fmt, orc)
    for span in (3, 40, 400):
        mv = rnd_mvs(rng, nbh, nbv, span)
        mvp = mv.ctypes.data_as(C.POINTER(A.MV))
        inp = mk_frame(rng, w, h, fmt, orc)
        ia, ib = A.BorderedFrame(w, h, fmt), A.BorderedFrame(w, h, fmt)
        ia.buf[:] = inp.buf
        ib.buf[:] = inp.buf
        da, db = A.BorderedFrame(w, h, fmt), A.BorderedFrame(w, h, fmt)
        orc.orc_sub_pred(mvp, C.byref(prm), da.ptr(), ia.ptr(), reff.ptr())
        A.chk(prod, prod.dsvg_op_sub_pred(mvp, C.byref(prm), db.ptr(), ib.ptr(), reff.ptr()))
        for c in range(3):
            A.assert_same("prediction plane %d span %d" % (c, span), db.plane(c), da.plane(c), db.plane(c).shape)
            A.assert_same("residual plane %d span %d" % (c, span), ib.plane(c), ia.plane(c), ib.plane(c).shape)
        A.assert_same("prediction frame bytes", db.raw(), da.raw())
        A.assert_same("residual frame bytes", ib.raw(), ia.raw())
        oa, ob = A.BorderedFrame(w, h, fmt), A.BorderedFrame(w, h, fmt)
        orc.orc_add_pred(mvp, C.byref(prm), ia.ptr(), oa.ptr(), reff.ptr())
        A.chk(prod, prod.dsvg_op_add_pred(mvp, C.byref(prm), ib.ptr(), ob.ptr(), reff.ptr()))
        A.assert_same("add_pred", ob.raw(), oa.raw())
        orc.orc_frame_add(ia.ptr(), da.ptr())
        A.chk(prod, prod.dsvg_op_frame_add(ib.ptr(), db.ptr()))
        A.assert_same("frame_add", ib.raw(), ia.raw())


def test_frame_ops(prod, orc):
    rng = np.random.default_rng(5)
    for (w, h, fmt) in [(352, 288, A.SUBSAMP_420), (101, 77, A.SUBSAMP_420), (64, 48, A.SUBSAMP_444), (1920, 1080, A.SUBSAMP_420)]:
        fa = A.BorderedFrame(w, h, fmt)
        for i in range(3):
            fa.plane(i)[:, :] = rnd_plane(rng, *fa.dims[i])
        fb = A.BorderedFrame(w, h, fmt)
        fb.buf[:] = fa.buf
        orc.orc_frame_extend(fa.ptr())
        A.chk(prod, prod.dsvg_op_extend_frame(fb.ptr()))
        A.assert_same("extend_frame", fb.raw(), fa.raw())
        avg = C.c_int(0)
        A.chk(prod, prod.dsvg_op_frame_avg_luma(fb.ptr(), C.byref(avg)))
        assert avg.value == orc.orc_frame_avg_luma(fa.ptr())
        w2, h2 = A.rshift_up(w, 1), A.rshift_up(h, 1)
        da, db = A.BorderedFrame(w2, h2, fmt), A.BorderedFrame(w2, h2, fmt)
        orc.orc_frame_ds2x_luma(da.ptr(), fa.ptr())
        orc.orc_frame_extend_luma(da.ptr())
        A.chk(prod, prod.dsvg_op_ds2x_frame_luma(db.ptr(), fb.ptr()))
        A.chk(prod, prod.dsvg_op_extend_frame_luma(db.ptr()))
        A.assert_same("ds2x+extend_luma", db.raw(), da.raw())


def build_pyramid(orc, f0, levels):
    out = [f0]
    for i in range(levels):
        f = A.BorderedFrame(A.rshift_up(f0.w, i + 1), A.rshift_up(f0.h, i + 1), f0.fmt)
        orc.orc_frame_ds2x_luma(f.ptr(), out[-1].ptr())
        orc.orc_frame_extend_luma(f.ptr())
        out.append(f)
    return out


@pytest.mark.parametrize("w,h,style,levels", [(352, 288, 0, 3), (352, 288, 1, 3), (352, 288, 2, 4), (704, 480, 2, 3),
                                               (1920, 1080, 2, 4), (1920, 1080, 0, 4)])
def test_hme(prod, orc, w, h, style, levels):
    fmt = A.SUBSAMP_420
    clip = A.gen_clip(w, h, fmt, 0xC1F001 + style, 3, style=style)
    bw, bh, nbh, nbv = A.block_dims(w, h)
    meta = A.Meta(w, h, fmt, 30, 1, 1, 1)
    prm = A.Params(C.pointer(meta), 1, 1, bw, bh, nbh, nbv)
    frames = []
    for t in range(3):
        f = A.BorderedFrame(w, h, fmt)
        f.load_planar(clip[t])
        orc.orc_frame_extend(f.ptr())
        frames.append(build_pyramid(orc, f, levels))
    for t in (1, 2):
        ha, hb = A.HME(), A.HME()
        for hm in (ha, hb):
            hm.params = C.pointer(prm)
            hm.levels = levels
            for l in range(levels + 1):
                hm.src[l] = C.pointer(frames[t][l].c)
                hm.ref[l] = C.pointer(frames[t - 1][l].c)
        pa = orc.orc_hme_run(C.byref(ha))
        pb = C.c_int(-1)
        A.chk(prod, prod.dsvg_op_hme(C.byref(hb), C.byref(pb)))
        for l in range(levels, -1, -1):
            a = np.ctypeslib.as_array(C.cast(ha.mvf[l], C.POINTER(C.c_uint8)), shape=(nbh * nbv * 12,)).copy().view(A.MV_DTYPE)
            b = np.ctypeslib.as_array(C.cast(hb.mvf[l], C.POINTER(C.c_uint8)), shape=(nbh * nbv * 12,)).copy().view(A.MV_DTYPE)
            for k in ("x", "y", "mode", "submask", "lo_var", "lo_tex", "high_detail"):
                A.assert_same("hme level %d frame %d field %s" % (l, t, k), b[k], a[k], (nbv, nbh))
        assert pa == pb.value
        for l in range(levels + 1):
            C.CDLL(None).free(ha.mvf[l])
            prod.dsv_free(C.cast(hb.mvf[l], C.c_void_p))       # the product's fields come from its dsv_alloc, as the reference's do (dsv_encoder.c:239-244)
