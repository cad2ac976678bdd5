"""Definitions of the golden-vector cases shared by tools/make_goldens.py (which runs them on the REAL
reference, in the build container) and the tests (which run them on the oracle and on the HIP path and
compare with the committed hashes in tests/golden/)."""
import ctypes as C
import hashlib

import numpy as np

import _cabi as A

# name -> (w, h, fmt, frames, style, seed, reference CLI flags, kwargs for orc_cfg / make_encoder_cfg)
STREAM_CASES = {
    # BASELINE config 1: CIF intra-only with the CLI defaults (ABR!) and with CRF
    "cfg1_cif_intra_abr": (352, 288, A.SUBSAMP_420, 8, 0, 0x00C1F001, ["-gop0", "-qp85"], dict(qp=85, gop=0, rc_mode_cli=0)),
    "cfg1_cif_intra_crf": (352, 288, A.SUBSAMP_420, 8, 0, 0x00C1F001, ["-gop0", "-qp85", "-rc_mode1"], dict(qp=85, gop=0, rc_mode_cli=1)),
    "cif_gop12_style2": (352, 288, A.SUBSAMP_420, 6, 2, 0x00C1F002, ["-gop12", "-qp85", "-rc_mode1"], dict(qp=85, gop=12, rc_mode_cli=1)),
    "cif_gop12_style1_abr": (352, 288, A.SUBSAMP_420, 14, 1, 0x00C1F003, ["-gop12", "-qp60"], dict(qp=60, gop=12, rc_mode_cli=0)),
    "cif_forced_intra": (352, 288, A.SUBSAMP_420, 6, 1, 0x00C1F004, ["-gop12", "-qp85", "-rc_mode1", "-ipct20"], dict(qp=85, gop=12, rc_mode_cli=1, ipct=20)),
    "qvga_444": (320, 240, A.SUBSAMP_444, 6, 2, 0x00C1F005, ["-gop12", "-qp95", "-rc_mode1"], dict(qp=95, gop=12, rc_mode_cli=1)),
    "qvga_422_abr": (320, 240, A.SUBSAMP_422, 6, 1, 0x00C1F006, ["-gop12", "-qp85", "-kbps800"], dict(qp=85, gop=12, rc_mode_cli=0, kbps=800)),
    "cif_411": (352, 288, A.SUBSAMP_411, 4, 0, 0x00C1F007, ["-gop12", "-qp85", "-rc_mode1"], dict(qp=85, gop=12, rc_mode_cli=1)),
    "lowq_chroma_cap": (352, 288, A.SUBSAMP_420, 5, 2, 0x00C1F008, ["-gop12", "-qp10", "-rc_mode1"], dict(qp=10, gop=12, rc_mode_cli=1)),
    # every partial intra submask (clip style 6: static noise, cells that change in chosen quadrants)
    "cif_submasks_style6": (352, 288, A.SUBSAMP_420, 5, 6, 0x00C1F009, ["-gop12", "-qp85", "-rc_mode1"], dict(qp=85, gop=12, rc_mode_cli=1)),
    # BASELINE config 2 / 3 shapes, short
    "cfg2_1080p_intra": (1920, 1080, A.SUBSAMP_420, 2, 1, 0x10800001, ["-gop0", "-qp85", "-rc_mode1"], dict(qp=85, gop=0, rc_mode_cli=1)),
    "cfg3_1080p_gop12": (1920, 1080, A.SUBSAMP_420, 13, 0, 0x10800003, ["-gop12", "-qp85", "-rc_mode1"], dict(qp=85, gop=12, rc_mode_cli=1)),
    "cfg3_1080p_gop12_style2": (1920, 1080, A.SUBSAMP_420, 4, 2, 0x10800004, ["-gop12", "-qp85", "-rc_mode1"], dict(qp=85, gop=12, rc_mode_cli=1)),
    # BASELINE config 4 / 5 shapes (3840x2160), short
    "cfg4_4k_gop12": (3840, 2160, A.SUBSAMP_420, 7, 0, 0x21600004, ["-gop12", "-qp85", "-rc_mode1", "-scd0"], dict(qp=85, gop=12, rc_mode_cli=1, scd=0)),
    "cfg5_4k_444_abr": (3840, 2160, A.SUBSAMP_444, 7, 0, 0x21600005, ["-gop30", "-qp85", "-kbps20000"], dict(qp=85, gop=30, rc_mode_cli=0, kbps=20000)),
}

# Full-length streams of the BASELINE configs, hashes only (tests/golden/long_streams.json), run under -m gpu only: the CPU
# oracle would take minutes on them.  Same tuple layout as STREAM_CASES.
LONG_STREAM_CASES = {
    # config 5 complete: 60 frames = 2 GOPs of 30, 4K 4:4:4, ABR feedback from every packet into the next quantiser, the CLI's
    # stable_refresh of 14 against a GOP of 30
    "cfg5_4k_444_abr_60": (3840, 2160, A.SUBSAMP_444, 60, 0, 0x21600005, ["-gop30", "-qp85", "-kbps20000"], dict(qp=85, gop=30, rc_mode_cli=0, kbps=20000)),
    # config 4: two closed 4K GOPs (also coded GOP-sharded and in chain mode: all three must give these bytes)
    "cfg4_4k_gop12_24": (3840, 2160, A.SUBSAMP_420, 24, 0, 0x21600004, ["-gop12", "-qp85", "-rc_mode1", "-scd0"], dict(qp=85, gop=12, rc_mode_cli=1, scd=0)),
    # one 1080p stream with scene cuts, CLI defaults apart from CRF: plain GOP sharding is NOT exact here, chain mode is
    "1080p_gop12_scenecuts_36": (1920, 1080, A.SUBSAMP_420, 36, 5, 0x10800333, ["-gop12", "-qp85", "-rc_mode1"], dict(qp=85, gop=12, rc_mode_cli=1)),
    # GOP 30 CRF: the stability accumulators (refresh every 14 P pictures) cross GOP boundaries
    "1080p_gop30_crf_45": (1920, 1080, A.SUBSAMP_420, 45, 5, 0x10800030, ["-gop30", "-qp85", "-rc_mode1"], dict(qp=85, gop=30, rc_mode_cli=1)),
}

# operator-level known answers: name -> dict describing a seeded input
OP_CASES = {
    "sbt_352x288_P": dict(op="sbt", w=352, h=288, isP=1, seed=11),
    "sbt_352x288_I": dict(op="sbt", w=352, h=288, isP=0, seed=12),
    "sbt_250x130_P": dict(op="sbt", w=250, h=130, isP=1, seed=13),
    "sbt_960x540_I": dict(op="sbt", w=960, h=540, isP=0, seed=14),
    "sbt_1920x1080_P": dict(op="sbt", w=1920, h=1080, isP=1, seed=15),
    "sbt_3840x2160_I": dict(op="sbt", w=3840, h=2160, isP=0, seed=16),
    "hzcc_960x540_overlap": dict(op="hzcc", w=960, h=540, isP=1, q=313, cur_plane=1, seed=21),
    "hzcc_250x130_overlap2": dict(op="hzcc", w=250, h=130, isP=0, q=100, cur_plane=0, seed=22),
    "hzcc_352x288_minq": dict(op="hzcc", w=352, h=288, isP=1, q=16, cur_plane=0, seed=23),
    "bmc_352x288_420": dict(op="bmc", w=352, h=288, fmt=A.SUBSAMP_420, span=40, seed=31),
    "bmc_360x200_422_far": dict(op="bmc", w=360, h=200, fmt=A.SUBSAMP_422, span=400, seed=32),
    "hme_352x288_style2": dict(op="hme", w=352, h=288, style=2, levels=3, seed=0xC1F041),
    "hme_704x480_style1": dict(op="hme", w=704, h=480, style=1, levels=3, seed=0xC1F042),
    # 3840x2160: 64x64 blocks (the widest rows-per-lane path of the search, 32x32 chroma blocks)
    "bmc_3840x2160_420": dict(op="bmc", w=3840, h=2160, fmt=A.SUBSAMP_420, span=90, seed=33),
    "hme_3840x2160_style2": dict(op="hme", w=3840, h=2160, style=2, levels=5, seed=0xC1F043),
}


# ---- the seeded geometry / parameter sweep of tests/test_gpu_fuzz.py (defined here so that tools/make_goldens.py can
# probe, in the build container, which of its inputs the REFERENCE itself cannot encode) --------------------------------
def fuzz_cases():
    rng = np.random.default_rng(0xD5F1)
    fmts = [A.SUBSAMP_420, A.SUBSAMP_420, A.SUBSAMP_444, A.SUBSAMP_422, A.SUBSAMP_411]
    out = []
    for i in range(72):
        big = i % 6 == 5                               # a few frames beyond every block-size threshold (352/704/1024/1280)
        w = int(rng.integers(16, 700 if big else 215)) * 2
        h = int(rng.integers(16, 400 if big else 150)) * 2
        fmt = fmts[int(rng.integers(0, len(fmts)))]
        if fmt == A.SUBSAMP_411:
            w = (w + 3) & ~3
        kw = dict(qp=int(rng.integers(15, 100)), gop=[0, 3, 12, 12][int(rng.integers(0, 4))], rc_mode_cli=int(rng.integers(0, 4) != 0))
        if rng.integers(0, 3) == 0:
            kw["scd"] = 0
        out.append((w, h, fmt, 3 if big else int(rng.integers(3, 6)), int(rng.integers(0, 3)), kw, 0xF00D00 + i))
    return out + [
        # the smallest frames the reference accepts: chroma planes with only 3 / 4 / 5 transform levels
        (32, 32, A.SUBSAMP_420, 4, 2, dict(qp=80, gop=12, rc_mode_cli=1), 0xF00E01),
        (32, 32, A.SUBSAMP_411, 4, 1, dict(qp=90, gop=0, rc_mode_cli=1), 0xF00E02),
        (40, 32, A.SUBSAMP_411, 4, 0, dict(qp=70, gop=3, rc_mode_cli=1), 0xF00E03),
        (32, 64, A.SUBSAMP_420, 4, 2, dict(qp=85, gop=12, rc_mode_cli=0), 0xF00E04),
    ]


def fuzz_id(c):
    return "%dx%d_f%x_%s" % (c[0], c[1], c[2], "_".join("%s%s" % kv for kv in sorted(c[5].items())))


EXTREME_KINDS = ["noise01", "checker_flip", "stripes"]
EXTREME_QPS = [99, 85, 60]
EXTREME_GEOM = (352, 288, A.SUBSAMP_420, 5)
EXTREME_KW = lambda qp: dict(qp=qp, gop=12, rc_mode_cli=1, scd=0, ipct=101)       # keep every inter candidate a P picture


def extreme_clip(kind, qp):
    """residuals at the edge of the 8-bit range: binary noise, flipping checkerboards, moving stripes"""
    w, h, fmt, n = EXTREME_GEOM
    rng = np.random.default_rng(7 + qp)
    fb = A.frame_bytes(w, h, fmt)
    clip = np.empty((n, fb), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for t in range(n):
        if kind == "noise01":
            y = (rng.integers(0, 2, size=(h, w)) * 255).astype(np.uint8)
        elif kind == "checker_flip":
            y = ((((xx >> (t % 3)) + (yy >> (t % 2)) + t) & 1) * 255).astype(np.uint8)
        else:
            y = ((((xx + 3 * t) // (1 + t)) & 1) * 255).astype(np.uint8)
        c = (rng.integers(0, 2, size=(fb - w * h)) * 255).astype(np.uint8)
        clip[t, : w * h] = y.reshape(-1)
        clip[t, w * h:] = c
    return clip


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _plane(rng, w, h):
    base = rng.integers(0, 256, size=(h // 8 + 2, w // 8 + 2)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8)))[:h, :w] + rng.integers(-6, 7, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def _frame(rng, w, h, fmt, extend):
    f = A.BorderedFrame(w, h, fmt)
    for i in range(3):
        f.plane(i)[:, :] = _plane(rng, *f.dims[i])
    extend(f.ptr())
    return f


def run_op_case(case, impl, L, orc=None):
    """run one operator case with implementation `impl` in {"ref","orc","prod"}; returns {name: sha256}.
    Inputs are built with numpy only (identical everywhere); border extension uses the oracle/ref."""
    fn = {
        "ref": dict(fwd="dsv_fwd_sbt", inv="dsv_inv_sbt", enc="dsv_encode_plane", sub="dsv_sub_pred", hme="dsv_hme",
                    ext="dsv_extend_frame", ds="dsv_ds2x_frame_luma", extl="dsv_extend_frame_luma"),
        "orc": dict(fwd="orc_fwd_sbt", inv="orc_inv_sbt", enc="orc_encode_plane", sub="orc_sub_pred", hme="orc_hme_run",
                    ext="orc_frame_extend", ds="orc_frame_ds2x_luma", extl="orc_frame_extend_luma"),
        "prod": dict(fwd="dsvg_op_fwd_sbt", inv="dsvg_op_inv_sbt", enc="dsvg_op_encode_plane", sub="dsvg_op_sub_pred",
                     hme="dsvg_op_hme"),
    }[impl]
    helper = L if impl != "prod" else orc          # input preparation (extend / pyramid) never uses the product
    hfn = fn if impl != "prod" else {"ext": "orc_frame_extend", "ds": "orc_frame_ds2x_luma", "extl": "orc_frame_extend_luma"}
    rng = np.random.default_rng(case["seed"])
    out = {}
    if case["op"] == "sbt":
        w, h, isP = case["w"], case["h"], case["isP"]
        f = _frame(rng, w, h, A.SUBSAMP_444, getattr(helper, hfn["ext"]))
        co = np.zeros(w * h, dtype=np.int32)
        getattr(L, fn["fwd"])(C.byref(f.c.planes[0]), C.byref(A.Coefs(A.i32p(co), w, h)), isP)
        out["coefs"] = _sha(co)
        q = 313
        co2 = co.copy()
        co2[1:] = (co2[1:] // 24) * 24
        for c in (0, 1):
            o = A.BorderedFrame(w, h, A.SUBSAMP_444)
            tmp = co2.copy()
            getattr(L, fn["inv"])(C.byref(o.c.planes[c]), C.byref(A.Coefs(A.i32p(tmp), w, h)), q, isP, c)
            out["inv_c%d" % c] = _sha(o.plane(c))
    elif case["op"] == "hzcc":
        w, h, isP, q, cp = case["w"], case["h"], case["isP"], case["q"], case["cur_plane"]
        fw, fh = (w * 2, h * 2) if cp else (w, h)
        bw, bh, nbh, nbv = A.block_dims(fw, fh)
        meta = A.Meta(fw, fh, A.SUBSAMP_420, 30, 1, 1, 1)
        prm = A.Params(C.pointer(meta), 1, isP, bw, bh, nbh, nbv)
        sb = rng.integers(0, 4, size=nbh * nbv).astype(np.uint8)
        st = A.Stability(C.pointer(prm), A.u8p(sb), cp, isP)
        co = rng.laplace(0, 40, size=(h, w)).astype(np.int32)
        co[: h // 8, : w // 8] *= 16
        co = co.reshape(-1).copy()
        buf = np.zeros(w * h * 8 + 64, dtype=np.uint8)
        bs = A.BS(A.u8p(buf), 0)
        getattr(L, fn["enc"])(C.byref(bs), C.byref(A.Coefs(A.i32p(co), w, h)), q, C.byref(st))
        out["bits"] = _sha(buf[: bs.pos // 8])
        out["dequant"] = _sha(co)
    elif case["op"] == "bmc":
        w, h, fmt, span = case["w"], case["h"], case["fmt"], case["span"]
        bw, bh, nbh, nbv = A.block_dims(w, h)
        meta = A.Meta(w, h, fmt, 30, 1, 1, 1)
        prm = A.Params(C.pointer(meta), 1, 1, bw, bh, nbh, nbv)
        ext = getattr(helper, hfn["ext"])
        reff = _frame(rng, w, h, fmt, ext)
        inp = _frame(rng, w, h, fmt, ext)
        mv = np.zeros(nbh * nbv, dtype=A.MV_DTYPE)
        mv["x"] = rng.integers(-span, span + 1, size=nbh * nbv)
        mv["y"] = rng.integers(-span, span + 1, size=nbh * nbv)
        intra = rng.random(nbh * nbv) < 0.3
        mv["mode"] = intra
        mv["submask"] = np.where(intra, rng.integers(1, 16, size=nbh * nbv), 0)
        dif = A.BorderedFrame(w, h, fmt)
        getattr(L, fn["sub"])(mv.ctypes.data_as(C.POINTER(A.MV)), C.byref(prm), dif.ptr(), inp.ptr(), reff.ptr())
        out["pred"] = _sha(dif.raw())
        out["resid"] = _sha(inp.raw())
    elif case["op"] == "hme":
        w, h, levels = case["w"], case["h"], case["levels"]
        fmt = A.SUBSAMP_420
        clip = A.gen_clip(w, h, fmt, case["seed"], 2, style=case["style"])
        bw, bh, nbh, nbv = A.block_dims(w, h)
        meta = A.Meta(w, h, fmt, 30, 1, 1, 1)
        prm = A.Params(C.pointer(meta), 1, 1, bw, bh, nbh, nbv)
        pyr = []
        for t in range(2):
            f = A.BorderedFrame(w, h, fmt)
            f.load_planar(clip[t])
            getattr(helper, hfn["ext"])(f.ptr())
            lv = [f]
            for i in range(levels):
                g = A.BorderedFrame(A.rshift_up(w, i + 1), A.rshift_up(h, i + 1), fmt)
                getattr(helper, hfn["ds"])(g.ptr(), lv[-1].ptr())
                getattr(helper, hfn["extl"])(g.ptr())
                lv.append(g)
            pyr.append(lv)
        hm = A.HME()
        hm.params = C.pointer(prm)
        hm.levels = levels
        for l in range(levels + 1):
            hm.src[l] = C.pointer(pyr[1][l].c)
            hm.ref[l] = C.pointer(pyr[0][l].c)
        if impl == "prod":
            pct = C.c_int(0)
            rc = L.dsvg_op_hme(C.byref(hm), C.byref(pct))
            assert rc == 0, L.dsvg_last_error()
        else:
            getattr(L, fn["hme"])(C.byref(hm))
        for l in range(levels + 1):
            a = np.ctypeslib.as_array(C.cast(hm.mvf[l], C.POINTER(C.c_uint8)), shape=(nbh * nbv * 12,)).copy().view(A.MV_DTYPE)
            fields = np.stack([a[k].astype(np.int32) for k in ("x", "y", "mode", "submask", "lo_var", "lo_tex", "high_detail")])
            out["mv_level%d" % l] = _sha(fields)
            if impl == "prod":
                L.dsv_free(C.cast(hm.mvf[l], C.c_void_p))
    return out
