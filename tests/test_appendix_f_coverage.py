"""the parity clips must be CHECKED to reach the decision points they are meant to reach.

The oracle counts the branches of the level-0 motion search and of the encoder's per-picture decisions while it encodes
(oracle/orc.h ORC_COV_*, hme.c:544-721, dsv_encoder.c:236-252,330-408,538-554).  The clips below are the ones the GPU parity
tests run (tests/golden_cases.py STREAM_CASES up to 704 pixels wide, and the style clips of test_gpu_recon / test_gpu_chain /
tools/soak.py at the sizes used there); every bin must be non-empty over the set, and the bins a clip style exists for must be
hit by that style.  CPU only: the oracle is the checker, nothing of the product runs here."""
import ctypes as C

import numpy as np
import pytest

import _cabi as A
import golden_cases as G

NAMES = (["HP_SKIPPED", "HP_KEPT_FULLPEL", "HP_REFINED",
          "NB0_PLAIN", "NB1_PLAIN", "NB2_PLAIN", "NB3_PLAIN", "NB0_HD", "NB1_HD", "NB2_HD", "NB3_HD",
          "INTRA_ZEROVAR", "INTRA_REFVAR", "INTRA_FLATSRC", "INTRA_AVG", "INTRA_BADSAD", "INTRA_CHROMA", "INTRA_NONE",
          "VETO_TAKEN", "VETO_NOT_TAKEN", "LOWTEX_ALL_INTRA", "QUAD_VOTE"]
         + ["SUBMASK%d" % i for i in range(16)]
         + ["LO_TEX", "LO_VAR", "LO_NEITHER", "FORCED_INTRA_IPCT", "FORCED_INTRA_SCENE", "P_KEPT", "STAB_REFRESH", "STAB_RESET_LO",
            "STABLE_BY_HD", "STABLE_BY_AVG", "UNSTABLE_INTER", "INTRA_BLOCK_FLAG", "STABLE_I", "UNSTABLE_I"])


def _cov(L):
    v = (C.c_ulonglong * len(NAMES))()
    n = L.orc_cov_read(v, len(NAMES))
    assert n == len(NAMES), "ORC_COV_N changed: update NAMES (%d != %d)" % (n, len(NAMES))
    return {NAMES[i]: int(v[i]) for i in range(len(NAMES))}


def _encode(L, w, h, fmt, frames, style, seed, **kw):
    """oracle encode with the counters cleared first; returns (counters, submasks of the intra blocks of all P pictures)"""
    L.orc_cov_reset()
    clip = A.gen_clip(w, h, fmt, seed, frames, style=style)
    cfg = A.orc_cfg(w, h, fmt, **kw)
    e = L.orc_enc_open(C.byref(cfg))
    out, n, cap = C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
    masks = set()
    for t in range(frames):
        L.orc_enc_frame(e, clip[t].ctypes.data, C.byref(out), C.byref(n), C.byref(cap), None)
        cnt = C.c_int(0)
        p = L.orc_enc_last_mvs(e, C.byref(cnt))
        if t and p and cnt.value:
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cnt.value * 12,)).reshape(cnt.value, 12)
            masks |= set(int(m) for m in a[a[:, 4] != 0][:, 5])          # DSV_MV: mode at byte 4, submask at byte 5 (dsv.h:137-150)
    C.CDLL(None).free(out)
    L.orc_enc_close(e)
    return _cov(L), masks


@pytest.fixture(scope="module")
def orc():
    L = A.load_orc()
    L.orc_cov_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    L.orc_enc_last_mvs.restype = C.c_void_p
    return L


CLIPS = [(k, v) for k, v in G.STREAM_CASES.items() if v[0] <= 704] + [
    ("style1_704x480", (704, 480, A.SUBSAMP_420, 6, 1, 0xABC1, [], dict(qp=85, gop=12, rc_mode_cli=1))),
    ("style2_704x480", (704, 480, A.SUBSAMP_420, 6, 2, 0xABC2, [], dict(qp=85, gop=12, rc_mode_cli=1))),
    ("style4_704x480", (704, 480, A.SUBSAMP_420, 5, 4, 0xABC4, [], dict(qp=70, gop=12, rc_mode_cli=1))),
    ("style3_cif_16", (352, 288, A.SUBSAMP_420, 16, 3, 0xABC3, [], dict(qp=85, gop=12, rc_mode_cli=1))),
    ("style5_cif_16", (352, 288, A.SUBSAMP_420, 16, 5, 0xABC5, [], dict(qp=85, gop=12, rc_mode_cli=1))),
]


def test_every_appendix_f_bin_is_hit_by_the_parity_clips(orc):
    total = {k: 0 for k in NAMES}
    per = {}
    for name, (w, h, fmt, frames, style, seed, _, kw) in CLIPS:
        c, _ = _encode(orc, w, h, fmt, frames, style, seed, **kw)
        per[name] = c
        for k in NAMES:
            total[k] += c[k]
    empty = [k for k in NAMES if total[k] == 0]
    assert not empty, "no parity clip reaches: %s" % empty
    # the bins each style exists for (so that a change of the generator that loses them is noticed where it happens)
    assert per["cif_forced_intra"]["FORCED_INTRA_IPCT"] >= 1                       # -ipct20 on style 1: dsv_encoder.c:248-252
    assert per["style3_cif_16"]["FORCED_INTRA_SCENE"] >= 1 and per["style3_cif_16"]["STAB_REFRESH"] >= 1      # scene cuts, :546-551; 12 P pictures: refresh, :345-348
    assert per["style5_cif_16"]["UNSTABLE_INTER"] > 100 and per["style5_cif_16"]["UNSTABLE_I"] > 100          # fast pan: accumulators leave zero
    s1 = per["cif_gop12_style1_abr"]
    assert s1["VETO_TAKEN"] > 0 and s1["VETO_NOT_TAKEN"] > 0 and s1["LOWTEX_ALL_INTRA"] > 0 and s1["QUAD_VOTE"] > 0   # hme.c:685-692
    assert s1["LO_TEX"] > 0 and s1["LO_VAR"] > 0 and s1["STAB_RESET_LO"] > 0                                   # flat objects: :388-391
    assert min(s1["NB%d_%s" % (n, k)] for n in range(4) for k in ("PLAIN", "HD")) > 0                           # hme.c:621-648, all eight combinations in ONE clip
    assert per["style2_704x480"]["HP_SKIPPED"] > 0 and per["style2_704x480"]["HP_KEPT_FULLPEL"] > 0 and per["style2_704x480"]["HP_REFINED"] > 0   # hme.c:551-591
    causes = ["INTRA_ZEROVAR", "INTRA_REFVAR", "INTRA_FLATSRC", "INTRA_AVG", "INTRA_BADSAD", "INTRA_CHROMA"]
    assert all(total[k] > 0 for k in causes)                                       # each of the six tests of hme.c:652-682 fires first somewhere


def test_style6_reaches_every_partial_submask(orc):
    """hme.c:689-716: the 4-bit submask of an intra block -- every value, and the vote that clears all four bits (the block
    stays inter), in the committed golden clip cif_submasks_style6; the pictures stay P pictures (a third of the blocks intra)"""
    w, h, fmt, frames, style, seed, _, kw = G.STREAM_CASES["cif_submasks_style6"]
    c, masks = _encode(orc, w, h, fmt, frames, style, seed, **kw)
    assert all(c["SUBMASK%d" % m] > 0 for m in range(16)), {m: c["SUBMASK%d" % m] for m in range(16)}
    assert masks == set(range(1, 16))                                             # ... and they arrive in the motion fields the pictures are coded with
    assert c["P_KEPT"] == frames - 1 and c["FORCED_INTRA_IPCT"] == 0 and c["FORCED_INTRA_SCENE"] == 0
