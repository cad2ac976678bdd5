"""Seeded random geometry / parameter sweep of the whole GPU path against the oracle: ragged plane sizes (HZCC scan
regions that overlap, partial edge blocks, odd chroma widths), every chroma format, intra-only / short / long GOPs, CRF
and ABR.  Encode must be byte-identical to the oracle's stream; the GPU decoder must give the oracle's frames."""
import importlib
import os

import numpy as np
import pytest

import _cabi as A
from test_gpu_stream import _decode_and_compare, explain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


def _cases():
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
    return out


def _skip_if_reference_dies(clip, w, h, fmt, kw):
    """The reference itself dies on a few inputs (SIGFPE: a 1-pixel-wide chroma edge block has a 0x0 quadrant,
    bmc.c:176-189; heap overflow of the picture buffer when binary noise is coded at top quality, bs.c:53) and the
    oracle restates that faithfully: probe it in a forked child and skip what the reference cannot encode -- there is
    no answer to be bit-exact with."""
    pid = os.fork()
    if pid == 0:
        try:
            A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
        finally:
            os._exit(0)
    _, status = os.waitpid(pid, 0)
    if os.WIFSIGNALED(status):
        pytest.skip("the reference crashes on this input (signal %d)" % os.WTERMSIG(status))


CASES = _cases() + [
    # the smallest frames the reference accepts: chroma planes with only 3 / 4 / 5 transform levels
    (32, 32, A.SUBSAMP_420, 4, 2, dict(qp=80, gop=12, rc_mode_cli=1), 0xF00E01),
    (32, 32, A.SUBSAMP_411, 4, 1, dict(qp=90, gop=0, rc_mode_cli=1), 0xF00E02),
    (40, 32, A.SUBSAMP_411, 4, 0, dict(qp=70, gop=3, rc_mode_cli=1), 0xF00E03),
    (32, 64, A.SUBSAMP_420, 4, 2, dict(qp=85, gop=12, rc_mode_cli=0), 0xF00E04),
]


@pytest.mark.parametrize("case", range(len(CASES)), ids=["%dx%d_f%x_%s" % (c[0], c[1], c[2], "_".join("%s%s" % kv for kv in sorted(c[5].items()))) for c in CASES])
def test_fuzz_encode_decode(pkg, orc, case):
    w, h, fmt, n, style, kw, seed = CASES[case]
    clip = A.gen_clip(w, h, fmt, seed, n, style=style)
    _skip_if_reference_dies(clip, w, h, fmt, kw)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    got = pkg.encode_clip(clip, w, h, fmt, **kw)
    assert got == want, explain(got, want)
    _decode_and_compare(pkg, w, h, fmt, n, style, kw, seed, check_recon=False)


@pytest.mark.parametrize("kind", ["noise01", "checker_flip", "stripes"])
@pytest.mark.parametrize("qp", [99, 85, 60])
def test_extreme_content(pkg, orc, kind, qp):
    """residuals at the edge of the 8-bit range (the packed int16 level-1 inverse and the int16 symbol planes must
    hold the largest coefficients a real input can produce)"""
    w, h, fmt, n = 352, 288, A.SUBSAMP_420, 5
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
    kw = dict(qp=qp, gop=12, rc_mode_cli=1, scd=0, ipct=101)       # keep every inter candidate a P picture
    _skip_if_reference_dies(clip, w, h, fmt, kw)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    got = pkg.encode_clip(clip, w, h, fmt, **kw)
    assert got == want, explain(got, want)
