"""CPU: the range bound behind the int16 symbol planes -- checked, not asserted in a comment.

The encoder stores the quantised symbols of transform levels 1-3 as int16 and runs level 1 of the P-picture luma inverse
on int16 pairs (k_sbt.hip: pk_nudge / inv_l1_item).  Worst cases, from the filter gains:
  * samples enter as x - 128 in [-128, 127] (p2sbc sbt.c:576-592; a P picture's residual is clamp(src - pred + 128) first,
    bmc.c:43-55);
  * Haar level (sbt.c:268-349): detail = x0 -+ x1 +- x2 -+ x3 (gain 4), LL = sum (gain 4), times 4/5 truncating where
    LVL_TEST holds (I pictures: every level; P pictures: levels > 1, sbt.c:22);
  * I pictures, level 1 = B4T (sbt.c:91-126,166-201): L = round2(3(x0 + x1) - x-1 - x2), H = round2(x-1 - 3x0 + 3x1 - x2),
    rows then columns: gain 8 / 2 = 4 per pass;
  * quantisers (hzcc.c:64-135): levels 2, 3 divide by 2q with q >= MINQUANT = 16; level 1 shifts.
Each bound is computed here from those definitions, checked against the ORACLE transform on the sign patterns that attain it
(so the bound is tight, not just safe), and the chain of the packed inverse is followed to its widest intermediate."""
import ctypes as C

import numpy as np
import pytest

import _cabi as A

MINQUANT = 16
LO, HI = -128, 127


def haar_bounds(is_p):
    """max |detail| and max |LL| per transform level 1..3 for inputs in [LO, HI]"""
    det, ll = {}, {}
    amp = max(-LO, HI)                                   # |x| <= 128, but a difference of four reaches 2*127 + 2*128
    if is_p:
        det[1] = 2 * HI + 2 * (-LO)                      # x0 - x1 + x2 - x3
        ll[1] = 4 * amp                                  # level 1 of a P picture is not scaled
    else:
        row = (3 * 2 * amp + 2 * amp + 1) // 2           # round2 of gain 8
        det[1] = ll[1] = (8 * row + 1) // 2              # the column pass on row-pass outputs
    for lv in (2, 3):
        det[lv] = 4 * ll[lv - 1]
        ll[lv] = (4 * ll[lv - 1]) * 4 // 5
    return det, ll


@pytest.mark.parametrize("is_p", [0, 1])
def test_detail_coefficients_and_symbols_fit_int16(orc, is_p):
    det, ll = haar_bounds(is_p)
    assert det == ({1: 510, 2: 2048, 3: 6552} if is_p else {1: 2048, 2: 8192, 3: 26212}), det
    for lv in (1, 2, 3):
        assert det[lv] <= 32767 and ll[lv] <= 32767
    # symbols: levels 2, 3 (scan levels 1, 0) quantise by (2|v| + 1) / (2q), q >= MINQUANT; level 1 shifts right
    assert (2 * det[3] + 1) // (2 * MINQUANT) <= 32767 and det[1] <= 32767
    # the oracle on extremal sign patterns: never above the bound, and within 2 % of it (rounding) for every level
    w = h = 64
    best = {1: 0, 2: 0, 3: 0}
    pats = []
    for px in (1, 2, 4, 8, 0):
        for py in (1, 2, 4, 8, 0):
            yy, xx = np.mgrid[0:h, 0:w]
            sx = ((xx // px) & 1) if px else np.zeros_like(xx)
            sy = ((yy // py) & 1) if py else np.zeros_like(yy)
            pats.append(np.where((sx ^ sy) == 1, 255, 0).astype(np.uint8))
            pats.append(np.where((sx ^ sy) == 1, 0, 255).astype(np.uint8))
    # B4T's four-tap high pass peaks on the pattern (-, +, -, +) . (1, -3, 3, -1): period-2 checkerboards above cover it
    for pat in pats:
        f = A.BorderedFrame(w, h, A.SUBSAMP_444)
        f.plane(0)[:, :] = pat
        co = np.zeros(w * h, dtype=np.int32)
        orc.orc_fwd_sbt(C.byref(f.c.planes[0]), C.byref(A.Coefs(A.i32p(co), w, h)), is_p)
        co = co.reshape(h, w)
        for lv in (1, 2, 3):
            s = w >> lv
            band = np.abs(np.concatenate([co[:s, s:2 * s].ravel(), co[s:2 * s, :s].ravel(), co[s:2 * s, s:2 * s].ravel()]))
            m = int(band.max())
            assert m <= det[lv], "level %d: |coefficient| %d above the bound %d" % (lv, m, det[lv])
            best[lv] = max(best[lv], m)
    # Tightness.  P pictures: plain Haar all the way, the block patterns attain every level's bound.  I pictures: level 1's bound
    # is attained; the chained bounds of levels 2, 3 assume four neighbouring LL1 values of full size and alternating sign, which
    # overlapping B4T supports cannot all deliver -- there the bound is an upper bound only (what the int16 claim needs).
    for lv in ((1, 2, 3) if is_p else (1,)):
        assert best[lv] >= 0.98 * det[lv], "level %d: bound %d is not attained (best %d): the gain model is wrong" % (lv, det[lv], best[lv])
    if not is_p:
        assert best[3] >= 8192, "the patterns do not stress level 3 of an I picture (best %d)" % best[3]


def test_dequantised_value_is_at_most_twice_its_coefficient():
    """hzcc.c:94-128: a non-zero symbol needs 2|v| > q, so dequant(quant(v)) = |s| q + q / 2 <= 2|v| -- every quantiser the
    frame quantiser can produce (dsv_get_quant :77-92: up to 2047 * 3 / 2 for P pictures, tmq4pos halves / quarters it)"""
    v = np.arange(1, 26213, dtype=np.int64)
    worst = 0.0
    for q in list(range(MINQUANT, 3072, 7)) + [3070, 3071, 2047, 1023, 511]:
        m = v << 1
        s = np.where(m <= q, 0, (m + 1) // (2 * q))
        dq = (s * (2 * q) + q) >> 1
        dq = np.where(s == 0, 0, dq)
        assert (dq <= 2 * v).all(), q
        worst = max(worst, float((dq / v).max()))
    assert worst > 1.4                                    # (the factor is real: the bound cannot be tightened to |v|)


def test_packed_level1_inverse_stays_inside_int16():
    """the int16-pair level 1 of the P-picture luma inverse (k_sbt.hip inv_l1_item): every intermediate of the widest chain"""
    det, ll = haar_bounds(1)
    d1r, d2r, d3r = 2 * det[1], 2 * det[2], 2 * det[3]            # dequantised details (<= twice the coefficient)
    ll3r = 2 * ll[3]                                              # reconstructed LL3 (levels >= 4 run in int32: same factor)
    ll2r = (ll3r * 5 // 4 + 3 * d3r) // 4                         # inverse Haar: (LL * 5 / 4 +- LH +- HL +- HH) / 4
    ll1r = (ll2r * 5 // 4 + 3 * d2r) // 4
    assert ll1r <= 7200
    widest = {
        "lp - ln (neighbouring LL1 values)": 2 * ll1r,
        "mn - mx (both clamped differences)": 4 * ll1r,
        "2 * detail before the nudge's rdiv2": 2 * d1r + 2 * ll1r,
        "LL + HL + LH + HH before div4": ll1r + 3 * d1r + 2 * (2 * ll1r // 4 + 1),
    }
    for what, v in widest.items():
        assert v <= 32767, "%s reaches %d" % (what, v)
