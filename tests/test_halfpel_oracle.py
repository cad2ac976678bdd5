"""CPU: the half-pel clips of tests/halfpel_cases.py do what they are for -- the oracle encoder picks vectors with every
(horizontal, vertical) half-pel phase, for luma and for chroma -- and the oracle is pinned to the reference CLI on them
(skipped where oracle/_ref is absent), so the GPU parity test in test_gpu_halfpel.py stands on the reference itself."""
import tempfile

import numpy as np
import pytest

import _cabi as A
import halfpel_cases as H


@pytest.mark.parametrize("case", range(len(H.CASES)))
def test_clips_cover_every_phase(case):
    w, h, fmt, seed, cli = H.CASES[case]
    clip = H.halfpel_clip(w, h, fmt, seed)
    lum, chrm, intra = H.oracle_phases(clip, w, h, fmt, **cli)
    assert (lum > 0).all(), "luma phases [yh][xh] %s" % lum.tolist()
    assert (chrm > 0).all(), "chroma phases [yh][xh] %s" % chrm.tolist()
    assert intra < lum.sum() // 10


@pytest.mark.ref
@pytest.mark.parametrize("case", range(len(H.CASES)))
def test_oracle_matches_ref_cli_on_halfpel_clips(case):
    if not A.have_ref():
        pytest.skip("no reference build")
    w, h, fmt, seed, cli = H.CASES[case]
    clip = H.halfpel_clip(w, h, fmt, seed)
    flags = ["-gop%d" % cli["gop"], "-qp%d" % cli["qp"], "-rc_mode%d" % cli["rc_mode_cli"], "-scd%d" % cli["scd"]]
    with tempfile.TemporaryDirectory() as td:
        want = A.ref_cli_encode(clip, w, h, A.FMT_CLI[fmt], flags, td)
        got, recs = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli), want_recon=True)
        assert got == want
        dec_ref = A.ref_cli_decode(want, td).reshape(clip.shape[0], -1)
        for t in range(clip.shape[0]):
            assert np.array_equal(dec_ref[t], recs[t]), "reference decode != oracle reconstruction at frame %d" % t
