"""GPU parity on clips with known half-pel motion (tests/halfpel_cases.py): every phase combination of the luma and chroma
compensation filters, lanes with different phases in one wave of the lean forward kernel -- stream bytes against the oracle
(which test_halfpel_oracle.py pins to the reference CLI on the same clips), whole clip as one batch and frame by frame, and
the product decoder on the product's stream."""
import importlib

import numpy as np
import pytest

import _cabi as A
import halfpel_cases as H
from test_gpu_stream import product_decode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


@pytest.mark.parametrize("case", range(len(H.CASES)))
def test_halfpel_clips_equal_oracle(pkg, case):
    w, h, fmt, seed, cli = H.CASES[case]
    clip = H.halfpel_clip(w, h, fmt, seed)
    n = clip.shape[0]
    want, recs = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli), want_recon=True, eos=False)
    for per_call in (n, 1):
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), 1, per_call)
        try:
            got = b""
            for t in range(0, n, per_call):
                got += b.encode(clip[t:t + per_call].reshape(1, per_call, -1))[0]
        finally:
            b.close()
        assert len(got) == len(want), "frames per call %d: %d bytes against %d" % (per_call, len(got), len(want))
        assert got == want, "frames per call %d: first difference at byte %d" % (per_call, next(i for i in range(len(want)) if got[i] != want[i]))
    dec = product_decode(pkg, want)
    assert len(dec) == n
    for t in range(n):
        assert np.array_equal(dec[t], recs[t]), "decoded frame %d differs from the oracle's reconstruction" % t
