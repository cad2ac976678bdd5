"""The product decoders take the block size from the stream (dsv_decoder.c:335-360): the drop-in dsv_dec builds a context
for the size a picture announces and carries its reference picture over when the size changes in mid-GOP; the batched
decoder follows the streams' size while no stream holds a reference."""
import importlib

import numpy as np
import pytest

import _cabi as A
import blocksize_cases as B
from test_gpu_stream import product_decode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


@pytest.mark.parametrize("case", range(len(B.CASES)))
def test_dsv_dec_follows_stream_block_size(pkg, case):
    w, h, fmt, n, stream = B.make_stream(case)
    want = A.orc_decode(stream, w, h, fmt)
    got = product_decode(pkg, stream)
    assert len(got) == len(want) == n
    for t in range(n):
        A.assert_same("decoded frame %d" % t, got[t], want[t])


@pytest.mark.parametrize("case", [0, 1])
def test_batched_decoder_follows_stream_block_size(pkg, case):
    w, h, fmt, n, stream = B.make_stream(case)
    want = A.orc_decode(stream, w, h, fmt)
    pk = A.split_packets(stream)
    S = 3
    d = pkg.DecBatch(w, h, fmt, S)
    got = []
    try:
        for p in pk:
            out, status, fnum = d.decode([p] * S)
            if p[5] & 4:
                assert list(status) == [0] * S, list(status)
                for s in range(1, S):
                    assert (out[s] == out[0]).all()
                got.append(out[0].copy())
    finally:
        d.close()
    assert len(got) == n
    for t in range(n):
        A.assert_same("decoded frame %d" % t, got[t], want[t])
