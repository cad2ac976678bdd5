"""GPU path vs the committed golden vectors of the REAL reference (tests/golden/)."""
import hashlib
import importlib
import json
import os

import pytest

import _cabi as A
import golden_cases as G

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


with open(os.path.join(GOLD, "streams.json")) as f:
    STREAMS = json.load(f)
with open(os.path.join(GOLD, "ops.json")) as f:
    OPS = json.load(f)


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


@pytest.mark.parametrize("name", sorted(G.STREAM_CASES))
def test_gpu_stream_matches_reference_golden(pkg, name):
    w, h, fmt, n, style, seed, flags, kw = G.STREAM_CASES[name]
    clip = A.gen_clip(w, h, fmt, seed, n, style=style)
    got = pkg.encode_clip(clip, w, h, fmt, **kw)
    want = STREAMS[name]
    assert len(got) == want["len"]
    assert [sha(p) for p in A.split_packets(got)] == want["packets"]
    assert sha(got) == want["sha256"]


@pytest.mark.parametrize("name", sorted(G.OP_CASES))
def test_gpu_operator_matches_reference_golden(pkg, orc, name):
    got = G.run_op_case(G.OP_CASES[name], "prod", A.load_prod(), orc=orc)
    assert got == OPS[name]
