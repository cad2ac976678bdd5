"""Golden vectors (tests/golden/, generated from the REAL reference by tools/make_goldens.py):
the oracle restatement must reproduce every one of them.  CPU only, needs neither /root/reference
nor oracle/_ref -- this is what pins the oracle on the GPU box."""
import hashlib
import json
import os

import numpy as np
import pytest

import _cabi as A
import golden_cases as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


with open(os.path.join(GOLD, "streams.json")) as f:
    STREAMS = json.load(f)
with open(os.path.join(GOLD, "ops.json")) as f:
    OPS = json.load(f)

SMALL = [k for k, v in G.STREAM_CASES.items() if v[0] <= 704]
BIG = [k for k, v in G.STREAM_CASES.items() if v[0] > 704]


@pytest.mark.parametrize("name", SMALL + BIG)
def test_oracle_stream_matches_reference_golden(orc, name):
    w, h, fmt, n, style, seed, flags, kw = G.STREAM_CASES[name]
    clip = A.gen_clip(w, h, fmt, seed, n, style=style)
    got, recs = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw), want_recon=True)
    want = STREAMS[name]
    assert len(got) == want["len"]
    assert [sha(p) for p in A.split_packets(got)] == want["packets"]
    assert sha(got) == want["sha256"]
    dec = A.orc_decode(got, w, h, fmt)
    assert [sha(d) for d in dec] == want["decoded"]
    assert [sha(r) for r in recs] == want["decoded"], "encoder reconstruction != reference decoder output"


def test_committed_stream_file_is_the_reference_output(orc):
    w, h, fmt, n, style, seed, flags, kw = G.STREAM_CASES["cif_gop12_style2"]
    with open(os.path.join(GOLD, "cif_gop12.dsv"), "rb") as f:
        want = f.read()
    assert sha(want) == STREAMS["cif_gop12_style2"]["sha256"]
    clip = A.gen_clip(w, h, fmt, seed, n, style=style)
    got, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **kw))
    assert got == want
    frames = A.orc_decode(want, w, h, fmt)
    assert len(frames) == n


@pytest.mark.parametrize("name", sorted(G.OP_CASES))
def test_oracle_operator_matches_reference_golden(orc, name):
    got = G.run_op_case(G.OP_CASES[name], "orc", orc)
    assert got == OPS[name]
