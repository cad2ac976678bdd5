"""Device-resident average-bitrate rate control (round 4: include/dsvg_rc.h, k_rc.hip, dsvg_code_batch_rc).

quality2quant (dsv_encoder.c:70-168) and the statistics of dsv_enc (:816-848) run on the device between the frame steps of a
call: a whole batch of ABR streams is enqueued like a CRF batch.  Every stream must equal the oracle encoder's bytes -- which
are pinned to the reference's (tests/test_oracle_vs_ref.py, the ABR goldens of tests/golden/streams.json) -- and the frame-serial
host path of rounds 1-3 (DSV1_ABR_SERIAL=1)."""
import importlib
import os

import numpy as np
import pytest

import _cabi as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


def _batch_streams(pkg, clips, w, h, fmt, F, calls, serial, pipelined=False, **kw):
    """clips [S][calls*F][bytes] through one Batch, F frames per call; returns the S byte strings"""
    S = clips.shape[0]
    old = os.environ.get("DSV1_ABR_SERIAL")
    os.environ["DSV1_ABR_SERIAL"] = "1" if serial else "0"
    try:
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **kw), S, F)
    finally:
        if old is None:
            del os.environ["DSV1_ABR_SERIAL"]
        else:
            os.environ["DSV1_ABR_SERIAL"] = old
    out = [b""] * S
    try:
        if pipelined:
            # two calls in flight: the second is enqueued before the first is collected -- the streams' rate-control state
            # goes from call to call on the device
            b.submit(np.ascontiguousarray(clips[:, :F]))
            for k in range(1, calls):
                b.submit(np.ascontiguousarray(clips[:, k * F:(k + 1) * F]))
                for s, o in enumerate(b.collect()):
                    out[s] += o
            for s, o in enumerate(b.collect()):
                out[s] += o
        else:
            for k in range(calls):
                for s, o in enumerate(b.encode(np.ascontiguousarray(clips[:, k * F:(k + 1) * F]))):
                    out[s] += o
    finally:
        b.close()
    return out


CASES = [
    # w, h, fmt, streams, F, calls, kwargs
    (352, 288, A.SUBSAMP_420, 3, 7, 3, dict(qp=60, gop=12, rc_mode_cli=0)),                      # P pictures over / under the budget, GOP starts inside calls
    (320, 240, A.SUBSAMP_422, 2, 6, 2, dict(qp=85, gop=12, rc_mode_cli=0, kbps=800)),
    (352, 288, A.SUBSAMP_420, 2, 5, 2, dict(qp=85, gop=0, rc_mode_cli=0)),                        # intra-only ABR (BASELINE config 1: the CLI's defaults)
    (704, 480, A.SUBSAMP_444, 18, 4, 2, dict(qp=70, gop=12, rc_mode_cli=0, kbps=3000)),           # 18 streams: two coding streams, each with its own k_rc
    (352, 288, A.SUBSAMP_420, 2, 6, 2, dict(qp=85, gop=12, rc_mode_cli=0, ipct=20)),              # forced-intra pictures (quality2quant's forced_intra branch)
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_abr_batch_on_the_device_equals_oracle_and_the_serial_path(pkg, case):
    w, h, fmt, S, F, calls, kw = CASES[case]
    styles = [1, 2, 0, 4, 5, 3]
    clips = np.stack([A.gen_clip(w, h, fmt, 0xAB400 + 17 * s + case, F * calls, style=styles[s % len(styles)]) for s in range(S)])
    want = [A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, **kw), eos=False)[0] for s in range(S)]
    got = _batch_streams(pkg, clips, w, h, fmt, F, calls, serial=False, **kw)
    for s in range(S):
        assert got[s] == want[s], "stream %d (device rate control) differs from the oracle" % s
    if case < 3:
        ser = _batch_streams(pkg, clips, w, h, fmt, F, calls, serial=True, **kw)
        assert ser == got


def test_abr_state_stays_on_the_device_between_pipelined_calls(pkg):
    w, h, fmt, S, F, calls = 352, 288, A.SUBSAMP_420, 4, 5, 4
    kw = dict(qp=60, gop=12, rc_mode_cli=0)
    clips = np.stack([A.gen_clip(w, h, fmt, 0xAB500 + s, F * calls, style=[1, 2, 4, 0][s]) for s in range(S)])
    want = [A.orc_encode(clips[s], A.orc_cfg(w, h, fmt, **kw), eos=False)[0] for s in range(S)]
    got = _batch_streams(pkg, clips, w, h, fmt, F, calls, serial=False, pipelined=True, **kw)
    for s in range(S):
        assert got[s] == want[s], "stream %d differs" % s
