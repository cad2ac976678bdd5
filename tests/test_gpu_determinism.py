"""Many streams side by side must not influence one another: the same clip encoded as stream s and as stream s % 4 of one
batch gives the same bytes, and a second run gives the first run's bytes.  (A race between the waves of one workgroup in
the chunk-list entropy kernels -- one wave taking chunk flags down while another was still reading them -- showed only
here: one picture in five thousand, never in the single-stream parity tests.)  Stream 0 is also compared with the oracle."""
import importlib

import numpy as np
import pytest

import _cabi as A

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,streams,frames,per_batch,reps", [(704, 480, 96, 5, 1, 3), (1920, 1080, 48, 4, 1, 2), (704, 480, 64, 6, 3, 2)])
def test_twin_streams_and_repeats(w, h, streams, frames, per_batch, reps):
    pkg = importlib.import_module("digital-subband-video-1_amd")
    assert pkg.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    fmt = A.SUBSAMP_420
    cli = dict(qp=85, gop=12, rc_mode_cli=1)
    clips = [A.gen_clip(w, h, fmt, 0xD7E0 + g, frames, style=(0, 2, 0, 1)[g]) for g in range(4)]
    first = None
    for r in range(reps):
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), streams, per_batch)
        try:
            got = [b""] * streams
            for t in range(0, frames, per_batch):
                fr = np.stack([clips[s % 4][t:t + per_batch] for s in range(streams)]).reshape(streams, per_batch, -1)
                pk = b.encode(fr)
                for s in range(streams):
                    assert pk[s] == pk[s % 4], "rep %d frames %d..: stream %d differs from its twin %d" % (r, t, s, s % 4)
                    got[s] += pk[s]
        finally:
            b.close()
        if first is None:
            first = got[:4]
            want = A.orc_encode(clips[0], A.orc_cfg(w, h, fmt, **cli), eos=False)[0]
            assert got[0] == want, "stream 0 differs from the oracle"
        else:
            assert got[:4] == first, "repetition %d differs from the first run" % r
