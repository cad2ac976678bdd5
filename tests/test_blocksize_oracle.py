"""The oracle decoder follows the block size a picture announces exactly as the reference decoder does
(dsv_decoder.c:335-360) -- checked against the reference CLI on streams with other block sizes than the encoder's rule,
including a change in mid-GOP.  (The GPU decoders are then checked against the oracle: test_gpu_blocksize.py.)"""
import numpy as np
import pytest

import _cabi as A
import blocksize_cases as B

pytestmark = pytest.mark.skipif(not A.have_ref(), reason="needs the compiled reference (build container)")


@pytest.mark.parametrize("case", range(len(B.CASES)))
def test_oracle_decoder_follows_stream_block_size(case, tmp_path):
    w, h, fmt, n, stream = B.make_stream(case)
    want = np.asarray(A.ref_cli_decode(stream, str(tmp_path))).reshape(n, -1)
    got = A.orc_decode(stream, w, h, fmt)
    assert len(got) == n
    for t in range(n):
        A.assert_same("frame %d" % t, got[t], want[t])
