"""Every environment switch the library reads (grep getenv over csrc/) selects a live alternate path or a debug print.  Each is set here, in a
process of its own, over tests/switch_probe.py's workload -- five reference goldens, the batched pipeline on device / copied / host clips, chain
mode, the drop-in dsv_enc, dsv_dec and the batched decoder -- and must leave every hash unchanged (verdict round 5, item 5).  A switch that
appears in csrc/ and not in SWITCHES fails test_every_switch_of_the_library_is_listed (CPU)."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")

# path-changing switches: name -> value
SWITCHES = {
    "DSV1_NO_MC_FUSION": "1", "DSV1_NO_INPLACE_PRED": "1", "DSV1_NO_DEC_SYM": "1", "DSV1_NO_DEC_SYM_I": "1", "DSV1_NO_DEC_SYM_OV": "1",
    "DSV1_NO_PATCH_KERNEL": "1", "DSV1_NO_LIST_PACK": "1", "DSV1_NO_LAZY_BORDER": "1", "DSV1_NO_LLQ": "1", "DSV1_NO_CHROMA_IN_PLACE": "1",
    "DSV1_NO_LUMA_IN_PLACE": "1", "DSV1_NO_FUSE_LEVEL2": "1", "DSV1_NO_UNPACK_SIDES": "1", "DSV1_NO_LEVEL_SIDES": "1", "DSV1_NO_CHROMA_SUMS": "1",
    "DSV1_NO_INV54_ALL": "1", "DSV1_NO_SMALL_SPLIT": "1", "DSV1_NO_FETCH_FAST": "1", "DSV1_DEC_ONE_STREAM": "1", "DSV1_NO_MC_PATCH": "1",
    "DSV1_NO_XCD_ORDER": "1", "DSV1_NO_FUSED_BORDER": "1", "DSV1_NO_PATCH_PART": "1", "DSV1_NO_EDGE_TILES": "1", "DSV1_DEC_NO_POOL": "1",
    "DSV1_NO_STREAM_PROBE": "1", "DSV1_NO_PRIO_COPY_STREAM": "1", "DSV1_NO_PAR_ENQUEUE": "1", "DSV1_ABR_SERIAL": "1", "DSV1_RECON_ALL": "1",
    "DSV1_NO_RECYCLE": "1", "DSV1_RECYCLE_MAX_MB": "1", "DSV1_HOST_THREADS": "3", "DSV1_ENC_PIPELINE": "0", "DSV1_ENC_LOOKAHEAD": "5",
}
CODE_STREAMS = ("1", "3", "4")            # DSV1_CODE_STREAMS: the default is 2
# switches that only print / time: set together
DEBUG = {"DSV1_BORDER_DEBUG": "1", "DSV1_STREAM_DEBUG": "1", "DSV1_DEC_VERBOSE": "1", "DSV1_DEC_PROF": "1", "DSV1_HOST_PROF": "1", "DSV1_TIMELINE": "1",
         "DSV1_DEBUG_LINK_REPEAT": "2"}        # (the large copies issued twice: a slow link emulated, same bytes)
# not the library's own: set by launchers (shard.py / torch.distributed.run), covered by tests/test_rank_cores.py
LAUNCHER = {"DSV1_CORES_PINNED", "LOCAL_WORLD_SIZE"}


def switches_in_sources():
    names = set()
    for f in glob.glob(os.path.join(ROOT, "digital-subband-video-1_amd", "csrc", "**", "*"), recursive=True):
        if f.endswith((".hip", ".hpp", ".c", ".h")):
            names |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(f).read()))
    return names


def test_every_switch_of_the_library_is_listed():
    known = set(SWITCHES) | set(DEBUG) | LAUNCHER | {"DSV1_CODE_STREAMS"}
    assert switches_in_sources() <= known, "untested switches: %s" % sorted(switches_in_sources() - known)
    assert known <= switches_in_sources() | LAUNCHER, "listed but gone from the sources: %s" % sorted(known - switches_in_sources() - LAUNCHER)


def probe(extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("DSV1_")}
    env.update(extra)
    p = subprocess.run([sys.executable, os.path.join(HERE, "switch_probe.py")], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, "switch_probe.py failed under %s:\n%s" % (extra, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def plain():
    got = probe({})
    with open(os.path.join(GOLD, "streams.json")) as f:
        gold = json.load(f)
    for k, v in got.items():
        if k.startswith("golden:"):
            assert v == gold[k[7:]]["sha256"], "%s differs from the reference's golden WITHOUT any switch" % k
    return got


def _same(plain, got, what):
    diff = sorted(k for k in plain if got.get(k) != plain[k])
    assert not diff and set(got) == set(plain), "%s changes the output of: %s" % (what, diff)


@pytest.mark.gpu
@pytest.mark.timeout(1000)
@pytest.mark.parametrize("name", sorted(SWITCHES))
def test_switch_leaves_every_output_unchanged(plain, name):
    _same(plain, probe({name: SWITCHES[name]}), "%s=%s" % (name, SWITCHES[name]))


@pytest.mark.gpu
@pytest.mark.timeout(1000)
@pytest.mark.parametrize("n", CODE_STREAMS)
def test_code_streams_leave_every_output_unchanged(plain, n):
    _same(plain, probe({"DSV1_CODE_STREAMS": n}), "DSV1_CODE_STREAMS=%s" % n)


@pytest.mark.gpu
@pytest.mark.timeout(1000)
def test_debug_prints_leave_every_output_unchanged(plain):
    _same(plain, probe(DEBUG), "the debug / profiling switches")
