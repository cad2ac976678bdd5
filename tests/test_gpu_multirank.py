"""The N>1 path on the GPU that is there (one MI355X per gpurun box): two ranks launched the way the driver launches
bench.py (torch.distributed.run, one process per rank) sharing device 0 through DSV1_BENCH_DEBUG_SHARED_GPU, and the
GOP sharding of shard.py with the REAL encoder -- two encoder contexts with their own frame-number ranges, gathered and
joined, equal the serial stream (dsv_encoder.c:624-641: closed GOPs are the unit of sharding)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import _cabi as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    m = importlib.import_module("digital-subband-video-1_amd")
    assert m.lib().dsvg_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return m


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("ranks,gops", [(2, 8), (8, 4)])
def test_bench_ranks_on_one_gpu(ranks, gops):
    """2 ranks, and the 8 ranks of a full node (their core split, 8 stream-placement probes on one device, the barrier and
    the MAX over 8 processes) -- the latter with 4 GOPs per rank so that eight contexts fit beside each other"""
    env = dict(os.environ)
    env["DSV1_BENCH_DEBUG_SHARED_GPU"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    steps = 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(A.ROOT, "bench.py"), "--gpus", str(ranks), "--steps", str(steps), "--warmup", "1",
           "--gops", str(gops), "--cpu-gops", "2", "--cfg4-gops", "16"]
    r = subprocess.run(cmd, cwd=A.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE line, the other ranks none
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == steps and d["scaling"] == "weak"
    assert d["bit_exact_vs_cpu"] is True
    assert d["cpu_baseline"] is None                     # the timed CPU sample belongs to N=1
    assert "shapes" not in d
    # config 4's leg: 16 closed 4K GOPs sharded over the ranks (gop_range), gathered, joined, equal to the serial stream
    c4 = d["cfg4_sharded"]
    assert "error" not in c4, c4
    assert c4["bit_exact_vs_cpu"] is True and c4["gops"] == 16 and c4["n_gpus"] == ranks and c4["gops_per_gpu"] == 16 // ranks and c4["Mpix_s"] > 0
    pix = gops * 12 * 1920 * 1080
    want = ranks * pix * steps / (d["ms_per_step"] * 1e-3 * steps) / 1e6      # whole job: all ranks' pixels over the MAX time
    assert abs(d["value"] - want) <= 0.01 * want
    assert d["config"]["frames_per_step"] == ranks * gops * 12
    ncores = len(os.sched_getaffinity(0))
    if ncores >= ranks:
        # shard.pin_rank_to_cores: this rank's share only (cores / ranks as a plain slice; by NUMA node where sysfs shows one GPU per rank)
        assert 1 <= d["config"]["host_cores_rank0"] <= max(ncores // ranks, ncores - (ranks - 1) if ranks == 1 else 2 * (ncores // ranks))
    assert d["config"]["streams_on_own_hw_queue"] >= 0                     # (0: several processes share the device: the probe keeps what it got)


def test_gop_sharding_over_two_encoder_contexts_equals_serial(pkg):
    shard = importlib.import_module("digital-subband-video-1_amd.shard")
    w, h, fmt, gop, ngops = 352, 288, A.SUBSAMP_420, 6, 8
    clip = A.gen_clip(w, h, fmt, 0x5A4E, gop * ngops, style=0)
    cli = dict(qp=85, gop=gop, rc_mode_cli=1, scd=0)
    world = 2
    local = []
    for rank in range(world):                            # what each rank of a node does with its own GPU
        lo, hi = shard.gop_range(ngops, world, rank)
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), hi - lo, gop)
        try:
            for s in range(hi - lo):
                b.set_fnum(s, (lo + s) * gop)
            parts = b.encode(clip[lo * gop:hi * gop].reshape(hi - lo, gop, -1))
        finally:
            b.close()
        local.append([(lo + s, parts[s]) for s in range(hi - lo)])
    joined = pkg.concat_gops(shard.gather_streams([p for part in local for p in part]))
    serial = pkg.encode_clip(clip, w, h, fmt, **cli)
    want, _ = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli))
    assert serial == want
    assert joined == serial
