"""N>1 path on CPU: two gloo processes shard the GOPs of one clip, gather the per-GOP streams on rank 0
and join them with the product's host-side dsv1_concat_gops (plain C, no GPU) -- the result must equal
the serial stream.  The per-GOP encoder here is the oracle (test infrastructure) standing in for the
GPU; the sharding, gathering and link fix-up code under test is the product's."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import _cabi as A

W, H, FMT, GOP, NGOPS = 176, 144, A.SUBSAMP_420, 6, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, A.ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("digital-subband-video-1_amd.shard")
    pkg = importlib.import_module("digital-subband-video-1_amd")
    clip = A.gen_clip(W, H, FMT, 0x5A4D, GOP * NGOPS, style=0)
    lo, hi = shard.gop_range(NGOPS, world, rank)
    cfg = A.orc_cfg(W, H, FMT, qp=85, gop=GOP, rc_mode_cli=1)
    local = []
    for g in range(lo, hi):
        s, _ = A.orc_encode(clip[g * GOP:(g + 1) * GOP], cfg, start_fnum=g * GOP, eos=False)
        local.append((g, s))
    parts = shard.gather_streams(local, dist)
    dist.barrier()
    if rank == 0:
        joined = pkg.concat_gops(parts)
        serial, _ = A.orc_encode(clip, cfg)
        q.put((joined == serial, len(joined), len(serial), [len(p) for p in parts]))
    dist.destroy_process_group()


def test_gop_partition_is_balanced_and_complete():
    shard = importlib.import_module("digital-subband-video-1_amd.shard")
    for n in (1, 5, 8, 64, 65):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard.gop_range(n, world, r)
                assert 0 <= hi - lo <= -(-n // world)
                cover += list(range(lo, hi))
            assert cover == list(range(n))


@pytest.mark.timeout(300)
def test_two_rank_gop_sharding_equals_serial_stream():
    if not os.path.exists(A.PROD_SO):
        import __graft_entry__ as g
        g.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, n_joined, n_serial, sizes = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len(sizes) == NGOPS
    assert ok, "sharded %d bytes vs serial %d bytes" % (n_joined, n_serial)


def test_rank_core_pinning_splits_the_allowed_cores():
    import subprocess
    code = ("import os,sys,importlib; sys.path.insert(0, %r); sh = importlib.import_module('digital-subband-video-1_amd.shard'); "
            "a = sorted(os.sched_getaffinity(0)); m = sh.pin_rank_to_cores(1, 2); "
            "print(len(a), len(m), sorted(os.sched_getaffinity(0)) == m)" % A.ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, check=True)
    n, mine, ok = r.stdout.split()
    assert ok == "True" and (int(n) < 2 or int(mine) == int(n) // 2)
