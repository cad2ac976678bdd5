"""bench.py --gpus N (verdict round 5, item 8): the flag decides the number of ranks.  Under a launcher the line must not claim
another n_gpus than the ranks that ran (a mismatch exits non-zero BEFORE anything touches a GPU: checked here on the CPU); started
plainly with --gpus N > 1 the script launches its own N child ranks (GPU test: both share device 0, as in test_gpu_multirank.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0
    assert "WORLD_SIZE=2" in p.stderr and "--gpus 4" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_gpus_zero_is_refused():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "0"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_plain_start_with_gpus_2_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DSV1_BENCH_DEBUG_SHARED_GPU"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gops", "4", "--cpu-gops", "2",
                        "--cfg4-gops", "0"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1100)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 2 * 4 * 12 and d["bit_exact_vs_cpu"] is True
