"""bench.py keeps its contract: one JSON line with the driver's keys, the roofline and cpu_baseline objects, and the
bit-exactness spot check -- run small (8 GOPs per step, one timed step)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--gops", "8", "--cpu-gops", "2", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["bit_exact_vs_cpu"] is True
    # the bytes of the last TIMED step, all streams, against reference-derived bytes (round 4)
    t = d["bit_exact_timed_output"]
    assert t["equal"] is True and t["sha256"] == t["sha256_expected"] and t["streams"] == 8 and t["first_frame_number"] >= 36
    assert "int16" in d["dtype"] and "int32" in d["dtype"]
    r = d["roofline"]
    assert "regime" in r and all("regime" in o for o in r["others_exclusive"])
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] == "GB/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_bench_line_carries_the_other_shapes():
    """after the headline: the PCIe-inclusive figure, configs 2 / 4 / 5 and the batched decoder, each bit-exact"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--gops", "8", "--cpu-gops", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["value_host_pinned"]["value"] > 0 and d["value_host_pinned"]["value"] < d["value"] * 1.5
    sh = d["shapes"]
    assert "error" not in sh, sh
    for k in ("cfg2_1080p_intra", "cfg4_4k_gop12", "cfg4_8gops", "cfg3_4gops", "cfg5_4k_444_abr", "headline_mixed16", "cfg3_worstcase", "decode_1080p_batched"):
        assert sh[k]["Mpix_s"] > 0 and sh[k]["bit_exact_vs_cpu"] is True, (k, sh[k])
    assert sh["cfg3_worstcase"]["intra_blocks_pct_of_P_pictures"] > 25          # the clip really leaves the lean path
    for k in ("cfg5_abr_x8", "cfg5_abr_x16"):
        assert sh[k]["Mpix_s"] > 0, (k, sh[k])
    # round 6: the line explains itself -- host phases, device-side phase sums and idle time, the link as the library's copies see it
    sb = d["step_breakdown"]
    assert sb["host_batches"] == 1 and sb["device_ms_per_batch"]["coding_phases_seen"] == 1
    assert sb["device_ms_per_batch"]["coding_stream0"] > 0 and sb["device_ms_per_batch"]["motion_search"] > 0 and sb["device_idle_ms_per_batch"] >= 0
    assert sum(sb["host_ms_per_batch"].values()) > 0 and sb["fetch_bytes_per_call"] > 0
    assert d["link"]["h2d_GBs"] > 1 and d["link"]["d2h_GBs"] > 1
    assert d["value_host_pinned"]["frac_of_link"] <= 1.05
    assert d["box"]["placement"]["chosen"] in d["box"]["placement"]["measured"]
    assert d["value_recon_all"]["value"] > 0 and d["value_recon_all"]["bit_exact_vs_cpu"] is True
    if "pipeline" in d:                                  # (needs profiles/pmc_traffic.json with the per-step sum)
        assert d["pipeline"]["bound"] == "hbm" and 0 < d["pipeline"]["frac"] < 1
        assert d["pipeline"]["valu"]["wave_instr_per_step"] > 0 and d["pipeline"]["binding_roof"] in ("valu", "hbm")
    c = d["cpu_baseline"]
    assert c["nproc"] >= 1 and isinstance(c["cpu_model"], str)


def test_cfg4_sharded_leg_on_one_gpu():
    """BASELINE config 4's own harness (closed 4K GOPs sharded by shard.gop_range, gathered, joined, compared with the serial
    stream derived from one reference encode) on the one GPU of the box: 6 GOPs instead of 64"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--gops", "4", "--cpu-gops", "0",
                        "--cfg4-sharded", "--cfg4-gops", "6"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, DSV1_BENCH_SKIP_SHAPES="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = d["cfg4_sharded"]
    assert "error" not in c, c
    assert c["bit_exact_vs_cpu"] is True and c["sha256"] == c["sha256_expected"] and c["gops"] == 6 and c["Mpix_s"] > 0
