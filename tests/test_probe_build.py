"""The diagnostic builds DESIGN.md's evidence comes from still compile: the in-kernel clock / stage probe (DSVG_CLOCK_PROBE, tools/ab/clock_probe.sh)
and the motion search's ablation and sensitivity switches (AB_HME_*).  Compile only (gfx950 cross-compile, no GPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "digital-subband-video-1_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,flags", [
    ("k_hme.hip", ["-DDSVG_CLOCK_PROBE"]),
    ("k_hme.hip", ["-DAB_HME_NO_CAND", "-DAB_HME_NO_NINE", "-DAB_HME_NO_STATS", "-DAB_HME_NO_CHROMA", "-DAB_HME_NO_HP", "-DAB_HME_DUMMY_SALU=8", "-DAB_HME_DUMMY_VALU=8",
                   "-DHME_NINE_LDS=0"]),
])
def test_diagnostic_variants_compile(tmp_path, src, flags):
    out = tmp_path / "o.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", str(out),
                        "-I" + os.path.join(ROOT, "include")] + flags, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists() and out.stat().st_size > 10000
