"""CPU: the build's ISA check (csrc/Makefile) refuses an object that contains gfx950's v_ashr_pk_u8_i32 -- the instruction keeps
bits 31:16 of its destination (tools/ubench/ashr_pk.hip, measured on MI355X) while this compiler merges its result into wider
values as if they were zero.  A probe source with the instruction in inline asm must fail to build through
the same rule that builds the kernels; a harmless one must pass."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "digital-subband-video-1_amd", "csrc")

PROBE = """#include <hip/hip_runtime.h>
__global__ void zz_probe(const int *a, unsigned *o) { unsigned d = 0; %s o[0] = d + (unsigned)a[0]; }
"""


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
@pytest.mark.parametrize("bad", [True, False])
def test_makefile_refuses_ashr_pk(bad):
    name = "zz_guard_probe_%d" % int(bad)
    src = os.path.join(CSRC, name + ".hip")
    obj = os.path.join(CSRC, "build", name + ".o")
    body = 'asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 8" : "+v"(d) : "v"(a[0]), "v"(a[1]));' if bad else ""
    try:
        with open(src, "w") as f:
            f.write(PROBE % body)
        r = subprocess.run(["make", "-C", CSRC, obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if bad:
            assert r.returncode != 0 and "refused" in r.stdout, r.stdout[-600:]
            assert not os.path.exists(obj)
        else:
            assert r.returncode == 0 and os.path.exists(obj), r.stdout[-600:]
    finally:
        for p in (src, obj):
            if os.path.exists(p):
                os.remove(p)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
@pytest.mark.parametrize("how", ["no_objdump", "other_arch_name"])
def test_makefile_isa_check_fails_closed(how):
    """the check must not pass because it could not be made: a missing llvm-objdump, or a bundle entry that is not named
    for the architecture asked for, refuses the object"""
    name = "zz_guard_closed_%s" % how
    src = os.path.join(CSRC, name + ".hip")
    obj = os.path.join(CSRC, "build", name + ".o")
    try:
        with open(src, "w") as f:
            f.write(PROBE % "")
        if how == "no_objdump":
            cmd = ["make", "-C", CSRC, obj, "OBJDUMP=/nonexistent/llvm-objdump"]
        else:
            # compile for gfx950 but make the rule look for another architecture's code object
            cmd = ["make", "-C", CSRC, obj, "ARCH=gfx950", "HIPFLAGS=--offload-arch=gfx942 -O3 -std=c++17 -fPIC"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode != 0 and "refused" in r.stdout, r.stdout[-800:]
        assert not os.path.exists(obj)
    finally:
        for p in (src, obj):
            if os.path.exists(p):
                os.remove(p)
