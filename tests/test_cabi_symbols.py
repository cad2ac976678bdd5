"""The drop-in boundary: libdsv1_mi355x.so must load (no GPU needed for that) and export every
function declared in include/dsvg.h and include/dsv1_api.h.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

import _cabi as A

INC = os.path.join(A.ROOT, "include")
DECL = re.compile(r"^\s*(?:/\*.*?\*/\s*)?(?:const\s+)?(?:unsigned\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**\s*\*?\s*((?:dsvg|dsv1|dsv)_[a-z0-9_]+|estimate_bitrate|conv444to422|conv422to420)\s*\(", re.M)


def declared(header):
    with open(os.path.join(INC, header)) as f:
        src = f.read()
    names = set(DECL.findall(src))
    return sorted(n for n in names if not n.endswith("_t"))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(A.PROD_SO):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(A.PROD_SO)


@pytest.mark.parametrize("header", ["dsvg.h", "dsv1_api.h"])
def test_every_declared_symbol_is_exported(lib, header):
    names = declared(header)
    assert len(names) > 15, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/%s but not exported: %s" % (header, missing)


def test_reference_cli_symbols_are_present(lib):
    """every function dsv_main.c binds"""
    need = ["dsv_enc_init", "dsv_enc_set_metadata", "dsv_enc_start", "dsv_enc", "dsv_enc_end_of_stream", "dsv_enc_free",
            "dsv_dec", "dsv_get_metadata", "dsv_dec_free", "dsv_yuv_read", "dsv_yuv_write", "dsv_load_planar_frame",
            "dsv_mk_frame", "dsv_frame_ref_dec", "dsv_mk_buf", "dsv_buf_free", "dsv_free", "dsv_set_log_level",
            "dsv_memory_report", "estimate_bitrate", "conv444to422", "conv422to420"]
    assert not [n for n in need if not hasattr(lib, n)]


def test_no_device_fails_loudly_not_silently(lib):
    """without a HIP device the operator calls must return an error code, never a CPU result"""
    lib.dsvg_device_count.restype = C.c_int
    if lib.dsvg_device_count() > 0:
        pytest.skip("a HIP device is present")
    lib.dsvg_last_error.restype = C.c_char_p
    import numpy as np
    w = h = 32
    px = np.zeros(w * h, dtype=np.uint8)
    co = np.zeros(w * h, dtype=np.int32)
    pl = A.Plane(A.u8p(px), w * h, 0, w, w, h, 0, 0)
    rc = lib.dsvg_op_fwd_sbt(C.byref(pl), C.byref(A.Coefs(A.i32p(co), w, h)), 1)
    assert rc != 0 and lib.dsvg_last_error()
    assert not co.any()
    # the session-level entry points refuse as well: batch encoder and batched decoder
    import importlib
    pkg = importlib.import_module("digital-subband-video-1_amd")
    pkg.lib()
    cfg = pkg.make_encoder_cfg(64, 64, A.SUBSAMP_420)
    h = C.c_void_p(None)
    assert lib.dsv1_batch_open(C.byref(h), C.byref(cfg), 0, 1, 1) != 0 and not h.value
    m = pkg.Meta()
    m.width, m.height, m.subsamp = 64, 64, A.SUBSAMP_420
    assert lib.dsv1_decbatch_open(C.byref(h), 0, C.byref(m), 2) != 0 and not h.value


def test_host_only_helpers_work_without_gpu(lib):
    """scalar helpers that are part of the boundary but not of the hot path"""
    lib.dsvg_get_quant.restype = C.c_int
    assert lib.dsvg_get_quant(313, 0, 0) == 313 and lib.dsvg_get_quant(313, 1, 1) == 312
    assert lib.dsvg_get_quant(313, 0, 1) == 208 and lib.dsvg_get_quant(313, 0, 2) == 469
    assert lib.dsvg_lb2(1920) == 11 and lib.dsvg_lb2(1) == 0
    orc = A.load_orc()
    for q in (1, 16, 40, 313, 2047):
        for isP in (0, 1):
            for lvl in (0, 1, 2):
                assert lib.dsvg_get_quant(q, isP, lvl) == orc.orc_get_quant(q, isP, lvl)
