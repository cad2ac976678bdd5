"""MI355X-native DSV1 hot path: thin ctypes binding over the C ABI (include/dsvg.h, include/dsv1_api.h).

The product is libdsv1_mi355x.so (HIP kernels for gfx950 + the C session layer).  This module only
loads it; there is no Python or CPU fallback -- if the library is missing, or no HIP device is
usable, calls fail loudly."""
import ctypes as _C
import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))
SO_PATH = _os.path.join(_HERE, "libdsv1_mi355x.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _os.path.exists(SO_PATH):
            raise RuntimeError("HIP extension %s is not built; run __graft_entry__.build()" % SO_PATH)
        _lib = _C.CDLL(SO_PATH)
        _lib.dsvg_last_error.restype = _C.c_char_p
    return _lib
