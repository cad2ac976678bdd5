"""The drop-in allocator (dsv_alloc / dsv_free, dsv.c:41-96) with its round-4 recycling of large blocks: a recycled block
handed out by dsv_alloc is zeroed like a fresh one, accounting stays balanced, and the parked blocks can be given back.
CPU only: host code of the C-ABI library, no kernel is launched."""
import ctypes as C

import _cabi as A


def _lib():
    L = C.CDLL(A.PROD_SO)
    L.dsv_alloc.restype = C.c_void_p
    L.dsv_alloc.argtypes = [C.c_int]
    L.dsv_free.argtypes = [C.c_void_p]
    L.dsv1_release_parked.restype = None
    L.dsv1_parked_bytes.restype = C.c_size_t
    L.dsv1_recycle_hold.argtypes = [C.c_int]
    return L


def test_recycled_blocks_come_back_zeroed_and_sized():
    L = _lib()
    L.dsv1_release_parked()
    L.dsv1_recycle_hold(1)                     # (what an open batch does)
    try:
        _recycled_blocks(L)
    finally:
        L.dsv1_recycle_hold(-1)


def _recycled_blocks(L):
    n = 3 << 20
    p = L.dsv_alloc(n)
    assert p
    buf = (C.c_ubyte * n).from_address(p)
    assert not any(buf[i] for i in range(0, n, 4099))
    C.memset(p, 0xA5, n)
    L.dsv_free(p)
    # a slightly smaller request takes the parked block (same pages), and sees zeros again
    q = L.dsv_alloc(n - 1000)
    assert q == p, "the parked block was not reused"
    buf = (C.c_ubyte * (n - 1000)).from_address(q)
    assert not any(buf[i] for i in range(0, n - 1000, 4099)) and buf[n - 1001] == 0
    L.dsv_free(q)
    # a much smaller request does not take a block far beyond twice its size
    r = L.dsv_alloc(512 << 10)
    assert r != p
    L.dsv_free(r)
    # small blocks are never parked
    s = L.dsv_alloc(1000)
    L.dsv_free(s)
    L.dsv1_release_parked()
    t = L.dsv_alloc(n)
    assert t
    L.dsv_free(t)
    L.dsv1_release_parked()


def test_many_blocks_stay_bounded():
    L = _lib()
    L.dsv1_release_parked()
    L.dsv1_recycle_hold(1)
    try:
        ps = [L.dsv_alloc(300 << 10) for _ in range(1200)]       # far more than the 96 blocks a size class parks
        assert all(ps)
        for p in ps:
            L.dsv_free(p)
        assert 0 < L.dsv1_parked_bytes() <= 96 * (512 << 10)
        qs = [L.dsv_alloc(300 << 10) for _ in range(1200)]
        assert all(qs) and len(set(qs)) == 1200
        for q in qs:
            L.dsv_free(q)
    finally:
        L.dsv1_recycle_hold(-1)
    assert L.dsv1_parked_bytes() == 0


def test_nothing_stays_parked_without_an_open_batch():
    """a drop-in library must not keep a finished encoder's memory (advisor round 4): blocks are parked only while a batch /
    session holds the recycler, and the last holder's release gives everything back"""
    L = _lib()
    L.dsv1_release_parked()
    p = L.dsv_alloc(3 << 20)
    L.dsv_free(p)
    assert L.dsv1_parked_bytes() == 0          # nobody holds: freed for good
    assert L.dsv1_recycle_hold(1) == 1
    assert L.dsv1_recycle_hold(1) == 2
    p = L.dsv_alloc(3 << 20)
    L.dsv_free(p)
    assert L.dsv1_parked_bytes() >= 3 << 20
    assert L.dsv1_recycle_hold(-1) == 1
    assert L.dsv1_parked_bytes() >= 3 << 20    # one holder left
    assert L.dsv1_recycle_hold(-1) == 0
    assert L.dsv1_parked_bytes() == 0
