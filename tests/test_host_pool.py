"""The session layer's worker pool (csrc/host/dsv1_util.c): foreground loops (dsv1_par_for) and, since round 6, ONE background loop beside them
(dsv1_par_bg_begin / _end: the packet prefixes of a CRF batch are written by idle workers while the session thread waits for the GPU).
Host logic only: no GPU call."""
import ctypes as C
import os
import threading
import time

import pytest

import _cabi as A

FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(A.PROD_SO):
        import __graft_entry__ as g
        g.build()
    os.environ.setdefault("DSV1_HOST_THREADS", "4")        # (read once per process: four threads whatever the box has)
    lib = C.CDLL(A.PROD_SO)
    lib.dsv1_par_for.argtypes = [C.c_int, FN, C.c_void_p]
    lib.dsv1_par_bg_begin.argtypes = [C.c_int, FN, C.c_void_p]
    lib.dsv1_par_bg_end.restype = None
    lib.dsv1_par_bg_pending.restype = C.c_int
    return lib


def test_background_loop_runs_every_item_once_beside_foreground_loops(L):
    bg = [0] * 300
    fg = [0] * 64
    tids = set()
    lock = threading.Lock()

    def bg_item(ctx, s, tid):
        time.sleep(0.0005)
        with lock:
            bg[s] += 1
            tids.add(tid)

    def fg_item(ctx, s, tid):
        with lock:
            fg[s] += 1
    cb_bg, cb_fg = FN(bg_item), FN(fg_item)
    L.dsv1_par_bg_begin(len(bg), cb_bg, None)
    assert L.dsv1_par_bg_pending() == 1
    for _ in range(5):                                   # foreground loops while the background one is out: they finish, every item once
        for i in range(len(fg)):
            fg[i] = 0
        L.dsv1_par_for(len(fg), cb_fg, None)
        assert fg == [1] * len(fg)
    time.sleep(0.05)                                     # the session thread "waits for the GPU": idle workers take background items meanwhile
    with lock:
        done_early = sum(bg)
    L.dsv1_par_bg_end()
    assert L.dsv1_par_bg_pending() == 0
    assert bg == [1] * len(bg)
    assert done_early > 0, "no background item ran before the join: the workers never picked the loop up"
    L.dsv1_par_bg_end()                                  # (joining twice is harmless)


def test_a_second_background_loop_joins_the_first(L):
    a, b = [0] * 40, [0] * 40
    lock = threading.Lock()

    def mk(arr):
        def item(ctx, s, tid):
            time.sleep(0.001)
            with lock:
                arr[s] += 1
        return FN(item)
    ca, cb = mk(a), mk(b)
    L.dsv1_par_bg_begin(len(a), ca, None)
    L.dsv1_par_bg_begin(len(b), cb, None)                # joins the first loop before it starts the second
    assert a == [1] * len(a)
    L.dsv1_par_bg_end()
    assert b == [1] * len(b)
