"""MI355X-native DSV1 hot path: thin ctypes binding over the C ABI (include/dsvg.h, include/dsv1_api.h).

The product is libdsv1_mi355x.so (HIP kernels for gfx950 + the C session layer).  This module only
loads it and marshals arguments; there is no Python or CPU fallback -- if the library is missing,
or no HIP device is usable, calls fail loudly."""
import ctypes as _C
import os as _os

import numpy as _np

_HERE = _os.path.dirname(_os.path.abspath(__file__))
# DSV1_SO: another build of the same library (A/B variants of tools/ab/variants.sh); never a different implementation
SO_PATH = _os.environ.get("DSV1_SO") or _os.path.join(_HERE, "libdsv1_mi355x.so")
_lib = None

SUBSAMP_444, SUBSAMP_422, SUBSAMP_420, SUBSAMP_411 = 0x0, 0x4, 0x5, 0x8
MAX_QUALITY = 2047


class Meta(_C.Structure):
    _fields_ = [("width", _C.c_int), ("height", _C.c_int), ("subsamp", _C.c_int), ("fps_num", _C.c_int),
                ("fps_den", _C.c_int), ("aspect_num", _C.c_int), ("aspect_den", _C.c_int)]


class Encoder(_C.Structure):
    """DSV_ENCODER (dsv_encoder.h:58-110 field order)"""
    _fields_ = [("quality", _C.c_int), ("gop", _C.c_int), ("do_scd", _C.c_int), ("rc_mode", _C.c_int),
                ("rc_high_motion_nudge", _C.c_int), ("bitrate", _C.c_uint), ("max_q_step", _C.c_int),
                ("min_quality", _C.c_int), ("max_quality", _C.c_int), ("min_I_frame_quality", _C.c_int),
                ("intra_pct_thresh", _C.c_int), ("scene_change_delta", _C.c_int), ("stable_refresh", _C.c_uint),
                ("pyramid_levels", _C.c_int),
                ("rc_quant", _C.c_uint), ("bpf_total", _C.c_uint), ("bpf_reset", _C.c_uint), ("bpf_avg", _C.c_int),
                ("total_P_frame_q", _C.c_int), ("avg_P_frame_q", _C.c_int), ("last_P_frame_over", _C.c_int),
                ("back_into_range", _C.c_int), ("next_fnum", _C.c_uint32), ("ref", _C.c_void_p), ("vidmeta", Meta),
                ("prev_link", _C.c_int), ("force_metadata", _C.c_int), ("stability", _C.c_void_p),
                ("refresh_ctr", _C.c_uint), ("stable_blocks", _C.c_void_p), ("prev_gop", _C.c_uint32),
                ("prev_avg_luma", _C.c_int)]


class Buf(_C.Structure):
    _fields_ = [("data", _C.POINTER(_C.c_uint8)), ("len", _C.c_uint)]


def lib():
    global _lib
    if _lib is None:
        if not _os.path.exists(SO_PATH):
            raise RuntimeError("HIP extension %s is not built; run __graft_entry__.build()" % SO_PATH)
        L = _C.CDLL(SO_PATH)
        L.dsvg_last_error.restype = _C.c_char_p
        L.dsv1_batch_open.argtypes = [_C.POINTER(_C.c_void_p), _C.POINTER(Encoder), _C.c_int, _C.c_int, _C.c_int]
        L.dsv1_batch_close.argtypes = [_C.c_void_p]
        L.dsv1_batch_set_fnum.argtypes = [_C.c_void_p, _C.c_int, _C.c_uint32]
        L.dsv1_batch_dropped_recons.restype = _C.c_long
        L.dsv1_batch_recon_all.argtypes = [_C.c_void_p, _C.c_int]
        L.dsv1_batch_dropped_recons.argtypes = [_C.c_void_p, _C.POINTER(_C.c_long)]
        L.dsv1_batch_encode.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_int, _C.POINTER(Buf)]
        L.dsv1_batch_submit.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_int, _C.POINTER(Buf)]
        L.dsv1_batch_collect.argtypes = [_C.c_void_p, _C.POINTER(Buf)]
        L.dsv1_batch_eos.argtypes = [_C.c_void_p, _C.c_int, _C.POINTER(Buf)]
        L.dsv1_concat_gops.argtypes = [_C.POINTER(Buf), _C.c_int, _C.POINTER(Buf)]
        L.dsv1_batch_ctx.restype = _C.c_void_p
        L.dsv1_batch_ctx.argtypes = [_C.c_void_p]
        L.dsv_free.argtypes = [_C.c_void_p]
        L.estimate_bitrate.restype = _C.c_uint
        L.estimate_bitrate.argtypes = [_C.c_int, _C.c_int, _C.POINTER(Meta)]
        L.dsvg_dev_alloc.argtypes = [_C.c_void_p, _C.POINTER(_C.c_void_p), _C.c_size_t]
        L.dsvg_dev_free.argtypes = [_C.c_void_p, _C.c_void_p]
        L.dsvg_dev_upload.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_size_t]
        L.dsvg_ctx_sync.argtypes = [_C.c_void_p]
        L.dsvg_ctx_code_streams.argtypes = [_C.c_void_p, _C.c_int]
        L.dsvg_ctx_streams_apart.argtypes = [_C.c_void_p]
        L.dsvg_ctx_copy_queue.argtypes = [_C.c_void_p]
        L.dsvg_ctx_stream.restype = _C.c_void_p
        L.dsvg_ctx_stream.argtypes = [_C.c_void_p]
        L.dsvg_ctx_join.argtypes = [_C.c_void_p, _C.c_void_p]
        L.dsvg_ctx_tile_stats.argtypes = [_C.c_void_p, _C.POINTER(_C.c_ulonglong), _C.c_int]
        L.dsvg_ctx_tile_stats2.argtypes = [_C.c_void_p, _C.POINTER(_C.c_ulonglong), _C.c_int]
        L.dsvg_dev_download.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_size_t]
        L.dsvg_host_alloc.argtypes = [_C.c_void_p, _C.POINTER(_C.c_void_p), _C.c_size_t]
        L.dsvg_host_free.argtypes = [_C.c_void_p, _C.c_void_p]
        L.dsv1_batch_stage.argtypes = [_C.c_void_p, _C.c_void_p]
        L.dsv1_decbatch_open.argtypes = [_C.POINTER(_C.c_void_p), _C.c_int, _C.POINTER(Meta), _C.c_int]
        L.dsv1_decbatch_decode.argtypes = [_C.c_void_p, _C.POINTER(Buf), _C.c_void_p, _C.c_size_t, _C.c_int,
                                           _C.POINTER(_C.c_int), _C.POINTER(_C.c_uint32)]
        L.dsv1_decbatch_close.argtypes = [_C.c_void_p]
        L.dsv1_decbatch_ctx.restype = _C.c_void_p
        L.dsv1_decbatch_ctx.argtypes = [_C.c_void_p]
        L.dsvg_prof_enable.argtypes = [_C.c_void_p, _C.c_ulonglong]
        L.dsvg_ctx_mark.argtypes = [_C.c_void_p, _C.c_int]
        L.dsvg_ctx_mark_ms.argtypes = [_C.c_void_p, _C.POINTER(_C.c_float)]
        L.dsvg_ctx_timeline.argtypes = [_C.c_void_p, _C.c_int]
        L.dsvg_ctx_timeline_get.argtypes = [_C.c_void_p, _C.POINTER(_C.c_double)]
        L.dsvg_ctx_fetch_prof.argtypes = [_C.c_void_p, _C.POINTER(_C.c_double), _C.c_int]
        L.dsv1_host_prof_enable.argtypes = [_C.c_int]
        L.dsv1_host_prof_enable.restype = None
        L.dsv1_host_prof_get.argtypes = [_C.POINTER(_C.c_double), _C.c_int, _C.POINTER(_C.c_long)]
        L.dsv1_host_prof_name.restype = _C.c_char_p
        L.dsv1_host_prof_name.argtypes = [_C.c_int]
        L.dsvg_link_probe.argtypes = [_C.c_int, _C.c_size_t, _C.c_int, _C.POINTER(_C.c_double)]
        L.dsv1_batch_encoder.restype = _C.c_void_p
        L.dsv1_batch_encoder.argtypes = [_C.c_void_p, _C.c_int]
        L.dsvg_prof_reset.argtypes = [_C.c_void_p]
        L.dsvg_prof_get.argtypes = [_C.c_void_p, _C.c_int, _C.POINTER(_C.c_double), _C.POINTER(_C.c_long),
                                    _C.POINTER(_C.c_double)]
        L.dsvg_prof_kernel_name.restype = _C.c_char_p
        _lib = L
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed rc=%d: %s" % (what, rc, lib().dsvg_last_error().decode()))


def make_encoder_cfg(w, h, fmt, qp=85, gop=12, rc_mode_cli=1, kbps=0, scd=1, ipct=50, pyrlevels=0, stabref=0,
                     fps_num=30, fps_den=1):
    """fill a DSV_ENCODER exactly like the reference CLI does for these flags (dsv_main.c:423-489):
    rc_mode_cli 0 = ABR, 1 = CRF (CLI numbering); kbps 0 = auto; stabref 0 = auto"""
    L = lib()
    e = Encoder()
    L.dsv_enc_init(_C.byref(e))
    e.vidmeta = Meta(w, h, fmt, fps_num, fps_den, 1, 1)
    e.gop = gop
    e.scene_change_delta = 4
    e.do_scd = scd
    e.intra_pct_thresh = ipct
    e.quality = MAX_QUALITY * qp // 100
    e.rc_mode = 0 if rc_mode_cli == 1 else 1
    e.bitrate = kbps * 1024 if kbps else L.estimate_bitrate(e.quality * 100 // MAX_QUALITY, gop, _C.byref(e.vidmeta))
    if e.rc_mode == 1:
        e.quality = min(max(e.quality * 3 // 2, 0), MAX_QUALITY)
    e.max_q_step = MAX_QUALITY // 200
    e.min_quality = MAX_QUALITY * 1 // 100
    e.max_quality = MAX_QUALITY
    e.min_I_frame_quality = MAX_QUALITY * 5 // 100
    e.rc_high_motion_nudge = 1
    e.pyramid_levels = pyrlevels
    e.stable_refresh = stabref if stabref else min(max(gop - 1, 1), 14)
    return e


def _take(buf):
    data = _C.string_at(buf.data, buf.len) if buf.len else b""
    if buf.data:
        lib().dsv_free(_C.cast(buf.data, _C.c_void_p))
    return data


class StreamBytes:
    """The finished packets of one stream exactly where the library assembled them (a dsv_alloc'd host buffer): no copy
    into a Python bytes object.  len(), bytes(), memoryview() work; the buffer is freed with the object."""

    def __init__(self, buf):
        self._ptr = _C.cast(buf.data, _C.c_void_p).value
        self._len = int(buf.len) if self._ptr else 0

    def __len__(self):
        return self._len

    def __bytes__(self):
        return _C.string_at(self._ptr, self._len) if self._len else b""

    def view(self):
        return memoryview((_C.c_ubyte * self._len).from_address(self._ptr)) if self._len else memoryview(b"")

    def __eq__(self, other):
        return bytes(self) == bytes(other)

    def __del__(self):
        if getattr(self, "_ptr", None):
            try:
                lib().dsv_free(_C.c_void_p(self._ptr))
            except Exception:      # interpreter shutdown
                pass
            self._ptr = None


class Batch:
    """nstreams independent encoder streams, frames_per_call frames each per encode() call"""

    def __init__(self, cfg, nstreams, frames_per_call, device=0, chains=0):
        """chains > 0: ONE stream in chain mode (dsv1_stream_open): frames_per_call consecutive frames per call, the chains of
        pictures between I pictures coded side by side"""
        self.L = lib()
        self.h = _C.c_void_p(None)
        self.nstreams, self.F = nstreams, frames_per_call
        if chains:
            assert nstreams == 1
            _chk(self.L.dsv1_stream_open(_C.byref(self.h), _C.byref(cfg), device, frames_per_call, chains), "dsv1_stream_open")
        else:
            _chk(self.L.dsv1_batch_open(_C.byref(self.h), _C.byref(cfg), device, nstreams, frames_per_call), "dsv1_batch_open")
        self.ctx = self.L.dsv1_batch_ctx(self.h)
        m = cfg.vidmeta
        self.frame_bytes = m.width * m.height + 2 * _chroma_size(m.width, m.height, m.subsamp)
        self._dev = []
        self._pin = []

    def set_fnum(self, stream, fnum):
        self.L.dsv1_batch_set_fnum(self.h, stream, fnum)

    def dropped_recons(self):
        """(dropped, remedied): reference pictures coded without a reconstruction because nobody predicts from them / coded again
        because a renumbered stream did after all (include/dsv1_api.h, dsv1_batch_dropped_recons)"""
        r = _C.c_long(0)
        n = self.L.dsv1_batch_dropped_recons(self.h, _C.byref(r))
        return int(n), int(r.value)

    def recon_all(self, on=True):
        """reconstruct every reference picture (on) / drop the ones nobody predicts from (off, the default); between batches"""
        _chk(self.L.dsv1_batch_recon_all(self.h, 1 if on else 0), "dsv1_batch_recon_all")

    def encoder(self, stream):
        """the stream's DSV_ENCODER (owned by the batch): its public parameter fields may be changed between submits"""
        p = self.L.dsv1_batch_encoder(self.h, stream)
        if not p:
            raise IndexError("no stream %d" % stream)
        return Encoder.from_address(p)

    def upload(self, clip):
        """keep a raw clip (numpy uint8, any shape) resident in HBM; returns the device pointer"""
        a = _np.ascontiguousarray(clip, dtype=_np.uint8)
        p = _C.c_void_p(None)
        _chk(self.L.dsvg_dev_alloc(self.ctx, _C.byref(p), a.nbytes), "dsvg_dev_alloc")
        _chk(self.L.dsvg_dev_upload(self.ctx, p, a.ctypes.data, a.nbytes), "dsvg_dev_upload")
        self._dev.append(p)
        return p

    def pinned(self, shape):
        """uint8 numpy array in pinned host memory (dsvg_host_alloc): uploads from it are asynchronous"""
        n = int(_np.prod(shape))
        p = _C.c_void_p(None)
        _chk(self.L.dsvg_host_alloc(self.ctx, _C.byref(p), n), "dsvg_host_alloc")
        self._pin.append(p)
        return _np.ctypeslib.as_array(_C.cast(p, _C.POINTER(_C.c_uint8)), shape=(n,)).reshape(shape)

    def stage(self, yuv):
        """queue the upload of the host clip of a coming submit() (up to two; submit() must get the same memory)"""
        assert yuv.dtype == _np.uint8 and yuv.flags["C_CONTIGUOUS"]
        self._keep_staged = (getattr(self, "_keep_staged", []) + [yuv])[-3:]
        _chk(self.L.dsv1_batch_stage(self.h, yuv.ctypes.data), "dsv1_batch_stage")

    def encode(self, yuv, on_device=False, eos=False):
        """yuv: numpy [nstreams][F][frame_bytes] (host) or a device pointer from upload();
        returns one bytes object per stream"""
        bufs = (Buf * self.nstreams)()
        if on_device:
            ptr = yuv
        else:
            a = _np.ascontiguousarray(yuv, dtype=_np.uint8)
            if a.size != self.nstreams * self.F * self.frame_bytes:      # (the C entry point reads nstreams x F frames whatever it is given)
                raise ValueError("a batch is %d streams x %d frames x %d bytes, got %d bytes" % (self.nstreams, self.F, self.frame_bytes, a.size))
            ptr = a.ctypes.data
        _chk(self.L.dsv1_batch_encode(self.h, ptr, 1 if on_device else 0, bufs), "dsv1_batch_encode")
        if eos:
            for s in range(self.nstreams):
                _chk(self.L.dsv1_batch_eos(self.h, s, _C.byref(bufs[s])), "dsv1_batch_eos")
        return [_take(bufs[s]) for s in range(self.nstreams)]

    def submit(self, yuv, on_device=False, held=False):
        """pipelined form: enqueue one batch (returns while its residual coding still runs on the GPU).
        At most two batches may be in flight: steady state is submit(i+1); collect(i).
        Device clips: by default (the contract of dsv1_api.h's plain yuv_on_device = 1) the clip is copied whole and may change
        as soon as submit() returns; held=True (DSV1_CLIP_HELD, what bench.py times): the caller keeps the clip unchanged until
        collect() of this batch returned -- its chroma is then read in place"""
        if on_device:
            ptr = yuv
        else:
            self._keep = _np.ascontiguousarray(yuv, dtype=_np.uint8)
            if self._keep.size != self.nstreams * self.F * self.frame_bytes:
                raise ValueError("a batch is %d streams x %d frames x %d bytes, got %d bytes" % (self.nstreams, self.F, self.frame_bytes, self._keep.size))
            ptr = self._keep.ctypes.data
        if not hasattr(self, "_abr"):
            self._abr = []
        bufs = (Buf * self.nstreams)()
        _chk(self.L.dsv1_batch_submit(self.h, ptr, (2 if held else 1) if on_device else 0, bufs), "dsv1_batch_submit")
        self._abr.append(bufs)

    def collect(self, copy=True):
        """fetch + assemble the oldest submitted batch -> one bytes object per stream (copy=False: StreamBytes
        objects that keep the packets in the buffers the library assembled them in)"""
        bufs = self._abr.pop(0)
        _chk(self.L.dsv1_batch_collect(self.h, bufs), "dsv1_batch_collect")
        if not copy:
            return [StreamBytes(bufs[s]) for s in range(self.nstreams)]
        return [_take(bufs[s]) for s in range(self.nstreams)]

    def sync(self):
        _chk(self.L.dsvg_ctx_sync(self.ctx), "dsvg_ctx_sync")

    def code_streams(self, n=0):
        """set (n >= 1) / query (n = 0) the number of coding streams; returns the previous value"""
        return self.L.dsvg_ctx_code_streams(self.ctx, n)

    def tile_stats(self, enable=True):
        """inverse-transform tiles of P pictures since the previous call: dict general_luma / general_chroma (computed)
        and zero_luma / zero_chroma (found empty: reconstruction = prediction); counting continues only if `enable`.
        Syncs and clears the counters."""
        v = (_C.c_ulonglong * 8)()
        _chk(self.L.dsvg_ctx_tile_stats2(self.ctx, v, 1 if enable else 0), "dsvg_ctx_tile_stats2")
        return {"general_luma": v[0], "general_chroma": v[1], "zero_luma": v[2], "zero_chroma": v[3],
                "flagged_patches_luma": v[4], "flagged_patches_chroma": v[5], "moved_unflagged_patches_chroma": v[6],
                "fused_border_bytes": v[7]}

    def kernel_names(self):
        return [self.L.dsvg_prof_kernel_name(i).decode() for i in range(self.L.dsvg_prof_kernels())]

    def mark(self, which):
        """HIP-event mark on the first coding stream (0: a timed region starts, 1: behind its last enqueued coding work)"""
        _chk(self.L.dsvg_ctx_mark(self.ctx, which), "dsvg_ctx_mark")

    def mark_ms(self):
        ms = _C.c_float(0)
        _chk(self.L.dsvg_ctx_mark_ms(self.ctx, _C.byref(ms)), "dsvg_ctx_mark_ms")
        return ms.value

    def breakdown_start(self):
        """start the per-step accounting of a timed loop: host phases of submit / collect (process-wide), the fetch's host side and the
        device-side marks of the pipeline's streams (dsvg_ctx_timeline)"""
        self.L.dsv1_host_prof_enable(1)
        v = (_C.c_double * 5)()
        _chk(self.L.dsvg_ctx_fetch_prof(self.ctx, v, 1), "dsvg_ctx_fetch_prof")
        _chk(self.L.dsvg_ctx_timeline(self.ctx, 1), "dsvg_ctx_timeline")

    def breakdown_stop(self, steps):
        """-> dict of ms per step (call after sync()): where the host thread spent a step, what the pipeline's streams did meanwhile"""
        n = self.L.dsv1_host_prof_get(None, 0, None)
        ms = (_C.c_double * n)()
        nb = _C.c_long(0)
        self.L.dsv1_host_prof_get(ms, n, _C.byref(nb))
        f = (_C.c_double * 5)()
        _chk(self.L.dsvg_ctx_fetch_prof(self.ctx, f, 1), "dsvg_ctx_fetch_prof")
        t = (_C.c_double * 12)()
        _chk(self.L.dsvg_ctx_timeline_get(self.ctx, t), "dsvg_ctx_timeline_get")
        _chk(self.L.dsvg_ctx_timeline(self.ctx, 0), "dsvg_ctx_timeline")
        self.L.dsv1_host_prof_enable(0)
        k = float(max(steps, 1))
        host = {self.L.dsv1_host_prof_name(i).decode(): round(ms[i], 3) for i in range(n)}
        nph = max(t[0], 1.0)
        return {"host_ms_per_batch": host, "host_batches": int(nb.value),
                "fetch_host_ms_per_call": {"wait_for_coding": round(f[0], 3), "sizes_round_trip": round(f[1], 3), "gather_copy_assembly": round(f[2], 3)},
                "fetch_bytes_per_call": int(f[3]), "fetch_calls": int(f[4]),
                "device_ms_per_batch": {"coding_phases_seen": int(t[0]), "clip_upload": round(t[2] / nph, 3), "load_pyramid": round(t[3] / nph, 3), "motion_search": round(t[4] / nph, 3),
                                        "table_uploads": round(t[5] / nph, 3), "coding_stream0": round(t[6] / nph, 3), "coding_stream1": round(t[7] / nph, 3),
                                        "fetch_gather_copy": round(t[8] / nph, 3), "coding_overlapped_by_load_or_search": round(t[11] / nph, 3)},
                "device_span_ms": round(t[1], 3),
                "device_idle_ms_per_batch": round(t[10] / nph, 3),
                "device_idle_note": "device time inside [first coding start, last coding end] with no load / motion-search / table-upload / coding phase in flight on any pipeline stream: the chip waiting for the host"}

    def prof_enable(self, kernels):
        """kernels: iterable of kernel names whose launches get HIP-event brackets (empty = off)"""
        names = self.kernel_names()
        mask = 0
        for k in kernels:
            mask |= 1 << names.index(k)
        _chk(self.L.dsvg_prof_enable(self.ctx, mask), "dsvg_prof_enable")
        _chk(self.L.dsvg_prof_reset(self.ctx), "dsvg_prof_reset")

    def prof_get(self, kernel):
        """(total ms, launches, algorithmic bytes) of one kernel since the last prof_enable"""
        kid = self.kernel_names().index(kernel)
        ms, n, by = _C.c_double(0), _C.c_long(0), _C.c_double(0)
        _chk(self.L.dsvg_prof_get(self.ctx, kid, _C.byref(ms), _C.byref(n), _C.byref(by)), "dsvg_prof_get")
        return ms.value, n.value, by.value

    def close(self):
        if self.h:
            for p in self._dev:
                self.L.dsvg_dev_free(self.ctx, p)
            self._dev = []
            self.L.dsvg_ctx_sync(self.ctx)
            for p in self._pin:
                self.L.dsvg_host_free(self.ctx, p)
            self._pin = []
            self.L.dsv1_batch_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecBatch:
    """nstreams independent streams of one geometry, one packet per stream per decode() call (dsv1_decbatch_*)"""

    def __init__(self, w, h, fmt, nstreams, device=0):
        self.L = lib()
        self.h = _C.c_void_p(None)
        self.nstreams = nstreams
        m = Meta()
        m.width, m.height, m.subsamp = w, h, fmt
        _chk(self.L.dsv1_decbatch_open(_C.byref(self.h), device, _C.byref(m), nstreams), "dsv1_decbatch_open")
        self.frame_bytes = w * h + 2 * _chroma_size(w, h, fmt)
        self._dev = None

    @property
    def ctx(self):
        """the device context (not cached: the batch builds a new one when its streams announce another block size)"""
        return self.L.dsv1_decbatch_ctx(self.h)

    def decode(self, packets, out=None, on_device=False):
        """packets: one bytes object per stream.  Host output: returns (frames [nstreams][frame_bytes] uint8, status,
        fnum); on_device=True leaves the frames in a device buffer owned by this object (returns its pointer)."""
        S = self.nstreams
        assert len(packets) == S
        keep = [_np.frombuffer(bytes(p) + b"\0" * 16, dtype=_np.uint8).copy() for p in packets]
        bufs = (Buf * S)()
        for s in range(S):
            bufs[s].data = keep[s].ctypes.data_as(_C.POINTER(_C.c_uint8))
            bufs[s].len = len(packets[s])
        status = (_C.c_int * S)()
        fnum = (_C.c_uint32 * S)()
        if on_device and out is not None:
            dst = out                                  # a device buffer of the caller's (nstreams x frame_bytes: dev_alloc())
        elif on_device:
            if self._dev is None:
                self._dev = _C.c_void_p(None)
                _chk(self.L.dsvg_dev_alloc(self.ctx, _C.byref(self._dev), self.frame_bytes * S), "dsvg_dev_alloc")
            dst = self._dev
        else:
            if out is None:
                out = _np.zeros((S, self.frame_bytes), dtype=_np.uint8)
            dst = out.ctypes.data
        _chk(self.L.dsv1_decbatch_decode(self.h, bufs, dst, self.frame_bytes, 1 if on_device else 0, status, fnum), "dsv1_decbatch_decode")
        return (dst if on_device else out), list(status), list(fnum)

    def sync(self):
        _chk(self.L.dsvg_ctx_sync(self.ctx), "dsvg_ctx_sync")

    def dev_alloc(self):
        """a device buffer for one call's frames (decode(.., out=buffer, on_device=True)); freed with the context"""
        p = _C.c_void_p(None)
        _chk(self.L.dsvg_dev_alloc(self.ctx, _C.byref(p), self.frame_bytes * self.nstreams), "dsvg_dev_alloc")
        return p

    def download(self, dev=None):
        """host copy of the device output buffer of the last decode(on_device=True), or of `dev`"""
        self.sync()
        out = _np.zeros((self.nstreams, self.frame_bytes), dtype=_np.uint8)
        _chk(self.L.dsvg_dev_download(self.ctx, out.ctypes.data, dev if dev is not None else self._dev, out.nbytes), "dsvg_dev_download")
        return out

    def close(self):
        if self.h:
            self.L.dsvg_ctx_sync(self.ctx)
            if self._dev is not None:
                self.L.dsvg_dev_free(self.ctx, self._dev)
                self._dev = None
            self.L.dsv1_decbatch_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _chroma_size(w, h, fmt):
    """chroma plane samples for a DSV_SUBSAMP_* format code (dsv.h:62-75: bits 2-3 horizontal, 0-1 vertical shift)"""
    hs, vs = (fmt >> 2) & 3, fmt & 3
    return ((w + (1 << hs) - 1) >> hs) * ((h + (1 << vs) - 1) >> vs)


def encode_clip(clip, w, h, fmt, device=0, eos=True, start_fnum=0, **cli):
    """one serial stream on the GPU: the whole clip in one batch call -> .dsv bytes"""
    cfg = make_encoder_cfg(w, h, fmt, **cli)
    n = clip.shape[0]
    b = Batch(cfg, 1, n, device)
    try:
        if start_fnum:
            b.set_fnum(0, start_fnum)
        return b.encode(clip.reshape(1, n, -1), eos=eos)[0]
    finally:
        b.close()


def encode_stream(clip, w, h, fmt, frames_per_call, chains, device=0, eos=True, **cli):
    """one stream through the GOP-parallel chain mode (dsv1_stream_open): the clip in calls of frames_per_call frames
    (clip.shape[0] must be a multiple), two calls in flight -> .dsv bytes, byte for byte the serial encoder's"""
    cfg = make_encoder_cfg(w, h, fmt, **cli)
    n = clip.shape[0]
    assert n % frames_per_call == 0
    b = Batch(cfg, 1, frames_per_call, device, chains=chains)
    try:
        out = b""
        calls = [clip[i:i + frames_per_call].reshape(1, frames_per_call, -1) for i in range(0, n, frames_per_call)]
        b.submit(calls[0])
        for k in range(1, len(calls)):
            b.submit(calls[k])
            out += bytes(b.collect()[0])
        out += bytes(b.collect()[0])
        if eos:
            e = Buf()
            _chk(b.L.dsv1_batch_eos(b.h, 0, _C.byref(e)), "dsv1_batch_eos")
            out += _take(e)
        return out
    finally:
        b.close()


def concat_gops(streams):
    """join independently encoded closed GOPs exactly as one serial encode would have linked them"""
    L = lib()
    arr = (Buf * len(streams))()
    keep = []
    for i, s in enumerate(streams):
        a = _np.frombuffer(s, dtype=_np.uint8).copy()
        keep.append(a)
        arr[i].data = a.ctypes.data_as(_C.POINTER(_C.c_uint8))
        arr[i].len = a.size
    out = Buf()
    _chk(L.dsv1_concat_gops(arr, len(streams), _C.byref(out)), "dsv1_concat_gops")
    return _take(out)


def encode_gops(clip, w, h, fmt, gop, device=0, **cli):
    """GOP-sharded encode: clip (N frames, N % gop == 0) -> N/gop independent closed GOPs in ONE batch,
    each seeded with its frame number, then joined.  Bit-exact with the serial stream when GOPs are
    independent (CRF, stable_refresh == gop-1, no forced-intra P frames: SURVEY.md 8e)."""
    n = clip.shape[0]
    assert n % gop == 0
    g = n // gop
    cfg = make_encoder_cfg(w, h, fmt, gop=gop, **cli)
    b = Batch(cfg, g, gop, device)
    try:
        for s in range(g):
            b.set_fnum(s, s * gop)
        parts = b.encode(clip.reshape(g, gop, -1))
    finally:
        b.close()
    return concat_gops(parts)
