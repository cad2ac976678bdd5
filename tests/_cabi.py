"""ctypes mirrors of the DSV1 C ABI shared by the three libraries the tests drive:

  * oracle/_ref/libdsv1ref.so  -- the real reference, compiled from /root/reference (checker)
  * oracle/liborc.so           -- our scalar C restatement (checker / CPU baseline)
  * digital-subband-video-1_amd/libdsv1_mi355x.so -- THE PRODUCT (HIP kernels behind the C ABI)

Struct layouts follow dsv.h:86-198, dsv_internal.h:39-49, dsv_encoder.h:124-130 of the reference.
"""
import ctypes as C
import os
import subprocess
import hashlib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libdsv1ref.so")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "dsv1")
ORC_SO = os.path.join(ROOT, "oracle", "liborc.so")
CLIPGEN_SO = os.path.join(ROOT, "tools", "clipgen", "libclipgen.so")
PKG_DIR = os.path.join(ROOT, "digital-subband-video-1_amd")
PROD_SO = os.path.join(PKG_DIR, "libdsv1_mi355x.so")

SUBSAMP_444, SUBSAMP_422, SUBSAMP_420, SUBSAMP_411 = 0x0, 0x4, 0x5, 0x8
BORDER = 64


class Meta(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("subsamp", C.c_int),
                ("fps_num", C.c_int), ("fps_den", C.c_int),
                ("aspect_num", C.c_int), ("aspect_den", C.c_int)]


class Plane(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_int), ("format", C.c_int),
                ("stride", C.c_int), ("w", C.c_int), ("h", C.c_int), ("hs", C.c_int), ("vs", C.c_int)]


class Coefs(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_int32)), ("width", C.c_int), ("height", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("alloc", C.POINTER(C.c_uint8)), ("planes", Plane * 3), ("refcount", C.c_int),
                ("format", C.c_int), ("width", C.c_int), ("height", C.c_int), ("border", C.c_int)]


class _MVxy(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16)]


class _MVu(C.Union):
    _fields_ = [("mv", _MVxy), ("all", C.c_int32)]


class MV(C.Structure):
    _fields_ = [("u", _MVu), ("mode", C.c_uint8), ("submask", C.c_uint8),
                ("lo_var", C.c_uint8), ("lo_tex", C.c_uint8), ("high_detail", C.c_uint8)]


assert C.sizeof(MV) == 12

MV_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("mode", "u1"), ("submask", "u1"),
                     ("lo_var", "u1"), ("lo_tex", "u1"), ("high_detail", "u1"), ("pad", "u1", (3,))])
assert MV_DTYPE.itemsize == 12


class Params(C.Structure):
    _fields_ = [("vidmeta", C.POINTER(Meta)), ("is_ref", C.c_int), ("has_ref", C.c_int),
                ("blk_w", C.c_int), ("blk_h", C.c_int), ("nblocks_h", C.c_int), ("nblocks_v", C.c_int)]


class Stability(C.Structure):
    _fields_ = [("params", C.POINTER(Params)), ("stable_blocks", C.POINTER(C.c_uint8)),
                ("cur_plane", C.c_uint8), ("isP", C.c_uint8)]


class BS(C.Structure):
    _fields_ = [("start", C.POINTER(C.c_uint8)), ("pos", C.c_uint)]


class HME(C.Structure):
    _fields_ = [("params", C.POINTER(Params)), ("src", C.POINTER(Frame) * 6), ("ref", C.POINTER(Frame) * 6),
                ("mvf", C.POINTER(MV) * 6), ("levels", C.c_int)]


class Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_uint)]


class EncCfg(C.Structure):          # orc_enc_cfg / dsvg_enc_cfg
    _fields_ = [("meta", Meta), ("quality", C.c_int), ("gop", C.c_int), ("do_scd", C.c_int),
                ("rc_mode", C.c_int), ("rc_high_motion_nudge", C.c_int), ("bitrate", C.c_uint),
                ("max_q_step", C.c_int), ("min_quality", C.c_int), ("max_quality", C.c_int),
                ("min_I_frame_quality", C.c_int), ("intra_pct_thresh", C.c_int),
                ("scene_change_delta", C.c_int), ("stable_refresh", C.c_uint), ("pyramid_levels", C.c_int)]


def u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def rshift_up(x, s):
    return (x + (1 << s) - 1) >> s


def hshift(fmt):
    return (fmt >> 2) & 3


def vshift(fmt):
    return fmt & 3


def chroma_dims(w, h, fmt):
    return rshift_up(w, hshift(fmt)), rshift_up(h, vshift(fmt))


def coef_dims(w, h, fmt, c):
    if c == 0:
        return w, h
    cw, ch = chroma_dims(w, h, fmt)
    return (cw + 1) & ~1, (ch + 1) & ~1


def frame_bytes(w, h, fmt):
    cw, ch = chroma_dims(w, h, fmt)
    return w * h + 2 * cw * ch


def block_dims(w, h):
    """size4dim + clamp, dsv_encoder.c:556-595"""
    def s4(d):
        s = 64 if d > 1280 else 48 if d > 1024 else 32 if d > 704 else 24 if d > 352 else 16
        return min(max(s & ~7, 16), 64)
    bw, bh = s4(w), s4(h)
    return bw, bh, (w + bw - 1) // bw, (h + bh - 1) // bh


class BorderedFrame:
    """numpy-owned frame with exactly the reference layout (frame.c:63-120): one zeroed allocation,
    Y,U,V back to back, 64-px border, stride round16(w+128); a zeroed guard page in front."""

    GUARD = 8192

    def __init__(self, w, h, fmt, border=True):
        self.w, self.h, self.fmt = w, h, fmt
        ext = BORDER if border else 0
        cw, ch = chroma_dims(w, h, fmt)
        self.dims = [(w, h), (cw, ch), (cw, ch)]
        self.strides = [((d[0] + 2 * ext + 15) & ~15) for d in self.dims]
        self.lens = [s * (d[1] + 2 * ext) for s, d in zip(self.strides, self.dims)]
        self.buf = np.zeros(self.GUARD + sum(self.lens) + 4096, dtype=np.uint8)
        self.ext = ext
        self.c = Frame()
        base = self.buf.ctypes.data + self.GUARD
        self.c.alloc = C.cast(base, C.POINTER(C.c_uint8))
        self.c.refcount = 1
        self.c.format = fmt
        self.c.width, self.c.height, self.c.border = w, h, 1 if border else 0
        off = 0
        self.offs = []
        for i in range(3):
            p = self.c.planes[i]
            p.format = fmt
            p.w, p.h = self.dims[i]
            p.stride = self.strides[i]
            p.len = self.lens[i]
            p.hs = hshift(fmt) if i else 0
            p.vs = vshift(fmt) if i else 0
            o = off + self.strides[i] * ext + ext
            self.offs.append(self.GUARD + o)
            p.data = C.cast(base + o, C.POINTER(C.c_uint8))
            off += self.lens[i]
        self.total = off

    def plane(self, i):
        """interior view (h, w) of plane i"""
        w, h = self.dims[i]
        s = self.strides[i]
        o = self.offs[i]
        return np.lib.stride_tricks.as_strided(self.buf[o:], shape=(h, w), strides=(s, 1))

    def raw(self):
        """whole allocation incl. borders (what must match byte for byte)"""
        return self.buf[self.GUARD:self.GUARD + self.total]

    def load_planar(self, yuv):
        o = 0
        for i in range(3):
            w, h = self.dims[i]
            self.plane(i)[:, :] = yuv[o:o + w * h].reshape(h, w)
            o += w * h

    def to_planar(self):
        return np.concatenate([self.plane(i).reshape(-1) for i in range(3)])

    def ptr(self):
        return C.byref(self.c)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


_libs = {}


def have_ref():
    return os.path.exists(REF_SO)


def load_ref():
    if "ref" not in _libs:
        _libs["ref"] = C.CDLL(REF_SO)
        L = _libs["ref"]
        L.dsv_get_quant.restype = C.c_int
        L.dsv_hme.restype = C.c_int
        L.dsv_alloc.restype = C.c_void_p
        L.dsv_free.argtypes = [C.c_void_p]
    return _libs["ref"]


def load_orc():
    if "orc" not in _libs:
        if not os.path.exists(ORC_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"],
                                  stdout=subprocess.DEVNULL)
        L = C.CDLL(ORC_SO)
        L.orc_enc_open.restype = C.c_void_p
        L.orc_enc_open.argtypes = [C.POINTER(EncCfg)]
        L.orc_enc_frame.restype = C.c_size_t
        L.orc_enc_frame.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_size_t), C.c_void_p]
        L.orc_enc_eos.restype = C.c_size_t
        L.orc_enc_eos.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_enc_close.argtypes = [C.c_void_p]
        L.orc_enc_set_next_fnum.argtypes = [C.c_void_p, C.c_uint]
        L.orc_enc_set_params.argtypes = [C.c_void_p, C.POINTER(EncCfg)]
        L.orc_enc_force_metadata.argtypes = [C.c_void_p]
        L.orc_enc_last_mvs.restype = C.POINTER(MV)
        L.orc_enc_last_mvs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.orc_enc_last_stable.restype = C.POINTER(C.c_uint8)
        L.orc_enc_last_stable.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.orc_dec_open.restype = C.c_void_p
        L.orc_dec_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint)]
        L.orc_dec_close.argtypes = [C.c_void_p]
        L.orc_dec_get_meta.argtypes = [C.c_void_p, C.POINTER(Meta)]
        L.orc_hme_run.restype = C.c_int
        L.orc_get_quant.restype = C.c_int
        _libs["orc"] = L
    return _libs["orc"]


def load_prod():
    """the product library; fails loudly if it is missing (no fallback)"""
    if "prod" not in _libs:
        if not os.path.exists(PROD_SO):
            raise RuntimeError("product library %s is not built (run python -c 'import __graft_entry__ as g; g.build()')" % PROD_SO)
        L = C.CDLL(PROD_SO)
        L.dsvg_last_error.restype = C.c_char_p
        L.dsvg_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.dsvg_ctx_destroy.argtypes = [C.c_void_p]
        L.dsv_free.argtypes = [C.c_void_p]
        L.dsv_free.restype = None
        _libs["prod"] = L
    return _libs["prod"]


def chk(L, rc):
    if rc != 0:
        raise RuntimeError("dsvg call failed rc=%d: %s" % (rc, L.dsvg_last_error().decode()))


def assert_same(name, got, want, shape=None):
    """bit-exact comparison with a useful report"""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    if np.array_equal(got, want):
        return
    g = got.reshape(-1)
    w = want.reshape(-1)
    bad = np.nonzero(g != w)[0]
    msg = ["%s: %d of %d elements differ" % (name, bad.size, g.size)]
    for i in bad[:12]:
        where = str(np.unravel_index(i, shape)) if shape is not None else str(i)
        msg.append("  at %s: got %s want %s" % (where, g[i], w[i]))
    if shape is not None and bad.size:
        ys, xs = np.unravel_index(bad, shape)
        msg.append("  bbox rows %d..%d cols %d..%d" % (ys.min(), ys.max(), xs.min(), xs.max()))
    raise AssertionError("\n".join(msg))


def load_clipgen():
    """the synthetic clip generator (tools/clipgen: neither product nor oracle)"""
    if "clipgen" not in _libs:
        src = os.path.join(ROOT, "tools", "clipgen", "clipgen.c")
        # (re)built when missing or OLDER than its source: a stale generator would feed the tests other inputs than the goldens were
        # made from, with no hint why (advisor, round 3).  The GPU box has gcc too.
        if not os.path.exists(CLIPGEN_SO) or os.path.getmtime(CLIPGEN_SO) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", CLIPGEN_SO, src])
        L = C.CDLL(CLIPGEN_SO)
        L.clipgen_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int]
        _libs["clipgen"] = L
    return _libs["clipgen"]


def gen_clip(w, h, fmt, seed, nframes, style=0, start=0):
    """synthetic clip (nframes, frame_bytes) via the integer generator tools/clipgen/clipgen.c
    (style 0 pan + texture, 1 + flat moving objects, 2 static + textured square + luma step, 3 style 0 with scene cuts)"""
    L = load_clipgen()
    fb = frame_bytes(w, h, fmt)
    out = np.empty((nframes, fb), dtype=np.uint8)
    for t in range(nframes):
        L.clipgen_frame(out[t].ctypes.data, w, h, fmt, seed, start + t, style)
    return out


def orc_cfg(w, h, fmt, qp=85, gop=12, rc_mode_cli=1, kbps=0, scd=1, ipct=50, pyrlevels=0, stabref=0):
    cfg = EncCfg()
    load_orc().orc_cfg_from_cli(C.byref(cfg), w, h, fmt, qp, gop, rc_mode_cli, kbps, scd, ipct, pyrlevels, stabref)
    return cfg


# a caller's changes of the encoder's public fields between two dsv_enc calls: {frame index: {field: value | "force_metadata": True}}
# (the same table drives the oracle, the reference library and the product: orc_encode, drive_dsv_enc)
PARAM_FIELDS = ("quality", "bitrate", "rc_high_motion_nudge", "max_q_step", "min_quality", "max_quality", "min_I_frame_quality")


def orc_encode(clip, cfg, want_recon=False, start_fnum=0, eos=True, changes=None):
    """encode a clip with the oracle session layer; returns (stream bytes, [recon frames])"""
    L = load_orc()
    e = L.orc_enc_open(C.byref(cfg))
    L.orc_enc_set_next_fnum(e, start_fnum)
    out = C.c_void_p(None)
    n = C.c_size_t(0)
    cap = C.c_size_t(0)
    recs = []
    cur = EncCfg.from_buffer_copy(cfg)
    for t in range(clip.shape[0]):
        ch = (changes or {}).get(t)
        if ch:
            for k, v in ch.items():
                if k != "force_metadata":
                    assert k in PARAM_FIELDS, k
                    setattr(cur, k, v)
            L.orc_enc_set_params(e, C.byref(cur))
            if ch.get("force_metadata"):
                L.orc_enc_force_metadata(e)
        rec = np.empty(clip.shape[1], dtype=np.uint8) if want_recon else None
        L.orc_enc_frame(e, clip[t].ctypes.data, C.byref(out), C.byref(n), C.byref(cap),
                        rec.ctypes.data if want_recon else None)
        if want_recon:
            recs.append(rec)
    if eos:
        L.orc_enc_eos(e, C.byref(out), C.byref(n), C.byref(cap))
    data = C.string_at(out.value, n.value)
    C.CDLL(None).free(out)
    L.orc_enc_close(e)
    return data, recs


def split_packets(stream):
    """split a .dsv byte string into packets using the next_link field (dsv_main.c:567-612)"""
    pk = []
    o = 0
    while o + 14 <= len(stream):
        assert stream[o:o + 4] == b"DSV1", "bad fourcc at %d" % o
        nxt = int.from_bytes(stream[o + 10:o + 14], "big")
        ln = nxt if nxt else 14
        pk.append(stream[o:o + ln])
        o += ln
    return pk


def orc_decode(stream, w, h, fmt):
    L = load_orc()
    d = L.orc_dec_open()
    fb = frame_bytes(w, h, fmt)
    frames = []
    for p in split_packets(stream):
        out = np.zeros(fb, dtype=np.uint8)
        fn = C.c_uint(0)
        rc = L.orc_dec_packet(d, p, len(p), out.ctypes.data, C.byref(fn))
        if rc == 0 and (p[5] & 4):
            frames.append(out)
    L.orc_dec_close(d)
    return frames


def ref_cli_encode(clip, w, h, fmt_cli, extra, tmpdir):
    """run the real reference CLI (oracle/_ref/dsv1 e ...) on a clip; returns the .dsv bytes"""
    inp = os.path.join(tmpdir, "in.yuv")
    outp = os.path.join(tmpdir, "out.dsv")
    clip.tofile(inp)
    cmd = [REF_CLI, "e", "-y", "-inp_" + inp, "-out_" + outp, "-w%d" % w, "-h%d" % h, "-fmt%d" % fmt_cli] + extra
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(outp, "rb") as f:
        return f.read()


def ref_cli_decode(stream, tmpdir):
    inp = os.path.join(tmpdir, "d.dsv")
    outp = os.path.join(tmpdir, "d.yuv")
    with open(inp, "wb") as f:
        f.write(stream)
    subprocess.run([REF_CLI, "d", "-y", "-inp_" + inp, "-out_" + outp], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return np.fromfile(outp, dtype=np.uint8)


FMT_CLI = {SUBSAMP_444: 0, SUBSAMP_422: 1, SUBSAMP_420: 2, SUBSAMP_411: 3}


def drive_dsv_enc(L, enc, clip, w, h, fmt, changes=None, flush_calls=False):
    """the reference's frame-at-a-time API (dsv_encoder.h:112-121) on library L (the product or oracle/_ref's libdsv1ref.so: same
    struct layout) driven the way dsv_main.c:506-537 does; `changes` rewrites the encoder's public fields between calls.
    Returns (stream bytes, buffers returned per call)."""
    L.dsv_load_planar_frame.restype = C.c_void_p
    L.dsv_load_planar_frame.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.dsv_enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.dsv_enc.restype = C.c_int
    L.dsv_enc_start(C.byref(enc))
    out = b""
    counts = []
    bufs = (Buf * 4)()
    scratch = np.empty_like(clip[0])

    def take(nb):
        nonlocal out
        for i in range(nb):
            out += C.string_at(bufs[i].data, bufs[i].len)
            L.dsv_buf_free(C.byref(bufs[i]))
    for t in range(clip.shape[0]):
        ch = (changes or {}).get(t)
        if ch:
            for k, v in ch.items():
                if k == "force_metadata":
                    L.dsv_enc_force_metadata(C.byref(enc))
                else:
                    assert k in PARAM_FIELDS, k
                    setattr(enc, k, v)
        scratch[...] = clip[t]                           # the caller reuses ONE picture buffer (dsv_main.c:506-520)
        frame = L.dsv_load_planar_frame(fmt, scratch.ctypes.data, w, h)
        nb = L.dsv_enc(C.byref(enc), frame, bufs) & 3
        counts.append(nb)
        take(nb)
        scratch[...] = 0xA5                              # ... and overwrites it right after the call
    if flush_calls:
        while True:
            nb = L.dsv_enc(C.byref(enc), None, bufs) & 3
            if not nb:
                break
            take(nb)
    L.dsv_enc_end_of_stream(C.byref(enc), bufs)
    take(1)
    L.dsv_enc_free(C.byref(enc))
    return out, counts
