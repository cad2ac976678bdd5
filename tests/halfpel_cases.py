"""Clips whose motion is known in HALF-PEL units: a smooth texture synthesised at twice the picture's resolution and sampled
at a per-frame offset, the left and the right half of the picture moving differently.  Every frame pair then has one
(horizontal phase, vertical phase) per half -- all four combinations for luma and for chroma over a clip -- and the waves of
the lean forward kernel (64 patches = 512 pixels of a row) see lanes with different phases side by side: the paths
hme.c:551-591 (half-pel search) -> bmc.c:58-174 (hpel / hpelL) -> k_fwd_mc_fast's packed filters."""
import ctypes as C

import numpy as np

import _cabi as A

# (left half step, right half step) per frame, in half-pel units (x, y)
STEPS = [((1, 0), (0, 1)), ((1, 1), (2, 0)), ((0, 1), (3, 2)), ((3, 0), (1, 1)), ((2, 1), (1, 3)), ((2, 2), (3, 1)), ((1, 0), (1, 0))]

CASES = [
    # w, h, fmt, seed, cli
    (704, 288, A.SUBSAMP_420, 11, dict(qp=85, gop=12, rc_mode_cli=1, scd=0)),
    (640, 240, A.SUBSAMP_444, 12, dict(qp=60, gop=12, rc_mode_cli=1, scd=0)),
    (768, 192, A.SUBSAMP_422, 13, dict(qp=90, gop=12, rc_mode_cli=1, scd=0)),
    (1280, 144, A.SUBSAMP_420, 14, dict(qp=75, gop=12, rc_mode_cli=1, scd=0)),
]


def _field(rng, W2, H2):
    yy, xx = np.mgrid[0:H2, 0:W2].astype(np.float64)
    f = np.zeros((H2, W2))
    for _ in range(10):
        fx, fy = rng.uniform(0.004, 0.09, 2) * rng.choice([-1, 1], 2)
        f += rng.uniform(10, 30) * np.sin(2 * np.pi * (fx * xx + fy * yy) + rng.uniform(0, 6.28))
    f += rng.normal(0, 2.0, f.shape)
    return np.clip(128 + f, 0, 255)


def halfpel_clip(w, h, fmt, seed, steps=STEPS):
    rng = np.random.default_rng(seed)
    n = len(steps) + 1
    offs = [np.cumsum(np.array([(0, 0)] + [s[k] for s in steps]), axis=0) for k in (0, 1)]
    pad = int(max(np.abs(o).max() for o in offs)) + 8
    cw, ch = A.chroma_dims(w, h, fmt)
    sx, sy = w // cw, h // ch
    fields = [_field(rng, 2 * w + 2 * pad, 2 * h + 2 * pad), _field(rng, 2 * cw + 2 * pad, 2 * ch + 2 * pad),
              _field(rng, 2 * cw + 2 * pad, 2 * ch + 2 * pad)]
    out = np.empty((n, A.frame_bytes(w, h, fmt)), np.uint8)
    for t in range(n):
        planes = []
        for c, (pw, ph) in enumerate([(w, h), (cw, ch), (cw, ch)]):
            f = fields[c]
            img = np.empty((ph, pw))
            for half in (0, 1):
                ox, oy = offs[half][t]
                if c:
                    ox, oy = int(round(ox / sx)), int(round(oy / sy))
                full = f[pad + oy:pad + oy + 2 * ph:2, pad + ox:pad + ox + 2 * pw:2]
                x0, x1 = (0, pw // 2) if half == 0 else (pw // 2, pw)
                img[:, x0:x1] = full[:, x0:x1]
            planes.append(np.rint(img).astype(np.uint8).ravel())
        out[t] = np.concatenate(planes)
    return out


def oracle_phases(clip, w, h, fmt, **cli):
    """(luma phase histogram [yh][xh], chroma phase histogram, intra blocks) of the oracle encoder's vectors"""
    cfg = A.orc_cfg(w, h, fmt, **cli)
    L = A.load_orc()
    e = L.orc_enc_open(C.byref(cfg))
    L.orc_enc_set_next_fnum(e, 0)
    out, n, cap = C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
    lum, chrm, intra = np.zeros((2, 2), int), np.zeros((2, 2), int), 0
    hs, vs = A.hshift(fmt), A.vshift(fmt)
    for t in range(clip.shape[0]):
        L.orc_enc_frame(e, clip[t].ctypes.data, C.byref(out), C.byref(n), C.byref(cap), None)
        cnt = C.c_int(0)
        p = L.orc_enc_last_mvs(e, C.byref(cnt))
        if t == 0 or not p or cnt.value == 0:
            continue
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cnt.value * 12,)).reshape(cnt.value, 12).copy()
        mv = a[:, :4].copy().view(np.int16).reshape(-1, 2).astype(int)
        inter = a[:, 4] == 0
        intra += int((~inter).sum())
        for x, y in mv[inter]:
            lum[y & 1, x & 1] += 1
            chrm[(y >> vs) & 1, (x >> hs) & 1] += 1
    C.CDLL(None).free(out)
    L.orc_enc_close(e)
    return lum, chrm, intra
