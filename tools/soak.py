#!/usr/bin/env python3
"""Random stream-parity soak on the GPU box (encoder; every third case the decoder too): geometries (with an emphasis on multiples of 8 / 16 / 128, where the edge
tiles of the fast inverse kernel and the 16-byte border paths apply), formats, quantisers, GOP lengths, content styles and
batch shapes (several streams side by side, frames per call; host clips, device clips with in-place chroma, one stream in
GOP-parallel chain mode) -- product stream against the oracle's, byte for byte.
usage: soak.py [cases] [seed]      (the oracle is the slow side: ~0.1-1 s per case; SOAK_MODE=host|device|chain|piped: every case in that mode)
piped (round 6, advisor round 5): the loop bench.py times -- submit(held=True) / collect with TWO batches in flight on two device clips that are
rewritten in place, each scribbled over as soon as its collect has returned -- on geometries around the limits of the in-place luma ring
(width a multiple of 16 near 2*16*ring_x16 + 64, height a multiple of 4 near 2*4*ring_y4 + 64, partial edge blocks, two pyramid levels)."""
import importlib, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
from test_gpu_stream import product_decode
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WS = [1920, 128, 144, 160, 256, 320, 352, 384, 400, 512, 640, 704, 720, 768, 800, 960, 1024, 1280]
HS = [1080, 64, 72, 96, 128, 144, 176, 240, 256, 288, 320, 360, 384, 400, 480, 512, 576, 600, 720]
FMTS = [A.SUBSAMP_420, A.SUBSAMP_420, A.SUBSAMP_444, A.SUBSAMP_422, A.SUBSAMP_411]
bad = 0
t0 = time.time()
for k in range(N):
    w, h = rng.choice(WS), rng.choice(HS)
    if rng.random() < 0.2:
        w += rng.choice([2, 4, 6, 10, 14]); h += rng.choice([2, 4, 6])
    fmt = rng.choice(FMTS)
    # (the reference divides by zero when a chroma edge block is one pixel wide or high, bmc.c:176-189: the oracle follows it)
    bw, bh = A.block_dims(w, h)[:2]
    cw, ch = A.chroma_dims(w, h, fmt)
    if cw % max(1, bw >> A.hshift(fmt)) == 1 or ch % max(1, bh >> A.vshift(fmt)) == 1 or w % bw == 1 or h % bh == 1:
        continue
    n = rng.choice([3, 4, 5, 7])
    S = rng.choice([1, 1, 2, 5, 17, 33, 64])
    F = rng.choice([f for f in (1, 2, 3, n) if n % f == 0])        # (the batch API takes whole batches)
    style = rng.choice([0, 1, 2, 3, 4, 5, 6, 7])
    mode = os.environ.get('SOAK_MODE') or rng.choice(['host', 'host', 'device', 'chain', 'piped'])      # batch from host memory / from a device clip (in-place chroma) / one stream in chain mode / the pipelined held loop
    if mode == 'piped':
        w, h = 16 * rng.randrange(9, 30), 4 * rng.randrange(36, 110)     # 144 .. 464 x 144 .. 436: both sides of the ring limits of 16 / 24-pixel blocks
        fmt = rng.choice([A.SUBSAMP_420, A.SUBSAMP_420, A.SUBSAMP_422, A.SUBSAMP_444])
        bw, bh = A.block_dims(w, h)[:2]
        cw, ch = A.chroma_dims(w, h, fmt)
        if cw % max(1, bw >> A.hshift(fmt)) == 1 or ch % max(1, bh >> A.vshift(fmt)) == 1 or w % bw == 1 or h % bh == 1:
            continue
    cli = dict(qp=rng.choice([20, 50, 70, 85, 95]), gop=rng.choice([0, 3, 12]), rc_mode_cli=1, scd=rng.choice([0, 1]))
    if mode != 'chain' and rng.random() < 0.3:                 # round 4: ABR streams (rate control on the device; chain mode refuses them)
        cli['rc_mode_cli'] = 0
        cli['kbps'] = rng.choice([0, 300, 2000])
    seed = rng.randrange(1 << 30)
    if os.environ.get('SOAK_VERBOSE'):
        print('case %d: %s %dx%d fmt %d n %d S %d F %d style %d %s seed %d' % (k, mode, w, h, fmt, n, S, F, style, cli, seed), flush=True)
    clips = [A.gen_clip(w, h, fmt, seed + s, n, style=style) for s in range(min(S, 3))]
    try:
        want = [A.orc_encode(c, A.orc_cfg(w, h, fmt, **cli), eos=False)[0] for c in clips]
    except Exception as e:
        print("case %d skipped (oracle: %s)" % (k, e)); continue
    if mode == 'chain':
        # ONE stream, GOP-parallel (dsv1_stream_open): calls of F frames, a random number of chains side by side
        n2 = rng.choice([8, 12, 18])
        Fc = rng.choice([f for f in (2, 3, 4, 6) if n2 % f == 0])
        clip = A.gen_clip(w, h, fmt, seed, n2, style=style)
        try:
            want1 = A.orc_encode(clip, A.orc_cfg(w, h, fmt, **cli))[0]
        except Exception as e:
            print("case %d skipped (oracle: %s)" % (k, e)); continue
        got1 = pkg.encode_stream(clip, w, h, fmt, Fc, rng.choice([1, 2, 3, 5]), **cli)
        if got1 != want1:
            bad += 1
            print("CHAIN MISMATCH case %d: %dx%d fmt %d n %d F %d style %d %s seed %d" % (k, w, h, fmt, n2, Fc, style, cli, seed))
        continue
    if mode == 'piped':
        cli.pop('kbps', None); cli['rc_mode_cli'] = 1
        cli['gop'] = rng.choice([3, 4, 12])
        if rng.random() < 0.4:
            cli['pyrlevels'] = 2
        Fp, ncalls, Sp = rng.choice([2, 3, 4]), rng.choice([3, 4, 5]), rng.choice([1, 2, 5, 17])
        clipsP = [A.gen_clip(w, h, fmt, seed + s, Fp * ncalls, style=style) for s in range(min(Sp, 3))]
        try:
            wantP = [A.orc_encode(c, A.orc_cfg(w, h, fmt, **cli), eos=False)[0] for c in clipsP]
        except Exception as e:
            print("case %d skipped (oracle: %s)" % (k, e)); continue
        b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), Sp, Fp)
        try:
            fbp = A.frame_bytes(w, h, fmt)
            junk = np.full((Sp, Fp, fbp), 0xA5, dtype=np.uint8)
            dev = [b.upload(junk), b.upload(junk)]
            gotP = [b""] * Sp
            def fill(c):
                fr = np.ascontiguousarray(np.stack([clipsP[s % len(clipsP)][c * Fp:(c + 1) * Fp] for s in range(Sp)]).reshape(Sp, Fp, -1))
                pkg._chk(b.L.dsvg_dev_upload(b.ctx, dev[c & 1], fr.ctypes.data, fr.nbytes), "dsvg_dev_upload")
            def scribble(c):
                junk[...] = rng.randrange(256)
                pkg._chk(b.L.dsvg_dev_upload(b.ctx, dev[c & 1], junk.ctypes.data, junk.nbytes), "dsvg_dev_upload")
            fill(0)
            b.submit(dev[0], on_device=True, held=True)
            for c in range(1, ncalls):
                fill(c)
                b.submit(dev[c & 1], on_device=True, held=True)
                pk = b.collect()                      # batch c - 1: its clip is the caller's again ...
                scribble(c - 1)                       # ... and is overwritten while batch c is in flight
                for s in range(Sp):
                    gotP[s] += pk[s]
            pk = b.collect()
            for s in range(Sp):
                gotP[s] += pk[s]
            for s in range(Sp):
                if gotP[s] != wantP[s % len(clipsP)]:
                    bad += 1
                    print("PIPED MISMATCH case %d: %dx%d fmt %d F %d calls %d S %d style %d %s seed %d stream %d" % (k, w, h, fmt, Fp, ncalls, Sp, style, cli, seed, s))
                    break
        finally:
            b.close()
        continue
    b = pkg.Batch(pkg.make_encoder_cfg(w, h, fmt, **cli), S, F)
    try:
        got = [b""] * S
        t = 0
        while t < n:
            f = F
            fr = np.stack([clips[s % len(clips)][t:t + f] for s in range(S)]).reshape(S, f, -1)
            pk = b.encode(b.upload(fr), on_device=True) if mode == 'device' else b.encode(fr)
            for s in range(S):
                got[s] += pk[s]
            t += f
        for s in range(S):
            if got[s] != want[s % len(clips)]:
                bad += 1
                print("MISMATCH case %d (%s): %dx%d fmt %d n %d S %d F %d style %d %s seed %d stream %d" % (k, mode, w, h, fmt, n, S, F, style, cli, seed, s))
                break
    finally:
        b.close()
    if k % 3 == 0:                                   # every third case also through the drop-in decoder
        frames = product_decode(pkg, want[0])
        ref = A.orc_decode(want[0], w, h, fmt)
        if len(frames) != len(ref) or any((x != y).any() for x, y in zip(frames, ref)):
            bad += 1
            print("DECODE MISMATCH case %d: %dx%d fmt %d n %d style %d %s seed %d" % (k, w, h, fmt, n, style, cli, seed))
print("soak: %d cases, %d mismatches, %.0f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
