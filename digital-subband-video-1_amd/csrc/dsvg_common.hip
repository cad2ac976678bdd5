// dsvg_common.hip -- geometry / quantiser derivation and small runtime helpers (host side of the shim).
#include <stdarg.h>
#include <stdlib.h>
#include "dsvg_host.hpp"

static thread_local char g_err[512] = "";

// (round 5: the library does NOT touch GPU_MAX_HW_QUEUES -- in a fresh process small contexts ran 2.4 times slower with eight hardware queues
// (config 5: 6.2 -> 15 ms per step, profiles/r05_hw_queues_small_shapes.txt).  What a host-fed context needs of the queues it arranges itself:
// pick_streams, dsvg_pipe.hip.)

void dsvg_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dsvg_last_error(void) { return g_err; }

extern "C" int dsvg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The host <-> device link as the pipeline's own copies see it (verdict round 5: a figure measured through another allocator's pinned memory read
// 29 GB/s on a box where the staged clip upload of the same process then ran at 37.5).  Pinned memory from hipHostMalloc -- allocated and first
// touched by the calling thread, i.e. on the NUMA node its affinity mask says --, ONE asynchronous copy of `bytes` per repetition on a stream of its
// own, HIP events around the repetitions: exactly what dsvg_ingest_begin / the packet fetch do.  gbs[0] = host -> device, gbs[1] = device -> host.
// Meant to be called from a short-lived child process whose affinity was set first (shard.py: link_probe), before the parent touches the GPU.
extern "C" int dsvg_link_probe(int device, size_t bytes, int reps, double gbs[2])
{
    if (!gbs || bytes < 4096 || reps < 1) { dsvg_set_error("bad link probe arguments"); return DSVG_ERR_ARG; }
    gbs[0] = gbs[1] = 0.0;
    void *h = nullptr, *d = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DSVG_OK;
    auto fail = [&](const char *what, hipError_t e) { dsvg_set_error("link probe: %s: %s", what, hipGetErrorString(e)); rc = e == hipErrorOutOfMemory ? DSVG_ERR_NOMEM : DSVG_ERR_HIP; };
    hipError_t e;
    do {
        if ((e = hipSetDevice(device)) != hipSuccess) { fail("hipSetDevice", e); break; }
        if ((e = hipHostMalloc(&h, bytes, hipHostMallocDefault)) != hipSuccess) { fail("hipHostMalloc", e); break; }
        memset(h, 0x5a, bytes);
        if ((e = hipMalloc(&d, bytes)) != hipSuccess) { fail("hipMalloc", e); break; }
        if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) { fail("hipStreamCreate", e); break; }
        if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) { fail("hipEventCreate", e); break; }
        for (int dir = 0; dir < 2 && rc == DSVG_OK; dir++) {
            void *dst = dir ? h : d;
            const void *src = dir ? d : h;
            const hipMemcpyKind kind = dir ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
            if ((e = hipMemcpyAsync(dst, src, bytes, kind, st)) != hipSuccess) { fail("warm-up copy", e); break; }
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps; r++)
                if ((e = hipMemcpyAsync(dst, src, bytes, kind, st)) != hipSuccess) { fail("copy", e); break; }
            (void)hipEventRecord(e1, st);
            if ((e = hipStreamSynchronize(st)) != hipSuccess) { fail("synchronise", e); break; }
            float ms = 0.f;
            if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess || ms <= 0.f) { fail("elapsed time", e); break; }
            gbs[dir] = (double)bytes * reps / (ms * 1e-3) / 1e9;
        }
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    return rc;
}

static int g_op_device = 0;
extern "C" int dsvg_set_device(int device)
{
    if (device < 0 || device >= dsvg_device_count()) {
        dsvg_set_error("no such HIP device %d", device);
        return DSVG_ERR_NODEVICE;
    }
    g_op_device = device;
    return DSVG_OK;
}
int dsvg_op_device() { return g_op_device; }

int lb2u(unsigned n) { return dsvg_lb2u(n); }            // dsv_lb2 hzcc.c:437-447 (dsvg_dev.hpp: shared with the device)
extern "C" int dsvg_lb2(unsigned n) { return lb2u(n); }

int get_quant(int q, int isP, int level) { return dsvg_level_quant(q, isP, level); }   // dsv_get_quant hzcc.c:77-92
extern "C" int dsvg_get_quant(int q, int isP, int level) { return get_quant(q, isP, level); }

void make_frame_layout(FrameLayout &L, int fmt, int w, int h)
{
    memset(&L, 0, sizeof(L));
    const int ext = DSVG_BORDER;
    L.hs = fmt_hs(fmt);
    L.vs = fmt_vs(fmt);
    const int cw = rsu(w, L.hs), ch = rsu(h, L.vs);
    size_t off = 0;
    for (int c = 0; c < 3; c++) {
        L.w[c] = c ? cw : w;
        L.h[c] = c ? ch : h;
        L.stride[c] = (L.w[c] + 2 * ext + 15) & ~15;
        const size_t len = (size_t)L.stride[c] * (L.h[c] + 2 * ext);
        L.off[c] = off + (size_t)L.stride[c] * ext + ext;
        off += len;
    }
    L.bytes = off;
    L.pitch = (off + 255) & ~(size_t)255;
}

int layout_from_host(FrameLayout &L, const dsvg_frame *f, const uint8_t **base, size_t *bytes)
{
    memset(&L, 0, sizeof(L));
    const uint8_t *b = f->alloc ? f->alloc : f->planes[0].data;
    size_t end = 0;
    L.hs = fmt_hs(f->format);
    L.vs = fmt_vs(f->format);
    for (int c = 0; c < 3; c++) {
        const dsvg_plane *p = &f->planes[c];
        L.w[c] = p->w; L.h[c] = p->h; L.stride[c] = p->stride;
        if (p->data < b) { dsvg_set_error("plane %d lies before the frame base", c); return DSVG_ERR_ARG; }
        L.off[c] = (size_t)(p->data - b);
        const int ext = f->border ? DSVG_BORDER : 0;
        const size_t e = L.off[c] + (size_t)p->stride * (p->h + ext - 1) + p->w + ext;
        if (e > end) end = e;
        if ((size_t)p->len > 0) {
            const size_t start = L.off[c] - (size_t)p->stride * ext - ext;
            if (start + (size_t)p->len > end) end = start + (size_t)p->len;
        }
    }
    L.bytes = end;
    L.pitch = (end + 255) & ~(size_t)255;
    *base = b;
    *bytes = end;
    return DSVG_OK;
}

void make_coef_layout(CoefLayout &C, int fmt, int w, int h)
{
    memset(&C, 0, sizeof(C));
    const int cw = (rsu(w, fmt_hs(fmt)) + 1) & ~1, ch = (rsu(h, fmt_vs(fmt)) + 1) & ~1;
    size_t off = 0, o3 = 0, o1 = 0, o5 = 0;
    for (int c = 0; c < 3; c++) {
        C.w[c] = c ? cw : w;
        C.h[c] = c ? ch : h;
        const int mx = C.w[c] > C.h[c] ? C.w[c] : C.h[c];
        C.lvls[c] = lb2u((unsigned)mx);
        C.w3[c] = rsu(C.w[c], 3); C.h3[c] = rsu(C.h[c], 3);
        C.w1[c] = rsu(C.w[c], 1); C.h1[c] = rsu(C.h[c], 1);
        C.off[c] = off;   off += (size_t)C.w[c] * C.h[c];
        C.s3off[c] = o3;  o3 += ((size_t)C.w3[c] * C.h3[c] + 3) & ~(size_t)3;
        C.s1off[c] = o1;  o1 += ((size_t)C.w1[c] * C.h1[c] + 3) & ~(size_t)3;
        C.w5[c] = rsu(C.w[c], 5); C.h5[c] = rsu(C.h[c], 5);
        C.s5off[c] = o5;  o5 += ((size_t)C.w5[c] * C.h5[c] + 3) & ~(size_t)3;
    }
    C.total = off; C.s3total = o3; C.s1total = o1; C.s5total = o5;
}

void make_sbt_geo(SbtGeo &g, int W, int H, int pw, int ph, int pstride, size_t poff, size_t coff, size_t s3off, size_t s1off, size_t s5off)
{
    memset(&g, 0, sizeof(g));
    g.W = W; g.H = H; g.pw = pw; g.ph = ph; g.pstride = pstride;
    g.poff = poff; g.coff = coff; g.s3off = s3off; g.s1off = s1off; g.s5off = s5off;
    g.lvls = lb2u((unsigned)(W > H ? W : H));
    g.w3 = rsu(W, 3); g.h3 = rsu(H, 3);
    g.w1 = rsu(W, 1); g.h1 = rsu(H, 1);
    g.w5 = rsu(W, 5); g.h5 = rsu(H, 5);
}

void make_hz_plane(HzPlane &hp, int w, int h, int q, int isP, int cur_plane, int nbh, int nbv)
{
    memset(&hp, 0, sizeof(hp));
    int n = 0, base = 0;
    HzRegion *r = hp.r;
    r[n].x0 = 0; r[n].y0 = 0; r[n].sw = rsu(w, 3); r[n].sh = rsu(h, 3);
    r[n].level = -1; r[n].base = 0;
    base += r[n].sw * r[n].sh;
    n++;
    for (int l = 0; l < 3; l++) {
        const int sw = rsu(w, 3 - l), sh = rsu(h, 3 - l);
        hp.s_w[l] = sw; hp.s_h[l] = sh;
        for (int s = 1; s < 4; s++, n++) {
            r[n].x0 = (s & 1) ? sw : 0;
            r[n].y0 = (s & 2) ? sh : 0;
            r[n].sw = sw; r[n].sh = sh;
            r[n].level = l;
            r[n].dbx = (nbh << 14) / sw;
            r[n].dby = (nbv << 14) / sh;
            r[n].base = base;
            base += sw * sh;
        }
    }
    hp.nscan = base;
    hp.nchunks = (base + HZ_CHUNK - 1) / HZ_CHUNK;
    hp.w = w; hp.h = h; hp.nbh = nbh;
    dsvg_plane_set_quant(hp, q, isP, cur_plane);               // the region quantisers (shared with the device: dsvg_dev.hpp)
}

void make_hqp(int hqp[16], int q, int isP) { dsvg_set_hqp(hqp, q, isP); }      // sbt.c:677-696

void block_geometry(int w, int h, int *bw, int *bh, int *nbh, int *nbv)
{
    auto s4 = [](int d) {
        int s = d > 1280 ? 64 : d > 1024 ? 48 : d > 704 ? 32 : d > 352 ? 24 : 16;
        s &= ~7;
        return s < 16 ? 16 : (s > 64 ? 64 : s);
    };
    *bw = s4(w); *bh = s4(h);
    *nbh = (w + *bw - 1) / *bw;
    *nbv = (h + *bh - 1) / *bh;
}

int auto_pyramid_levels(int w, int h, int nbh, int nbv)
{
    int lv = lb2u((unsigned)(w < h ? w : h));
    const int nb = nbh > nbv ? nbh : nbv;
    while ((1 << lv) > nb) lv--;
    return lv < 3 ? 3 : (lv > DSVG_MAX_PYRAMID ? DSVG_MAX_PYRAMID : lv);
}

int Slab::alloc(size_t n, bool zero)
{
    bytes = n;
    hipError_t e = hipMalloc((void **)&raw, n + 2 * GUARD_BYTES);
    if (e != hipSuccess) { dsvg_set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); return DSVG_ERR_HIP; }
    if (zero) {
        e = hipMemset(raw, 0, n + 2 * GUARD_BYTES);
        if (e == hipSuccess) e = hipDeviceSynchronize();      // the users' streams are non-blocking: no implicit order against this NULL-stream fill
        if (e != hipSuccess) { dsvg_set_error("hipMemset failed: %s", hipGetErrorString(e)); return DSVG_ERR_HIP; }
    }
    p = raw + GUARD_BYTES;
    return DSVG_OK;
}
void Slab::release()
{
    if (raw) (void)hipFree(raw);
    raw = p = nullptr;
    bytes = 0;
}

hipEvent_t Prof::get()
{
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void Prof::begin(hipStream_t st, int kid, double alg_bytes)
{
    open = false;
    if (!want(kid)) return;
    Rec r; r.kid = kid; r.a = get(); r.b = get(); r.bytes = alg_bytes;
    (void)hipEventRecord(r.a, st);
    recs.push_back(r);
    open = true;
}
void Prof::end(hipStream_t st)
{
    if (!open) return;
    (void)hipEventRecord(recs.back().b, st);
    open = false;
}
void Prof::collect()
{
    for (auto &r : recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.kid] += t; bytes[r.kid] += r.bytes; launches[r.kid]++; }
        pool.push_back(r.a); pool.push_back(r.b);
    }
    recs.clear();
}
void Prof::reset()
{
    collect();
    for (int i = 0; i < KID_N; i++) { ms[i] = 0; bytes[i] = 0; launches[i] = 0; }
}
const char *kid_name(int kid)
{
    static const char *n[KID_N] = {
#define X(id, name) name,
        DSVG_KERNEL_IDS(X)
#undef X
    };
    return (kid >= 0 && kid < KID_N) ? n[kid] : "?";
}
