// dsvg_ops.hip -- OPERATOR-LEVEL C ABI: twins of the reference operator API (dsv_internal.h:94-109,
// dsv.h:169, dsv_encoder.h:132).  Host pointers in/out; every call stages its operands in HBM and
// runs the same HIP kernels the batched pipeline uses (job count 1).  No CPU fallback exists: with
// no HIP device every entry point fails with DSVG_ERR_NODEVICE / DSVG_ERR_HIP.
#include <stdlib.h>
#include "dsvg_host.hpp"

int dsvg_op_device();

namespace {

struct OpScope {                 // selects the device, owns a stream and every temp allocation
    hipStream_t st = nullptr;
    std::vector<void *> frees;
    std::vector<Slab> slabs;
    int init()
    {
        if (dsvg_device_count() <= 0) { dsvg_set_error("no HIP device available"); return DSVG_ERR_NODEVICE; }
        HIPCHK(hipSetDevice(dsvg_op_device()));
        HIPCHK(hipStreamCreate(&st));
        sbt_set_func_attributes();
        return DSVG_OK;
    }
    template <typename T> int dev(T **p, size_t n, bool zero = true)
    {
        HIPCHK(hipMalloc((void **)p, n * sizeof(T) + 64));
        frees.push_back((void *)*p);
        if (zero) HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(T) + 64, st));
        return DSVG_OK;
    }
    int slab(uint8_t **p, size_t n)
    {
        Slab s;
        int rc = s.alloc(n);
        if (rc) return rc;
        slabs.push_back(s);
        *p = s.p;
        return DSVG_OK;
    }
    int sync()
    {
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        return DSVG_OK;
    }
    ~OpScope()
    {
        for (void *p : frees) (void)hipFree(p);
        for (auto &s : slabs) s.release();
        if (st) (void)hipStreamDestroy(st);
    }
};

#define OPCHK(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// upload a host frame verbatim (same linear layout) into a guarded device slab
int upload_frame(OpScope &S, const dsvg_frame *f, FrameLayout &L, uint8_t **dptr)
{
    const uint8_t *base; size_t bytes;
    OPCHK(layout_from_host(L, f, &base, &bytes));
    OPCHK(S.slab(dptr, L.pitch + 4096));
    HIPCHK(hipMemcpyAsync(*dptr, base, bytes, hipMemcpyHostToDevice, S.st));
    return DSVG_OK;
}
int download_frame(OpScope &S, dsvg_frame *f, const FrameLayout &L, const uint8_t *dptr)
{
    uint8_t *base = f->alloc ? f->alloc : f->planes[0].data;
    HIPCHK(hipMemcpyAsync(base, dptr, L.bytes, hipMemcpyDeviceToHost, S.st));
    return DSVG_OK;
}

int check_plane_dims(int W, int H, int isP)
{
    if (W < 8 || H < 8) { dsvg_set_error("plane %dx%d too small (needs >= 3 transform levels)", W, H); return DSVG_ERR_UNSUPPORTED; }
    if (!isP && ((W | H) & 1)) {
        dsvg_set_error("intra transform needs even plane dims (reference leaves stale temp words for odd n)");
        return DSVG_ERR_UNSUPPORTED;
    }
    SbtGeo g; make_sbt_geo(g, W, H, W, H, W, 0, 0, 0, 0);
    if (!sbt_tail_supported(g)) { dsvg_set_error("plane %dx%d: outside the LDS tail limits", W, H); return DSVG_ERR_UNSUPPORTED; }
    return DSVG_OK;
}

struct HostBits {                // MSB-first reader/writer for plane framing (bs.c semantics)
    uint8_t *p; unsigned pos;
    unsigned bit() { unsigned b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u; pos++; return b; }
    unsigned bits(int n) { unsigned v = 0; while (n--) v = (v << 1) | bit(); return v; }
    unsigned ueg() { unsigned m = 1; while (!bit()) m = (m << 1) | bit(); return m - 1; }
    int seg() { int v = (int)ueg(); return (v && bit()) ? -v : v; }
    int neg() { int v = (int)ueg() + 1; return bit() ? -v : v; }
    void align() { pos = (pos + 7u) & ~7u; }
    void put(int n, unsigned v) { while (n--) { if ((v >> n) & 1u) p[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7)); pos++; } }
    void put_ueg(unsigned v)
    {
        const unsigned m = v + 1; int k = 31 - __builtin_clz(m);
        while (k-- > 0) { pos++; put(1, (m >> k) & 1u); }
        put(1, 1);
    }
    void put_seg(int v) { unsigned m = v < 0 ? (unsigned)-v : (unsigned)v; put_ueg(m); if (m) put(1, v < 0); }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int dsvg_op_fwd_sbt(const dsvg_plane *src, dsvg_coefs *dst, int isP)
{
    if (!src || !dst || !src->data || !dst->data) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    const int W = dst->width, H = dst->height;
    OPCHK(check_plane_dims(W, H, isP));
    OpScope S; OPCHK(S.init());
    const int ph = src->h < H ? src->h : H;
    // re-stride into an aligned staging plane; W bytes per row are exactly what p2sbc reads (sbt.c:584-590)
    const int dstride = ((W + 15) & ~15) + 16;
    uint8_t *dpx; OPCHK(S.slab(&dpx, (size_t)dstride * (ph + 8) + 4096));
    HIPCHK(hipMemcpy2DAsync(dpx, (size_t)dstride, src->data, (size_t)src->stride, (size_t)W, (size_t)ph,
                            hipMemcpyHostToDevice, S.st));
    int32_t *dco, *ds3, *ds1, *ds5;
    OPCHK(S.dev(&dco, (size_t)W * H));
    OPCHK(S.dev(&ds3, (size_t)rsu(W, 3) * rsu(H, 3) + 8));
    OPCHK(S.dev(&ds1, (size_t)rsu(W, 1) * rsu(H, 1) + 8));
    OPCHK(S.dev(&ds5, (size_t)rsu(W, 5) * rsu(H, 5) + 8));
    JobDev jb; memset(&jb, 0, sizeof(jb));
    jb.src = dpx; jb.xf = dpx; jb.coef = dco; jb.s3 = ds3; jb.s1 = ds1; jb.s5 = ds5; jb.isP = isP;
    jb.srcp[0] = dpx; jb.srcs[0] = dstride;
    JobDev *djb; OPCHK(S.dev(&djb, 1, false));
    HIPCHK(hipMemcpyAsync(djb, &jb, sizeof(jb), hipMemcpyHostToDevice, S.st));
    SbtGeo3 G; memset(&G, 0, sizeof(G));
    make_sbt_geo(G.g[0], W, H, src->w, ph, dstride, 0, 0, 0, 0);
    launch_fwd_sbt(S.st, djb, 1, G, 0, 1, isP, 0);
    HIPCHK(hipMemcpyAsync(dst->data, dco, (size_t)W * H * 4, hipMemcpyDeviceToHost, S.st));
    return S.sync();
}

extern "C" int dsvg_op_inv_sbt(dsvg_plane *dst, dsvg_coefs *src, int q, int isP, int c)
{
    if (!src || !dst || !src->data || !dst->data) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    const int W = src->width, H = src->height;
    OPCHK(check_plane_dims(W, H, isP));
    OpScope S; OPCHK(S.init());
    const int pc = c ? 1 : 0;
    const int ph = dst->h < H ? dst->h : H, pw = dst->w < W ? dst->w : W;
    const int dstride = (W + 15) & ~15;
    uint8_t *dpx; OPCHK(S.slab(&dpx, (size_t)dstride * (H + 8) + 4096));
    int32_t *dco, *ds3, *ds1, *ds5;
    OPCHK(S.dev(&dco, (size_t)W * H, false));
    OPCHK(S.dev(&ds3, (size_t)rsu(W, 3) * rsu(H, 3) + 8));
    OPCHK(S.dev(&ds1, (size_t)rsu(W, 1) * rsu(H, 1) + 8));
    OPCHK(S.dev(&ds5, (size_t)rsu(W, 5) * rsu(H, 5) + 8));
    HIPCHK(hipMemcpyAsync(dco, src->data, (size_t)W * H * 4, hipMemcpyHostToDevice, S.st));
    JobDev jb; memset(&jb, 0, sizeof(jb));
    jb.xf = dpx; jb.coef = dco; jb.s3 = ds3; jb.s1 = ds1; jb.s5 = ds5; jb.isP = isP; jb.quant = q;
    make_hqp(jb.hqp, q, isP);
    JobDev *djb; OPCHK(S.dev(&djb, 1, false));
    HIPCHK(hipMemcpyAsync(djb, &jb, sizeof(jb), hipMemcpyHostToDevice, S.st));
    SbtGeo3 G; memset(&G, 0, sizeof(G));
    make_sbt_geo(G.g[pc], W, H, pw, ph, dstride, 0, 0, 0, 0);
    launch_inv_sbt(S.st, djb, 1, G, pc, 1, isP);
    HIPCHK(hipMemcpy2DAsync(dst->data, (size_t)dst->stride, dpx, (size_t)dstride, (size_t)pw, (size_t)ph,
                            hipMemcpyDeviceToHost, S.st));
    return S.sync();
}

// shared by encode/decode: one-plane job with the quantiser tables of `stab`
static int make_plane_job(OpScope &S, JobDev &jb, const dsvg_coefs *co, int q, const dsvg_stability *stab, int32_t **dco)
{
    const int W = co->width, H = co->height, c = stab->cur_plane ? 1 : 0;
    const dsvg_params *p = stab->params;
    const int nblk = p->nblocks_h * p->nblocks_v;
    memset(&jb, 0, sizeof(jb));
    make_hz_plane(jb.hz[c], W, H, q, stab->isP, stab->cur_plane, p->nblocks_h, p->nblocks_v);
    if (jb.hz[c].nchunks > hz_scan_items_max()) { dsvg_set_error("plane too large for the scan kernel"); return DSVG_ERR_UNSUPPORTED; }
    OPCHK(S.dev(dco, (size_t)W * H, false));
    HIPCHK(hipMemcpyAsync(*dco, co->data, (size_t)W * H * 4, hipMemcpyHostToDevice, S.st));
    uint8_t *dst; OPCHK(S.dev(&dst, (size_t)nblk, false));
    HIPCHK(hipMemcpyAsync(dst, stab->stable_blocks, (size_t)nblk, hipMemcpyHostToDevice, S.st));
    jb.coef = *dco; jb.stable = dst; jb.isP = stab->isP; jb.quant = q;
    return DSVG_OK;
}

extern "C" int dsvg_op_encode_plane(dsvg_bs *bs, dsvg_coefs *src, int q, const dsvg_stability *stab)
{
    if (!bs || !src || !stab || !stab->params || !stab->stable_blocks) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    OpScope S; OPCHK(S.init());
    const int W = src->width, H = src->height, c = stab->cur_plane ? 1 : 0;
    JobDev jb; int32_t *dco;
    OPCHK(make_plane_job(S, jb, src, q, stab, &dco));
    const HzPlane &hp = jb.hz[c];
    const size_t cap = (size_t)W * H * 4 + 64;
    OPCHK(S.dev(&jb.nzpos, (size_t)hp.nchunks * HZ_CHUNK, false));
    OPCHK(S.dev(&jb.nzval, (size_t)hp.nchunks * HZ_CHUNK, false));
    OPCHK(S.dev(&jb.chunks, (size_t)hp.nchunks + 1));
    OPCHK(S.dev(&jb.psum, 3));
    OPCHK(S.dev(&jb.bits, cap + 64));
    jb.bits_cap[c] = cap;
    JobDev *djb; OPCHK(S.dev(&djb, 1, false));
    HIPCHK(hipMemcpyAsync(djb, &jb, sizeof(jb), hipMemcpyHostToDevice, S.st));
    launch_hz_encode(S.st, djb, 1, hp.nchunks);
    HzPlaneSum ps[3];
    HIPCHK(hipMemcpyAsync(ps, jb.psum, sizeof(ps), hipMemcpyDeviceToHost, S.st));
    HIPCHK(hipMemcpyAsync(src->data, dco, (size_t)W * H * 4, hipMemcpyDeviceToHost, S.st));
    OPCHK(S.sync());
    if (ps[c].overflow) { dsvg_set_error("packed plane exceeds %zu bytes", cap); return DSVG_ERR_OVERFLOW; }
    const size_t nbytes = (size_t)((ps[c].total_bits + 7) >> 3);

    // plane framing (dsv_encode_plane hzcc.c:449-476, hzcc_enc head/tail hzcc.c:150-155,287-292)
    HostBits hb{bs->start, bs->pos};
    hb.align();
    const unsigned startp = hb.pos >> 3;
    hb.put(32, 0);
    hb.put_seg(ps[c].dc);
    hb.align();
    hb.put(32, ps[c].nruns);
    hb.align();
    if (nbytes) HIPCHK(hipMemcpy(bs->start + (hb.pos >> 3), jb.bits, nbytes, hipMemcpyDeviceToHost));
    hb.pos += (unsigned)ps[c].total_bits;
    hb.align();
    hb.put(8, 0x55);
    hb.align();
    const unsigned endp = hb.pos >> 3;
    hb.pos = startp * 8;
    hb.put(32, endp - startp - 4);
    bs->pos = endp * 8;
    return DSVG_OK;
}

extern "C" int dsvg_op_decode_plane(uint8_t *in, unsigned len, dsvg_coefs *dst, int q, const dsvg_stability *stab)
{
    if (!in || !dst || !stab || !stab->params || !stab->stable_blocks) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    OpScope S; OPCHK(S.init());
    const int c = stab->cur_plane ? 1 : 0;
    JobDev jb; int32_t *dco;
    OPCHK(make_plane_job(S, jb, dst, q, stab, &dco));
    const HzPlane &hp = jb.hz[c];

    // the host reads the plane header only (SEG(DC), run count); the code chain is parsed on the device
    HostBits hb{in, 0};
    const int dc = hb.seg();
    hb.align();
    const int runs = (int)hb.bits(32);
    hb.align();
    jb.dec_dc[c] = dc; jb.dec_runs[c] = runs; jb.dec_len[c] = (int)len; jb.dec_bitpos[c] = (long long)hb.pos;
    uint8_t *dpay; OPCHK(S.dev(&dpay, (size_t)len + 64, true));
    HIPCHK(hipMemcpyAsync(dpay, in, len, hipMemcpyHostToDevice, S.st));
    jb.bits = dpay; jb.bits_off[c] = 0;
    OPCHK(S.dev(&jb.dec_meta[c], (size_t)len / 16 + 2, false));
    OPCHK(S.dev(&jb.nzpos, (size_t)hp.nchunks * HZ_CHUNK, false));
    OPCHK(S.dev(&jb.nzval, (size_t)hp.nchunks * HZ_CHUNK, false));
    JobDev *djb; OPCHK(S.dev(&djb, 1, false));
    HIPCHK(hipMemcpyAsync(djb, &jb, sizeof(jb), hipMemcpyHostToDevice, S.st));
    // parse, then three ordered scatter phases so that a later region's non-zero overwrites an earlier one
    launch_hz_parse_scatter(S.st, djb, 1, c, 1, std::min(std::max(runs, 0), hp.nchunks * HZ_CHUNK - 1) + 1, (int)(len / 16 + 2));
    HIPCHK(hipMemcpyAsync(dst->data, dco, (size_t)dst->width * dst->height * 4, hipMemcpyDeviceToHost, S.st));
    OPCHK(S.sync());
    dst->data[0] = dc;
    return DSVG_OK;
}

// ------------------------------------------------------------------------------------------------
static int mc_geo_from(McGeo &G, const dsvg_params *p, const FrameLayout &L)
{
    memset(&G, 0, sizeof(G));
    G.blk_w = p->blk_w; G.blk_h = p->blk_h; G.nbh = p->nblocks_h; G.nbv = p->nblocks_v;
    G.hs = L.hs; G.vs = L.vs;
    for (int c = 0; c < 3; c++) { G.w[c] = L.w[c]; G.h[c] = L.h[c]; G.stride[c] = L.stride[c]; G.off[c] = L.off[c]; }
    if (p->blk_w > 64 || p->blk_h > 64 || p->blk_w < 16 || p->blk_h < 16) { dsvg_set_error("bad block size"); return DSVG_ERR_ARG; }
    return DSVG_OK;
}
static int same_layout(const FrameLayout &a, const FrameLayout &b)
{
    for (int c = 0; c < 3; c++)
        if (a.w[c] != b.w[c] || a.h[c] != b.h[c] || a.stride[c] != b.stride[c] || a.off[c] != b.off[c]) return 0;
    return 1;
}

static int op_pred(const dsvg_mv *mv, const dsvg_params *p, dsvg_frame *dif, dsvg_frame *io, const dsvg_frame *ref, int sub)
{
    if (!mv || !p || !dif || !io || !ref || !p->vidmeta) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    OpScope S; OPCHK(S.init());
    FrameLayout Ld, Li, Lr;
    uint8_t *dd, *di, *dr;
    OPCHK(upload_frame(S, dif, Ld, &dd));
    OPCHK(upload_frame(S, io, Li, &di));
    OPCHK(upload_frame(S, ref, Lr, &dr));
    if (!same_layout(Ld, Li) || !same_layout(Ld, Lr)) { dsvg_set_error("frames must share one layout"); return DSVG_ERR_ARG; }
    McGeo G; OPCHK(mc_geo_from(G, p, Ld));
    const int nblk = p->nblocks_h * p->nblocks_v;
    DMV *dmv; OPCHK(S.dev(&dmv, (size_t)nblk, false));
    HIPCHK(hipMemcpyAsync(dmv, mv, (size_t)nblk * sizeof(DMV), hipMemcpyHostToDevice, S.st));
    JobDev jb; memset(&jb, 0, sizeof(jb));
    jb.ref = dr; jb.mvs = dmv;
    if (sub) { jb.pred = dd; jb.src = di; jb.xf = di; }          // dif <- prediction, inp <- residual (in place)
    else     { jb.pred = di; jb.src = di; jb.xf = di; }          // out <- prediction, then out += dif - 128
    for (int pl = 0; pl < 3; pl++) { jb.srcp[pl] = jb.src + G.off[pl]; jb.srcs[pl] = G.stride[pl]; }
    JobDev *djb; OPCHK(S.dev(&djb, 1, false));
    HIPCHK(hipMemcpyAsync(djb, &jb, sizeof(jb), hipMemcpyHostToDevice, S.st));
    launch_mc(S.st, djb, 1, G, sub);
    if (!sub) launch_frame_add(S.st, di, Li, dd, Ld);
    OPCHK(download_frame(S, io, Li, di));
    if (sub) OPCHK(download_frame(S, dif, Ld, dd));
    return S.sync();
}

extern "C" int dsvg_op_sub_pred(const dsvg_mv *mv, const dsvg_params *p, dsvg_frame *dif, dsvg_frame *inp, const dsvg_frame *ref)
{
    return op_pred(mv, p, dif, inp, ref, 1);
}
extern "C" int dsvg_op_add_pred(const dsvg_mv *mv, const dsvg_params *p, dsvg_frame *dif, dsvg_frame *out, const dsvg_frame *ref)
{
    return op_pred(mv, p, dif, out, ref, 0);
}

extern "C" int dsvg_op_frame_add(dsvg_frame *dst, const dsvg_frame *src)
{
    if (!dst || !src) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    OpScope S; OPCHK(S.init());
    FrameLayout Ld, Ls; uint8_t *dd, *ds;
    OPCHK(upload_frame(S, dst, Ld, &dd));
    OPCHK(upload_frame(S, src, Ls, &ds));
    launch_frame_add(S.st, dd, Ld, ds, Ls);
    OPCHK(download_frame(S, dst, Ld, dd));
    return S.sync();
}

extern "C" int dsvg_op_extend_frame(dsvg_frame *f)
{
    if (!f) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    if (!f->border) return DSVG_OK;
    OpScope S; OPCHK(S.init());
    FrameLayout L; uint8_t *d;
    OPCHK(upload_frame(S, f, L, &d));
    launch_extend(S.st, d, L, 0, 1, 3, nullptr);
    OPCHK(download_frame(S, f, L, d));
    return S.sync();
}
extern "C" int dsvg_op_extend_frame_luma(dsvg_frame *f)
{
    if (!f) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    if (!f->border) return DSVG_OK;
    OpScope S; OPCHK(S.init());
    FrameLayout L; uint8_t *d;
    OPCHK(upload_frame(S, f, L, &d));
    launch_extend(S.st, d, L, 0, 1, 1, nullptr);
    OPCHK(download_frame(S, f, L, d));
    return S.sync();
}
extern "C" int dsvg_op_ds2x_frame_luma(dsvg_frame *dst, const dsvg_frame *src)
{
    if (!dst || !src) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    OpScope S; OPCHK(S.init());
    FrameLayout Ld, Ls; uint8_t *dd, *ds;
    OPCHK(upload_frame(S, dst, Ld, &dd));
    OPCHK(upload_frame(S, src, Ls, &ds));
    launch_ds2x(S.st, ds, Ls, dd, Ld, 0, 1);
    OPCHK(download_frame(S, dst, Ld, dd));
    return S.sync();
}
extern "C" int dsvg_op_frame_avg_luma(const dsvg_frame *f, int *avg)
{
    if (!f || !avg) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    OpScope S; OPCHK(S.init());
    FrameLayout L; uint8_t *d; unsigned *ds;
    OPCHK(upload_frame(S, f, L, &d));
    OPCHK(S.dev(&ds, 4));
    launch_luma_sum(S.st, d, L, 0, 1, ds);
    unsigned sum = 0;
    HIPCHK(hipMemcpyAsync(&sum, ds, 4, hipMemcpyDeviceToHost, S.st));
    OPCHK(S.sync());
    *avg = (int)sum / (L.w[0] * L.h[0]);
    return DSVG_OK;
}

// Memory the operator calls hand to their caller (dsvg_op_hme's motion fields) comes from the allocator the caller FREES with: the reference's
// caller releases hme.mvf[] with dsv_free (dsv_encoder.c:239-244).  Default: the library's own dsv_alloc / dsv_free (dsv1_util.c); a build that
// keeps the reference's dsv.c -- whose dsv_free steps back over a 16-byte header when DSV_MEMORY_STATS is on (dsv.c:41-66) -- passes its own pair.
extern "C" void *dsv_alloc(int size);
extern "C" void dsv_free(void *ptr);
static void *(*g_op_alloc)(int) = dsv_alloc;
static void (*g_op_free)(void *) = dsv_free;
extern "C" void dsvg_set_allocator(void *(*alloc_fn)(int), void (*free_fn)(void *))
{
    g_op_alloc = alloc_fn && free_fn ? alloc_fn : dsv_alloc;
    g_op_free = alloc_fn && free_fn ? free_fn : dsv_free;
}

extern "C" int dsvg_op_hme(dsvg_hme *h, int *intra_pct)
{
    if (!h || !h->params || !h->params->vidmeta) { dsvg_set_error("null argument"); return DSVG_ERR_ARG; }
    if (h->levels < 0 || h->levels > DSVG_MAX_PYRAMID) { dsvg_set_error("bad level count"); return DSVG_ERR_ARG; }
    OpScope S; OPCHK(S.init());
    const dsvg_params *p = h->params;
    const int nblk = p->nblocks_h * p->nblocks_v;
    HmeArgs A; memset(&A, 0, sizeof(A));
    A.levels = h->levels; A.nxb = p->nblocks_h; A.nyb = p->nblocks_v; A.nblk = nblk;
    A.blk_w = p->blk_w; A.blk_h = p->blk_h;
    for (int l = 0; l <= h->levels; l++) {
        FrameLayout Ls, Lr; const uint8_t *bs_, *br_; size_t ns, nr;
        OPCHK(layout_from_host(Ls, h->src[l], &bs_, &ns));
        OPCHK(layout_from_host(Lr, h->ref[l], &br_, &nr));
        if (!same_layout(Ls, Lr)) { dsvg_set_error("src/ref layouts differ at level %d", l); return DSVG_ERR_ARG; }
        uint8_t *slab; OPCHK(S.slab(&slab, 2 * Ls.pitch + 4096));
        HIPCHK(hipMemcpyAsync(slab, bs_, ns, hipMemcpyHostToDevice, S.st));
        HIPCHK(hipMemcpyAsync(slab + Ls.pitch, br_, nr, hipMemcpyHostToDevice, S.st));
        A.L[l] = Ls; A.slab[l] = slab;
    }
    int *dslots; OPCHK(S.dev(&dslots, 2));
    const int slots[2] = {0, 1};
    HIPCHK(hipMemcpyAsync(dslots, slots, sizeof(slots), hipMemcpyHostToDevice, S.st));
    A.cur_slots = dslots; A.ref_slots = dslots + 1;
    {   // chroma planes of the two level-0 frames (slots 0, 1 of the level's slab)
        unsigned long long hcu[2], hcv[2]; int hcs[2];
        for (int k = 0; k < 2; k++) {
            hcu[k] = (unsigned long long)(uintptr_t)(A.slab[0] + (size_t)k * A.L[0].pitch + A.L[0].off[1]);
            hcv[k] = (unsigned long long)(uintptr_t)(A.slab[0] + (size_t)k * A.L[0].pitch + A.L[0].off[2]);
            hcs[k] = A.L[0].stride[1];
        }
        unsigned long long *dcu; int *dcs;
        OPCHK(S.dev(&dcu, 4)); OPCHK(S.dev(&dcs, 2));
        HIPCHK(hipMemcpyAsync(dcu, hcu, sizeof(hcu), hipMemcpyHostToDevice, S.st));
        HIPCHK(hipMemcpyAsync(dcu + 2, hcv, sizeof(hcv), hipMemcpyHostToDevice, S.st));
        HIPCHK(hipMemcpyAsync(dcs, hcs, sizeof(hcs), hipMemcpyHostToDevice, S.st));
        A.slot_cu = dcu; A.slot_cv = dcu + 2; A.slot_cs = dcs;
    }
    OPCHK(S.dev(&A.mvf, (size_t)(h->levels + 1) * nblk));
    OPCHK(S.dev(&A.aux_tex, (size_t)nblk));
    OPCHK(S.dev(&A.aux_var, (size_t)nblk));
    launch_hme(S.st, A, 1);
    for (int l = 0; l <= h->levels; l++) h->mvf[l] = nullptr;
    auto drop = [&](int rc) { for (int l = 0; l <= h->levels; l++) { if (h->mvf[l]) g_op_free(h->mvf[l]); h->mvf[l] = nullptr; } return rc; };
    for (int l = 0; l <= h->levels; l++) {
        h->mvf[l] = (dsvg_mv *)g_op_alloc((int)((size_t)nblk * sizeof(dsvg_mv)));       // (as hme.c:736-741: the caller owns them)
        if (!h->mvf[l]) { dsvg_set_error("out of host memory for the motion field of level %d", l); return drop(DSVG_ERR_NOMEM); }
        if (hipMemcpyAsync(h->mvf[l], A.mvf + (size_t)l * nblk, (size_t)nblk * sizeof(DMV), hipMemcpyDeviceToHost, S.st) != hipSuccess) { dsvg_set_error("copying the motion field back failed"); return drop(DSVG_ERR_HIP); }
    }
    { const int rc_ = S.sync(); if (rc_) return drop(rc_); }
    if (intra_pct) {
        int n = 0;
        for (int i = 0; i < nblk; i++) n += h->mvf[0][i].mode != 0;
        *intra_pct = n * 100 / nblk;
    }
    return DSVG_OK;
}
