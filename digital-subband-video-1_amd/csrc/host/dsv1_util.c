#define _GNU_SOURCE
#include <sched.h>
/* dsv1_util.c -- host glue the drop-in API needs around the GPU hot path: counting allocator
 * (dsv.c:41-96: zeroed, 16-byte header), log level (dsv.c:19-39), DSV_BUF (dsv.c:172-187), host frame
 * containers in the reference layout (frame.c:63-197), planar YUV file access (dsv.c:98-170), the
 * motion-vector predictor (dsv.c:189-231) and the two CLI helpers of util.c.  None of this is hot. */
#include <stdarg.h>
#include <stdio.h>
#include <pthread.h>
#include <unistd.h>
#include "dsv1_host.h"

/* dsv.h:221 of the reference: the logging macros of callers compiled against the reference headers
 * (dsv_main.c) index this table directly */
char *dsv_lvlname[5] = {"NONE", "ERROR", "WARNING", "INFO", "DEBUG"};

static int g_level = 1;
static unsigned g_nalloc, g_nfree, g_balloc, g_bfree;
int dsv1_device = 0;

void dsv_set_log_level(int level) { g_level = level; }
int dsv_get_log_level(void) { return g_level; }
void dsv1_set_device(int device) { dsv1_device = device; }

void dsv1_log(int level, const char *fmt, ...)
{
    va_list ap;
    if (level > g_level) return;
    printf("[DSV][%s] ", dsv_lvlname[level < 0 ? 0 : (level > 4 ? 4 : level)]);
    va_start(ap, fmt);
    vprintf(fmt, ap);
    va_end(ap);
    printf("\n");
}

/* Large blocks are recycled.  A batch of high-rate streams hands the caller a few hundred megabytes of packets per call in
 * fresh buffers and gets them back through dsv_free a call later: as calloc / free that is a page fault per 4 KB on first
 * touch (or a memset of the whole block) and an unmap per buffer, every call -- BASELINE config 2 (intra only, 239 MB of
 * packets per 768 pictures) spent a third of its step there (10.4-10.9 ms against 7.7-8.0 with the allocator told not to give
 * memory back, same box).  Blocks of DSV1_RECYCLE_MIN bytes and more keep their pages: dsv_free parks them by size class (bounded count
 * and bytes), dsv_alloc takes a parked block of at least the size asked for and at most twice that, and zeroes only what was
 * asked for -- or nothing, for the buffers the library fills itself (dsv1_alloc_raw).
 * Blocks are parked only while an encoder batch / session is open (dsv1_recycle_hold, counted by dsv1_batch_open / _close): when the
 * last one closes the parked blocks go back to the system, and a buffer the caller frees after that is freed for good -- a drop-in
 * library must not sit on a gigabyte of a process that has finished encoding (advisor round 4).  DSV1_RECYCLE_MAX_MB bounds the
 * parked bytes (default 1024).
 * Header (16 bytes in front of the data, dsv.c:41-96 has the size there): int32 size as requested, uint32 capacity of the block. */
#define DSV1_RECYCLE_MIN (256u << 10)
#define DSV1_RECYCLE_GRAIN_LOG 18                 /* capacities are multiples of 256 KB: class = capacity / 256 KB - 1 */
#define DSV1_RECYCLE_CLASSES 256                  /* ... up to 64 MB; larger blocks go straight back to the system */
#define DSV1_RECYCLE_DEPTH 96                     /* parked blocks per class */
static struct { pthread_mutex_t mu; size_t bytes; int users; int n[DSV1_RECYCLE_CLASSES]; uint8_t *p[DSV1_RECYCLE_CLASSES][DSV1_RECYCLE_DEPTH]; } g_park = {PTHREAD_MUTEX_INITIALIZER, 0, 0, {0}, {{NULL}}};
static size_t park_max_bytes(void)
{
    static size_t v = 0;
    if (!v) { const char *e = getenv("DSV1_RECYCLE_MAX_MB"); long mb = e ? atol(e) : 1024; if (mb < 1) mb = 1; if (mb > 65536) mb = 65536; v = (size_t)mb << 20; }
    return v;
}

static int park_on(void)                      /* DSV1_NO_RECYCLE=1: plain calloc / free (A/B switch, and for callers that want every byte back at once) */
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("DSV1_NO_RECYCLE"); v = !(e && atoi(e) != 0); }
    return v;
}
/* a parked block of at least `need` bytes (a multiple of the grain) and at most twice that: the first non-empty class from need's own upwards */
static uint8_t *park_take(size_t need)
{
    uint8_t *p = NULL;
    int k = (int)(need >> DSV1_RECYCLE_GRAIN_LOG) - 1, kmax = 2 * (k + 1) - 1;
    if (k < 0 || k >= DSV1_RECYCLE_CLASSES) return NULL;
    if (kmax >= DSV1_RECYCLE_CLASSES) kmax = DSV1_RECYCLE_CLASSES - 1;
    pthread_mutex_lock(&g_park.mu);
    for (; k <= kmax; k++)
        if (g_park.n[k] > 0) { p = g_park.p[k][--g_park.n[k]]; g_park.bytes -= *(uint32_t *)(p + 4); break; }
    pthread_mutex_unlock(&g_park.mu);
    return p;
}
static int park_put(uint8_t *p)
{
    const size_t cap = *(uint32_t *)(p + 4);
    const int k = (int)(cap >> DSV1_RECYCLE_GRAIN_LOG) - 1;
    int ok = 0;
    if (k < 0 || k >= DSV1_RECYCLE_CLASSES || (cap & (((size_t)1 << DSV1_RECYCLE_GRAIN_LOG) - 1))) return 0;
    pthread_mutex_lock(&g_park.mu);
    if (g_park.users > 0 && g_park.n[k] < DSV1_RECYCLE_DEPTH && g_park.bytes + cap <= park_max_bytes()) { g_park.p[k][g_park.n[k]++] = p; g_park.bytes += cap; ok = 1; }
    pthread_mutex_unlock(&g_park.mu);
    return ok;
}
static void *alloc_impl(int size, int zero)
{
    uint8_t *p = NULL;
    size_t cap = (size_t)size;
    if (size < 0) return NULL;
    if (cap >= DSV1_RECYCLE_MIN && cap < 0xfff00000u && park_on()) {
        cap = (cap + (((size_t)1 << DSV1_RECYCLE_GRAIN_LOG) - 1)) & ~(((size_t)1 << DSV1_RECYCLE_GRAIN_LOG) - 1);       /* sizes differ a little from call to call: round, so that blocks fit again */
        p = park_take(cap);
        if (p) { cap = *(uint32_t *)(p + 4); if (zero) memset(p + 16, 0, (size_t)size); }
    }
    if (!p) {
        p = (uint8_t *)(zero ? calloc(1, cap + 16) : malloc(cap + 16));
        if (!p) return NULL;
        if (!zero) memset(p, 0, 16);
    }
    *(int32_t *)p = size;
    *(uint32_t *)(p + 4) = (uint32_t)cap;
    __atomic_fetch_add(&g_nalloc, 1u, __ATOMIC_RELAXED);       /* the batch encoder allocates from worker threads */
    __atomic_fetch_add(&g_balloc, (unsigned)size, __ATOMIC_RELAXED);
    return p + 16;
}
void *dsv_alloc(int size) { return alloc_impl(size, 1); }
/* the same block, contents unspecified: for buffers the library fills itself (a stream's packets) */
void *dsv1_alloc_raw(int size) { return alloc_impl(size, 0); }

void dsv_free(void *ptr)
{
    uint8_t *p;
    if (!ptr) return;
    p = (uint8_t *)ptr - 16;
    __atomic_fetch_add(&g_nfree, 1u, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_bfree, (unsigned)*(int32_t *)p, __ATOMIC_RELAXED);
    if (*(uint32_t *)(p + 4) >= DSV1_RECYCLE_MIN && park_on() && park_put(p)) return;
    free(p);
}
/* give the parked blocks back (a caller that wants its memory; the last batch / session closing) */
static void release_parked_locked(void)
{
    int k;
    for (k = 0; k < DSV1_RECYCLE_CLASSES; k++)
        while (g_park.n[k] > 0) free(g_park.p[k][--g_park.n[k]]);
    g_park.bytes = 0;
}
void dsv1_release_parked(void)
{
    pthread_mutex_lock(&g_park.mu);
    release_parked_locked();
    pthread_mutex_unlock(&g_park.mu);
}
/* +1: an encoder batch / session opens (blocks freed from now on are parked); -1: it closes -- the last one out releases the
 * parked blocks.  Returns the number of holders left. */
int dsv1_recycle_hold(int delta)
{
    int n;
    pthread_mutex_lock(&g_park.mu);
    g_park.users += delta;
    if (g_park.users < 0) g_park.users = 0;
    if (g_park.users == 0) release_parked_locked();
    n = g_park.users;
    pthread_mutex_unlock(&g_park.mu);
    return n;
}
size_t dsv1_parked_bytes(void)
{
    size_t b;
    pthread_mutex_lock(&g_park.mu);
    b = g_park.bytes;
    pthread_mutex_unlock(&g_park.mu);
    return b;
}

void dsv_memory_report(void)
{
    dsv1_log(4, "n alloc: %u  n freed: %u  alloc bytes: %u  freed bytes: %u  not freed: %d",
             g_nalloc, g_nfree, g_balloc, g_bfree, (int)(g_balloc - g_bfree));
}

void dsv_mk_buf(DSV_BUF *buf, int size)
{
    memset(buf, 0, sizeof(*buf));
    buf->data = (unsigned char *)dsv_alloc(size);
    buf->len = (unsigned)size;
}

void dsv_buf_free(DSV_BUF *buf)
{
    if (buf && buf->data) {
        dsv_free(buf->data);
        buf->data = NULL;
    }
}

/* make room for n more bytes (one allocation instead of a chain of doublings) */
int dsv1_buf_reserve(DSV_BUF *b, unsigned n)
{
    unsigned cap = b->data ? (unsigned)*(int32_t *)(b->data - 16) : 0;
    if (b->len + n > cap) {
        unsigned char *nd = (unsigned char *)dsv1_alloc_raw((int)(b->len + n + 64));      /* (filled by the caller: not zeroed) */
        if (!nd) return -1;
        if (b->data) {
            memcpy(nd, b->data, b->len);
            dsv_free(b->data);
        }
        b->data = nd;
    }
    return 0;
}

int dsv1_buf_append(DSV_BUF *b, const uint8_t *src, unsigned n)
{
    unsigned cap = b->data ? (unsigned)*(int32_t *)(b->data - 16) : 0;
    if (b->len + n > cap) {
        unsigned ncap = cap ? cap * 2 : (1u << 16);
        unsigned char *nd;
        while (ncap < b->len + n) ncap *= 2;
        nd = (unsigned char *)dsv1_alloc_raw((int)ncap);
        if (!nd) return -1;
        if (b->data) {
            memcpy(nd, b->data, b->len);
            dsv_free(b->data);
        }
        b->data = nd;
    }
    memcpy(b->data + b->len, src, n);
    b->len += n;
    return 0;
}

/* ---- host frames ---------------------------------------------------------------------------- */
/* Every frame this library hands out is a DSV_FRAME followed by two hidden words: the pinned pool its pixel memory belongs to (the
 * decoder's output frames, dsv1_pool_frame) or NULL (pixel memory from dsv_alloc, or the caller's).  dsv_frame_ref_dec gives a pool
 * frame's buffer back to its pool instead of freeing it. */
typedef struct { DSV_FRAME f; dsv1_frame_pool *pool; int idx; } frame_ext;
struct dsv1_frame_pool {
    int refs;                       /* the session + every frame out (atomic) */
    int n, device;
    size_t bytes;
    uint8_t *buf[DSV1_POOL_FRAMES];
    int busy[DSV1_POOL_FRAMES];     /* (atomic) */
};

static DSV_FRAME *frame_shell(int format, int w, int h)
{
    DSV_FRAME *f = (DSV_FRAME *)dsv_alloc((int)sizeof(frame_ext));      /* zeroed: no pool */
    f->refcount = 1;
    f->format = format;
    f->width = w;
    f->height = h;
    return f;
}

DSV_FRAME *dsv_mk_frame(int format, int width, int height, int border)
{
    DSV_FRAME *f = frame_shell(format, width, height);
    const int ext = border ? DSVG_FRAME_BORDER : 0;
    const int hs = (format >> 2) & 3, vs = format & 3;
    int c, total = 0, off = 0;
    f->border = !!border;
    for (c = 0; c < 3; c++) {
        DSV_PLANE *p = &f->planes[c];
        p->format = format;
        p->w = c ? (width + (1 << hs) - 1) >> hs : width;
        p->h = c ? (height + (1 << vs) - 1) >> vs : height;
        p->hs = c ? hs : 0;
        p->vs = c ? vs : 0;
        p->stride = (p->w + 2 * ext + 15) & ~15;
        p->len = p->stride * (p->h + 2 * ext);
        total += p->len;
    }
    f->alloc = (uint8_t *)dsv_alloc(total);
    for (c = 0; c < 3; c++) {
        DSV_PLANE *p = &f->planes[c];
        p->data = f->alloc + off + p->stride * ext + ext;
        off += p->len;
    }
    return f;
}

DSV_FRAME *dsv_load_planar_frame(int format, void *data, int width, int height)
{
    DSV_FRAME *f = frame_shell(format, width, height);
    const int hs = (format >> 2) & 3, vs = format & 3;
    uint8_t *d = (uint8_t *)data;
    int c;
    for (c = 0; c < 3; c++) {
        DSV_PLANE *p = &f->planes[c];
        p->format = format;
        p->w = c ? (width + (1 << hs) - 1) >> hs : width;
        p->h = c ? (height + (1 << vs) - 1) >> vs : height;
        p->hs = c ? hs : 0;
        p->vs = c ? vs : 0;
        p->stride = p->w;
        p->len = p->stride * p->h;
        p->data = d;
        d += p->len;
    }
    return f;
}

DSV_FRAME *dsv_frame_ref_inc(DSV_FRAME *frame)
{
    frame->refcount++;
    return frame;
}

void dsv_frame_ref_dec(DSV_FRAME *frame)
{
    if (!frame || frame->refcount <= 0) {
        dsv1_log(1, "assert: frame refcount");
        exit(-1);
    }
    if (--frame->refcount == 0) {
        frame_ext *x = (frame_ext *)frame;
        if (x->pool) {                                   /* the pixel memory goes back to the decoder's pinned pool */
            dsv1_frame_pool *pl = x->pool;
            __atomic_store_n(&pl->busy[x->idx], 0, __ATOMIC_RELEASE);
            dsv1_pool_unref(pl);
        } else if (frame->alloc) dsv_free(frame->alloc);
        dsv_free(frame);
    }
}

/* ---- the decoder's output frames: a few buffers of PINNED host memory in the reference's frame layout (frame.c:63-120 -- the layout
 * the device keeps its reconstructions in), so that a decoded picture reaches the caller by ONE asynchronous device-to-host copy of the
 * reconstruction slot: no packing kernel, no staging copy, no row-by-row copy into a freshly zeroed frame (dsv_dec spent most of a
 * call's 700 us there).  A frame the caller still holds keeps its buffer; with every buffer out the decoder falls back to plain frames. */
dsv1_frame_pool *dsv1_pool_new(dsvg_ctx *ctx, int device, size_t bytes, int n)
{
    dsv1_frame_pool *pl = (dsv1_frame_pool *)calloc(1, sizeof(*pl));
    int i;
    if (!pl) return NULL;
    if (n > DSV1_POOL_FRAMES) n = DSV1_POOL_FRAMES;
    pl->refs = 1; pl->device = device; pl->bytes = bytes;
    for (i = 0; i < n; i++) {
        void *p = NULL;
        if (dsvg_host_alloc(ctx, &p, bytes)) break;
        memset(p, 0, bytes);
        pl->buf[pl->n++] = (uint8_t *)p;
    }
    if (!pl->n) { free(pl); return NULL; }
    return pl;
}
void dsv1_pool_unref(dsv1_frame_pool *pl)
{
    int i;
    if (!pl || __atomic_sub_fetch(&pl->refs, 1, __ATOMIC_ACQ_REL) > 0) return;
    for (i = 0; i < pl->n; i++) dsvg_host_free_on(pl->device, pl->buf[i]);
    free(pl);
}
/* a frame in the reference layout (dsv_mk_frame with a border) on a free buffer of the pool, or NULL when every buffer is out */
DSV_FRAME *dsv1_pool_frame(dsv1_frame_pool *pl, int format, int width, int height)
{
    const int ext = DSVG_FRAME_BORDER, hs = (format >> 2) & 3, vs = format & 3;
    DSV_FRAME *f;
    frame_ext *x;
    size_t off = 0;
    int i, c, idx = -1;
    if (!pl) return NULL;
    for (i = 0; i < pl->n && idx < 0; i++) {
        int zero = 0;
        if (__atomic_compare_exchange_n(&pl->busy[i], &zero, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) idx = i;
    }
    if (idx < 0) return NULL;
    f = frame_shell(format, width, height);
    f->border = 1;
    for (c = 0; c < 3; c++) {
        DSV_PLANE *p = &f->planes[c];
        p->format = format;
        p->w = c ? (width + (1 << hs) - 1) >> hs : width;
        p->h = c ? (height + (1 << vs) - 1) >> vs : height;
        p->hs = c ? hs : 0;
        p->vs = c ? vs : 0;
        p->stride = (p->w + 2 * ext + 15) & ~15;
        p->len = p->stride * (p->h + 2 * ext);
        p->data = pl->buf[idx] + off + (size_t)p->stride * ext + ext;
        off += (size_t)p->len;
    }
    if (off > pl->bytes) {                               /* (not this pool's geometry) */
        __atomic_store_n(&pl->busy[idx], 0, __ATOMIC_RELEASE);
        dsv_free(f);
        return NULL;
    }
    f->alloc = pl->buf[idx];
    x = (frame_ext *)f;
    x->pool = pl; x->idx = idx;
    __atomic_add_fetch(&pl->refs, 1, __ATOMIC_ACQ_REL);
    return f;
}

/* ---- planar YUV files ------------------------------------------------------------------------- */
int dsv_yuv_write(FILE *out, int fno, DSV_PLANE *p)
{
    size_t fsz;
    int c, y;
    if (!out || fno < 0) return -1;
    fsz = (size_t)p[0].w * p[0].h + (size_t)p[1].w * p[1].h + (size_t)p[2].w * p[2].h;
    if (fseek(out, (long)(fno * fsz), SEEK_SET)) return -1;
    for (c = 0; c < 3; c++)
        for (y = 0; y < p[c].h; y++)
            if (fwrite(p[c].data + (size_t)y * p[c].stride, (size_t)p[c].w, 1, out) != 1) return -1;
    return 0;
}

int dsv_yuv_read(FILE *in, int fno, uint8_t *o, int w, int h, int subsamp)
{
    size_t npix = (size_t)w * h, chr, off;
    if (!in || fno < 0) return -1;
    switch (subsamp) {
        case DSV_SUBSAMP_444: off = fno * npix * 3; chr = npix; break;
        case DSV_SUBSAMP_422: off = fno * npix * 2; chr = (size_t)(w / 2) * h; break;
        case DSV_SUBSAMP_420:
        case DSV_SUBSAMP_411: off = fno * npix * 3 / 2; chr = npix / 4; break;
        default: dsv1_log(1, "unsupported format"); exit(-1);
    }
    if (fseek(in, (long)off, SEEK_SET)) return -1;
    return fread(o, 1, npix + 2 * chr, in) == npix + 2 * chr ? 0 : -1;
}

/* ---- motion vector predictor -------------------------------------------------------------------- */
static int mvp_pick(int left, int top, int topleft)
{
    const int g = left + top - topleft;
    return abs(g - left) < abs(g - top) ? left : top;
}

void dsv_movec_pred(DSV_MV *vecs, DSV_PARAMS *p, int x, int y, int *px, int *py)
{
    const int W = p->nblocks_h;
    int vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0}, k;
    const DSV_MV *nb[3];
    nb[0] = x > 0 ? &vecs[y * W + x - 1] : NULL;
    nb[1] = y > 0 ? &vecs[(y - 1) * W + x] : NULL;
    nb[2] = (x > 0 && y > 0) ? &vecs[(y - 1) * W + x - 1] : NULL;
    for (k = 0; k < 3; k++)
        if (nb[k] && nb[k]->mode == 0) { vx[k] = nb[k]->u.mv.x; vy[k] = nb[k]->u.mv.y; }
    *px = mvp_pick(vx[0], vx[1], vx[2]);
    *py = mvp_pick(vy[0], vy[1], vy[2]);
}

/* ---- CLI helpers (util.c) ---------------------------------------------------------------------- */
unsigned estimate_bitrate(int quality, int gop, DSV_META *md)
{
    const int fps = (md->fps_num + md->fps_den / 2) / md->fps_den;
    int bpf = 352 * 288 * 3 / 2, ratio;
    if (md->subsamp == DSV_SUBSAMP_444) bpf = 352 * 288 * 3;
    else if (md->subsamp == DSV_SUBSAMP_422) bpf = 352 * 288 * 2;
    if (gop == DSV_GOP_INTRA) bpf *= 4;
    if (md->width < 320 && md->height < 240) bpf /= 4;
    ratio = (((md->width + md->height) / 2) << 8) / 352;
    bpf = bpf * ratio >> 8;
    return (unsigned)(((bpf * fps) / (26 - quality / 4)) * 3 / 2);
}

void conv444to422(DSV_PLANE *s, DSV_PLANE *d)
{
    int x, y;
    for (y = 0; y < s->h; y++) {
        const uint8_t *a = s->data + (size_t)y * s->stride;
        uint8_t *o = d->data + (size_t)y * d->stride;
        for (x = 0; x < s->w; x += 2) {
            const int n = x < s->w - 1 ? x + 1 : s->w - 1;
            o[x >> 1] = (uint8_t)((a[x] + a[n] + 1) >> 1);
        }
    }
}

void conv422to420(DSV_PLANE *s, DSV_PLANE *d)
{
    int x, y;
    for (x = 0; x < s->w; x++)
        for (y = 0; y < s->h; y += 2) {
            const int n = y < s->h - 1 ? y + 1 : s->h - 1;
            d->data[(size_t)d->stride * (y >> 1) + x] =
                (uint8_t)((s->data[(size_t)s->stride * y + x] + s->data[(size_t)s->stride * n + x] + 1) >> 1);
        }
}

/* ---- parallel loop over independent items (side information, packet assembly, packet parsing, job records) ---- */
/* Worker threads of a rank, from the host's cores: up to 12 (measured with the worker pool, 1080p GOP 12, 160 streams: 6 threads
 * 24.6 ms per step, 8: 24.1, 12: 23.9, 16: 24.0), never more than half of this process's share of the cores.  The share:
 * the affinity mask when the launcher made it this rank's own (shard.pin_rank_to_cores exports DSV1_CORES_PINNED=1); else the
 * allowed cores -- the whole host, or a mask narrowed by taskset / a cgroup that ALL ranks of the node run under -- divided by
 * the ranks of the node (one process per GPU: LOCAL_WORLD_SIZE of the launcher). */
int dsv1_host_threads_rule(long online, long allowed, long ranks, int pinned_by_launcher)
{
    long cores, n;
    if (ranks < 1) ranks = 1;
    if (online < 1) online = 1;
    if (allowed < 1 || allowed > online) allowed = online;
    cores = pinned_by_launcher ? allowed : allowed / ranks;
    if (cores < 1) cores = 1;
    /* round 6: a small share is used whole (one core left to the runtime's helper threads) -- with 4 cores the old rule (half of them)
     * left the session layer 2 threads and the GPU idle a fifth of every step (profiles/r06_cpu_starved.txt) */
    n = cores <= 2 ? cores : (cores <= 16 ? cores - 1 : cores / 2);
    if (n > 12) n = 12;
    if (n < 1) n = 1;
    return (int)n;
}
/* the CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cpu.cfs_quota_us): a box can show 256 cores and grant 4 -- twelve busy
 * threads on such a quota are throttled for the rest of every 100 ms period, the GPU with them.  0 = no limit / unknown. */
static long cgroup_cpu_limit(void)
{
    long quota = -1, period = 0;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[64];
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(f);
    } else {
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) { if (fscanf(f, "%ld", &quota) != 1) quota = -1; fclose(f); }
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) { if (fscanf(f, "%ld", &period) != 1) period = 0; fclose(f); }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (quota + period - 1) / period;
}
static int par_threads(int S)
{
    static int n = 0;
    if (!n) {
        const char *e = getenv("DSV1_HOST_THREADS");
        if (e) n = atoi(e);
        else {
            const char *lw = getenv("LOCAL_WORLD_SIZE"), *pin = getenv("DSV1_CORES_PINNED");
            long online = sysconf(_SC_NPROCESSORS_ONLN), allowed = online;
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof(set), &set) == 0) allowed = CPU_COUNT(&set);
            {
                /* a CPU quota below the visible cores is what the process really has; with the launcher's pinning it is still the CONTAINER's (all
                 * ranks'), so it only caps this rank's own share there */
                const long lim = cgroup_cpu_limit(), ranks = lw && atol(lw) > 0 ? atol(lw) : 1;
                if (lim > 0) {
                    if (pin && atoi(pin) != 0) {                     /* this rank's own mask: its share of the container's quota caps it */
                        const long share = lim / ranks > 0 ? lim / ranks : 1;
                        if (share < allowed) allowed = share;
                    } else if (lim < allowed) allowed = lim;        /* everybody's cores: the rule divides by the ranks */
                }
            }
            n = dsv1_host_threads_rule(online, allowed, lw ? atol(lw) : 1, pin && atoi(pin) != 0);
        }
        if (n < 1) n = 1;
        if (n > 64) n = 64;
    }
    return n < S ? n : S;
}
int dsv1_host_threads(void) { return par_threads(1 << 20); }      /* how many threads (the caller's included) a parallel loop of the session layer uses */
/* A pool of workers that lives as long as the process: a fork/join per call spent ~40 us per thread in pthread_create /
 * join, five times per batch.  Items are handed out one at a time (atomic counter): the streams of a batch differ in work.
 * The calling thread works too and waits for the stragglers; calls are serialised (one session thread per process is the
 * rule; a second caller simply waits its turn). */
static struct {
    pthread_mutex_t mu, call;
    pthread_cond_t go, done;
    int nworkers, started;
    unsigned long gen;              /* job generation the workers wait for */
    dsv1_par_fn fn; void *ctx; int S;
    volatile int next;              /* next item to hand out */
    int active;                     /* workers still inside the current job */
    /* ONE background job beside the foreground ones (round 6, dsv1_par_bg_begin / _end): its items are taken by workers that have nothing
     * else to do -- while the session thread waits for the GPU -- and by the session thread itself when it joins; all fields under mu */
    dsv1_par_fn bg_fn; void *bg_ctx; int bg_S, bg_next, bg_done;
    pthread_cond_t bg_cv;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, NULL, NULL, 0, 0, 0,
            NULL, NULL, 0, 0, 0, PTHREAD_COND_INITIALIZER};

/* a forked child has none of the workers: it starts its own on first need */
static void pool_after_fork_child(void)
{
    pthread_mutex_init(&g_pool.mu, NULL); pthread_mutex_init(&g_pool.call, NULL);
    pthread_cond_init(&g_pool.go, NULL); pthread_cond_init(&g_pool.done, NULL); pthread_cond_init(&g_pool.bg_cv, NULL);
    g_pool.started = 0; g_pool.nworkers = 0; g_pool.active = 0; g_pool.gen = 0;
    g_pool.bg_fn = NULL; g_pool.bg_S = g_pool.bg_next = g_pool.bg_done = 0;       /* (the parent's background job is the parent's) */
}
/* take background items until none is left to hand out or a foreground loop this worker has not looked at yet has started (`seen`: the
 * generation it served last) -- foreground loops go first, the session thread waits for them; called and left with mu HELD */
static void pool_run_bg_locked(int tid, unsigned long seen)
{
    while (g_pool.bg_fn && g_pool.bg_next < g_pool.bg_S && g_pool.gen == seen) {
        const int s = g_pool.bg_next++;
        const dsv1_par_fn fn = g_pool.bg_fn;
        void *ctx = g_pool.bg_ctx;
        pthread_mutex_unlock(&g_pool.mu);
        fn(ctx, s, tid);
        pthread_mutex_lock(&g_pool.mu);
        if (++g_pool.bg_done == g_pool.bg_S) pthread_cond_broadcast(&g_pool.bg_cv);
    }
}
static void pool_run_items(int tid)
{
    for (;;) {
        const int s = __sync_fetch_and_add(&g_pool.next, 1);
        if (s >= g_pool.S) break;
        g_pool.fn(g_pool.ctx, s, tid);
    }
}
static void *pool_worker(void *p)
{
    const int tid = (int)(size_t)p;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen && !(g_pool.bg_fn && g_pool.bg_next < g_pool.bg_S)) pthread_cond_wait(&g_pool.go, &g_pool.mu);
        if (g_pool.gen != seen) {
            seen = g_pool.gen;
            if (tid <= g_pool.nworkers) {                /* (else: this job uses fewer workers) */
                pthread_mutex_unlock(&g_pool.mu);
                pool_run_items(tid);
                pthread_mutex_lock(&g_pool.mu);
                if (--g_pool.active == 0) pthread_cond_signal(&g_pool.done);
            }
            continue;                                   /* (a newer foreground job may have come in meanwhile: look again before background work) */
        }
        pool_run_bg_locked(tid, seen);
    }
    return NULL;
}
static void par_for_min(int S, dsv1_par_fn fn, void *ctx, int min_items);
void dsv1_par_for(int S, dsv1_par_fn fn, void *ctx) { par_for_min(S, fn, ctx, 4); }   /* (a wake-up costs ~30 us: small loops stay on the caller) */
/* the same for a FEW LONG items (the coding streams' launch sequences of a call: milliseconds each): two items already go parallel */
void dsv1_par_for_long(int S, dsv1_par_fn fn, void *ctx) { par_for_min(S, fn, ctx, 2); }
/* workers are created on first need and never leave; called with mu held */
static void pool_start_workers_locked(int nthr)
{
    static int atfork_set = 0;
    if (!atfork_set) { atfork_set = 1; pthread_atfork(NULL, NULL, pool_after_fork_child); }
    while (g_pool.started < nthr - 1) {
        pthread_t th;
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        if (pthread_create(&th, &at, pool_worker, (void *)(size_t)(g_pool.started + 1)) != 0) { pthread_attr_destroy(&at); break; }
        pthread_attr_destroy(&at);
        g_pool.started++;
    }
}
/* A BACKGROUND loop over S items (round 6): returns at once; the items are run by workers that have no foreground work -- i.e. while the
 * session thread waits for the GPU or runs serial code -- and whatever is left when dsv1_par_bg_end is called is finished there with the
 * caller's help.  One at a time (a second begin joins the first); fn / ctx must stay valid until the end call.  A host that was short of
 * cores spent every step's packet prefixes (23 core-ms per 320-GOP batch) between the coding enqueue and the fetch, with the GPU waiting. */
void dsv1_par_bg_end(void)
{
    pthread_mutex_lock(&g_pool.mu);
    if (g_pool.bg_fn) {
        while (g_pool.bg_next < g_pool.bg_S) {
            const int s = g_pool.bg_next++;
            const dsv1_par_fn fn = g_pool.bg_fn;
            void *ctx = g_pool.bg_ctx;
            pthread_mutex_unlock(&g_pool.mu);
            fn(ctx, s, 0);
            pthread_mutex_lock(&g_pool.mu);
            g_pool.bg_done++;
        }
        while (g_pool.bg_done < g_pool.bg_S) pthread_cond_wait(&g_pool.bg_cv, &g_pool.mu);
        g_pool.bg_fn = NULL; g_pool.bg_ctx = NULL; g_pool.bg_S = g_pool.bg_next = g_pool.bg_done = 0;
    }
    pthread_mutex_unlock(&g_pool.mu);
}
int dsv1_par_bg_pending(void) { int r; pthread_mutex_lock(&g_pool.mu); r = g_pool.bg_fn != NULL; pthread_mutex_unlock(&g_pool.mu); return r; }
void dsv1_par_bg_begin(int S, dsv1_par_fn fn, void *ctx)
{
    const int nthr = par_threads(S);
    int i;
    dsv1_par_bg_end();
    if (S <= 0 || !fn) return;
    if (nthr <= 1) { for (i = 0; i < S; i++) fn(ctx, i, 0); return; }      /* no workers: now */
    pthread_mutex_lock(&g_pool.mu);
    pool_start_workers_locked(nthr);
    if (g_pool.started < 1) {                            /* (thread creation failed) */
        pthread_mutex_unlock(&g_pool.mu);
        for (i = 0; i < S; i++) fn(ctx, i, 0);
        return;
    }
    g_pool.bg_fn = fn; g_pool.bg_ctx = ctx; g_pool.bg_S = S; g_pool.bg_next = 0; g_pool.bg_done = 0;
    pthread_cond_broadcast(&g_pool.go);
    pthread_mutex_unlock(&g_pool.mu);
}
static void par_for_min(int S, dsv1_par_fn fn, void *ctx, int min_items)
{
    const int nthr = par_threads(S);
    int i;
    if (nthr <= 1 || S < min_items) { for (i = 0; i < S; i++) fn(ctx, i, 0); return; }
    pthread_mutex_lock(&g_pool.call);
    pthread_mutex_lock(&g_pool.mu);
    pool_start_workers_locked(nthr);
    g_pool.fn = fn; g_pool.ctx = ctx; g_pool.S = S; g_pool.next = 0;
    g_pool.nworkers = g_pool.started < nthr - 1 ? g_pool.started : nthr - 1;
    g_pool.active = g_pool.nworkers;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.go);
    pthread_mutex_unlock(&g_pool.mu);
    pool_run_items(0);
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.active > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_pool.call);
}

