/* dsv1_host.h -- private helpers of the C session layer: MSB-first bit writer/reader with the
 * interleaved exp-Golomb codes (bs.c:28-267 semantics: the writer ORs into a zeroed buffer) and a
 * growable output buffer.  Host-only scalar code for headers and side information; the coefficient
 * payloads themselves are packed on the GPU. */
#ifndef DSV1_HOST_H
#define DSV1_HOST_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/dsv1_api.h"

/* `end` bounds the READER (bits): past it every bit reads as 1 -- which ends any exp-Golomb prefix -- and `over` is set,
 * so a truncated or hostile packet can neither run the reader off its buffer nor loop; the writer ignores it */
typedef struct { uint8_t *p; unsigned pos, end; int over; } bitw;

static inline void bw_init(bitw *b, uint8_t *buf) { b->p = buf; b->pos = 0; b->end = ~0u; b->over = 0; }
static inline void br_init(bitw *b, uint8_t *buf, unsigned nbytes) { b->p = buf; b->pos = 0; b->end = nbytes > 0x1fffffffu ? ~0u : nbytes * 8u; b->over = 0; }
static inline void bw_align(bitw *b) { b->pos = (b->pos + 7u) & ~7u; }
static inline unsigned bw_bytes(const bitw *b) { return b->pos >> 3; }
static inline void bw_bit(bitw *b, unsigned one)
{
    if (one) b->p[b->pos >> 3] |= (uint8_t)(0x80u >> (b->pos & 7));
    b->pos++;
}
static inline void bw_bits(bitw *b, int n, unsigned v) { while (n--) bw_bit(b, (v >> n) & 1u); }
static inline void bw_ueg(bitw *b, unsigned v)
{
    const unsigned m = v + 1;
    int k = 31 - __builtin_clz(m);
    while (k-- > 0) { b->pos++; bw_bit(b, (m >> k) & 1u); }
    bw_bit(b, 1);
}
static inline void bw_seg(bitw *b, int v)
{
    const unsigned m = v < 0 ? (unsigned)-v : (unsigned)v;
    bw_ueg(b, m);
    if (m) bw_bit(b, v < 0);
}
static inline void bw_bytes_in(bitw *b, const uint8_t *src, unsigned n)
{
    memcpy(b->p + (b->pos >> 3), src, n);
    b->pos += n * 8u;
}

static inline unsigned br_bit(bitw *b)
{
    unsigned v;
    if (b->pos >= b->end) { b->over = 1; return 1u; }
    v = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1u;
    b->pos++;
    return v;
}
static inline unsigned br_bits(bitw *b, int n) { unsigned v = 0; while (n--) v = (v << 1) | br_bit(b); return v; }
static inline unsigned br_ueg(bitw *b) { unsigned m = 1; int k = 0; while (!br_bit(b) && k++ < 32) m = (m << 1) | br_bit(b); return m - 1; }
static inline int br_seg(bitw *b) { int v = (int)br_ueg(b); return (v && br_bit(b)) ? -v : v; }

/* zero-bit run-length coder (bs.c:222-267) */
typedef struct { bitw b; int nz; } zrle;
static inline void zr_init(zrle *z, uint8_t *buf) { bw_init(&z->b, buf); z->nz = 0; }
static inline void zr_init_rd(zrle *z, uint8_t *buf, unsigned nbytes) { br_init(&z->b, buf, nbytes); z->nz = 0; }
static inline void zr_put(zrle *z, int bit) { if (bit) { bw_ueg(&z->b, (unsigned)z->nz); z->nz = 0; } else z->nz++; }
static inline int zr_end(zrle *z) { bw_ueg(&z->b, (unsigned)z->nz); z->nz = 0; bw_align(&z->b); return (int)bw_bytes(&z->b); }
static inline int zr_get(zrle *z)
{
    if (z->nz == 0) z->nz = (int)br_ueg(&z->b); else z->nz--;
    return z->nz == 0;
}

static inline void put_be32(uint8_t *p, unsigned v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}
static inline unsigned get_be32(const uint8_t *p)
{
    return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3];
}

/* append n bytes to a dsv_alloc-backed growing DSV_BUF (capacity kept in a hidden word before data) */
int  dsv1_buf_append(DSV_BUF *b, const uint8_t *src, unsigned n);
int  dsv1_buf_reserve(DSV_BUF *b, unsigned n);
/* dsv_alloc without the zeroing (a buffer the library fills itself); freed with dsv_free like any other */
void *dsv1_alloc_raw(int size);
/* the decoder's pinned output frames (dsv1_util.c) */
#define DSV1_POOL_FRAMES 6
typedef struct dsv1_frame_pool dsv1_frame_pool;
dsv1_frame_pool *dsv1_pool_new(dsvg_ctx *ctx, int device, size_t bytes, int n);
void dsv1_pool_unref(dsv1_frame_pool *pl);
DSV_FRAME *dsv1_pool_frame(dsv1_frame_pool *pl, int format, int width, int height);
int dsv1_recycle_hold(int delta);       /* dsv1_util.c: +1 a batch opens, -1 it closes (the last one out releases the parked blocks) */
void dsv1_log(int level, const char *fmt, ...);
extern int dsv1_device;
/* parallel loop over S independent items on a persistent worker pool: fn(ctx, s, worker) for every s; DSV1_HOST_THREADS workers (default min(12,
 * half of this process's share of the cores)) */
typedef void (*dsv1_par_fn)(void *ctx, int s, int tid);
void dsv1_par_for(int S, dsv1_par_fn fn, void *ctx);
void dsv1_par_for_long(int S, dsv1_par_fn fn, void *ctx);      /* the same for a few long items (two already run in parallel) */
/* ONE background loop beside the foreground ones (round 6): begin returns at once, idle workers take the items (foreground loops go first), end joins
 * with the caller's help; a second begin joins the first.  fn / ctx must stay valid until the end call. */
void dsv1_par_bg_begin(int S, dsv1_par_fn fn, void *ctx);
void dsv1_par_bg_end(void);
int dsv1_par_bg_pending(void);
/* the default size of that pool as a function of the host (dsv1_util.c); exported so that the rule can be tested without the host it is for */
int dsv1_host_threads_rule(long online, long allowed, long ranks, int pinned_by_launcher);

#define CLAMPI(v, lo, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))

#endif
