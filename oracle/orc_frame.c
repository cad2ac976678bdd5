/*
 * orc_frame.c -- ORACLE (test infrastructure): frame containers and border handling.
 *
 * The memory layout is load-bearing: one zeroed allocation, planes Y,U,V
 * back to back, a 64-pixel border on every side of every plane, stride = round16(w + 128)
 * (dsv_mk_frame frame.c:63-120).  Motion search / compensation address up to one pixel
 * beyond the border, which lands in the neighbouring row or plane; identical layout gives
 * identical bytes.
 *
 *   dsv_mk_coefs frame.c:29-61, dsv_load_planar_frame frame.c:122-164,
 *   dsv_frame_copy frame.c:199-221, dsv_frame_avg_luma frame.c:223-238,
 *   dsv_ds2x_frame_luma frame.c:240-261, dsv_extend_frame(_luma) frame.c:263-327
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define HDR 16   /* the reference allocator keeps a 16-byte zeroed-ish header in front (dsv.c:47-56) */

static int round_up_pow2(int x, int p) { return (x + (1 << p) - 1) & ~((1 << p) - 1); }

orc_frame *orc_frame_new(int format, int w, int h, int border)
{
    orc_frame *f = (orc_frame *)calloc(1, sizeof(*f));
    const int ext = border ? ORC_BORDER : 0;
    const int cw = ORC_RSHIFT_UP(w, ORC_HSHIFT(format));
    const int ch = ORC_RSHIFT_UP(h, ORC_VSHIFT(format));
    size_t off = 0;

    f->refcount = 1;
    f->format = format;
    f->width = w;
    f->height = h;
    f->border = !!border;
    for (int c = 0; c < 3; c++) {
        orc_plane *p = &f->planes[c];
        p->format = format;
        p->w = c ? cw : w;
        p->h = c ? ch : h;
        p->hs = c ? ORC_HSHIFT(format) : 0;
        p->vs = c ? ORC_VSHIFT(format) : 0;
        p->stride = round_up_pow2(p->w + 2 * ext, 4);
        p->len = p->stride * (p->h + 2 * ext);
        off += (size_t)p->len;
    }
    /* header in front mimics the reference allocator so that "row -65" style reads stay in bounds */
    uint8_t *base = (uint8_t *)calloc(1, off + HDR + 64);
    f->alloc = base + HDR;
    off = 0;
    for (int c = 0; c < 3; c++) {
        orc_plane *p = &f->planes[c];
        p->data = f->alloc + off + (size_t)p->stride * ext + ext;
        off += (size_t)p->len;
    }
    return f;
}

void orc_frame_free(orc_frame *f)
{
    if (!f) return;
    if (f->alloc) free(f->alloc - HDR);
    free(f);
}

void orc_frame_wrap_planar(orc_frame *f, int format, uint8_t *data, int w, int h)
{
    memset(f, 0, sizeof(*f));
    f->refcount = 1;
    f->format = format;
    f->width = w;
    f->height = h;
    const int cw = ORC_RSHIFT_UP(w, ORC_HSHIFT(format));
    const int ch = ORC_RSHIFT_UP(h, ORC_VSHIFT(format));
    uint8_t *p = data;
    for (int c = 0; c < 3; c++) {
        orc_plane *pl = &f->planes[c];
        pl->format = format;
        pl->w = c ? cw : w;
        pl->h = c ? ch : h;
        pl->stride = pl->w;
        pl->len = pl->stride * pl->h;
        pl->hs = c ? ORC_HSHIFT(format) : 0;
        pl->vs = c ? ORC_VSHIFT(format) : 0;
        pl->data = p;
        p += pl->len;
    }
}

static void extend_plane(orc_plane *c)
{
    const int w = c->w, h = c->h;
    const size_t span = (size_t)w + 2 * ORC_BORDER;
    for (int y = 0; y < h; y++) {
        uint8_t *line = c->data + (size_t)y * c->stride;
        memset(line - ORC_BORDER, line[0], ORC_BORDER);
        memset(line + w, line[w - 1], ORC_BORDER);
    }
    const uint8_t *top = c->data - ORC_BORDER;
    const uint8_t *bot = c->data + (size_t)(h - 1) * c->stride - ORC_BORDER;
    for (int j = 1; j <= ORC_BORDER; j++) {
        memcpy(c->data - (size_t)j * c->stride - ORC_BORDER, top, span);
        memcpy(c->data + (size_t)(h - 1 + j) * c->stride - ORC_BORDER, bot, span);
    }
}

void orc_frame_extend(orc_frame *f)
{
    if (!f->border) return;
    for (int c = 0; c < 3; c++) extend_plane(&f->planes[c]);
}

void orc_frame_extend_luma(orc_frame *f)
{
    if (!f->border) return;
    extend_plane(&f->planes[0]);
}

void orc_frame_copy(orc_frame *dst, const orc_frame *src)
{
    /* copies src->w bytes of dst->h rows (frame.c:210-214), then re-extends */
    for (int c = 0; c < 3; c++) {
        const orc_plane *s = &src->planes[c];
        orc_plane *d = &dst->planes[c];
        for (int y = 0; y < d->h; y++)
            memcpy(d->data + (size_t)y * d->stride, s->data + (size_t)y * s->stride, (size_t)s->w);
    }
    orc_frame_extend(dst);
}

int orc_frame_avg_luma(const orc_frame *f)
{
    const orc_plane *p = &f->planes[0];
    int acc = 0;
    for (int y = 0; y < p->h; y++) {
        const uint8_t *line = p->data + (size_t)y * p->stride;
        for (int x = 0; x < p->w; x++) acc += line[x];
    }
    return acc / (p->w * p->h);
}

void orc_frame_ds2x_luma(orc_frame *dst, const orc_frame *src)
{
    const orc_plane *s = &src->planes[0];
    orc_plane *d = &dst->planes[0];
    for (int y = 0; y < d->h; y++) {
        const uint8_t *a = s->data + (size_t)(2 * y) * s->stride;
        const uint8_t *b = a + s->stride;
        uint8_t *o = d->data + (size_t)y * d->stride;
        for (int x = 0; x < d->w; x++)
            o[x] = (uint8_t)((a[2 * x] + a[2 * x + 1] + b[2 * x] + b[2 * x + 1] + 2) >> 2);
    }
}

void orc_coefs_new(orc_coefs c[3], int format, int w, int h)
{
    int cw = round_up_pow2(ORC_RSHIFT_UP(w, ORC_HSHIFT(format)), 1);
    int ch = round_up_pow2(ORC_RSHIFT_UP(h, ORC_VSHIFT(format)), 1);
    size_t n0 = (size_t)w * h, n1 = (size_t)cw * ch;
    c[0].width = w;  c[0].height = h;
    c[1].width = cw; c[1].height = ch;
    c[2].width = cw; c[2].height = ch;
    c[0].data = (int32_t *)calloc(n0 + 2 * n1, sizeof(int32_t));
    c[1].data = c[0].data + n0;
    c[2].data = c[1].data + n1;
}
