/*
 * orc_bmc.c -- ORACLE (test infrastructure): half-pel block motion compensation.
 *
 * Restates bmc.c of the reference: per-block prediction (compensate bmc.c:204-302) with the
 * luma 4-tap (-1,9,9,-1) half-pel filter (hpelL bmc.c:124-174, hpfh/hpfv bmc.c:113-122), the
 * chroma bilinear filter (hpel bmc.c:58-110), intra (sub-)blocks predicted by the mean of the
 * co-located reference pixels (avgval bmc.c:176-189), and the two residual mappings
 * subf bmc.c:43-55 / addf bmc.c:29-41 (dsv_sub_pred bmc.c:318, dsv_add_pred bmc.c:333,
 * dsv_frame_add bmc.c:304).
 */
#include <string.h>
#include "orc.h"

static inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline int tap_h(const uint8_t *p)        { return 9 * (p[0] + p[1]) - (p[-1] + p[2]); }
static inline int tap_v(const uint8_t *p, int s) { return 9 * (p[0] + p[s]) - (p[-s] + p[2 * s]); }

static void luma_block(uint8_t *dst, int ds, const uint8_t *ref, int rs, int xh, int yh, int w, int h)
{
    if (!xh && !yh) {
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * ds, ref + (size_t)y * rs, (size_t)w);
    } else if (!xh) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                dst[(size_t)y * ds + x] = sat8((tap_v(ref + (size_t)y * rs + x, rs) + 8) >> 4);
    } else if (!yh) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                dst[(size_t)y * ds + x] = sat8((tap_h(ref + (size_t)y * rs + x) + 8) >> 4);
    } else {
        /* horizontal taps kept unrounded in 16 bits for rows -1..h+2, then vertical taps */
        int16_t mid[(64 + 4) * 64];
        for (int y = 0; y < h + 4; y++)
            for (int x = 0; x < w; x++)
                mid[y * w + x] = (int16_t)tap_h(ref + (ptrdiff_t)(y - 1) * rs + x);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int16_t *m = mid + y * w + x;
                int c = 9 * (m[w] + m[2 * w]) - (m[0] + m[3 * w]);
                dst[(size_t)y * ds + x] = sat8((c + 128) >> 8);
            }
    }
}

static void chroma_block(uint8_t *dst, int ds, const uint8_t *ref, int rs, int xh, int yh, int w, int h)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *r = ref + (size_t)y * rs;
        uint8_t *d = dst + (size_t)y * ds;
        for (int x = 0; x < w; x++) {
            if (xh && yh)  d[x] = (uint8_t)((r[x] + r[x + 1] + r[x + rs] + r[x + rs + 1] + 2) >> 2);
            else if (xh)   d[x] = (uint8_t)((r[x] + r[x + 1] + 1) >> 1);
            else if (yh)   d[x] = (uint8_t)((r[x] + r[x + rs] + 1) >> 1);
            else           d[x] = r[x];
        }
    }
}

static void fill_mean(uint8_t *dst, int ds, const uint8_t *ref, int rs, int w, int h)
{
    int sum = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) sum += ref[(size_t)y * rs + x];
    int m = sum / (w * h);
    for (int y = 0; y < h; y++) memset(dst + (size_t)y * ds, m, (size_t)w);
}

static void predict_plane(const orc_mv *vecs, const orc_params *p, int c, const orc_frame *ref, orc_plane *dp)
{
    const int sh = c ? ORC_HSHIFT(p->vidmeta->subsamp) : 0;
    const int sv = c ? ORC_VSHIFT(p->vidmeta->subsamp) : 0;
    const int bw = p->blk_w >> sh, bh = p->blk_h >> sv;
    const int limx = dp->w - bw + ORC_BORDER - 1, limy = dp->h - bh + ORC_BORDER - 1;
    const orc_plane *rp = &ref->planes[c];

    for (int j = 0; j < p->nblocks_v; j++) {
        const int y = j * bh;
        const int ch = (y + bh >= dp->h) ? dp->h - y : bh;
        for (int i = 0; i < p->nblocks_h; i++) {
            const int x = i * bw;
            const int cw = (x + bw >= dp->w) ? dp->w - x : bw;
            const orc_mv *mv = &vecs[i + j * p->nblocks_h];
            uint8_t *d = dp->data + x + (ptrdiff_t)y * dp->stride;

            if (mv->mode == 0) {
                int dx = mv->u.mv.x >> sh, dy = mv->u.mv.y >> sv;
                int px = clampi(x + (dx >> 1), -ORC_BORDER, limx);
                int py = clampi(y + (dy >> 1), -ORC_BORDER, limy);
                const uint8_t *r = rp->data + px + (ptrdiff_t)py * rp->stride;
                if (c == 0) luma_block(d, dp->stride, r, rp->stride, dx & 1, dy & 1, cw, ch);
                else        chroma_block(d, dp->stride, r, rp->stride, dx & 1, dy & 1, cw, ch);
            } else if (mv->submask == 0xF) {
                fill_mean(d, dp->stride, rp->data + x + (ptrdiff_t)y * rp->stride, rp->stride, cw, ch);
            } else {
                /* quadrants in mask-bit order TL,TR,BL,BR; each is (cw/2)x(ch/2) (bmc.c:266-294) */
                const int qw = cw / 2, qh = ch / 2;
                for (int k = 0; k < 4; k++) {
                    int ox = (k & 1) ? qw : 0, oy = (k & 2) ? qh : 0;
                    const uint8_t *r = rp->data + (x + ox) + (ptrdiff_t)(y + oy) * rp->stride;
                    uint8_t *q = d + ox + (ptrdiff_t)oy * dp->stride;
                    if (mv->submask & (1 << k))
                        fill_mean(q, dp->stride, r, rp->stride, qw, qh);
                    else
                        for (int r0 = 0; r0 < qh; r0++)
                            memcpy(q + (size_t)r0 * dp->stride, r + (size_t)r0 * rp->stride, (size_t)qw);
                }
            }
        }
    }
}

void orc_sub_pred(const orc_mv *mv, const orc_params *p, orc_frame *dif, orc_frame *inp, const orc_frame *ref)
{
    for (int c = 0; c < 3; c++) {
        orc_plane *d = &dif->planes[c], *i = &inp->planes[c];
        predict_plane(mv, p, c, ref, d);
        for (int y = 0; y < d->h; y++)
            for (int x = 0; x < d->w; x++) {
                uint8_t *v = i->data + (size_t)y * i->stride + x;
                *v = sat8(*v - d->data[(size_t)y * d->stride + x] + 128);
            }
    }
}

void orc_add_pred(const orc_mv *mv, const orc_params *p, orc_frame *dif, orc_frame *out, const orc_frame *ref)
{
    for (int c = 0; c < 3; c++) {
        orc_plane *d = &dif->planes[c], *o = &out->planes[c];
        predict_plane(mv, p, c, ref, o);
        for (int y = 0; y < o->h; y++)
            for (int x = 0; x < o->w; x++) {
                uint8_t *v = o->data + (size_t)y * o->stride + x;
                *v = sat8(*v + d->data[(size_t)y * d->stride + x] - 128);
            }
    }
}

void orc_frame_add(orc_frame *dst, const orc_frame *src)
{
    for (int c = 0; c < 3; c++) {
        const orc_plane *s = &src->planes[c];
        orc_plane *d = &dst->planes[c];
        for (int y = 0; y < d->h; y++)
            for (int x = 0; x < d->w; x++) {
                uint8_t *v = d->data + (size_t)y * d->stride + x;
                *v = sat8(*v + s->data[(size_t)y * s->stride + x] - 128);
            }
    }
}
