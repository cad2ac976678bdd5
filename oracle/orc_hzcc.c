/*
 * orc_hzcc.c -- ORACLE (test infrastructure): adaptive dead-zone quantisation and the
 * "hierarchical zero coefficient coder" of DSV1.
 *
 * Restates hzcc.c of the reference as a table-driven scan: the ten scan regions ("LL", then
 * level 0,1,2 x LH,HL,HH) are enumerated once (hzcc.c:30-48 geometry) and one loop walks
 * them for both the encoder (hzcc_enc hzcc.c:137-293) and the decoder (hzcc_dec
 * hzcc.c:295-435).  Regions may overlap for some plane sizes (e.g. 960x540);
 * walking them sequentially reproduces the reference's double processing.
 *
 *   quantiser derivation  dsv_get_quant hzcc.c:77-92, tmq4pos hzcc.c:64-74,
 *                         fix_quant hzcc.c:50-57, highest level shift hzcc.c:204-212
 *   quant/dequant         hzcc.c:94-135
 *   plane framing         dsv_encode_plane hzcc.c:449-476, dsv_decode_plane hzcc.c:479-496
 */
#include <limits.h>
#include "orc.h"

#define MINQ 16            /* MINQUANT hzcc.c:27 */
#define CHROMA_CAP 512     /* CHROMA_LIMIT hzcc.c:23 */
#define BLK_FP 14          /* BLOCK_P hzcc.c:59 */
#define EOP 0x55

int orc_lb2(unsigned n)
{
    unsigned i = 1;
    int l = 0;
    while (i < n) { i <<= 1; l++; }
    return l;
}

int orc_get_quant(int q, int isP, int level)
{
    if (isP) q = q * 3 / 2;
    if (level == 1) q = q * 2 / 3;
    else if (level == 2) q = q * 3 / 2;
    return q < MINQ ? MINQ : q;
}

static inline int q_lo(int v, int q)             /* quant hzcc.c:94-112 */
{
    int m = (v < 0 ? -v : v) << 1;
    if (m <= q) return 0;
    m = (m + 1) / (q << 1);
    return v < 0 ? -m : m;
}
static inline int dq_lo(int v, int q)            /* dequant hzcc.c:121-128 */
{
    return v < 0 ? -((-v * (q << 1) + q) >> 1) : (v * (q << 1) + q) >> 1;
}
static inline int q_hi(int v, int sh)            /* quantH hzcc.c:114-118 */
{
    return v < 0 ? -((-v) >> sh) : v >> sh;
}
static inline int dq_hi(int v, int sh)           /* dequantH hzcc.c:131-135 */
{
    return (int)((unsigned)v << sh);
}

typedef struct {
    int x0, y0, sw, sh;     /* rectangle inside the w x h coefficient plane */
    int level;              /* -1 = "LL" region, else 0..2 */
    int qp;                 /* base quantiser (level 2: shift) */
    int qp_h;               /* level 2: shift for flagged blocks */
    int dbx, dby;           /* 14-bit fixed point block steps (hzcc.c:196-197) */
} region;

static int build_regions(region r[10], int w, int h, int q, const orc_stability *st)
{
    int n = 0;
    if (st->cur_plane > 0 && q > CHROMA_CAP) q = CHROMA_CAP;

    r[n].x0 = r[n].y0 = 0;
    r[n].sw = ORC_RSHIFT_UP(w, ORC_MAXLVL);
    r[n].sh = ORC_RSHIFT_UP(h, ORC_MAXLVL);
    r[n].level = -1;
    r[n].qp = orc_get_quant(q, st->isP, 0);
    r[n].qp_h = 0; r[n].dbx = r[n].dby = 0;
    n++;
    for (int l = 0; l < ORC_MAXLVL; l++) {
        int sw = ORC_RSHIFT_UP(w, ORC_MAXLVL - l), sh = ORC_RSHIFT_UP(h, ORC_MAXLVL - l);
        int qp = orc_get_quant(q, st->isP, l), qp_h = 0;
        if (l == ORC_MAXLVL - 1) {
            qp = orc_lb2((unsigned)qp);
            qp_h = qp - (st->isP ? 1 : 3);
            qp_h = qp_h < 1 ? 1 : (qp_h > 24 ? 24 : qp_h);
        }
        for (int s = 1; s < 4; s++, n++) {
            r[n].x0 = (s & 1) ? sw : 0;
            r[n].y0 = (s & 2) ? sh : 0;
            r[n].sw = sw; r[n].sh = sh;
            r[n].level = l; r[n].qp = qp; r[n].qp_h = qp_h;
            r[n].dbx = (st->params->nblocks_h << BLK_FP) / sw;
            r[n].dby = (st->params->nblocks_v << BLK_FP) / sh;
        }
    }
    return n;
}

/* effective quantiser for a cell of region r at (x,y) inside the region */
static inline int cell_quant(const region *r, const orc_stability *st, int x, int y)
{
    if (r->level < 0) return r->qp;
    unsigned char flag = st->stable_blocks[((y * r->dby) >> BLK_FP) * st->params->nblocks_h
                                           + ((x * r->dbx) >> BLK_FP)];
    if (r->level == ORC_MAXLVL - 1)
        return flag ? r->qp_h : r->qp;
    int t = (flag & 2) ? r->qp >> 2 : (flag ? r->qp >> 1 : r->qp);
    return t < MINQ ? MINQ : t;
}

static void hzcc_write(orc_bs *bs, int32_t *src, int w, int h, int q, const orc_stability *st)
{
    region reg[10];
    int nreg, run = 0, nruns = 0, held = 0;
    unsigned count_at;

    orc_bs_align(bs);
    count_at = bs->pos;
    orc_bs_put_bits(bs, 32, 0);
    orc_bs_align(bs);

    nreg = build_regions(reg, w, h, q, st);
    src[0] = 0;                                   /* DC travels separately */

    for (int ri = 0; ri < nreg; ri++) {
        const region *r = &reg[ri];
        const int hi = (r->level == ORC_MAXLVL - 1);
        for (int y = 0; y < r->sh; y++) {
            int32_t *row = src + (size_t)(r->y0 + y) * w + r->x0;
            for (int x = 0; x < r->sw; x++) {
                int tq = cell_quant(r, st, x, y);
                int v = hi ? q_hi(row[x], tq) : q_lo(row[x], tq);
                if (v) {
                    row[x] = hi ? dq_hi(v, tq) : dq_lo(v, tq);
                    orc_bs_put_ueg(bs, (unsigned)run);
                    if (held) orc_bs_put_neg(bs, held);   /* value k-1 follows run k */
                    held = v;
                    nruns++;
                    run = 0;
                } else {
                    row[x] = 0;
                    run++;
                }
            }
        }
    }
    if (held) orc_bs_put_neg(bs, held);
    orc_bs_align(bs);

    unsigned end = bs->pos;
    bs->pos = count_at;
    orc_bs_put_bits(bs, 32, (unsigned)nruns);
    bs->pos = end;
}

static void hzcc_read(orc_bs *bs, unsigned bufsz, int32_t *out, int w, int h, int q, const orc_stability *st)
{
    region reg[10];
    int nreg, runs, run;

    orc_bs_align(bs);
    runs = (int)orc_bs_get_bits(bs, 32);
    orc_bs_align(bs);
    run = (runs-- > 0) ? (int)orc_bs_get_ueg(bs) : INT_MAX;

    nreg = build_regions(reg, w, h, q, st);
    for (int ri = 0; ri < nreg; ri++) {
        const region *r = &reg[ri];
        const int hi = (r->level == ORC_MAXLVL - 1);
        for (int y = 0; y < r->sh; y++) {
            int32_t *row = out + (size_t)(r->y0 + y) * w + r->x0;
            for (int x = 0; x < r->sw; x++) {
                if (run-- != 0) continue;
                run = (runs-- > 0) ? (int)orc_bs_get_ueg(bs) : INT_MAX;
                int v = orc_bs_get_neg(bs);
                if (orc_bs_bytepos(bs) >= bufsz) return;       /* hzcc.c:337-339 */
                int tq = cell_quant(r, st, x, y);
                row[x] = hi ? dq_hi(v, tq) : dq_lo(v, tq);
            }
        }
    }
    orc_bs_align(bs);
}

void orc_encode_plane(orc_bs *bs, orc_coefs *src, int q, const orc_stability *stab)
{
    int32_t *d = src->data;
    orc_bs_align(bs);
    unsigned start = orc_bs_bytepos(bs);
    orc_bs_put_bits(bs, 32, 0);
    int dc = d[0];
    orc_bs_put_seg(bs, dc);
    hzcc_write(bs, d, src->width, src->height, q, stab);
    d[0] = dc;
    orc_bs_put_bits(bs, 8, EOP);
    orc_bs_align(bs);
    unsigned end = orc_bs_bytepos(bs);
    bs->pos = start * 8;
    orc_bs_put_bits(bs, 32, end - start - 4);
    bs->pos = end * 8;
}

void orc_decode_plane(uint8_t *in, unsigned len, orc_coefs *dst, int q, const orc_stability *stab)
{
    orc_bs bs;
    orc_bs_init(&bs, in);
    int dc = orc_bs_get_seg(&bs);
    hzcc_read(&bs, len, dst->data, dst->width, dst->height, q, stab);
    (void)orc_bs_get_bits(&bs, 8);     /* 0x55 end-of-plane marker (reference only logs a mismatch) */
    dst->data[0] = dc;
}
