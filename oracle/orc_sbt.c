/*
 * orc_sbt.c -- ORACLE (test infrastructure): the DSV1 subband transform.
 *
 * Restates sbt.c of the reference with a different decomposition (one generic 2x2-cell
 * Haar evaluator with mirrored/zeroed missing samples instead of four hand-unrolled edge
 * loops; one strided 1-D B4T for rows and columns; per-call scratch instead of a static
 * temp buffer) -- results are byte-identical, see tests/test_oracle_vs_ref.py.
 *
 *   forward  : dsv_fwd_sbt sbt.c:630-651, p2sbc sbt.c:576-592, fwd sbt.c:268-349,
 *              fwd_b4t_{h,v,2d} sbt.c:91-126,166-201,240-251
 *   inverse  : dsv_inv_sbt sbt.c:654-714, inv sbt.c:438-574, inv_simple sbt.c:352-435,
 *              inv_b4t_{h,v,2d} sbt.c:129-163,204-238,253-265, sbc2int sbt.c:595-614
 *   rounding : round2/4/8 sbt.c:63-88; LL scaling x*4/5 and x*5/4 sbt.c:20-21 (C division
 *              truncates toward zero -- load-bearing)
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

static inline int rdiv2(int v) { return v < 0 ? -((1 - v) >> 1) : (v + 1) >> 1; }
static inline int rdiv4(int v) { return v < 0 ? -((2 - v) >> 2) : (v + 2) >> 2; }
static inline int rdiv8(int v) { return v < 0 ? -((4 - v) >> 3) : (v + 4) >> 3; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline int ll_down(int x) { return x * 4 / 5; }   /* FWD_SCALE */
static inline int ll_up(int x)   { return x * 5 / 4; }   /* INV_SCALE */

static int num_levels(int w, int h)                      /* sbt.c:617-628 */
{
    return orc_lb2((unsigned)(w > h ? w : h));
}

static void copy_region(int32_t *dst, const int32_t *src, int w, int h, int stride)
{
    for (int y = 0; y < h; y++)
        memcpy(dst + (size_t)y * stride, src + (size_t)y * stride, (size_t)w * sizeof(int32_t));
}

/* ---- Haar ---------------------------------------------------------------------------- */

/* One forward level.  A missing right/bottom sample (odd region size) behaves exactly like a
 * mirrored one in the reference's edge formulas (sbt.c:309-346); the subbands that do not
 * exist for that cell are simply not stored. */
static void haar_fwd_level(int32_t *buf, int32_t *tmp, int W, int H, int lvl, int scaled)
{
    const int ws = ORC_RSHIFT_UP(W, lvl - 1), hs = ORC_RSHIFT_UP(H, lvl - 1);
    const int wo = ORC_RSHIFT_UP(W, lvl), ho = ORC_RSHIFT_UP(H, lvl);

    for (int cy = 0; cy < ho; cy++) {
        const int y = 2 * cy, hasB = (y + 1 < hs);
        const int32_t *r0 = buf + (size_t)y * W;
        const int32_t *r1 = r0 + W;
        for (int cx = 0; cx < wo; cx++) {
            const int x = 2 * cx, hasR = (x + 1 < ws);
            int a = r0[x];
            int b = hasR ? r0[x + 1] : a;
            int c = hasB ? r1[x] : a;
            int d = hasB ? (hasR ? r1[x + 1] : c) : b;
            int ll = a + b + c + d;
            tmp[(size_t)cy * W + cx] = scaled ? ll_down(ll) : ll;
            if (hasR)
                tmp[(size_t)cy * W + wo + cx] = a - b + c - d;           /* LH */
            if (hasB)
                tmp[(size_t)(ho + cy) * W + cx] = a + b - c - d;         /* HL */
            if (hasR && hasB)
                tmp[(size_t)(ho + cy) * W + wo + cx] = a - b - c + d;    /* HH */
        }
    }
    copy_region(buf, tmp, ws, hs, W);
}

/* smoothing nudge of sbt.c:480-527: lp/ln are the LL values before/after the cell */
static inline int nudge_detail(int ll, int lp, int ln, int det, int hqp)
{
    int mx = ll - ln, mn = lp - ll;
    if (mn > mx) { int t = mn; mn = mx; mx = t; }
    if (mx > 0) mx = 0;
    if (mn < 0) mn = 0;
    if (mx != mn) {
        int t = rdiv4(lp - ln);
        int n = rdiv2(clampi(t, mx, mn) - (det << 1));
        det += clampi(n, -hqp, hqp);
    }
    return det;
}

/* One inverse level.  filt selects the luma "filtered" variant (sbt.c:438) over the plain one
 * (sbt.c:352).  Missing subbands count as zero; the filter only touches complete cells and
 * never the first/last complete cell of a row/column (inX/inY, sbt.c:463,468). */
static void haar_inv_level(int32_t *buf, int32_t *tmp, int W, int H, int lvl, int scaled, int filt, int hqp)
{
    const int ws = ORC_RSHIFT_UP(W, lvl - 1), hs = ORC_RSHIFT_UP(H, lvl - 1);
    const int wo = ORC_RSHIFT_UP(W, lvl), ho = ORC_RSHIFT_UP(H, lvl);
    const int wfull = ws & ~1, hfull = hs & ~1;     /* extent covered by complete cells */

    for (int cy = 0; cy < ho; cy++) {
        const int y = 2 * cy, hasB = (y + 1 < hs);
        const int32_t *pLL = buf + (size_t)cy * W;
        const int32_t *pHL = buf + (size_t)(ho + cy) * W;
        for (int cx = 0; cx < wo; cx++) {
            const int x = 2 * cx, hasR = (x + 1 < ws);
            int LL = scaled ? ll_up(pLL[cx]) : pLL[cx];
            int LH = hasR ? pLL[wo + cx] : 0;
            int HL = hasB ? pHL[cx] : 0;
            int HH = (hasR && hasB) ? pHL[wo + cx] : 0;

            if (filt && hasR && hasB) {
                if (x > 0 && x < wfull - 1) {
                    int lp = pLL[cx - 1], ln = pLL[cx + 1];
                    if (scaled) { lp = ll_up(lp); ln = ll_up(ln); }
                    LH = nudge_detail(LL, lp, ln, LH, hqp);
                }
                if (y > 0 && y < hfull - 1) {
                    int lp = pLL[cx - W], ln = pLL[cx + W];
                    if (scaled) { lp = ll_up(lp); ln = ll_up(ln); }
                    HL = nudge_detail(LL, lp, ln, HL, hqp);
                }
            }
            int32_t *o0 = tmp + (size_t)y * W + x;
            o0[0] = (LL + LH + HL + HH) / 4;
            if (hasR) o0[1] = (LL - LH + HL - HH) / 4;
            if (hasB) {
                o0[W] = (LL + LH - HL - HH) / 4;
                if (hasR) o0[W + 1] = (LL - LH - HL + HH) / 4;
            }
        }
    }
    copy_region(buf, tmp, ws, hs, W);
}

/* ---- biorthogonal 4-tap (1,3,3,1) level, intra frames level 1 only -------------------- */

/* n must be even (the reference leaves stale temp words for odd n: undefined, excluded) */
static void b4t_fwd_1d(int32_t *out, const int32_t *in, int n, int s)
{
    const int half = n >> 1;
    for (int k = 0; k < half; k++) {
        int im1 = 2 * k - 1, ip2 = 2 * k + 2;
        if (im1 < 0) im1 = 1;                 /* left: mirror  (sbt.c:97-100) */
        if (ip2 > n - 1) ip2 = n - 1;         /* right: clamp  (sbt.c:119-121) */
        int xm = in[(size_t)im1 * s], x0 = in[(size_t)(2 * k) * s];
        int x1 = in[(size_t)(2 * k + 1) * s], xp = in[(size_t)ip2 * s];
        out[(size_t)k * s] = rdiv2(3 * x0 + 3 * x1 - xm - xp);
        out[(size_t)(half + k) * s] = rdiv2(xm - 3 * x0 + 3 * x1 - xp);
    }
}

static void b4t_inv_1d(int32_t *out, const int32_t *in, int n, int s)
{
    const int half = n >> 1;
    for (int m = 0; m < half; m++) {
        int mp = m > 0 ? m - 1 : 0;
        int mn = m < half - 1 ? m + 1 : half - 1;
        int Lp = in[(size_t)mp * s], L = in[(size_t)m * s], Ln = in[(size_t)mn * s];
        int Hp = in[(size_t)(half + mp) * s], Hc = in[(size_t)(half + m) * s], Hn = in[(size_t)(half + mn) * s];
        out[(size_t)(2 * m) * s] = rdiv8(Lp + 3 * L + Hp - 3 * Hc);
        out[(size_t)(2 * m + 1) * s] = rdiv8(3 * L + Ln + 3 * Hc - Hn);
    }
}

static void b4t_fwd_2d(int32_t *buf, int32_t *tmp, int w, int h)
{
    for (int y = 0; y < h; y++) b4t_fwd_1d(tmp + (size_t)y * w, buf + (size_t)y * w, w, 1);
    for (int x = 0; x < w; x++) b4t_fwd_1d(buf + x, tmp + x, h, w);
}

static void b4t_inv_2d(int32_t *buf, int32_t *tmp, int w, int h)
{
    for (int x = 0; x < w; x++) b4t_inv_1d(tmp + x, buf + x, h, w);
    for (int y = 0; y < h; y++) b4t_inv_1d(buf + (size_t)y * w, tmp + (size_t)y * w, w, 1);
}

/* ---- drivers --------------------------------------------------------------------------- */

void orc_fwd_sbt(const orc_plane *src, orc_coefs *dst, int isP)
{
    const int w = dst->width, h = dst->height;
    int32_t *d = dst->data;

    /* p2sbc: only p->h rows are filled (a rounded-up last chroma row stays as allocated);
     * dc->width columns are read, so an odd chroma width reads the first border byte */
    for (int y = 0; y < src->h; y++) {
        const uint8_t *line = src->data + (size_t)y * src->stride;
        for (int x = 0; x < w; x++)
            d[(size_t)y * w + x] = (int)line[x] - 128;
    }

    int32_t *tmp = (int32_t *)calloc((size_t)(w + 2) * (h + 2), sizeof(int32_t));
    const int lvls = num_levels(w, h);
    for (int i = 1; i <= lvls; i++) {
        if (!isP && i == 1)
            b4t_fwd_2d(d, tmp, w, h);
        else
            haar_fwd_level(d, tmp, w, h, i, isP ? (i > 1) : 1);   /* LVL_TEST sbt.c:22 */
    }
    free(tmp);
}

/* hqp per level, sbt.c:677-696 */
static int level_hqp(int q, int isP, int lvl)
{
    if (lvl > 3)
        return orc_get_quant(q, isP, 0) / 2;
    int hqp = orc_get_quant(q, isP, ORC_MAXLVL - lvl);
    if (lvl == 1) {
        hqp = orc_lb2((unsigned)hqp);
        hqp = clampi(hqp - (isP ? 1 : 3), 1, 24);      /* DSV_QP_P / DSV_QP_I */
        hqp = (1 << hqp) >> 1;
    }
    return hqp / 2;
}

void orc_inv_sbt(orc_plane *dst, orc_coefs *src, int q, int isP, int c)
{
    const int w = src->width, h = src->height;
    int32_t *d = src->data;
    int32_t *tmp = (int32_t *)calloc((size_t)(w + 2) * (h + 2), sizeof(int32_t));
    const int lvls = num_levels(w, h);

    for (int i = lvls; i > 0; i--) {
        if (!isP && i == 1)
            b4t_inv_2d(d, tmp, w, h);
        else
            haar_inv_level(d, tmp, w, h, i, isP ? (i > 1) : 1, c == 0, c == 0 ? level_hqp(q, isP, i) : 0);
    }
    free(tmp);

    /* sbc2int */
    for (int y = 0; y < dst->h; y++) {
        uint8_t *line = dst->data + (size_t)y * dst->stride;
        for (int x = 0; x < dst->w; x++)
            line[x] = (uint8_t)clampi(d[(size_t)y * w + x] + 128, 0, 255);
    }
}
