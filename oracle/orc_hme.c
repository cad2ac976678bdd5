/*
 * orc_hme.c -- ORACLE (test infrastructure): hierarchical motion estimation + the level-0
 * mode decision of the DSV1 encoder.
 *
 * Restates hme.c of the reference (dsv_hme hme.c:730-741, refine_level hme.c:378-728):
 *   - candidate inheritance from the parent level (hme.c:452-480), best inherited by SAD
 *     (hme.c:482-510; strict '<', first minimum wins, the last candidate is the fallback),
 *   - 9-point +-1 full-pel search in the fixed order of hme.c:423-424 (hme.c:519-541),
 *   - level 0: optional 8-point half-pel search on the centred 14x14 window using a 32x32
 *     half-pel lattice (hpel hme.c:350-376, hpsad hme.c:302-314), block statistics with
 *     32-bit unsigned wrap (block_texture hme.c:181-211, block_analysis hme.c:213-245,
 *     y_sqrvar hme.c:247-267, c_maxvar hme.c:269-300), high_detail from the three causal
 *     neighbours (hme.c:621-648), the intra tests (hme.c:652-682), the representability veto
 *     (block_intra_test hme.c:147-179) and the per-quadrant vote (intra_metric hme.c:89-134).
 * Also the motion-vector predictor dsv_movec_pred dsv.c:189-231.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define WIN 14                 /* HP_SAD_SZ */
#define LAT 32                 /* HP_STRIDE: lattice row length, (WIN+2)*2 */

static inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int tap_h(const uint8_t *p)        { return 9 * (p[0] + p[1]) - (p[-1] + p[2]); }
static inline int tap_v(const uint8_t *p, int s) { return 9 * (p[0] + p[s]) - (p[-s] + p[2 * s]); }

static int sad(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h)
{
    int acc = 0;
    for (int y = 0; y < h; y++, a += as, b += bs)
        for (int x = 0; x < w; x++) acc += abs(a[x] - b[x]);
    return acc;
}

static int out_of_frame(const orc_frame *f, int x, int y, int w, int h)   /* invalid_block */
{
    int b = f->border * ORC_BORDER;
    return x < -b || y < -b || x + w > f->width + b || y + h > f->height + b;
}

/* 32x32 lattice of the 16x16 patch at `ref`: even rows F H F H.., odd rows V D V D.. */
static void build_lattice(uint8_t *lat, const uint8_t *ref, int rs)
{
    int16_t hrow[(WIN + 2 + 4) * (WIN + 2)];
    for (int r = 0; r < WIN + 2 + 4; r++)
        for (int i = 0; i < WIN + 2; i++)
            hrow[r * (WIN + 2) + i] = (int16_t)tap_h(ref + (ptrdiff_t)(r - 1) * rs + i);
    for (int j = 0; j < WIN + 2; j++) {
        const uint8_t *row = ref + (ptrdiff_t)j * rs;
        uint8_t *e = lat + (2 * j) * LAT, *o = e + LAT;
        for (int i = 0; i < WIN + 2; i++) {
            const int16_t *m = hrow + j * (WIN + 2) + i;
            int d = 9 * (m[WIN + 2] + m[2 * (WIN + 2)]) - (m[0] + m[3 * (WIN + 2)]);
            e[2 * i] = row[i];
            e[2 * i + 1] = sat8((tap_h(row + i) + 8) >> 4);
            o[2 * i] = sat8((tap_v(row + i, rs) + 8) >> 4);
            o[2 * i + 1] = sat8((d + 128) >> 8);
        }
    }
}

static int lattice_sad(const uint8_t *a, int as, const uint8_t *l)
{
    int acc = 0;
    for (int y = 0; y < WIN; y++, a += as, l += 2 * LAT)
        for (int x = 0; x < WIN; x++) acc += abs(a[x] - l[2 * x]);
    return acc;
}

/* horizontal/vertical gradient sums shared by both statistic functions */
static void grad_sums(const uint8_t *p, int s, int w, int h, unsigned *gh, unsigned *gv,
                      unsigned *sum, unsigned *sumsq)
{
    unsigned a = 0, b = 0, s1 = 0, s2 = 0;
    for (int y = 0; y < h; y++) {
        const uint8_t *r = p + (size_t)y * s;
        for (int x = 0; x < w; x++) {
            int px = r[x];
            if (x + 1 < w) a += (unsigned)abs(px - r[x + 1]);
            if (y > 0) b += (unsigned)abs(px - r[x - s]);
            s1 += (unsigned)px;
            s2 += (unsigned)(px * px);
        }
    }
    *gh = a; *gv = b; *sum = s1; *sumsq = s2;
}

static unsigned window_stats(const uint8_t *p, int s, int *avg, int *var)      /* block_texture */
{
    unsigned gh, gv, av, avs;
    grad_sums(p, s, WIN, WIN, &gh, &gv, &av, &avs);
    *avg = (int)(av / (WIN * WIN));
    *var = (int)(avs - (av * av) / (WIN * WIN));
    return ((gh + gv) / 2) / (WIN * WIN);
}

static unsigned block_stats(const uint8_t *p, int s, int w, int h, unsigned *texture)  /* block_analysis */
{
    unsigned gh, gv, s1, s2;
    grad_sums(p, s, w, h, &gh, &gv, &s1, &s2);
    *texture = ((gh + gv) / 2) / (unsigned)(w * h);
    return s2 - (s1 * s1) / (unsigned)(w * h);            /* wraps mod 2^32 for big blocks */
}

static unsigned sq_var(const uint8_t *p, int s, int w, int h)                  /* y_sqrvar */
{
    unsigned s1 = 0, s2 = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            unsigned px = p[(size_t)y * s + x];
            s1 += px; s2 += px * px;
        }
    return s2 - (s1 * s1) / (unsigned)(w * h);
}

static unsigned chroma_maxvar(const orc_plane *planes, int x, int y, int w, int h)   /* c_maxvar */
{
    const orc_plane *u = &planes[1], *v = &planes[2];
    unsigned vu = sq_var(u->data + x + (ptrdiff_t)y * u->stride, u->stride, w, h);
    unsigned vv = sq_var(v->data + x + (ptrdiff_t)y * v->stride, v->stride, w, h);
    return vu > vv ? vu : vv;
}

/* 1 if some pixel of the block cannot be reproduced by "mean + clamped residual" */
unsigned long long orc_cov[ORC_COV_N];
void orc_cov_reset(void) { memset(orc_cov, 0, sizeof orc_cov); }
int orc_cov_read(unsigned long long *out, int n)
{
    for (int i = 0; i < n && i < ORC_COV_N; i++) out[i] = orc_cov[i];
    return ORC_COV_N;
}

static int intra_unrepresentable(const uint8_t *src, int ss, const uint8_t *ref, int rs, int w, int h)
{
    int mean = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) mean += ref[(size_t)y * rs + x];
    mean /= (w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int px = src[(size_t)y * ss + x];
            int back = sat8(mean + sat8(px - mean + 128) - 128);
            if (back != px) return 1;
        }
    return 0;
}

/* "does the zero-motion reference do more good than evil" vote for one quadrant */
static int quadrant_prefers_inter(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h)
{
    unsigned good = 0, evil = 0;
    for (int y = 0; y < h; y++) {
        const uint8_t *ra = a + (size_t)y * as, *rb = b + (size_t)y * bs;
        const uint8_t *ua = y ? ra - as : ra, *ub = y ? rb - bs : rb;
        for (int x = 0; x < w; x++) {
            int pa = ra[x], pb = rb[x];
            int la = x ? ra[x - 1] : pa, lb = x ? rb[x - 1] : pb;
            int dif = abs(pa - pb);
            good += (unsigned)(abs(pa - la) + abs(pa - ua[x]) + abs(pb - lb) + abs(pb - ub[x]));
            if (dif == 0) good += 192;
            else if (dif == 1) good += 128;
            else if (dif == 2) good += 96;
            else evil += (unsigned)dif;
        }
    }
    return good >= (unsigned)((w + h) >> 1) * evil;
}

static const int FP_X[9] = { 0, 1, -1, 0, 0, -1, 1, -1, 1 };      /* hme.c:423-424 */
static const int FP_Y[9] = { 0, 0, 0, 1, -1, -1, -1, 1, 1 };
static const int HP_X[8] = { 1, -1, 0, 0, -1, 1, -1, 1 };          /* hme.c:426-427 */
static const int HP_Y[8] = { 0, 0, 1, -1, -1, -1, 1, 1 };
static const int PARENT_OFF[5][2] = { {0, 0}, {-2, 0}, {2, 0}, {0, -2}, {0, 2} };   /* hme.c:454 */

static int refine(orc_hme *hme, int level)
{
    const orc_params *prm = hme->params;
    const orc_frame *src = hme->src[level], *ref = hme->ref[level];
    const orc_plane *sp = &src->planes[0], *rp = &ref->planes[0];
    const int BW = prm->blk_w, BH = prm->blk_h;
    const int nxb = prm->nblocks_h, nyb = prm->nblocks_v;
    const int step = 1 << level;
    const unsigned pmask = ~(unsigned)((step << 1) - 1);
    const orc_mv *parent = level < hme->levels ? hme->mvf[level + 1] : NULL;
    orc_mv *mf = (orc_mv *)calloc((size_t)nxb * nyb, sizeof(orc_mv));
    int nintra = 0;

    hme->mvf[level] = mf;

    for (int j = 0; j < nyb; j += step) {
        for (int i = 0; i < nxb; i += step) {
            const int bx = (i * BW) >> level, by = (j * BH) >> level;
            if (bx >= src->width || by >= src->height)
                continue;                                   /* stays a zero inter vector */
            int remw = sp->w - bx, remh = sp->h - by;
            if (remw < 0) remw = 0;
            if (remh < 0) remh = 0;
            const int bw = remw < BW ? remw : BW, bh = remh < BH ? remh : BH;
            const uint8_t *sblk = sp->data + bx + (ptrdiff_t)by * sp->stride;
            const uint8_t *zref = rp->data + bx + (ptrdiff_t)by * rp->stride;

            /* candidate list: zero vector + unique non-zero parents */
            int32_t cand[8];
            int n = 0;
            cand[n++] = 0;
            if (parent) {
                int pi = (int)((unsigned)i & pmask), pj = (int)((unsigned)j & pmask);
                for (int m = 0; m < 5; m++) {
                    int x = pi + PARENT_OFF[m][0] * step, y = pj + PARENT_OFF[m][1] * step;
                    if (x < 0 || x >= nxb || y < 0 || y >= nyb) continue;
                    int32_t v = parent[x + y * nxb].u.all;
                    if (!v) continue;
                    int dup = 0;
                    for (int k = 0; k < n; k++) dup |= (cand[k] == v);
                    if (!dup) cand[n++] = v;
                }
            }
            int pick = n - 1;
            if (n > 1) {
                int best_score = INT_MAX;
                for (int k = 0; k < n; k++) {
                    orc_mv t; t.u.all = cand[k];
                    int dx = t.u.mv.x >> level, dy = t.u.mv.y >> level;
                    if (out_of_frame(src, bx, by, bw, bh)) continue;
                    if (out_of_frame(ref, bx + dx, by + dy, bw, bh)) continue;
                    int sc = sad(sblk, sp->stride, rp->data + (bx + dx) + (ptrdiff_t)(by + dy) * rp->stride,
                                 rp->stride, bw, bh);
                    if (best_score > sc) { best_score = sc; pick = k; }
                }
            }
            orc_mv start; start.u.all = cand[pick];
            int dx = clampi(start.u.mv.x >> level, -bw - bx, ref->width - bx);
            int dy = clampi(start.u.mv.y >> level, -bh - by, ref->height - by);

            int best = INT_MAX, m = 0;
            for (int k = 0; k < 9; k++) {
                int sc = sad(sblk, sp->stride,
                             rp->data + (bx + dx + FP_X[k]) + (ptrdiff_t)(by + dy + FP_Y[k]) * rp->stride,
                             rp->stride, bw, bh);
                if (best > sc) { best = sc; m = k; }
            }
            dx += FP_X[m];
            dy += FP_Y[m];

            orc_mv *mv = &mf[i + j * nxb];
            mv->mode = 0;
            mv->u.mv.x = (int16_t)(dx << level);
            mv->u.mv.y = (int16_t)(dy << level);
            if (level != 0) continue;

            /* ---------------- level 0: half-pel refinement ---------------- */
            const unsigned yarea = (unsigned)(bw * bh), yareasq = yarea * yarea;
            const int wx = bx + ((bw >> 1) - WIN / 2), wy = by + ((bh >> 1) - WIN / 2);
            const uint8_t *swin = sp->data + wx + (ptrdiff_t)wy * sp->stride;
            uint8_t refwin[WIN * WIN];
            int have_hp = 0;

            if (best > BW * BH) {
                uint8_t lat[(LAT + 2) * (LAT + 2)];
                int best_hp = (int)((unsigned)(best * (WIN * WIN)) / yarea);
                const uint8_t *rwin = rp->data + (wx + mv->u.mv.x) + (ptrdiff_t)(wy + mv->u.mv.y) * rp->stride;
                build_lattice(lat, rwin - 1 - rp->stride, rp->stride);
                const uint8_t *centre = lat + 2 + 2 * LAT;
                int hm = -1;
                for (int k = 0; k < 8; k++) {
                    int sc = lattice_sad(swin, sp->stride, centre + HP_X[k] + HP_Y[k] * LAT);
                    if (best_hp > sc) { best_hp = sc; hm = k; }
                }
                mv->u.mv.x = (int16_t)(mv->u.mv.x << 1);
                mv->u.mv.y = (int16_t)(mv->u.mv.y << 1);
                orc_cov[hm >= 0 ? ORC_COV_HP_REFINED : ORC_COV_HP_KEPT_FULLPEL]++;
                if (hm >= 0) {
                    const uint8_t *l = centre + HP_X[hm] + HP_Y[hm] * LAT;
                    mv->u.mv.x = (int16_t)(mv->u.mv.x + HP_X[hm]);
                    mv->u.mv.y = (int16_t)(mv->u.mv.y + HP_Y[hm]);
                    for (int y = 0; y < WIN; y++)
                        for (int x = 0; x < WIN; x++) refwin[y * WIN + x] = l[2 * x + y * 2 * LAT];
                    have_hp = 1;
                    best = (int)((unsigned)best_hp * yarea / (WIN * WIN));
                }
            } else {
                orc_cov[ORC_COV_HP_SKIPPED]++;
                mv->u.mv.x = (int16_t)(mv->u.mv.x << 1);
                mv->u.mv.y = (int16_t)(mv->u.mv.y << 1);
            }
            if (!have_hp) {
                const uint8_t *r = rp->data + (wx + (mv->u.mv.x >> 1)) + (ptrdiff_t)(wy + (mv->u.mv.y >> 1)) * rp->stride;
                for (int y = 0; y < WIN; y++) memcpy(refwin + y * WIN, r + (ptrdiff_t)y * rp->stride, WIN);
            }

            /* ---------------- statistics + flags ---------------- */
            unsigned luma_tex, luma_var = block_stats(sblk, sp->stride, bw, bh, &luma_tex);
            int src_avg, ref_avg, src_var, ref_var;
            int src_tex = (int)window_stats(swin, sp->stride, &src_avg, &src_var);
            int ref_tex = (int)window_stats(refwin, WIN, &ref_avg, &ref_var);
            unsigned thr_tex = 1;
            int thr_var = WIN * WIN;

            mv->lo_tex = (luma_tex <= 2);
            mv->lo_var = (luma_var < yareasq);

            if (i > 0) {
                const orc_mv *nb = &mf[j * nxb + i - 1];
                if (nb->mode == 0 && !nb->lo_tex && !nb->lo_var) { thr_var *= WIN; thr_tex++; }
            }
            if (j > 0) {
                const orc_mv *nb = &mf[(j - 1) * nxb + i];
                if (nb->mode == 0 && !nb->lo_tex && !nb->lo_var) { thr_var *= WIN; thr_tex++; }
            }
            if (i > 0 && j > 0) {
                const orc_mv *nb = &mf[(j - 1) * nxb + i - 1];
                if (nb->mode == 0 && !nb->lo_tex && !nb->lo_var) { thr_var *= WIN / 4; thr_tex++; }
            }
            mv->high_detail = (luma_tex > thr_tex && src_var > thr_var);
            orc_cov[(mv->high_detail ? ORC_COV_NB0_HD : ORC_COV_NB0_PLAIN) + (int)thr_tex - 1]++;
            orc_cov[mv->lo_tex ? ORC_COV_LO_TEX : (mv->lo_var ? ORC_COV_LO_VAR : ORC_COV_LO_NEITHER)]++;
            if (mv->lo_tex && mv->lo_var) orc_cov[ORC_COV_LO_VAR]++;

            /* ---------------- intra decision ---------------- */
            int want_intra = 0;
            if (src_tex < 2 && sq_var(zref, rp->stride, bw, bh) > luma_var * 2) want_intra = 1;
            else if (ref_var > src_var * 2) want_intra = 2;
            else if (src_tex == 0 && ref_tex != 0) want_intra = 3;
            else if (abs(src_avg - ref_avg) > 8) want_intra = 4;
            else if (luma_tex <= 10 && (unsigned)best > yareasq / 16) want_intra = 5;
            else {
                int fmt = prm->vidmeta->subsamp;
                int cbx = i * (BW >> ORC_HSHIFT(fmt)), cby = j * (BH >> ORC_VSHIFT(fmt));
                int cbw = bw >> ORC_HSHIFT(fmt), cbh = bh >> ORC_VSHIFT(fmt);
                unsigned cs = chroma_maxvar(src->planes, cbx, cby, cbw, cbh);
                unsigned cr = chroma_maxvar(ref->planes, cbx, cby, cbw, cbh);
                if (cr > 4 * cs) want_intra = 6;
            }
            orc_cov[want_intra ? ORC_COV_INTRA_ZEROVAR + want_intra - 1 : ORC_COV_INTRA_NONE]++;
            if (!want_intra) continue;
            if (intra_unrepresentable(sblk, sp->stride, zref, rp->stride, bw, bh)) { orc_cov[ORC_COV_VETO_TAKEN]++; continue; }
            orc_cov[ORC_COV_VETO_NOT_TAKEN]++;

            mv->submask = 0xF;
            orc_cov[src_tex > 1 ? ORC_COV_QUAD_VOTE : ORC_COV_LOWTEX_ALL_INTRA]++;
            if (src_tex > 1) {
                const int qw = bw / 2, qh = bh / 2;
                for (int k = 0; k < 4; k++) {
                    int ox = (k & 1) ? qw : 0, oy = (k & 2) ? qh : 0;
                    if (quadrant_prefers_inter(sblk + ox + (ptrdiff_t)oy * sp->stride, sp->stride,
                                               zref + ox + (ptrdiff_t)oy * rp->stride, rp->stride, qw, qh))
                        mv->submask &= (uint8_t)~(1u << k);
                }
            }
            if (src_tex > 1) orc_cov[ORC_COV_SUBMASK0 + mv->submask]++;
            if (mv->submask) {
                mv->mode = 1;
                nintra++;
            }
        }
    }
    return nintra;
}

int orc_hme_run(orc_hme *h)
{
    int nintra = 0;
    for (int l = h->levels; l >= 0; l--) nintra = refine(h, l);
    return nintra * 100 / (h->params->nblocks_h * h->params->nblocks_v);
}

static int pick_pred(int left, int top, int topleft)
{
    int grad = left + top - topleft;
    return abs(grad - left) < abs(grad - top) ? left : top;
}

void orc_mv_pred(const orc_mv *vecs, const orc_params *p, int x, int y, int *px, int *py)
{
    int vx[3] = { 0, 0, 0 }, vy[3] = { 0, 0, 0 };
    const int W = p->nblocks_h;
    const orc_mv *nb[3] = {
        x > 0 ? &vecs[y * W + x - 1] : NULL,
        y > 0 ? &vecs[(y - 1) * W + x] : NULL,
        (x > 0 && y > 0) ? &vecs[(y - 1) * W + x - 1] : NULL,
    };
    for (int k = 0; k < 3; k++)
        if (nb[k] && nb[k]->mode == 0) { vx[k] = nb[k]->u.mv.x; vy[k] = nb[k]->u.mv.y; }
    *px = pick_pred(vx[0], vx[1], vx[2]);
    *py = pick_pred(vy[0], vy[1], vy[2]);
}
