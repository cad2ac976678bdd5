/*
 * orc_enc.c -- ORACLE (test infrastructure): the DSV1 encoder session layer, restated.
 *
 * This is the CALLER of the hot path; it is restated here so that
 * whole .dsv streams can be compared byte for byte.  Follows dsv_encoder.c of the reference:
 *   frame flow            encode_one_frame dsv_encoder.c:574-694, dsv_enc dsv_encoder.c:780-854
 *   block size / pyramid  size4dim dsv_encoder.c:556-572, dsv_encoder.c:588-613, mk_pyramid :194-217
 *   scene change          check_scene_change dsv_encoder.c:538-554
 *   rate control          quality2quant dsv_encoder.c:70-168, statistics dsv_encoder.c:816-848
 *   stability side info   encode_stable_blocks dsv_encoder.c:330-408
 *   motion side info      encode_motion dsv_encoder.c:257-327
 *   packets               encode_packet_hdr :410-424, encode_metadata :427-461,
 *                         encode_picture :463-536, set_link_offsets :171-192, EOS :766-778
 *   CLI parameter mapping dsv_main.c:423-489, estimate_bitrate util.c:21-52
 * Unlike the reference it keeps no reference-counted per-frame records: only the previous
 * frame's padded source pyramid and reconstruction are retained.
 */
#include <limits.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "orc.h"

#define QMAX 2047                         /* DSV_MAX_QUALITY */
#define QPCT(p) (QMAX * (p) / 100)        /* DSV_QUALITY_PERCENT */
#define HDR_BYTES 14
#define BPF_RESET 256

typedef struct {
    orc_frame *padded;
    orc_frame *pyr[ORC_MAX_PYR];
    orc_frame *recon;
} pic_state;

struct orc_encoder {
    orc_enc_cfg c;
    unsigned next_fnum, prev_gop;
    int force_meta, prev_link, prev_avg_luma;
    unsigned rc_quant, bpf_total, bpf_reset;
    int bpf_avg, total_P_q, avg_P_q, last_P_over, back_in_range;
    int16_t *acc;                 /* x,y interleaved; 16-bit signed bit-fields dsv_encoder.h:101-104 */
    unsigned refresh_ctr;
    unsigned char *stable;
    orc_mv *last_mvs;
    int nblk;
    pic_state ref;                /* previous frame (valid when has_prev) */
    int has_prev;
};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static unsigned auto_bitrate(int quality_pct, int gop, const orc_meta *md)      /* util.c:21-52 */
{
    int fps = (md->fps_num + md->fps_den / 2) / md->fps_den;
    int bpf = 352 * 288 * 3 / 2;
    if (md->subsamp == 0) bpf = 352 * 288 * 3;
    else if (md->subsamp == 4) bpf = 352 * 288 * 2;
    if (gop == 0) bpf *= 4;
    if (md->width < 320 && md->height < 240) bpf /= 4;
    int ratio = (((md->width + md->height) / 2) << 8) / 352;
    bpf = bpf * ratio >> 8;
    int bps = bpf * fps;
    return (unsigned)((bps / (26 - quality_pct / 4)) * 3 / 2);
}

void orc_cfg_from_cli(orc_enc_cfg *cfg, int w, int h, int subsamp, int qp_pct, int gop,
                      int rc_mode_cli, int kbps, int scd, int ipct, int pyrlevels, int stabref)
{
    memset(cfg, 0, sizeof(*cfg));
    cfg->meta.width = w; cfg->meta.height = h; cfg->meta.subsamp = subsamp;
    cfg->meta.fps_num = 30; cfg->meta.fps_den = 1;
    cfg->meta.aspect_num = 1; cfg->meta.aspect_den = 1;
    cfg->gop = gop;
    cfg->scene_change_delta = 4;
    cfg->do_scd = scd;
    cfg->intra_pct_thresh = ipct;
    cfg->quality = QPCT(qp_pct);
    cfg->rc_mode = (rc_mode_cli == 1) ? 0 : 1;             /* CLI 1 = CRF -> library 0 */
    cfg->bitrate = kbps ? (unsigned)kbps * 1024u : auto_bitrate(cfg->quality * 100 / QMAX, gop, &cfg->meta);
    if (cfg->rc_mode == 1) cfg->quality = clampi(cfg->quality * 3 / 2, 0, QMAX);
    cfg->max_q_step = QMAX / 200;
    cfg->min_quality = QPCT(1);
    cfg->max_quality = QPCT(100);
    cfg->min_I_frame_quality = QPCT(5);
    cfg->rc_high_motion_nudge = 1;
    cfg->pyramid_levels = pyrlevels;
    cfg->stable_refresh = stabref ? (unsigned)stabref : (unsigned)clampi(gop - 1, 1, 14);
}

orc_encoder *orc_enc_open(const orc_enc_cfg *cfg)
{
    orc_encoder *e = (orc_encoder *)calloc(1, sizeof(*e));
    e->c = *cfg;
    e->prev_gop = (unsigned)-1;
    e->c.quality = clampi(e->c.quality, 0, QMAX);           /* dsv_enc_start :724-734 */
    if (e->c.rc_mode != 0) {
        e->rc_quant = (unsigned)e->c.quality;
        e->avg_P_q = e->c.quality * 4 / 5;
    }
    e->force_meta = 1;
    return e;
}

void orc_enc_set_next_fnum(orc_encoder *e, unsigned fnum) { e->next_fnum = fnum; }

/* what a library caller may do between two dsv_enc calls: rewrite the encoder's public fields (dsv_encoder.h:58-87).  The
 * reference reads them when it codes the next frame (quality2quant dsv_encoder.c:84-165, the GOP test :794-803), so the
 * change applies from that frame on.  Geometry and GOP structure stay what the encoder was opened with. */
void orc_enc_set_params(orc_encoder *e, const orc_enc_cfg *cfg)
{
    e->c.quality = cfg->quality;
    e->c.bitrate = cfg->bitrate;
    e->c.rc_high_motion_nudge = cfg->rc_high_motion_nudge;
    e->c.max_q_step = cfg->max_q_step;
    e->c.min_quality = cfg->min_quality;
    e->c.max_quality = cfg->max_quality;
    e->c.min_I_frame_quality = cfg->min_I_frame_quality;
}
void orc_enc_force_metadata(orc_encoder *e) { e->force_meta = 1; }       /* dsv_enc_force_metadata dsv_encoder.c:760-764 */

static void free_pic(pic_state *p)
{
    orc_frame_free(p->padded);
    for (int i = 0; i < ORC_MAX_PYR; i++) orc_frame_free(p->pyr[i]);
    orc_frame_free(p->recon);
    memset(p, 0, sizeof(*p));
}

void orc_enc_close(orc_encoder *e)
{
    if (!e) return;
    free_pic(&e->ref);
    free(e->acc); free(e->stable); free(e->last_mvs);
    free(e);
}

const orc_mv *orc_enc_last_mvs(const orc_encoder *e, int *nblk) { *nblk = e->nblk; return e->last_mvs; }
const unsigned char *orc_enc_last_stable(const orc_encoder *e, int *nblk) { *nblk = e->nblk; return e->stable; }

static int block_dim(int dim)                                  /* size4dim */
{
    int s = dim > 1280 ? 64 : dim > 1024 ? 48 : dim > 704 ? 32 : dim > 352 ? 24 : 16;
    return clampi(s & ~7, 16, 64);
}

static void out_reserve(uint8_t **out, size_t *len, size_t *cap, size_t extra)
{
    if (*len + extra > *cap) {
        size_t nc = (*cap ? *cap * 2 : 1 << 20);
        while (nc < *len + extra) nc *= 2;
        *out = (uint8_t *)realloc(*out, nc);
        *cap = nc;
    }
}

static void put_be32(uint8_t *p, unsigned v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

static void write_hdr(orc_bs *bs, int type)
{
    orc_bs_put_bits(bs, 8, 'D'); orc_bs_put_bits(bs, 8, 'S');
    orc_bs_put_bits(bs, 8, 'V'); orc_bs_put_bits(bs, 8, '1');
    orc_bs_put_bits(bs, 8, 0);
    orc_bs_put_bits(bs, 8, (unsigned)type);
    orc_bs_put_bits(bs, 32, 0);
    orc_bs_put_bits(bs, 32, 0);
}

static size_t write_meta_packet(orc_encoder *e, uint8_t *buf)
{
    orc_bs bs;
    const orc_meta *m = &e->c.meta;
    memset(buf, 0, 64);
    orc_bs_init(&bs, buf);
    write_hdr(&bs, 0x00);
    orc_bs_put_ueg(&bs, (unsigned)m->width);   orc_bs_put_ueg(&bs, (unsigned)m->height);
    orc_bs_put_ueg(&bs, (unsigned)m->subsamp);
    orc_bs_put_ueg(&bs, (unsigned)m->fps_num); orc_bs_put_ueg(&bs, (unsigned)m->fps_den);
    orc_bs_put_ueg(&bs, (unsigned)m->aspect_num); orc_bs_put_ueg(&bs, (unsigned)m->aspect_den);
    orc_bs_align(&bs);
    unsigned n = orc_bs_bytepos(&bs);
    put_be32(buf + 10, n);                    /* prev link of a metadata packet stays 0 */
    return n;
}

static void link_packet(orc_encoder *e, uint8_t *pkt, unsigned len, int eos)
{
    unsigned next = eos ? 0 : len;
    put_be32(pkt + 6, (unsigned)e->prev_link);
    put_be32(pkt + 10, next);
    e->prev_link = (int)next;
}

/* CRF: constant; ABR: proportional controller on the running bytes-per-frame average */
static int pick_quant(orc_encoder *e, int isP, int forced_intra)
{
    int q = (int)e->rc_quant;
    if (e->c.rc_mode != 0) {
        const orc_meta *vm = &e->c.meta;
        int fps = (vm->fps_num << 5) / vm->fps_den;
        if (fps == 0) fps = 1;
        int need = (int)(((e->c.bitrate << 5) / (unsigned)fps) >> 3);
        int bpf = e->bpf_avg ? e->bpf_avg : need;
        int dir = (bpf - need) > 0 ? -1 : 1;
        int delta = (abs(bpf - need) << 9) / need;
        int nudged = 0;
        if (dir == 1) delta *= 2;
        if (e->c.rc_high_motion_nudge) {
            if (isP && e->last_P_over) { delta = (delta + 1) * 2; dir = -1; nudged = 1; }
            else if (e->back_in_range)  { delta = (delta + 1) * 2; dir = 1;  nudged = 1; }
        }
        delta = (q * delta) >> 9;
        e->c.max_q_step = clampi(e->c.max_q_step, 1, QMAX);
        int cap = nudged ? e->c.max_q_step * 16 : e->c.max_q_step;
        if (delta > cap) delta = cap;
        q += delta * dir;
        int low_p = clampi(e->avg_P_q - QPCT(4), e->c.min_quality, e->c.max_quality);
        int minq = isP ? low_p : e->c.min_I_frame_quality;
        if (forced_intra) {
            if (q < QPCT(60)) q += QPCT(15);
            else if (q < QPCT(70)) q += QPCT(8);
            else if (q < QPCT(75)) q += QPCT(3);
            q = clampi(q, 0, e->c.max_quality - QPCT(5));
        }
        q = clampi(q, minq, e->c.max_quality);
        q = clampi(q, 0, QMAX);
    } else {
        q = e->c.quality;
    }
    e->rc_quant = (unsigned)q;
    return QMAX - ((QMAX - 5) * q / QMAX);
}

static void write_stability(orc_encoder *e, orc_bs *bs, const orc_mv *mvs, int isP)
{
    const int nblk = e->nblk;
    uint8_t *tmp = (uint8_t *)calloc((size_t)nblk * 32, 1);
    orc_zbrle rle;
    orc_rle_init(&rle, tmp);

    if (e->refresh_ctr >= e->c.stable_refresh) {
        e->refresh_ctr = 0;
        memset(e->acc, 0, sizeof(int16_t) * 2 * (size_t)nblk);
        orc_cov[ORC_COV_STAB_REFRESH]++;
    }
    int div = (int)e->refresh_ctr;
    if (div <= 0) div = 1;

    for (int i = 0; i < nblk; i++) {
        int16_t *a = &e->acc[2 * i];
        int stable = 0, intra = 0;
        if (isP) {
            const orc_mv *mv = &mvs[i];
            if (mv->mode == 0) {
                a[0] = (int16_t)(a[0] + (abs(mv->u.mv.x) >> 2));
                a[1] = (int16_t)(a[1] + (abs(mv->u.mv.y) >> 2));
                stable = mv->high_detail;
                stable |= (a[0] / div == 0 && a[1] / div == 0 && !mv->lo_tex && !mv->lo_var);
                orc_cov[mv->high_detail ? ORC_COV_STABLE_BY_HD : (stable ? ORC_COV_STABLE_BY_AVG : ORC_COV_UNSTABLE_INTER)]++;
            } else {
                intra = 1;
                orc_cov[ORC_COV_INTRA_BLOCK_FLAG]++;
            }
            if (mv->lo_tex || mv->lo_var) { a[0] = 0x3fff; a[1] = 0x3fff; orc_cov[ORC_COV_STAB_RESET_LO]++; }
        } else {
            stable = (a[0] / div == 0 && a[1] / div == 0);
            orc_cov[stable ? ORC_COV_STABLE_I : ORC_COV_UNSTABLE_I]++;
        }
        e->stable[i] = (unsigned char)(stable | (intra << 1));
        orc_rle_put(&rle, e->stable[i] & 1);
    }
    orc_bs_align(bs);
    int bytes = orc_rle_finish_write(&rle);
    orc_bs_put_ueg(bs, (unsigned)bytes);
    orc_bs_align(bs);
    orc_bs_append(bs, tmp, bytes);
    free(tmp);
}

static void write_motion(orc_encoder *e, orc_bs *bs, const orc_mv *mvs, const orc_params *p)
{
    const size_t cap = (size_t)e->nblk * 32;
    uint8_t *buf[4];
    orc_bs sub[4];
    orc_zbrle modes;
    for (int i = 0; i < 4; i++) {
        buf[i] = (uint8_t *)calloc(cap, 1);
        orc_bs_init(&sub[i], buf[i]);
    }
    orc_rle_init(&modes, buf[0]);

    for (int j = 0; j < p->nblocks_v; j++)
        for (int i = 0; i < p->nblocks_h; i++) {
            const orc_mv *mv = &mvs[i + j * p->nblocks_h];
            orc_rle_put(&modes, mv->mode);
            if (mv->mode == 0) {
                int px, py;
                orc_mv_pred(mvs, p, i, j, &px, &py);
                orc_bs_put_seg(&sub[1], mv->u.mv.x - px);
                orc_bs_put_seg(&sub[2], mv->u.mv.y - py);
            } else if (mv->submask == 0xF) {
                orc_bs_put_bits(&sub[3], 1, 1);
            } else {
                orc_bs_put_bits(&sub[3], 1, 0);
                orc_bs_put_bits(&sub[3], 4, mv->submask);
            }
        }
    for (int i = 0; i < 4; i++) {
        int bytes;
        orc_bs_align(bs);
        if (i == 0) bytes = orc_rle_finish_write(&modes);
        else { orc_bs_align(&sub[i]); bytes = (int)orc_bs_bytepos(&sub[i]); }
        orc_bs_put_ueg(bs, (unsigned)bytes);
        orc_bs_align(bs);
        orc_bs_append(bs, buf[i], bytes);
        free(buf[i]);
    }
}

size_t orc_enc_frame(orc_encoder *e, const uint8_t *yuv, uint8_t **out, size_t *outlen, size_t *outcap,
                     uint8_t *recon_out)
{
    const orc_meta *vm = &e->c.meta;
    const int w = vm->width, h = vm->height, fmt = vm->subsamp;
    const size_t len_before = *outlen;
    orc_frame in;
    orc_params prm;
    pic_state cur;
    orc_mv *mvs = NULL;
    int gop_start = 0, forced_intra = 0;
    const unsigned fnum = e->next_fnum++;

    memset(&cur, 0, sizeof(cur));
    orc_frame_wrap_planar(&in, fmt, (uint8_t *)yuv, w, h);
    orc_frame *xf = orc_frame_new(fmt, w, h, 1);
    orc_frame *pred = orc_frame_new(fmt, w, h, 1);

    memset(&prm, 0, sizeof(prm));
    prm.vidmeta = &e->c.meta;
    prm.blk_w = block_dim(w);
    prm.blk_h = block_dim(h);
    {   /* test vectors for decoders: streams whose block size is not the encoder's rule (the reference decoder takes it from
         * the packet, dsv_decoder.c:335-360).  ORC_BLK_OVERRIDE="WxH", multiples of 4 in 16..64 */
        const char *ov = getenv("ORC_BLK_OVERRIDE");
        int ow = 0, oh = 0;
        if (ov && sscanf(ov, "%dx%d", &ow, &oh) == 2 && ow >= 16 && ow <= 64 && oh >= 16 && oh <= 64 && !(ow & 3) && !(oh & 3)) {
            prm.blk_w = ow; prm.blk_h = oh;
        }
    }
    prm.nblocks_h = (w + prm.blk_w - 1) / prm.blk_w;
    prm.nblocks_v = (h + prm.blk_h - 1) / prm.blk_h;
    e->nblk = prm.nblocks_h * prm.nblocks_v;
    if (!e->acc) {
        e->acc = (int16_t *)calloc((size_t)e->nblk * 2, sizeof(int16_t));
        e->stable = (unsigned char *)calloc((size_t)e->nblk, 1);
    }
    if (e->c.pyramid_levels == 0) {
        int lv = orc_lb2((unsigned)(w < h ? w : h));
        int nb = prm.nblocks_h > prm.nblocks_v ? prm.nblocks_h : prm.nblocks_v;
        while ((1 << lv) > nb) lv--;
        e->c.pyramid_levels = clampi(lv, 3, ORC_MAX_PYR);
    }

    if (e->c.gop != 0) {
        cur.padded = orc_frame_new(fmt, w, h, 1);
        orc_frame_copy(cur.padded, &in);
        const orc_frame *prev = cur.padded;
        for (int i = 0; i < e->c.pyramid_levels; i++) {
            cur.pyr[i] = orc_frame_new(fmt, ORC_RSHIFT_UP(w, i + 1), ORC_RSHIFT_UP(h, i + 1), 1);
            orc_frame_ds2x_luma(cur.pyr[i], prev);
            orc_frame_extend_luma(cur.pyr[i]);
            prev = cur.pyr[i];
        }
    } else {
        cur.padded = orc_frame_new(fmt, w, h, 0);
        orc_frame_copy(cur.padded, &in);
    }
    if (e->force_meta || (unsigned)(e->prev_gop + (unsigned)e->c.gop) <= fnum) {
        gop_start = 1;
        e->prev_gop = fnum;
        e->force_meta = 0;
    }
    if (e->c.gop == 0) {
        prm.is_ref = 0;
        prm.has_ref = 0;
    } else {
        prm.is_ref = 1;
        prm.has_ref = !gop_start;
        if (e->c.do_scd) {
            int al = orc_frame_avg_luma(cur.pyr[e->c.pyramid_levels - 1]);
            if (abs(e->prev_avg_luma - al) > e->c.scene_change_delta) {
                if (prm.has_ref) orc_cov[ORC_COV_FORCED_INTRA_SCENE]++;
                prm.has_ref = 0;
                forced_intra = 1;
            }
            e->prev_avg_luma = al;
        }
    }
    if (prm.has_ref) {
        orc_hme hme;
        memset(&hme, 0, sizeof(hme));
        hme.levels = e->c.pyramid_levels;
        hme.params = &prm;
        hme.src[0] = cur.padded;
        hme.ref[0] = e->ref.padded;
        for (int i = 0; i < hme.levels; i++) {
            hme.src[i + 1] = cur.pyr[i];
            hme.ref[i + 1] = e->ref.pyr[i];
        }
        int pct = orc_hme_run(&hme);
        mvs = hme.mvf[0];
        for (int i = 1; i <= hme.levels; i++) free(hme.mvf[i]);
        forced_intra = 0;
        if (pct > e->c.intra_pct_thresh) {
            prm.has_ref = 0;
            forced_intra = 1;
        }
        orc_cov[forced_intra ? ORC_COV_FORCED_INTRA_IPCT : ORC_COV_P_KEPT]++;
    }
    const int isP = prm.has_ref;
    const int quant = pick_quant(e, isP, forced_intra);

    orc_frame_copy(xf, cur.padded);
    if (prm.has_ref)
        orc_sub_pred(mvs, &prm, pred, xf, e->ref.recon);

    /* ---- picture packet ---- */
    size_t bound = (size_t)w * h * (fmt == 0 ? 6 : fmt == 4 ? 4 : 2);
    uint8_t *pkt = (uint8_t *)calloc(bound + 4096, 1);
    orc_bs bs;
    orc_bs_init(&bs, pkt);
    write_hdr(&bs, 0x04 | (prm.is_ref << 1) | prm.has_ref);
    orc_bs_align(&bs);
    orc_bs_put_bits(&bs, 32, fnum);
    orc_bs_align(&bs);
    orc_bs_put_ueg(&bs, (unsigned)prm.blk_w >> 2);
    orc_bs_put_ueg(&bs, (unsigned)prm.blk_h >> 2);
    orc_bs_align(&bs);
    write_stability(e, &bs, mvs, isP);
    if (prm.has_ref) {
        orc_bs_align(&bs);
        write_motion(e, &bs, mvs, &prm);
    }
    orc_bs_align(&bs);
    orc_bs_put_bits(&bs, 11, (unsigned)quant);

    orc_stability st;
    orc_coefs co[3];
    st.params = &prm;
    st.stable_blocks = e->stable;
    st.isP = (unsigned char)isP;
    orc_coefs_new(co, fmt, w, h);
    for (int c = 0; c < 3; c++) {
        st.cur_plane = (unsigned char)c;
        orc_fwd_sbt(&xf->planes[c], &co[c], isP);
        orc_encode_plane(&bs, &co[c], quant, &st);
        orc_inv_sbt(&xf->planes[c], &co[c], quant, isP, c);
    }
    free(co[0].data);
    orc_bs_align(&bs);
    const unsigned pkt_len = orc_bs_bytepos(&bs);

    if (prm.has_ref)
        orc_frame_add(xf, pred);
    if (prm.is_ref && e->c.gop != 0) {
        cur.recon = orc_frame_new(fmt, w, h, 1);
        orc_frame_copy(cur.recon, xf);
    }
    if (recon_out) {
        uint8_t *o = recon_out;
        for (int c = 0; c < 3; c++)
            for (int y = 0; y < xf->planes[c].h; y++, o += xf->planes[c].w)
                memcpy(o, xf->planes[c].data + (size_t)y * xf->planes[c].stride, (size_t)xf->planes[c].w);
    }

    /* ---- emit: optional metadata packet, then the picture ---- */
    if (gop_start) {
        uint8_t mb[64];
        size_t n = write_meta_packet(e, mb);
        out_reserve(out, outlen, outcap, n);
        memcpy(*out + *outlen, mb, n);
        *outlen += n;
    }
    if (isP) e->refresh_ctr++;
    if (e->c.rc_mode != 0) {
        e->bpf_total += pkt_len;
        e->bpf_reset++;
        if (isP) {
            e->total_P_q += (int)e->rc_quant;
            e->avg_P_q = (int)((unsigned)e->total_P_q / e->bpf_reset);
            unsigned fps = (unsigned)(vm->fps_num << 5) / (unsigned)vm->fps_den;
            if (fps == 0) fps = 1;
            unsigned need = ((e->c.bitrate << 5) / fps) >> 3;
            int under = pkt_len < (need * 3 / 4);
            need = need * 7 / 8;
            int over = pkt_len > need;
            e->back_in_range = (e->last_P_over && under);
            e->last_P_over = over;
        } else {
            e->last_P_over = 0;
            e->back_in_range = 0;
        }
        e->bpf_avg = (int)(e->bpf_total / e->bpf_reset);
        if (e->bpf_reset >= BPF_RESET) {
            e->bpf_total = (unsigned)e->bpf_avg;
            e->total_P_q = (int)((unsigned)e->total_P_q / e->bpf_reset);
            e->bpf_reset = 1;
        }
    }
    link_packet(e, pkt, pkt_len, 0);
    out_reserve(out, outlen, outcap, pkt_len);
    memcpy(*out + *outlen, pkt, pkt_len);
    *outlen += pkt_len;
    free(pkt);

    /* ---- bookkeeping ---- */
    free(e->last_mvs);
    e->last_mvs = mvs;
    orc_frame_free(xf);
    orc_frame_free(pred);
    if (e->c.gop != 0) {
        free_pic(&e->ref);
        e->ref = cur;
        e->has_prev = 1;
    } else {
        free_pic(&cur);
    }
    return *outlen - len_before;
}

size_t orc_enc_eos(orc_encoder *e, uint8_t **out, size_t *outlen, size_t *outcap)
{
    uint8_t pkt[HDR_BYTES];
    orc_bs bs;
    memset(pkt, 0, sizeof(pkt));
    orc_bs_init(&bs, pkt);
    write_hdr(&bs, 0x10);
    link_packet(e, pkt, HDR_BYTES, 1);
    out_reserve(out, outlen, outcap, HDR_BYTES);
    memcpy(*out + *outlen, pkt, HDR_BYTES);
    *outlen += HDR_BYTES;
    return HDR_BYTES;
}
