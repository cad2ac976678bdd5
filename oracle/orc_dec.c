/*
 * orc_dec.c -- ORACLE (test infrastructure): the DSV1 decoder session layer, restated.
 *
 * Follows dsv_decoder.c of the reference: packet header dsv_decoder.c:21-49, metadata :51-71,
 * stability side info :127-145, motion side info :73-124, picture decode dsv_dec :286-472
 * (plane loop :379-412, prediction :422-436, reference keeping :438-456).  Debug overlays
 * (draw_info/drawvec) are out of scope.
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

struct orc_decoder {
    orc_meta meta;
    int got_meta;
    orc_frame *ref;          /* extended reconstruction of the last reference picture */
};

orc_decoder *orc_dec_open(void) { return (orc_decoder *)calloc(1, sizeof(orc_decoder)); }

void orc_dec_close(orc_decoder *d)
{
    if (!d) return;
    orc_frame_free(d->ref);
    free(d);
}

void orc_dec_get_meta(const orc_decoder *d, orc_meta *m) { *m = d->meta; }

static int round_even(int v) { return (v + 1) & ~1; }

int orc_dec_packet(orc_decoder *d, const uint8_t *pkt, size_t len, uint8_t *yuv_out, unsigned *fnum)
{
    orc_bs bs;
    (void)len;
    orc_bs_init(&bs, (uint8_t *)pkt);
    if (orc_bs_get_bits(&bs, 8) != 'D' || orc_bs_get_bits(&bs, 8) != 'S' ||
        orc_bs_get_bits(&bs, 8) != 'V' || orc_bs_get_bits(&bs, 8) != '1')
        return 1;
    (void)orc_bs_get_bits(&bs, 8);
    int type = (int)orc_bs_get_bits(&bs, 8);
    (void)orc_bs_get_bits(&bs, 32);
    (void)orc_bs_get_bits(&bs, 32);

    if (!(type & 0x4)) {
        if (type == 0x00) {
            orc_meta *m = &d->meta;
            m->width = (int)orc_bs_get_ueg(&bs);   m->height = (int)orc_bs_get_ueg(&bs);
            m->subsamp = (int)orc_bs_get_ueg(&bs);
            m->fps_num = (int)orc_bs_get_ueg(&bs); m->fps_den = (int)orc_bs_get_ueg(&bs);
            m->aspect_num = (int)orc_bs_get_ueg(&bs); m->aspect_den = (int)orc_bs_get_ueg(&bs);
            d->got_meta = 1;
            return 3;
        }
        return type == 0x10 ? 2 : 1;
    }
    if (!d->got_meta) return 0;

    orc_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.vidmeta = &d->meta;
    prm.has_ref = type & 1;
    const int is_ref = ((type & 0x6) == 0x6);
    const int w = d->meta.width, h = d->meta.height, fmt = d->meta.subsamp;

    orc_bs_align(&bs);
    *fnum = orc_bs_get_bits(&bs, 32);
    orc_bs_align(&bs);
    prm.blk_w = (int)orc_bs_get_ueg(&bs) << 2;
    prm.blk_h = (int)orc_bs_get_ueg(&bs) << 2;
    if (prm.blk_w < 16 || prm.blk_h < 16 || prm.blk_w > 64 || prm.blk_h > 64) return 1;
    prm.nblocks_h = (w + prm.blk_w - 1) / prm.blk_w;
    prm.nblocks_v = (h + prm.blk_h - 1) / prm.blk_h;
    const int nblk = prm.nblocks_h * prm.nblocks_v;

    unsigned char *stable = (unsigned char *)calloc((size_t)nblk, 1);
    orc_mv *mvs = NULL;
    {   /* stability flags */
        orc_zbrle r;
        orc_bs_align(&bs);
        unsigned n = orc_bs_get_ueg(&bs);
        orc_bs_align(&bs);
        orc_rle_init(&r, (uint8_t *)pkt + orc_bs_bytepos(&bs));
        bs.pos += n * 8;
        for (int i = 0; i < nblk; i++) stable[i] = (unsigned char)orc_rle_get(&r);
    }
    if (prm.has_ref) {
        orc_bs sub[4];
        orc_zbrle modes;
        mvs = (orc_mv *)calloc((size_t)nblk, sizeof(orc_mv));
        orc_bs_align(&bs);
        for (int i = 0; i < 4; i++) {
            unsigned n = orc_bs_get_ueg(&bs);
            orc_bs_align(&bs);
            orc_bs_init(&sub[i], (uint8_t *)pkt + orc_bs_bytepos(&bs));
            bs.pos += n * 8;
        }
        orc_rle_init(&modes, sub[0].start);
        for (int j = 0; j < prm.nblocks_v; j++)
            for (int i = 0; i < prm.nblocks_h; i++) {
                orc_mv *mv = &mvs[i + j * prm.nblocks_h];
                mv->mode = (uint8_t)orc_rle_get(&modes);
                if (mv->mode == 0) {
                    int px, py;
                    orc_mv_pred(mvs, &prm, i, j, &px, &py);
                    mv->u.mv.x = (int16_t)(orc_bs_get_seg(&sub[1]) + px);
                    mv->u.mv.y = (int16_t)(orc_bs_get_seg(&sub[2]) + py);
                } else {
                    mv->submask = orc_bs_get_bits(&sub[3], 1) ? 0xF : (uint8_t)orc_bs_get_bits(&sub[3], 4);
                    stable[i + j * prm.nblocks_h] |= 2;
                }
            }
    }
    orc_frame *resid = orc_frame_new(fmt, w, h, 1);
    orc_bs_align(&bs);
    const int quant = (int)orc_bs_get_bits(&bs, 11);

    orc_stability st;
    st.params = &prm;
    st.stable_blocks = stable;
    st.isP = (unsigned char)prm.has_ref;
    for (int c = 0; c < 3; c++) {
        orc_coefs co;
        orc_bs_align(&bs);
        int plen = (int)orc_bs_get_bits(&bs, 32);
        orc_bs_align(&bs);
        co.width = c ? round_even(resid->planes[c].w) : resid->planes[c].w;
        co.height = c ? round_even(resid->planes[c].h) : resid->planes[c].h;
        size_t bytes = (size_t)co.width * co.height * sizeof(int32_t);
        if (plen <= 0 || (size_t)plen > bytes * 2) break;
        uint8_t *enc = (uint8_t *)pkt + orc_bs_bytepos(&bs);
        bs.pos += (unsigned)plen * 8;
        co.data = (int32_t *)calloc(1, bytes);
        st.cur_plane = (unsigned char)c;
        orc_decode_plane(enc, (unsigned)plen, &co, quant, &st);
        orc_inv_sbt(&resid->planes[c], &co, quant, st.isP, c);
        free(co.data);
    }

    orc_frame *outf = orc_frame_new(fmt, w, h, 1);
    int rc = 0;
    if (prm.has_ref) {
        if (!d->ref) rc = 1;
        else orc_add_pred(mvs, &prm, resid, outf, d->ref);
    } else {
        orc_frame_copy(outf, resid);
    }
    if (rc == 0 && yuv_out) {
        uint8_t *o = yuv_out;
        for (int c = 0; c < 3; c++)
            for (int y = 0; y < outf->planes[c].h; y++, o += outf->planes[c].w)
                memcpy(o, outf->planes[c].data + (size_t)y * outf->planes[c].stride, (size_t)outf->planes[c].w);
    }
    if (rc == 0 && is_ref) {
        orc_frame_extend(outf);
        orc_frame_free(d->ref);
        d->ref = outf;
        outf = NULL;
    }
    orc_frame_free(outf);
    orc_frame_free(resid);
    free(mvs);
    free(stable);
    return rc;
}
