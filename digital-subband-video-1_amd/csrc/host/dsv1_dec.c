/* dsv1_dec.c -- decoder session layer in plain C (dsv_decoder.c:21-145,286-472 semantics).
 * The host parses packet headers and side information (a few hundred bytes), hands the three plane
 * payloads to the device (entropy parse on the host for now, scatter + dequantise + inverse
 * transform + motion compensation on the GPU), and copies the finished picture back.
 * Debug overlays (draw_info) are accepted and ignored. */
#include <stdio.h>
#include "dsv1_host.h"

typedef struct {
    dsvg_ctx *ctx;
    dsvg_geom g;
    int have_ref;
} dec_sess;

void dsv_dec_free(DSV_DECODER *d)
{
    if (d->ref) {
        dec_sess *s = (dec_sess *)d->ref;
        if (s->ctx) dsvg_ctx_destroy(s->ctx);
        free(s);
        d->ref = NULL;
    }
}

DSV_META *dsv_get_metadata(DSV_DECODER *d)
{
    DSV_META *m = (DSV_META *)dsv_alloc((int)sizeof(DSV_META));
    memcpy(m, &d->vidmeta, sizeof(DSV_META));
    return m;
}

int dsv_dec(DSV_DECODER *d, DSV_BUF *buffer, DSV_FRAME **out, DSV_FNUM *fn)
{
    bitw r;
    DSV_META *m = &d->vidmeta;
    dec_sess *ss;
    dsvg_dec_job job;
    DSV_MV *mvs = NULL;
    unsigned char *stable = NULL;
    uint8_t *packed = NULL;
    DSV_FRAME *f;
    int type, is_ref, has_ref, bw_, bh_, nbh, nbv, nblk, i, j, c, rc, ret = DSV_DEC_ERROR;

    *fn = (DSV_FNUM)-1;
    bw_init(&r, buffer->data);
    if (br_bits(&r, 8) != 'D' || br_bits(&r, 8) != 'S' || br_bits(&r, 8) != 'V' || br_bits(&r, 8) != '1') {
        dsv1_log(1, "bad 4cc");
        dsv_buf_free(buffer);
        return DSV_DEC_ERROR;
    }
    (void)br_bits(&r, 8);
    type = (int)br_bits(&r, 8);
    (void)br_bits(&r, 32);
    (void)br_bits(&r, 32);
    if (!(type & DSV_PT_PIC)) {
        if (type == DSV_PT_META) {
            m->width = (int)br_ueg(&r); m->height = (int)br_ueg(&r); m->subsamp = (int)br_ueg(&r);
            m->fps_num = (int)br_ueg(&r); m->fps_den = (int)br_ueg(&r);
            m->aspect_num = (int)br_ueg(&r); m->aspect_den = (int)br_ueg(&r);
            d->got_metadata = 1;
            ret = DSV_DEC_GOT_META;
        } else if (type == DSV_PT_EOS) {
            ret = DSV_DEC_EOS;
        }
        dsv_buf_free(buffer);
        return ret;
    }
    if (!d->got_metadata) {
        dsv1_log(2, "no metadata, skipping frame");
        dsv_buf_free(buffer);
        return DSV_DEC_OK;
    }
    has_ref = type & 1;
    is_ref = (type & 0x6) == 0x6;
    bw_align(&r);
    *fn = br_bits(&r, 32);
    bw_align(&r);
    bw_ = (int)br_ueg(&r) << 2;
    bh_ = (int)br_ueg(&r) << 2;
    if (bw_ < 16 || bh_ < 16 || bw_ > 64 || bh_ > 64) { dsv_buf_free(buffer); return DSV_DEC_ERROR; }

    if (!d->ref) {
        ss = (dec_sess *)calloc(1, sizeof(*ss));
        if ((rc = dsvg_ctx_create(&ss->ctx, dsv1_device, m->width, m->height, m->subsamp, 1, 1, 2, 1, 1))) {
            dsv1_log(1, "GPU session could not be opened: %s", dsvg_last_error());
            free(ss);
            dsv_buf_free(buffer);
            return DSV_DEC_ERROR;
        }
        dsvg_ctx_geom(ss->ctx, &ss->g);
        d->ref = ss;
    }
    ss = (dec_sess *)d->ref;
    nbh = (m->width + bw_ - 1) / bw_;
    nbv = (m->height + bh_ - 1) / bh_;
    if (bw_ != ss->g.blk_w || bh_ != ss->g.blk_h) {
        dsv1_log(1, "stream block size %dx%d differs from the encoder rule for this frame size", bw_, bh_);
        dsv_buf_free(buffer);
        return DSV_DEC_ERROR;
    }
    nblk = nbh * nbv;
    stable = (unsigned char *)calloc((size_t)nblk, 1);
    {   /* stability flags (decode_stability_blocks dsv_decoder.c:127-145) */
        zrle z;
        unsigned n;
        bw_align(&r);
        n = br_ueg(&r);
        bw_align(&r);
        zr_init(&z, buffer->data + bw_bytes(&r));
        r.pos += n * 8;
        for (i = 0; i < nblk; i++) stable[i] = (unsigned char)zr_get(&z);
    }
    if (has_ref) {   /* decode_motion dsv_decoder.c:73-124 */
        bitw sub[4];
        zrle modes;
        DSV_PARAMS prm;
        mvs = (DSV_MV *)calloc((size_t)nblk, sizeof(DSV_MV));
        memset(&prm, 0, sizeof(prm));
        prm.nblocks_h = nbh; prm.nblocks_v = nbv;
        bw_align(&r);
        for (i = 0; i < 4; i++) {
            const unsigned n = br_ueg(&r);
            bw_align(&r);
            bw_init(&sub[i], buffer->data + bw_bytes(&r));
            r.pos += n * 8;
        }
        zr_init(&modes, sub[0].p);
        for (j = 0; j < nbv; j++)
            for (i = 0; i < nbh; i++) {
                DSV_MV *mv = &mvs[i + j * nbh];
                mv->mode = (uint8_t)zr_get(&modes);
                if (mv->mode == 0) {
                    int px, py;
                    dsv_movec_pred(mvs, &prm, i, j, &px, &py);
                    mv->u.mv.x = (int16_t)(br_seg(&sub[1]) + px);
                    mv->u.mv.y = (int16_t)(br_seg(&sub[2]) + py);
                } else {
                    mv->submask = br_bit(&sub[3]) ? 0xF : (uint8_t)br_bits(&sub[3], 4);
                    stable[i + j * nbh] |= 2;
                }
            }
    }
    bw_align(&r);
    memset(&job, 0, sizeof(job));
    job.quant = (int)br_bits(&r, 11);
    job.mvs = (const dsvg_mv *)mvs;
    job.stable_blocks = stable;
    for (c = 0; c < 3; c++) {
        int plen;
        bw_align(&r);
        plen = (int)br_bits(&r, 32);
        bw_align(&r);
        if (plen <= 0 || (size_t)plen > ss->g.plane_out_cap[c] * 4 + 64) {
            dsv1_log(1, "plane length was strange: %d", plen);
            goto done;
        }
        job.plane_data[c] = buffer->data + bw_bytes(&r);
        job.plane_len[c] = (uint32_t)plen;
        r.pos += (unsigned)plen * 8;
    }
    if (has_ref && !ss->have_ref) {
        dsv1_log(2, "reference frame not found");
        goto done;
    }
    job.ref_recon_slot = has_ref ? 0 : -1;
    job.recon_slot = is_ref ? 0 : 1;
    if ((rc = dsvg_decode_pictures(ss->ctx, 1, &job))) {
        dsv1_log(1, "GPU decode failed: %s", dsvg_last_error());
        goto done;
    }
    packed = (uint8_t *)malloc(ss->g.frame_bytes);
    if ((rc = dsvg_download_recon(ss->ctx, job.recon_slot, packed))) {
        dsv1_log(1, "GPU download failed: %s", dsvg_last_error());
        goto done;
    }
    if (is_ref) ss->have_ref = 1;
    f = dsv_mk_frame(m->subsamp, m->width, m->height, 1);
    {
        const uint8_t *o = packed;
        for (c = 0; c < 3; c++)
            for (i = 0; i < f->planes[c].h; i++, o += f->planes[c].w)
                memcpy(f->planes[c].data + (size_t)i * f->planes[c].stride, o, (size_t)f->planes[c].w);
    }
    *out = f;
    ret = DSV_DEC_OK;
done:
    free(packed);
    free(mvs);
    free(stable);
    dsv_buf_free(buffer);
    return ret;
}
