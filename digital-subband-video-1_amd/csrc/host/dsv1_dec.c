/* dsv1_dec.c -- decoder session layer in plain C (dsv_decoder.c:21-145,286-472 semantics).
 * The host parses packet headers and side information (a few hundred bytes) and hands the three plane
 * payloads to the device: entropy parse, scatter + dequantise, inverse transform and motion compensation
 * run on the GPU.  dsv_dec() is the reference's one-picture-per-call entry (the frame is copied back into
 * a host DSV_FRAME); dsv1_decbatch_* decodes one picture of each of many independent streams per call.
 * Debug overlays (draw_info) are accepted and ignored. */
#include <stdio.h>
#include <time.h>
#include "dsv1_host.h"
#define DSV_MIN_BLOCK_SIZE 16          /* dsv.h:50-51 */
#define DSV_MAX_BLOCK_SIZE 64

typedef struct {
    dsvg_ctx *ctx;
    dsvg_geom g;
    int have_ref, rpar;         /* rpar: which of the two reference slots holds the current reference picture */
    dsv1_frame_pool *pool;      /* pinned output frames in the device's frame layout (round 5): a picture comes back by one asynchronous copy */
    unsigned char *stable;      /* [nblk] side information of the picture being decoded (kept from call to call) */
    DSV_MV *mvs;
} dec_sess;

static void sess_close(dec_sess *s)
{
    if (!s) return;
    if (s->ctx) dsvg_ctx_destroy(s->ctx);
    dsv1_pool_unref(s->pool);                           /* (frames the caller still holds keep their buffers alive) */
    free(s->stable); free(s->mvs);
    dsv1_recycle_hold(-1);
    free(s);
}

void dsv_dec_free(DSV_DECODER *d)
{
    if (d->ref) {
        sess_close((dec_sess *)d->ref);
        d->ref = NULL;
    }
}

DSV_META *dsv_get_metadata(DSV_DECODER *d)
{
    DSV_META *m = (DSV_META *)dsv_alloc((int)sizeof(DSV_META));
    memcpy(m, &d->vidmeta, sizeof(DSV_META));
    return m;
}

/* packet header (dsv_decoder.c:300-318): returns the packet type, leaves r after the two link words */
static int parse_packet_header(bitw *r, uint8_t *data, unsigned len, int *type)
{
    if (len < DSV_PACKET_HDR_SIZE) return -1;
    br_init(r, data, len);
    if (br_bits(r, 8) != 'D' || br_bits(r, 8) != 'S' || br_bits(r, 8) != 'V' || br_bits(r, 8) != '1') return -1;
    (void)br_bits(r, 8);
    *type = (int)br_bits(r, 8);
    (void)br_bits(r, 32);
    (void)br_bits(r, 32);
    return 0;
}

static void parse_meta(bitw *r, DSV_META *m)
{
    m->width = (int)br_ueg(r); m->height = (int)br_ueg(r); m->subsamp = (int)br_ueg(r);
    m->fps_num = (int)br_ueg(r); m->fps_den = (int)br_ueg(r);
    m->aspect_num = (int)br_ueg(r); m->aspect_den = (int)br_ueg(r);
}

/* picture packet after the header, first part (dsv_decoder.c:335-352): frame number and block size */
static int parse_picture_head(bitw *r, DSV_FNUM *fn, int *bw_, int *bh_)
{
    bw_align(r);
    *fn = br_bits(r, 32);
    bw_align(r);
    *bw_ = (int)br_ueg(r) << 2;
    *bh_ = (int)br_ueg(r) << 2;
    return (*bw_ < 16 || *bh_ < 16 || *bw_ > 64 || *bh_ > 64) ? -1 : 0;
}

/* rest of the picture packet: stability flags (decode_stability_blocks dsv_decoder.c:127-145), motion
 * (decode_motion dsv_decoder.c:73-124), quantiser and the three plane payloads -> a device job.
 * stable[nblk] and mvs[nblk] are caller storage (zeroed here).  Unlike the reference (which only checks
 * plen <= framesz * 2) every length the packet announces is held against the packet's own length `len`: a truncated or
 * hostile packet is refused, never read past. */
static int sub_fits(const bitw *r, unsigned n, unsigned len) { return !r->over && n <= len && bw_bytes(r) <= len - n; }
static int parse_picture_body(bitw *r, uint8_t *data, unsigned len, const dsvg_geom *g, int has_ref, unsigned char *stable, DSV_MV *mvs, dsvg_dec_job *job)
{
    const int nbh = g->nblocks_h, nbv = g->nblocks_v, nblk = nbh * nbv;
    int i, j, c;
    memset(stable, 0, (size_t)nblk);
    {
        zrle z;
        unsigned n;
        bw_align(r);
        n = br_ueg(r);
        bw_align(r);
        if (!sub_fits(r, n, len)) { dsv1_log(1, "stability block runs past the packet"); return -1; }
        zr_init_rd(&z, data + bw_bytes(r), n);
        r->pos += n * 8;
        for (i = 0; i < nblk; i++) stable[i] = (unsigned char)zr_get(&z);
    }
    if (has_ref) {
        bitw sub[4];
        zrle modes;
        DSV_PARAMS prm;
        memset(mvs, 0, (size_t)nblk * sizeof(DSV_MV));
        memset(&prm, 0, sizeof(prm));
        prm.nblocks_h = nbh; prm.nblocks_v = nbv;
        bw_align(r);
        for (i = 0; i < 4; i++) {
            const unsigned n = br_ueg(r);
            bw_align(r);
            if (!sub_fits(r, n, len)) { dsv1_log(1, "motion data run past the packet"); return -1; }
            br_init(&sub[i], data + bw_bytes(r), n);
            r->pos += n * 8;
        }
        zr_init_rd(&modes, sub[0].p, sub[0].end >> 3);
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
    bw_align(r);
    memset(job, 0, sizeof(*job));
    job->quant = (int)br_bits(r, 11);
    job->mvs = (const dsvg_mv *)mvs;
    job->stable_blocks = stable;
    for (c = 0; c < 3; c++) {
        int plen;
        bw_align(r);
        plen = (int)br_bits(r, 32);
        bw_align(r);
        /* the reference's bound (dsv_decoder.c:389-400: twice the int32 coefficient plane, chroma dimensions rounded up to
         * even) -- dsvg_decode_pictures applies the same; the packet's own length is the bound that keeps the reads safe */
        {
            const int hs = c ? (g->subsamp >> 2) & 3 : 0, vs = c ? g->subsamp & 3 : 0;
            size_t pw = (size_t)((g->width + (1 << hs) - 1) >> hs), phh = (size_t)((g->height + (1 << vs) - 1) >> vs);
            if (c) { pw = (pw + 1) & ~(size_t)1; phh = (phh + 1) & ~(size_t)1; }
            if (plen > 0 && (size_t)plen > pw * phh * 8) plen = -1;
        }
        if (plen <= 0 || !sub_fits(r, (unsigned)plen, len)) {
            dsv1_log(1, "plane length was strange: %d", plen);
            return -1;
        }
        job->plane_data[c] = data + bw_bytes(r);
        job->plane_len[c] = (uint32_t)plen;
        r->pos += (unsigned)plen * 8;
    }
    return 0;
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
    int type, is_ref, has_ref, bw_, bh_, nblk, i, c, rc, ret = DSV_DEC_ERROR;

    *fn = (DSV_FNUM)-1;
    if (parse_packet_header(&r, buffer->data, buffer->len, &type)) {
        dsv1_log(1, "bad 4cc");
        dsv_buf_free(buffer);
        return DSV_DEC_ERROR;
    }
    if (!(type & DSV_PT_PIC)) {
        if (type == DSV_PT_META) {
            parse_meta(&r, m);
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
    if (parse_picture_head(&r, fn, &bw_, &bh_)) { dsv_buf_free(buffer); return DSV_DEC_ERROR; }

    if (bw_ < DSV_MIN_BLOCK_SIZE || bh_ < DSV_MIN_BLOCK_SIZE || bw_ > DSV_MAX_BLOCK_SIZE || bh_ > DSV_MAX_BLOCK_SIZE) {
        dsv1_log(1, "bad block sizes %dx%d", bw_, bh_);       /* dsv_decoder.c:354-358 */
        dsv_buf_free(buffer);
        return DSV_DEC_ERROR;
    }
    ss = (dec_sess *)d->ref;
    if (!ss || bw_ != ss->g.blk_w || bh_ != ss->g.blk_h) {
        /* the block size is the stream's to choose, per picture (dsv_decoder.c:335-360): a context for this one; the
         * reference picture, if this picture needs one, moves over in the frame layout both contexts share */
        dec_sess *ns = (dec_sess *)calloc(1, sizeof(*ns));
        if (!ns || (rc = dsvg_ctx_create_blk(&ns->ctx, dsv1_device, m->width, m->height, m->subsamp, 1, 1, 3, 1, 1, bw_, bh_))) {
            dsv1_log(1, "GPU session could not be opened: %s", dsvg_last_error());
            free(ns);
            dsv_buf_free(buffer);
            return DSV_DEC_ERROR;
        }
        dsvg_ctx_geom(ns->ctx, &ns->g);
        dsv1_recycle_hold(+1);                           /* (frames that do not come from the pool: their blocks are parked, not unmapped) */
        ns->stable = (unsigned char *)calloc((size_t)ns->g.nblocks_h * ns->g.nblocks_v, 1);
        ns->mvs = (DSV_MV *)calloc((size_t)ns->g.nblocks_h * ns->g.nblocks_v, sizeof(DSV_MV));
        if (!ns->stable || !ns->mvs) { dsv1_log(1, "out of memory"); sess_close(ns); dsv_buf_free(buffer); return DSV_DEC_ERROR; }
        {
            const char *e = getenv("DSV1_DEC_NO_POOL");    /* (A/B: the packed download + row copy of rounds 1-4) */
            if (!(e && atoi(e) != 0)) ns->pool = dsv1_pool_new(ns->ctx, dsv1_device, ns->g.frame_alloc_bytes, DSV1_POOL_FRAMES);
        }
        if (ss) {
            if (ss->have_ref) {
                const size_t nb = ss->g.frame_alloc_bytes;
                uint8_t *raw = (uint8_t *)malloc(nb);
                if (!raw || dsvg_download_recon_raw(ss->ctx, ss->rpar, raw, nb) || dsvg_upload_recon_raw(ns->ctx, 0, raw, nb)) {
                    dsv1_log(1, "reference picture could not be carried over: %s", dsvg_last_error());
                    free(raw); sess_close(ns);
                    dsv_buf_free(buffer);
                    return DSV_DEC_ERROR;
                }
                free(raw);
                ns->have_ref = 1; ns->rpar = 0;
            }
            sess_close(ss);
        }
        d->ref = ss = ns;
    }
    /* DSV1_DEC_PROF=1: where a call's time goes (side information parse / dsvg_decode_pictures = uploads + launches / the wait for the picture) */
    static int prof = -1;
    static double pt[4];
    static long pn;
    double t0 = 0, t1 = 0, t2 = 0;
    if (prof < 0) { const char *e = getenv("DSV1_DEC_PROF"); prof = e && atoi(e) != 0; }
#define DEC_NOW(v) do { if (prof) { struct timespec ts_; clock_gettime(CLOCK_MONOTONIC, &ts_); v = ts_.tv_sec * 1e6 + ts_.tv_nsec * 1e-3; } } while (0)
    DEC_NOW(t0);
    nblk = ss->g.nblocks_h * ss->g.nblocks_v;
    stable = ss->stable; mvs = ss->mvs;
    memset(stable, 0, (size_t)nblk);
    memset(mvs, 0, (size_t)nblk * sizeof(DSV_MV));
    if (parse_picture_body(&r, buffer->data, buffer->len, &ss->g, has_ref, stable, mvs, &job)) goto done;
    if (has_ref && !ss->have_ref) {
        dsv1_log(2, "reference frame not found");
        goto done;
    }
    /* reference pictures alternate between slots 0 and 1 (the prediction of a P picture is written straight into the
     * slot its reconstruction will live in, dsvg_decode_pictures); other pictures go to slot 2 */
    job.ref_recon_slot = has_ref ? ss->rpar : -1;
    job.recon_slot = is_ref ? (ss->rpar ^ 1) : 2;
    DEC_NOW(t1);
    if ((rc = dsvg_decode_pictures(ss->ctx, 1, &job))) {
        dsv1_log(1, "GPU decode failed: %s", dsvg_last_error());
        goto done;
    }
    DEC_NOW(t2);
    /* the picture as the device holds it -- the reference's own frame layout -- straight into a pinned frame of the pool: one copy */
    f = dsv1_pool_frame(ss->pool, m->subsamp, m->width, m->height);
    if (f) {
        if ((rc = dsvg_download_recon_frame(ss->ctx, job.recon_slot, f->alloc, ss->g.frame_alloc_bytes))) {
            dsv1_log(1, "GPU download failed: %s", dsvg_last_error());
            dsv_frame_ref_dec(f);
            goto done;
        }
        if (is_ref) { ss->have_ref = 1; ss->rpar ^= 1; }
        *out = f;
        ret = DSV_DEC_OK;
        if (prof) {
            double t3 = 0;
            DEC_NOW(t3);
            pt[0] += t1 - t0; pt[1] += t2 - t1; pt[2] += t3 - t2;
            if (++pn % 24 == 0) fprintf(stderr, "[dsv_dec] %ld pictures: parse %.1f us, decode_pictures (uploads + launches) %.1f us, frame + download (wait) %.1f us per call\n",
                                        pn, pt[0] / pn, pt[1] / pn, pt[2] / pn);
        }
        goto done;
    }
    /* (every pool frame is still with the caller, or no pool: the packed download and a frame of its own) */
    packed = (uint8_t *)malloc(ss->g.frame_bytes);
    if (!packed || (rc = dsvg_download_recon(ss->ctx, job.recon_slot, packed))) {
        dsv1_log(1, "GPU download failed: %s", dsvg_last_error());
        goto done;
    }
    if (is_ref) { ss->have_ref = 1; ss->rpar ^= 1; }
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
    dsv_buf_free(buffer);
    return ret;
}

/* ---- batched decoder: one picture of each of nstreams independent streams per call ---- */
struct dsv1_decbatch {
    dsvg_ctx *ctx;
    dsvg_geom g;
    DSV_META meta;
    int nstreams, nblk, device;
    unsigned char *have_ref;         /* [nstreams] */
    unsigned char *rpar;             /* [nstreams] which of the stream's two reference slots holds its current reference */
    unsigned char *stable;           /* [nstreams][nblk] */
    DSV_MV *mvs;                     /* [nstreams][nblk] */
    dsvg_dec_job *pj;                /* [nstreams] parsed job of each stream's packet */
    dsvg_dec_job *jobs;              /* compacted: the picture packets of this call */
    int *slots;
};

void dsv1_decbatch_close(dsv1_decbatch *d)
{
    if (!d) return;
    if (d->ctx) dsvg_ctx_destroy(d->ctx);
    free(d->have_ref); free(d->rpar); free(d->stable); free(d->mvs); free(d->jobs); free(d->pj); free(d->slots);
    free(d);
}

int dsv1_decbatch_open(dsv1_decbatch **out, int device, const DSV_META *meta, int nstreams)
{
    dsv1_decbatch *d;
    int rc;
    if (!out || !meta || nstreams < 1) return DSVG_ERR_ARG;
    *out = NULL;
    d = (dsv1_decbatch *)calloc(1, sizeof(*d));
    if (!d) return DSVG_ERR_ARG;
    d->meta = *meta;
    d->nstreams = nstreams;
    d->device = device;
    /* stream s keeps its reference pictures alternately in reconstruction slots s and nstreams + s (ping-pong: a P picture's
     * prediction is written straight into its own slot); non-reference pictures go to slot 2 * nstreams + s */
    if ((rc = dsvg_ctx_create(&d->ctx, device, meta->width, meta->height, meta->subsamp, 1, 1, 3 * nstreams, nstreams, nstreams))) {
        free(d);
        return rc;
    }
    dsvg_ctx_geom(d->ctx, &d->g);
    d->nblk = d->g.nblocks_h * d->g.nblocks_v;
    d->have_ref = (unsigned char *)calloc((size_t)nstreams, 1);
    d->rpar = (unsigned char *)calloc((size_t)nstreams, 1);
    d->stable = (unsigned char *)calloc((size_t)nstreams * d->nblk, 1);
    d->mvs = (DSV_MV *)calloc((size_t)nstreams * d->nblk, sizeof(DSV_MV));
    d->jobs = (dsvg_dec_job *)calloc((size_t)nstreams, sizeof(dsvg_dec_job));
    d->pj = (dsvg_dec_job *)calloc((size_t)nstreams, sizeof(dsvg_dec_job));
    d->slots = (int *)calloc((size_t)nstreams, sizeof(int));
    *out = d;
    return DSVG_OK;
}

/* host part of one stream's packet: header + side information (a few hundred bytes) -> pj[s], status[s], fnum[s] */
typedef struct { dsv1_decbatch *d; const DSV_BUF *packets; int *status; DSV_FNUM *fnum; } parse_ctx;
static void parse_stream(void *vp, int s, int tid)
{
    parse_ctx *pc = (parse_ctx *)vp;
    dsv1_decbatch *d = pc->d;
    bitw r;
    int type, bw_, bh_, has_ref, is_ref;
    dsvg_dec_job *job = &d->pj[s];
    (void)tid;
    pc->status[s] = DSV_DEC_ERROR;
    pc->fnum[s] = (DSV_FNUM)-1;
    if (!pc->packets[s].data || parse_packet_header(&r, pc->packets[s].data, pc->packets[s].len, &type)) return;
    if (!(type & DSV_PT_PIC)) {
        if (type == DSV_PT_META) {
            DSV_META m;
            parse_meta(&r, &m);
            pc->status[s] = (m.width == d->meta.width && m.height == d->meta.height && m.subsamp == d->meta.subsamp) ? DSV_DEC_GOT_META : DSV_DEC_ERROR;
        } else if (type == DSV_PT_EOS) {
            pc->status[s] = DSV_DEC_EOS;
        }
        return;
    }
    has_ref = type & 1;
    is_ref = (type & 0x6) == 0x6;
    if (parse_picture_head(&r, &pc->fnum[s], &bw_, &bh_)) return;
    if (bw_ != d->g.blk_w || bh_ != d->g.blk_h) {
        /* (the batch has ONE geometry: dsv1_decbatch_decode follows the streams' block size while no stream holds a reference) */
        dsv1_log(1, "stream %d: block size %dx%d, the batch decodes %dx%d", s, bw_, bh_, d->g.blk_w, d->g.blk_h);
        return;
    }
    if (parse_picture_body(&r, pc->packets[s].data, pc->packets[s].len, &d->g, has_ref, d->stable + (size_t)s * d->nblk, d->mvs + (size_t)s * d->nblk, job)) return;
    if (has_ref && !d->have_ref[s]) {
        dsv1_log(2, "stream %d: reference frame not found", s);
        return;
    }
    job->ref_recon_slot = has_ref ? s + d->nstreams * d->rpar[s] : -1;
    job->recon_slot = is_ref ? s + d->nstreams * (d->rpar[s] ^ 1) : 2 * d->nstreams + s;
    if (is_ref) { d->have_ref[s] = 1; d->rpar[s] ^= 1; }
    pc->status[s] = DSV_DEC_OK;
}

int dsv1_decbatch_decode(dsv1_decbatch *d, const DSV_BUF *packets, void *yuv_out, size_t out_pitch, int out_on_device, int *status, DSV_FNUM *fnum)
{
    int s, n = 0, rc;
    parse_ctx pc;
    if (!d || !packets || !yuv_out || !status || !fnum) return DSVG_ERR_ARG;
    if (out_pitch == 0) out_pitch = d->g.frame_bytes;
    {   /* The block size is the streams' to choose (dsv_decoder.c:335-360); the batch shares one context, so it follows the
         * first picture packet of a call as long as no stream holds a reference picture (i.e. at the streams' start or where
         * every stream restarts with an I picture); a change in mid-GOP is an error of that stream (parse_stream). */
        int any_ref = 0;
        for (s = 0; s < d->nstreams; s++) any_ref |= d->have_ref[s];
        for (s = 0; s < d->nstreams && !any_ref; s++) {
            bitw r;
            int type, bw_, bh_;
            DSV_FNUM f_;
            if (!packets[s].data || parse_packet_header(&r, packets[s].data, packets[s].len, &type) || !(type & DSV_PT_PIC)) continue;
            if (parse_picture_head(&r, &f_, &bw_, &bh_)) continue;
            if ((bw_ != d->g.blk_w || bh_ != d->g.blk_h) && bw_ >= DSV_MIN_BLOCK_SIZE && bh_ >= DSV_MIN_BLOCK_SIZE &&
                bw_ <= DSV_MAX_BLOCK_SIZE && bh_ <= DSV_MAX_BLOCK_SIZE) {
                dsvg_ctx *nc = NULL;
                if ((rc = dsvg_ctx_create_blk(&nc, d->device, d->meta.width, d->meta.height, d->meta.subsamp, 1, 1, 3 * d->nstreams, d->nstreams,
                                              d->nstreams, bw_, bh_))) return rc;
                dsvg_ctx_destroy(d->ctx);
                d->ctx = nc;
                dsvg_ctx_geom(d->ctx, &d->g);
                d->nblk = d->g.nblocks_h * d->g.nblocks_v;
                free(d->stable); free(d->mvs);
                d->stable = (unsigned char *)calloc((size_t)d->nstreams * d->nblk, 1);
                d->mvs = (DSV_MV *)calloc((size_t)d->nstreams * d->nblk, sizeof(DSV_MV));
            }
            break;
        }
    }
    pc.d = d; pc.packets = packets; pc.status = status; pc.fnum = fnum;
    dsv1_par_for(d->nstreams, parse_stream, &pc);
    for (s = 0; s < d->nstreams; s++)
        if (status[s] == DSV_DEC_OK && fnum[s] != (DSV_FNUM)-1) {
            d->jobs[n] = d->pj[s];
            d->slots[n] = s;             /* whose job this is */
            n++;
        }
    if (!n) return DSVG_OK;
    /* device: all pictures of the call as one batch, then one packing pass into the caller's layout */
    if ((rc = dsvg_decode_pictures(d->ctx, n, d->jobs))) {
        dsv1_log(1, "GPU decode failed: %s", dsvg_last_error());
        for (s = 0; s < n; s++) status[d->slots[s]] = DSV_DEC_ERROR;
        return rc;
    }
    if (n == d->nstreams) {
        for (s = 0; s < n; s++) d->slots[s] = d->jobs[s].recon_slot;
        rc = dsvg_pack_recons(d->ctx, n, d->slots, yuv_out, out_pitch, out_on_device);
    } else {                             /* some streams had no picture this call: their output frames stay untouched */
        rc = DSVG_OK;
        for (s = 0; s < n && !rc; s++) {
            const int st = d->slots[s], slot = d->jobs[s].recon_slot;
            rc = dsvg_pack_recons(d->ctx, 1, &slot, (uint8_t *)yuv_out + (size_t)st * out_pitch, out_pitch, out_on_device);
        }
    }
    if (rc) dsv1_log(1, "GPU pack failed: %s", dsvg_last_error());
    return rc;
}

void *dsv1_decbatch_ctx(dsv1_decbatch *d) { return d ? (void *)d->ctx : NULL; }
