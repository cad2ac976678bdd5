/* dsv1_enc.c -- encoder session layer in plain C on top of the device pipeline (include/dsvg.h).
 *
 * Keeps the reference's semantics for everything that is NOT per-pixel work: GOP / forced-intra /
 * scene-change decisions (dsv_encoder.c:538-554,624-653), CRF/ABR quantiser control
 * (dsv_encoder.c:70-168,816-848), the stability accumulators and side-info coding
 * (dsv_encoder.c:257-408), packet framing and link offsets (dsv_encoder.c:171-192,410-536,766-778).
 *
 * Structure differs from the reference on purpose: frames live in HBM, and all source-only analysis
 * (padding, pyramid, mean luma, motion estimation -- SURVEY.md fact 3) of a whole batch
 * (streams x frames) runs first, then the reconstruction-dependent residual chain is enqueued frame
 * step by frame step across all streams without host synchronisation (CRF), and packets are
 * assembled once at the end.  ABR needs each packet's size before the next quantiser, so it runs
 * the same code one frame step at a time. */
#include <stdio.h>
#include "dsv1_host.h"
#include "dsvg_rc.h"              /* the rate control, shared with the device (k_rc.hip) */

typedef struct {
    DSV_FNUM fnum;
    int gop_start, is_ref, has_ref, forced_intra, isP, quant, n_intra;
    short reach[4];                  /* min / max of the inter blocks' mv.x >> 1, mv.y >> 1 */
    int cur_slot, ref_slot, out_slot;
    DSV_MV *mvs;
    unsigned char *stable;
    uint8_t *prefix;            /* packet bytes up to (not including) the 11-bit quantiser */
    unsigned prefix_len;
} pic_t;

typedef struct { uint8_t *pkt; size_t cap; } pkt_scratch;   /* packet staging buffer (one per assembling thread) */
typedef struct { dsv1_batch *b; pic_t *pics; volatile int rc; int nf; } side_ctx;   /* rc: a worker's failure (scratch allocation), checked after the parallel loop; nf: the frames of THIS batch (a background loop outlives the submit call) */
struct dsv1_batch {
    /* round 6: the packet prefixes of a CRF batch are written by a BACKGROUND loop of the worker pool (dsv1_par_bg_begin) that starts when the batch's coding
     * work has been enqueued and is joined when the batch is collected: bg_sc[parity] is that loop's context, bg_on[parity] says it has not been joined */
    side_ctx bg_sc[2];
    int bg_on[2];
    dsvg_ctx *ctx;
    dsvg_geom g;
    int nstreams, F, own_enc, nblk, prefix_cap, small_w, small_h;
    int rows;                   /* source-slot ring: rows x nstreams slots, rows = 2F+1 */
    unsigned gcount;            /* frames submitted so far per stream (ring position) */
    int parity;                 /* which half of the double-buffered per-batch state the next submit uses */
    int nf_cur;                 /* frames per stream of the batch being submitted (F, or fewer for a single stream's tail) */
    int nf_pending[2];          /* the same for the batches in flight */
    int pending[2];             /* batch submitted (device work enqueued) but not yet collected */
    DSV_ENCODER *enc;
    pic_t *pics;                /* [2][nstreams*F] */
    DSV_MV *mvpool;
    unsigned char *stabpool;
    uint8_t *prefixpool;
    const void *staged_host[2];      /* dsv1_batch_stage: FIFO of host clips whose upload is already queued */
    void *staged_dev[2];
    int nstaged;
    int *slots_cur, *slots_ref, *pair_pic, *out_slots;
    unsigned char *rpar;             /* per stream: which of its two reconstruction slots holds the current reference */
    unsigned char *has_recon;        /* per stream: a reference picture has been coded */
    unsigned char *border_skipped;   /* per stream: the last reconstruction was coded with border_hint (next picture: a GOP start) */
    /* Round 5: a picture nobody predicts from gets no reconstruction (dsvg_pic_job.recon_slot = -1: no inverse transform).  Inside a call
     * that is known exactly (the next picture of the stream has no reference: it starts a GOP or a scene).  For a call's LAST picture it
     * is known when the next frame number starts a GOP (dsv_encoder.c:702-708) -- unless the caller renumbers the stream in between
     * (dsv1_batch_set_fnum); the next submit then finds a P picture where a GOP start was promised and codes the dropped picture once
     * more, this time keeping its reconstruction (remedy_dropped: same source slot, vectors, flags and quantiser -- they are still in
     * the other half of pics[] -- the packet is discarded).  ABR streams keep their last reconstruction (their quantisers live on the device). */
    unsigned char *recon_dropped;    /* per stream: the last picture of the batch before was coded without a reconstruction */
    int nf_prev;                     /* frames per stream of the batch before */
    int keep_all;                    /* DSV1_RECON_ALL=1: reconstruct every reference picture, read or not */
    long n_dropped, n_remedied;
    unsigned *luma;
    DSV_MV *mv_tmp;
    dsvg_pic_job *jobs;
    dsvg_pic_out *outs;
    pkt_scratch sc0;                 /* packet staging of the serial (ABR / single-frame) path */
    /* ABR with the rate control ON THE DEVICE (round 4, include/dsvg_rc.h + k_rc): the whole call is enqueued like a CRF call, k_rc
     * turns every packet's size into the next picture's quantiser tables; the host replays the same code when it assembles the
     * packets and refuses a batch whose quantisers differ.  DSV1_ABR_SERIAL=1: the frame-by-frame host path of rounds 1-3. */
    int holds_recycler;              /* counted in dsv1_recycle_hold (dsv1_util.c): freed packet buffers are parked while a batch is open */
    int abr_dev;
    int rc_seeded;                   /* the device holds the streams' rate-control state (seeded from enc[] by the first call) */
    dsvg_rc_job *rcjobs;
    dsvg_rc_state *rc_dev;           /* [nstreams] the rate-control PARAMETERS the device holds (re-sent when the caller changed a stream's: advisor round 4) */
    dsvg_rc_state *rc_par;           /* [2][nstreams] the parameters each batch in flight was submitted with: what the host's replay of that batch uses */
    /* CHAIN MODE (dsv1_stream_open): ONE stream, the residual coding of a call's frames runs GOP-parallel.  A chain = an I
     * picture and the P pictures that follow it; `chains` = how many are coded side by side (0: the mode is off).  Everything
     * that decides what a chain is -- GOP starts, scene changes, forced-intra P pictures, the stability flags -- depends on
     * source pixels only and is replayed serially on the host first (the always-exact two-pass scheme). */
    int chains;
    int *ch_start, *ch_len, *ch_pair, *ch_cur;   /* per chain of the call: first picture, length, reconstruction slot pair, slot of the pair that holds its newest reconstruction */
    int carry_pair, carry_cur;       /* the pair / slot that holds the reconstruction of the call's last picture (the next call may predict from it) */
};

/* source slot of frame number g (per-stream counter) of stream s */
/* host-side phase timing (DSV1_HOST_PROF=1): where a batch's wall time goes inside submit / collect */
#include <time.h>
#include <stddef.h>
enum { HP_LOAD, HP_DECIDE, HP_ANALYSE, HP_SIDEINFO, HP_ENQUEUE, HP_PREFIX, HP_FETCH, HP_ASSEMBLE, HP_JOIN, HP_N };
static double hp_acc[HP_N];
static long hp_batches;
static int hp_on = -1;
static double hp_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
#define HP_BEGIN() double hp_t0_ = hp_on > 0 ? hp_now() : 0.0
#define HP_MARK(k) do { if (hp_on > 0) { const double n_ = hp_now(); hp_acc[k] += n_ - hp_t0_; hp_t0_ = n_; } } while (0)
static const char *const hp_nm[HP_N] = {"load+pyramid (enqueue, luma sums wait)", "GOP / scene-change decisions", "motion search (enqueue + GPU wait + D2H)",
                                         "intra decisions + stability flags", "job tables + coding enqueue", "packet prefixes (stability / motion bits; CRF batches: the start of the background loop)", "fetch (GPU wait + gather + D2H)", "packet assembly",
                                         "packet prefixes: joining the background loop at collect"};
static void hp_report(void)
{
    int k;
    if (hp_on != 1 || !hp_batches) return;              /* (1: DSV1_HOST_PROF, printed; 2: switched on by dsv1_host_prof_enable, read with dsv1_host_prof_get) */
    for (k = 0; k < HP_N; k++) fprintf(stderr, "[dsv1 host] %-44s %8.3f ms / batch\n", hp_nm[k], hp_acc[k] / (double)hp_batches);
}
/* the same figures as an API (verdict round 5: the bench line carries them): enable (clears the sums), then ms per submitted batch and phase */
void dsv1_host_prof_enable(int on)
{
    int k;
    for (k = 0; k < HP_N; k++) hp_acc[k] = 0.0;
    hp_batches = 0;
    hp_on = on ? 2 : (getenv("DSV1_HOST_PROF") != NULL);
}
int dsv1_host_prof_get(double *ms_per_batch, int n, long *batches)
{
    int k;
    if (batches) *batches = hp_batches;
    for (k = 0; ms_per_batch && k < n && k < HP_N; k++) ms_per_batch[k] = hp_batches ? hp_acc[k] / (double)hp_batches : 0.0;
    return HP_N;
}
const char *dsv1_host_prof_name(int k) { return k >= 0 && k < HP_N ? hp_nm[k] : NULL; }

/* The per-stream host phases (side info, packet assembly) are independent across streams: a
 * parallel loop over them on the worker pool (dsv1_par_for).  Workers: DSV1_HOST_THREADS, else min(12, cores / ranks on the node / 2); never more than streams. */
#include <pthread.h>
#include <unistd.h>
static int slot_of(const dsv1_batch *b, int s, unsigned g) { return (int)(g % (unsigned)b->rows) * b->nstreams + s; }

void *dsv1_batch_ctx(dsv1_batch *b) { return b ? (void *)b->ctx : NULL; }
int dsv1_batch_recon_slot(const dsv1_batch *b, int stream)
{
    if (!b || stream < 0 || stream >= b->nstreams || !b->has_recon[stream]) return -1;
    if (b->chains) return b->carry_cur;
    return stream + b->nstreams * b->rpar[stream];
}

/* the allocations of batch_open_on / dsv1_batch_open go through here so that a test can make the n-th one fail
 * (dsv1_debug_fail_alloc_at: 0 = off) and watch the unwinding */
static int b_alloc_fail_at, b_alloc_count;
void dsv1_debug_fail_alloc_at(int n) { b_alloc_fail_at = n; b_alloc_count = 0; }
static void *b_calloc(size_t n, size_t sz)
{
    if (b_alloc_fail_at && ++b_alloc_count == b_alloc_fail_at) return NULL;
    return calloc(n ? n : 1, sz ? sz : 1);
}

void dsv1_batch_close(dsv1_batch *b)
{
    hp_report();
    if (!b) return;
    if (b->bg_on[0] || b->bg_on[1]) dsv1_par_bg_end();  /* a background prefix loop still reads this batch's pictures */
    if (b->holds_recycler) dsv1_recycle_hold(-1);       /* the last batch out gives the parked packet buffers back */
    if (b->ctx) dsvg_ctx_destroy(b->ctx);
    if (b->own_enc && b->enc) {
        int s;
        for (s = 0; s < b->nstreams; s++) {
            if (b->enc[s].stability) dsv_free(b->enc[s].stability);
            if (b->enc[s].stable_blocks) dsv_free(b->enc[s].stable_blocks);
        }
        free(b->enc);
    }
    free(b->pics); free(b->mvpool); free(b->stabpool); free(b->prefixpool);
    free(b->slots_cur); free(b->slots_ref); free(b->pair_pic); free(b->out_slots);
    free(b->luma); free(b->mv_tmp); free(b->jobs); free(b->outs); free(b->rcjobs); free(b->sc0.pkt); free(b->rpar); free(b->has_recon); free(b->border_skipped); free(b->recon_dropped);
    free(b->ch_start); free(b->ch_len); free(b->ch_pair); free(b->ch_cur);
    free(b->rc_dev); free(b->rc_par);
    free(b);
}

static int batch_open_on(dsv1_batch **out, DSV_ENCODER *encs, int own, int device, int nstreams, int F, int chains)
{
    dsv1_batch *b;
    const DSV_META *m = &encs[0].vidmeta;
    int rc, i, np;
    if (!out || nstreams < 1 || F < 1 || chains < 0 || (chains && nstreams != 1)) return DSVG_ERR_ARG;
    b = (dsv1_batch *)b_calloc(1, sizeof(*b));
    if (!b) return DSVG_ERR_NOMEM;
    b->nstreams = nstreams; b->F = F; b->enc = encs; b->own_enc = own;
    if (chains > F) chains = F;
    b->chains = chains; b->carry_pair = -1; b->carry_cur = -1;
    np = nstreams * F;
    b->rows = 2 * F + 1;
    /* chain mode: `chains` pictures per frame step, a pair of reconstruction slots per chain + one pair for the chain the
     * call before left open */
    rc = dsvg_ctx_create(&b->ctx, device, m->width, m->height, m->subsamp, encs[0].pyramid_levels,
                         b->rows * nstreams, chains ? 2 * (chains + 1) : 2 * nstreams, chains ? chains : nstreams, 2 * np);
    if (rc) { b->enc = NULL; dsv1_batch_close(b); return rc; }       /* the caller still owns encs */
    dsvg_ctx_geom(b->ctx, &b->g);
    b->nblk = b->g.nblocks_h * b->g.nblocks_v;
    b->small_w = (m->width + (1 << b->g.pyramid_levels) - 1) >> b->g.pyramid_levels;
    b->small_h = (m->height + (1 << b->g.pyramid_levels) - 1) >> b->g.pyramid_levels;
    b->prefix_cap = 128 + b->nblk * 24;
    /* every allocation checked (verdict round 4): one failure unwinds the whole batch, context included */
#define B_ALLOC(field, type, n, sz) do { b->field = (type *)b_calloc((size_t)(n), (size_t)(sz)); if (!b->field) goto nomem; } while (0)
    B_ALLOC(pics, pic_t, 2 * (size_t)np, sizeof(pic_t));
    B_ALLOC(mvpool, DSV_MV, 2 * (size_t)np * b->nblk, sizeof(DSV_MV));
    B_ALLOC(stabpool, unsigned char, 2 * (size_t)np * b->nblk, 1);
    B_ALLOC(prefixpool, uint8_t, 2 * (size_t)np, b->prefix_cap);
    B_ALLOC(slots_cur, int, np, sizeof(int));
    B_ALLOC(slots_ref, int, np, sizeof(int));
    B_ALLOC(pair_pic, int, np, sizeof(int));
    B_ALLOC(out_slots, int, np, sizeof(int));
    B_ALLOC(rpar, unsigned char, nstreams, 1);
    B_ALLOC(has_recon, unsigned char, nstreams, 1);
    B_ALLOC(border_skipped, unsigned char, nstreams, 1);
    B_ALLOC(recon_dropped, unsigned char, nstreams, 1);
    B_ALLOC(luma, unsigned, (size_t)b->rows * nstreams, sizeof(unsigned));
    B_ALLOC(mv_tmp, DSV_MV, (size_t)np * b->nblk, sizeof(DSV_MV));
    B_ALLOC(jobs, dsvg_pic_job, np, sizeof(dsvg_pic_job));
    B_ALLOC(outs, dsvg_pic_out, np, sizeof(dsvg_pic_out));
    B_ALLOC(rcjobs, dsvg_rc_job, np, sizeof(dsvg_rc_job));
    B_ALLOC(rc_dev, dsvg_rc_state, nstreams, sizeof(dsvg_rc_state));
    B_ALLOC(rc_par, dsvg_rc_state, 2 * (size_t)nstreams, sizeof(dsvg_rc_state));
    { const char *e = getenv("DSV1_ABR_SERIAL"); b->abr_dev = !chains && !(e && atoi(e) != 0); }
    { const char *e = getenv("DSV1_RECON_ALL"); b->keep_all = e && atoi(e) != 0; }
    if (chains) {
        B_ALLOC(ch_start, int, (size_t)F + 1, sizeof(int));
        B_ALLOC(ch_len, int, (size_t)F + 1, sizeof(int));
        B_ALLOC(ch_pair, int, (size_t)F + 1, sizeof(int));
        B_ALLOC(ch_cur, int, (size_t)F + 1, sizeof(int));
    }
    b->sc0.cap = (size_t)b->prefix_cap + b->g.plane_out_cap[0] + 2 * b->g.plane_out_cap[1] + 256;
    b->sc0.pkt = (uint8_t *)b_calloc(1, b->sc0.cap);
    if (!b->sc0.pkt) goto nomem;
#undef B_ALLOC
    for (i = 0; i < 2 * np; i++) {
        b->pics[i].mvs = b->mvpool + (size_t)i * b->nblk;
        b->pics[i].stable = b->stabpool + (size_t)i * b->nblk;
        b->pics[i].prefix = b->prefixpool + (size_t)i * b->prefix_cap;
    }
    dsv1_recycle_hold(+1);
    b->holds_recycler = 1;
    *out = b;
    return DSVG_OK;
nomem:
    dsv1_log(1, "out of host memory while opening a batch of %d streams x %d frames", nstreams, F);
    b->enc = NULL;                                      /* the caller still owns encs */
    dsv1_batch_close(b);
    return DSVG_ERR_NOMEM;
}

int dsv1_batch_open(dsv1_batch **out, const DSV_ENCODER *cfg, int device, int nstreams, int frames_per_call)
{
    DSV_ENCODER *encs;
    int s, rc;
    if (!cfg || nstreams < 1) return DSVG_ERR_ARG;
    encs = (DSV_ENCODER *)b_calloc((size_t)nstreams, sizeof(DSV_ENCODER));
    if (!encs) return DSVG_ERR_NOMEM;
    for (s = 0; s < nstreams; s++) {
        encs[s] = *cfg;
        encs[s].ref = NULL; encs[s].stability = NULL; encs[s].stable_blocks = NULL;
        dsv_enc_start(&encs[s]);
    }
    rc = batch_open_on(out, encs, 1, device, nstreams, frames_per_call, 0);
    if (rc) {
        for (s = 0; s < nstreams; s++) {                /* what dsv_enc_start gave every stream */
            if (encs[s].stability) dsv_free(encs[s].stability);
            if (encs[s].stable_blocks) dsv_free(encs[s].stable_blocks);
        }
        free(encs);
    }
    return rc;
}

/* ONE stream, frames_per_call consecutive frames per call, residual coding GOP-parallel (chain mode, see struct dsv1_batch):
 * the same bytes as the frame-serial encoder for ANY CRF configuration -- scene changes, forced-intra P pictures, a
 * stable_refresh that does not line up with the GOP length (GOP 30), metadata forced in mid-stream.  max_chains = chains coded
 * side by side (more chains in a call simply take further rounds).  ABR streams are refused (serial by definition). */
int dsv1_stream_open(dsv1_batch **out, const DSV_ENCODER *cfg, int device, int frames_per_call, int max_chains)
{
    DSV_ENCODER *enc;
    int rc;
    if (!cfg || frames_per_call < 1 || max_chains < 1 || cfg->rc_mode != DSV_RATE_CONTROL_CRF) return DSVG_ERR_ARG;
    enc = (DSV_ENCODER *)calloc(1, sizeof(DSV_ENCODER));
    if (!enc) return DSVG_ERR_ARG;
    *enc = *cfg;
    enc->ref = NULL; enc->stability = NULL; enc->stable_blocks = NULL;
    dsv_enc_start(enc);
    rc = batch_open_on(out, enc, 1, device, 1, frames_per_call, max_chains);
    if (rc) {
        if (enc->stability) dsv_free(enc->stability);
        if (enc->stable_blocks) dsv_free(enc->stable_blocks);
        free(enc);
    }
    return rc;
}

DSV_ENCODER *dsv1_batch_encoder(dsv1_batch *b, int stream) { return b && stream >= 0 && stream < b->nstreams ? &b->enc[stream] : NULL; }
void dsv1_batch_set_fnum(dsv1_batch *b, int stream, DSV_FNUM next_fnum)
{
    if (b && stream >= 0 && stream < b->nstreams) b->enc[stream].next_fnum = next_fnum;
}

/* ---- rate control: quality2quant dsv_encoder.c:70-168 and the statistics of dsv_enc :816-848 live in include/dsvg_rc.h -- one
 * piece of code for this layer and for the device (k_rc); here they run on the encoder struct's own fields ------------------- */
static void rc_load(dsvg_rc_state *r, const DSV_ENCODER *e)
{
    r->rc_quant = e->rc_quant; r->bpf_total = e->bpf_total; r->bpf_reset = e->bpf_reset;
    r->bpf_avg = e->bpf_avg; r->total_P_frame_q = e->total_P_frame_q; r->avg_P_frame_q = e->avg_P_frame_q;
    r->last_P_frame_over = e->last_P_frame_over; r->back_into_range = e->back_into_range;
    r->bitrate = e->bitrate; r->fps_num = e->vidmeta.fps_num; r->fps_den = e->vidmeta.fps_den;
    r->rc_high_motion_nudge = e->rc_high_motion_nudge; r->max_q_step = e->max_q_step;
    r->min_quality = e->min_quality; r->max_quality = e->max_quality; r->min_I_frame_quality = e->min_I_frame_quality;
}
/* par: the parameter snapshot r was loaded with (NULL: the encoder's own fields).  quality2quant clamps max_q_step IN PLACE
 * (dsv_encoder.c:128); with a snapshot the clamp goes back only while the public field still holds the value the snapshot was taken
 * from -- a replay that runs after the caller has set a new max_q_step must not put the old one back */
static void rc_store(DSV_ENCODER *e, const dsvg_rc_state *r, const dsvg_rc_state *par)
{
    e->rc_quant = r->rc_quant; e->bpf_total = r->bpf_total; e->bpf_reset = r->bpf_reset;
    e->bpf_avg = r->bpf_avg; e->total_P_frame_q = r->total_P_frame_q; e->avg_P_frame_q = r->avg_P_frame_q;
    e->last_P_frame_over = r->last_P_frame_over; e->back_into_range = r->back_into_range;
    if (!par || e->max_q_step == par->max_q_step) e->max_q_step = r->max_q_step;
}
/* the PARAMETER half of the state (what the caller may change between frames): everything from `bitrate` on */
#define RC_PAR_OFF offsetof(dsvg_rc_state, bitrate)
static void rc_par_set(dsvg_rc_state *r, const dsvg_rc_state *par)
{
    if (par) memcpy((char *)r + RC_PAR_OFF, (const char *)par + RC_PAR_OFF, sizeof(*r) - RC_PAR_OFF);
}
static void rc_par_to_enc(DSV_ENCODER *e, const dsvg_rc_state *r)      /* the parameter half back into the public fields (not the frame rate: vidmeta) */
{
    e->bitrate = r->bitrate; e->rc_high_motion_nudge = r->rc_high_motion_nudge; e->max_q_step = r->max_q_step;
    e->min_quality = r->min_quality; e->max_quality = r->max_quality; e->min_I_frame_quality = r->min_I_frame_quality;
}
static int rc_par_differs(const dsvg_rc_state *a, const dsvg_rc_state *b)
{
    return memcmp((const char *)a + RC_PAR_OFF, (const char *)b + RC_PAR_OFF, sizeof(*a) - RC_PAR_OFF) != 0;
}
/* par: the parameters the picture's batch was submitted with (device-resident rate control: the replay must use what the
 * device used, whatever the caller has done to the public fields since), or NULL: the encoder's fields as they are now */
static int pick_quant_par(DSV_ENCODER *e, const dsvg_rc_state *par, int isP, int forced_intra)
{
    if (e->rc_mode != DSV_RATE_CONTROL_CRF) {
        dsvg_rc_state r;
        int fq;
        rc_load(&r, e);
        rc_par_set(&r, par);
        fq = dsvg_rc_pick(&r, isP, forced_intra);
        rc_store(e, &r, par);
        return fq;
    }
    e->rc_quant = (unsigned)e->quality;
    return DSV_MAX_QUALITY - ((DSV_MAX_QUALITY - 5) * e->quality / DSV_MAX_QUALITY);
}
static int pick_quant(DSV_ENCODER *e, int isP, int forced_intra) { return pick_quant_par(e, NULL, isP, forced_intra); }

static void rc_after_packet(DSV_ENCODER *e, const dsvg_rc_state *par, int isP, unsigned pkt_len)        /* dsv_encoder.c:816-848 */
{
    dsvg_rc_state r;
    if (e->rc_mode == DSV_RATE_CONTROL_CRF) return;
    rc_load(&r, e);
    rc_par_set(&r, par);
    dsvg_rc_after(&r, isP, pkt_len);
    rc_store(e, &r, par);
}

/* ---- side information --------------------------------------------------------------------------- */
static void write_pkt_hdr(bitw *w, int type)
{
    bw_bits(w, 8, 'D'); bw_bits(w, 8, 'S'); bw_bits(w, 8, 'V'); bw_bits(w, 8, '1');
    bw_bits(w, 8, 0);
    bw_bits(w, 8, (unsigned)type);
    bw_bits(w, 32, 0);
    bw_bits(w, 32, 0);
}

/* stability flags of one picture: updates the per-stream accumulators in coding order
 * (encode_stable_blocks dsv_encoder.c:330-408); the flags go to pc->stable (the coding kernels need them) */
static void stability_update(DSV_ENCODER *e, pic_t *pc, int nblk)
{
    int i, div;
    if (e->refresh_ctr >= e->stable_refresh) {
        e->refresh_ctr = 0;
        memset(e->stability, 0, sizeof(*e->stability) * (size_t)nblk);
    }
    div = (int)e->refresh_ctr;
    if (div <= 0) div = 1;
    for (i = 0; i < nblk; i++) {
        int stable = 0, intra = 0;
        /* (acc / div == 0 with C's truncating division and div >= 1 is -div < acc < div -- the accumulators are signed 16-bit
         * fields and may wrap: no division per block) */
#define NEAR0(a) ((a) < div && (a) > -div)
        if (pc->isP) {
            const DSV_MV *mv = &pc->mvs[i];
            if (mv->mode == 0) {
                e->stability[i].x += abs(mv->u.mv.x) >> 2;
                e->stability[i].y += abs(mv->u.mv.y) >> 2;
                stable = mv->high_detail;
                stable |= (NEAR0(e->stability[i].x) && NEAR0(e->stability[i].y) && !mv->lo_tex && !mv->lo_var);
            } else {
                intra = 1;
            }
            if (mv->lo_tex || mv->lo_var) {
                e->stability[i].x = 0x3fff;
                e->stability[i].y = 0x3fff;
            }
        } else {
            stable = (NEAR0(e->stability[i].x) && NEAR0(e->stability[i].y));
        }
#undef NEAR0
        e->stable_blocks[i] = (unsigned char)(stable | (intra << 1));
    }
    memcpy(pc->stable, e->stable_blocks, (size_t)nblk);
}
/* ... and their ZBRLE block in the packet prefix (from pc->stable: may run after the coding work was enqueued) */
static void stability_write(const pic_t *pc, int nblk, bitw *w, uint8_t *tmp)
{
    zrle z;
    int i, bytes;
    memset(tmp, 0, (size_t)nblk * 4 + 16);
    zr_init(&z, tmp);
    for (i = 0; i < nblk; i++) zr_put(&z, pc->stable[i] & 1);
    bw_align(w);
    bytes = zr_end(&z);
    bw_ueg(w, (unsigned)bytes);
    bw_align(w);
    bw_bytes_in(w, tmp, (unsigned)bytes);
}

/* block modes (ZBRLE), MV residuals against the neighbour predictor (SEG), intra sub-block masks
 * (encode_motion dsv_encoder.c:257-327) */
static void motion_pass(const dsv1_batch *b, pic_t *pc, bitw *w, uint8_t *tmp)
{
    const int nbh = b->g.nblocks_h, nbv = b->g.nblocks_v;
    const size_t cap = (size_t)b->nblk * 8 + 64;
    bitw sub[4];
    zrle modes;
    DSV_PARAMS prm;
    int i, j, k;
    memset(tmp, 0, cap * 4);
    for (k = 0; k < 4; k++) bw_init(&sub[k], tmp + cap * k);
    zr_init(&modes, tmp);
    memset(&prm, 0, sizeof(prm));
    prm.nblocks_h = nbh; prm.nblocks_v = nbv;
    for (j = 0; j < nbv; j++)
        for (i = 0; i < nbh; i++) {
            DSV_MV *mv = &pc->mvs[i + j * nbh];
            zr_put(&modes, mv->mode);
            if (mv->mode == 0) {
                int px, py;
                dsv_movec_pred(pc->mvs, &prm, i, j, &px, &py);
                bw_seg(&sub[1], mv->u.mv.x - px);
                bw_seg(&sub[2], mv->u.mv.y - py);
            } else if (mv->submask == 0xF) {
                bw_bit(&sub[3], 1);
            } else {
                bw_bit(&sub[3], 0);
                bw_bits(&sub[3], 4, mv->submask);
            }
        }
    for (k = 0; k < 4; k++) {
        int bytes;
        bw_align(w);
        if (k == 0) bytes = zr_end(&modes);
        else { bw_align(&sub[k]); bytes = (int)bw_bytes(&sub[k]); }
        bw_ueg(w, (unsigned)bytes);
        bw_align(w);
        bw_bytes_in(w, tmp + cap * k, (unsigned)bytes);
    }
}

static unsigned write_meta_packet(const DSV_ENCODER *e, uint8_t *buf)       /* encode_metadata :427-461 */
{
    bitw w;
    const DSV_META *m = &e->vidmeta;
    unsigned n;
    memset(buf, 0, 64);
    bw_init(&w, buf);
    write_pkt_hdr(&w, DSV_PT_META);
    bw_ueg(&w, (unsigned)m->width); bw_ueg(&w, (unsigned)m->height); bw_ueg(&w, (unsigned)m->subsamp);
    bw_ueg(&w, (unsigned)m->fps_num); bw_ueg(&w, (unsigned)m->fps_den);
    bw_ueg(&w, (unsigned)m->aspect_num); bw_ueg(&w, (unsigned)m->aspect_den);
    bw_align(&w);
    n = bw_bytes(&w);
    put_be32(buf + DSV_PACKET_NEXT_OFFSET, n);
    return n;
}

static void link_packet(DSV_ENCODER *e, uint8_t *pkt, unsigned len, int eos)   /* set_link_offsets :171-192 */
{
    const unsigned next = eos ? 0 : len;
    put_be32(pkt + DSV_PACKET_PREV_OFFSET, (unsigned)e->prev_link);
    put_be32(pkt + DSV_PACKET_NEXT_OFFSET, next);
    e->prev_link = (int)next;
}

/* picture packet = prefix + quantiser + three framed planes (encode_picture :518-536,
 * dsv_encode_plane hzcc.c:449-476); then metadata-first emission and RC statistics (dsv_enc :804-853) */
static int assemble(dsv1_batch *b, int s, pic_t *pc, const dsvg_pic_out *po, DSV_BUF *out, pkt_scratch *sc, const dsvg_rc_state *par)
{
    DSV_ENCODER *e = &b->enc[s];
    bitw w;
    unsigned len;
    int p;
    uint8_t *pkt;
    size_t need = (size_t)pc->prefix_len + 64;
    (void)sc;
    for (p = 0; p < 3; p++) need += po->nbytes[p] + 48;
    if (pc->gop_start) {
        uint8_t mb[64];
        const unsigned n = write_meta_packet(e, mb);
        if (dsv1_buf_append(out, mb, n)) return DSVG_ERR_ARG;
    }
    if (b->abr_dev && e->rc_mode != DSV_RATE_CONTROL_CRF) {
        /* the device chose this picture's quantiser (k_rc): replay the choice here -- it also advances this stream's host-side
         * state -- and refuse the batch if the two disagree (same code on both sides: include/dsvg_rc.h) */
        pc->quant = pick_quant_par(e, par, pc->isP, pc->forced_intra);
        if (pc->quant != po->rc_quant) {
            dsv1_log(1, "rate control: stream %d picture %u: the device coded with quantiser %d, the host replay says %d", s, (unsigned)pc->fnum, (int)po->rc_quant, pc->quant);
            return DSVG_ERR_RC;
        }
    }
    /* the packet is built in place at the end of the stream buffer: the payloads (the bulk) are copied once, straight
     * from the fetch buffer; only the few header bytes the bit writer ORs into are cleared first */
    if (dsv1_buf_reserve(out, (unsigned)need)) return DSVG_ERR_ARG;
    pkt = out->data + out->len;
    memcpy(pkt, pc->prefix, pc->prefix_len);
    memset(pkt + pc->prefix_len, 0, 8);
    bw_init(&w, pkt);
    w.pos = pc->prefix_len * 8;
    bw_bits(&w, 11, (unsigned)pc->quant);
    for (p = 0; p < 3; p++) {
        unsigned startp, endp;
        bw_align(&w);
        startp = bw_bytes(&w);
        memset(pkt + startp, 0, 24);
        bw_bits(&w, 32, 0);
        bw_seg(&w, po->dc[p]);
        bw_align(&w);
        bw_bits(&w, 32, po->nruns[p]);
        bw_align(&w);
        bw_bytes_in(&w, po->payload[p], po->nbytes[p]);
        memset(pkt + bw_bytes(&w), 0, 8);
        bw_bits(&w, 8, 0x55);
        bw_align(&w);
        endp = bw_bytes(&w);
        put_be32(pkt + startp, endp - startp - 4);
    }
    bw_align(&w);
    len = bw_bytes(&w);
    if (b->abr_dev && e->rc_mode != DSV_RATE_CONTROL_CRF && len != po->rc_pkt_len) {
        dsv1_log(1, "rate control: stream %d picture %u: packet of %u bytes, the device counted %u", s, (unsigned)pc->fnum, len, (unsigned)po->rc_pkt_len);
        return DSVG_ERR_RC;
    }
    rc_after_packet(e, par, pc->isP, len);
    link_packet(e, pkt, len, 0);
    out->len += len;
    return DSVG_OK;
}

/* what the coding work needs of the side information, per stream in coding order: intra decisions, the stability flags
 * (accumulators), the vectors' reach.  The bits of the packet prefix are written by prefix_stream AFTER the coding work
 * has been enqueued: the GPU has nothing else to do while this runs. */
static void side_stream(void *ctx, int s, int tid)
{
    side_ctx *c = (side_ctx *)ctx;
    dsv1_batch *b = c->b;
    const int F = b->F, nblk = b->nblk;
    DSV_ENCODER *e = &b->enc[s];
    int t;
    (void)tid;
    for (t = 0; t < b->nf_cur; t++) {
        pic_t *pc = &c->pics[s * F + t];
        if (pc->has_ref) {
            int nintra = 0, i;
            int x0 = 0, x1 = 0, y0 = 0, y1 = 0;           /* full-pel reach of the inter blocks' vectors */
            for (i = 0; i < nblk; i++) {
                const DSV_MV *m = &pc->mvs[i];
                if (m->mode != 0) { nintra++; continue; }
                {
                    const int dx = m->u.mv.x >> 1, dy = m->u.mv.y >> 1;
                    if (dx < x0) x0 = dx;
                    if (dx > x1) x1 = dx;
                    if (dy < y0) y0 = dy;
                    if (dy > y1) y1 = dy;
                }
            }
            pc->n_intra = nintra;
            pc->reach[0] = (short)x0; pc->reach[1] = (short)x1; pc->reach[2] = (short)y0; pc->reach[3] = (short)y1;
            pc->forced_intra = 0;
            if (nintra * 100 / nblk > e->intra_pct_thresh) { pc->has_ref = 0; pc->forced_intra = 1; }
        }
        pc->isP = pc->has_ref;
        stability_update(e, pc, nblk);
        if (pc->isP) e->refresh_ctr++;               /* dsv_enc dsv_encoder.c:812-814 */
    }
}
static void prefix_one(const dsv1_batch *b, pic_t *pc, uint8_t *tmp);
static void prefix_stream(void *ctx, int s, int tid)
{
    side_ctx *c = (side_ctx *)ctx;
    dsv1_batch *b = c->b;
    const int F = b->F, nblk = b->nblk;
    uint8_t *tmp = (uint8_t *)malloc(((size_t)nblk * 8 + 64) * 4 + 64);
    int t;
    (void)tid;
    if (!tmp) { c->rc = DSVG_ERR_ARG; return; }         /* (a stale prefix would go out as a corrupt packet: the submit fails instead) */
    for (t = 0; t < c->nf; t++) prefix_one(b, &c->pics[s * F + t], tmp);
    free(tmp);
}

/* the packet prefix of ONE picture (chain mode: a single stream, its pictures are the parallel items) */
static void prefix_one(const dsv1_batch *b, pic_t *pc, uint8_t *tmp)
{
    bitw w;
    memset(pc->prefix, 0, (size_t)b->prefix_cap);
    bw_init(&w, pc->prefix);
    write_pkt_hdr(&w, DSV_PT_PIC | (pc->is_ref << 1) | pc->has_ref);
    bw_align(&w);
    bw_bits(&w, 32, pc->fnum);
    bw_align(&w);
    bw_ueg(&w, (unsigned)b->g.blk_w >> 2);
    bw_ueg(&w, (unsigned)b->g.blk_h >> 2);
    bw_align(&w);
    stability_write(pc, b->nblk, &w, tmp);
    if (pc->has_ref) {
        bw_align(&w);
        motion_pass(b, pc, &w, tmp);
    }
    bw_align(&w);
    pc->prefix_len = bw_bytes(&w);
}
static void prefix_picture(void *ctx, int t, int tid)
{
    side_ctx *c = (side_ctx *)ctx;
    uint8_t *tmp = (uint8_t *)malloc(((size_t)c->b->nblk * 8 + 64) * 4 + 64);
    (void)tid;
    if (!tmp) { c->rc = DSVG_ERR_ARG; return; }
    prefix_one(c->b, &c->pics[t], tmp);
    free(tmp);
}

/* Chain mode, step 5 of a submit: the call's pictures -- decisions, motion fields and stability flags all known -- fall into
 * chains (a picture without a reference starts one; the call's first picture continues the chain the call before left open
 * when it is a P picture).  Chains do not depend on each other, so frame step k codes the k-th picture of every chain of a
 * round (at most `chains` side by side); consecutive steps with the same number of live chains go to the device as one
 * dsvg_code_batch call.  Calls are ordered against each other on the coding streams, so a slot pair may be reused by a later
 * round, and a chain may change its position from call to call. */
static int code_chains(dsv1_batch *b, pic_t *pics, int nf, int par)
{
    const int C = b->chains;
    DSV_ENCODER *e = &b->enc[0];
    int nch = 0, t, g0, rc, os = par * b->F;
    for (t = 0; t < nf; t++) {
        if (t == 0 || !pics[t].isP) { b->ch_start[nch] = t; b->ch_len[nch] = 0; nch++; }
        b->ch_len[nch - 1]++;
        pics[t].quant = pick_quant(e, pics[t].isP, pics[t].forced_intra);
    }
    if (pics[0].isP && b->carry_cur < 0) { dsv1_log(1, "a P picture without a reference picture on the device"); return DSVG_ERR_ARG; }
    for (g0 = 0; g0 < nch; g0 += C) {
        const int gn = nch - g0 < C ? nch - g0 : C;
        int c, L = 0, k = 0, nextpair = 0;
        for (c = g0; c < g0 + gn; c++) {
            if (b->ch_len[c] > L) L = b->ch_len[c];
            if (c == 0 && pics[0].isP) { b->ch_pair[c] = b->carry_pair; b->ch_cur[c] = b->carry_cur; continue; }
            /* C + 1 pairs: the one the open chain of the call before lives in is left alone while that chain goes on (round 0) */
            if (g0 == 0 && pics[0].isP && nextpair == b->carry_pair) nextpair++;
            b->ch_pair[c] = nextpair++;
            b->ch_cur[c] = -1;
        }
        while (k < L) {
            int nj = 0, k2, kk, idx = 0;
            for (c = g0; c < g0 + gn; c++) nj += b->ch_len[c] > k;
            for (k2 = k + 1; k2 < L; k2++) {
                int n2 = 0;
                for (c = g0; c < g0 + gn; c++) n2 += b->ch_len[c] > k2;
                if (n2 != nj) break;
            }
            for (kk = k; kk < k2; kk++)
                for (c = g0; c < g0 + gn; c++) {
                    pic_t *pc;
                    dsvg_pic_job *j;
                    int tt, last_of_chain;
                    if (b->ch_len[c] <= kk) continue;
                    tt = b->ch_start[c] + kk;
                    pc = &pics[tt];
                    j = &b->jobs[idx++];
                    memset(j, 0, sizeof(*j));
                    j->src_slot = pc->cur_slot;
                    j->ref_recon_slot = pc->isP ? b->ch_cur[c] : -1;
                    /* (a chain's last picture whose successor starts the next chain of this submit has no reader: no reconstruction,
                     * struct dsv1_batch, recon_dropped) */
                    if (pc->is_ref && kk + 1 == b->ch_len[c] && tt + 1 < nf && !b->keep_all) {
                        j->recon_slot = -1;
                        b->n_dropped++;
                    } else
                    if (pc->is_ref) {
                        /* the pair's other slot: the prediction is written straight into it (dsvg_code_batch) */
                        b->ch_cur[c] = 2 * b->ch_pair[c] + (b->ch_cur[c] == 2 * b->ch_pair[c] ? 1 : 0);
                        j->recon_slot = b->ch_cur[c];
                    } else j->recon_slot = -1;
                    j->quant = pc->quant;
                    j->mvs = (const dsvg_mv *)pc->mvs;
                    j->stable_blocks = pc->stable;
                    pc->out_slot = os++;
                    j->out_slot = pc->out_slot;
                    j->no_intra_blocks = pc->isP && pc->n_intra == 0;
                    j->has_reach = pc->isP;
                    memcpy(j->mv_reach, pc->reach, sizeof pc->reach);
                    /* who predicts from this reconstruction?  The chain's next picture -- in this call (the device then writes
                     * the border as far as that picture's vectors reach), or in a later one (0: the whole border).  The chain's
                     * last picture: nobody, when the stream's next picture is known to be an I picture (it starts the next chain
                     * of this submit, or it starts a GOP: dsv_encoder.c:702-708); else unknown */
                    last_of_chain = kk + 1 == b->ch_len[c];
                    if (!last_of_chain) j->border_hint = kk + 1 < k2;
                    else if (tt + 1 < nf) j->border_hint = 1;
                    else j->border_hint = e->force_metadata || (DSV_FNUM)(e->prev_gop + (DSV_FNUM)e->gop) <= e->next_fnum;
                    if (tt + 1 == nf) b->border_skipped[0] = (unsigned char)(j->border_hint && j->recon_slot >= 0);
                }
            if ((rc = dsvg_code_batch(b->ctx, k2 - k, nj, b->jobs))) return rc;
            k = k2;
        }
    }
    if (pics[nf - 1].is_ref) {
        b->carry_pair = b->ch_pair[nch - 1]; b->carry_cur = b->ch_cur[nch - 1];
        b->has_recon[0] = 1;
    }
    return DSVG_OK;
}

/* packet assembly of one stream of a collected batch (its own staging buffer per thread) */
typedef struct { dsv1_batch *b; pic_t *pics; DSV_BUF *out; int rc, nf, s0; const dsvg_rc_state *par; } asm_ctx;
static void asm_stream(void *ctx, int s, int tid)
{
    asm_ctx *c = (asm_ctx *)ctx;
    dsv1_batch *b = c->b;
    s += c->s0;                                  /* (a piece of the batch: streams s0 ..) */
    const int F = b->F;
    pkt_scratch sc = {NULL, 0};
    size_t need = 0;
    int t, p;
    (void)tid;
    for (t = 0; t < c->nf; t++) {
        const dsvg_pic_out *po = &b->outs[s * F + t];
        need += (size_t)c->pics[s * F + t].prefix_len + 192;
        for (p = 0; p < 3; p++) need += po->nbytes[p] + 32;
    }
    if (dsv1_buf_reserve(&c->out[s], (unsigned)need)) { c->rc = DSVG_ERR_ARG; return; }
    for (t = 0; t < c->nf; t++) {
        const int rc = assemble(b, s, &c->pics[s * F + t], &b->outs[s * F + t], &c->out[s], &sc, c->par ? &c->par[s] : NULL);
        if (rc) { c->rc = rc; break; }
    }
    free(sc.pkt);
}
/* dsvg_fetch_pictures_cb: pictures [first, first + count) have arrived (whole streams: the pieces end on multiples of nf) */
static void asm_piece(void *ctx, int first, int count)
{
    asm_ctx *c = (asm_ctx *)ctx;
    c->s0 = first / c->nf;
    dsv1_par_for(count / c->nf, asm_stream, c);
}

/* Submit one batch: all source-only analysis now (analysis stream), per-stream decisions and side info
 * on the host, then the residual chain of every frame step enqueued on the coding stream.  CRF returns
 * without waiting for the coding work; dsv1_batch_collect() fetches and assembles the packets.  ABR
 * (each quantiser needs the previous packet size) runs frame step by frame step and leaves nothing
 * pending except the already assembled packets. */
static int stage_n(dsv1_batch *b, const void *yuv_host, int nf);
static int code_chains(dsv1_batch *b, pic_t *pics, int nf, int par);
static void prefix_picture(void *ctx, int t, int tid);
/* the streams listed in b->out_slots[0..n) had the reconstruction of their last picture dropped and need it after all (struct dsv1_batch,
 * recon_dropped): code those pictures again, each into the free slot of its stream's pair.  Everything a picture was coded from is still
 * there -- its source frame (the ring keeps a call's last frames for the next call's motion search, copied whole), the reconstruction it
 * predicted from (its stream's current slot: nothing has been kept since), its vectors, flags and quantiser (the other half of pics[]).
 * The packets go to the first out slots of the half this submit is about to use and are never fetched.  Rare (a renumbered stream): the
 * device is drained before and after instead of ordering the extra call against the streams of its neighbours. */
static int remedy_dropped(dsv1_batch *b, int n, int par)
{
    const int S = b->nstreams, F = b->F;
    const pic_t *prev = b->pics + (size_t)(par ^ 1) * S * F;
    int i, rc;
    if (b->nf_prev < 1) return DSVG_ERR_ARG;
    if ((rc = dsvg_ctx_sync(b->ctx))) return rc;
    for (i = 0; i < n; i++) {
        const int s = b->out_slots[i];
        const pic_t *pp = &prev[s * F + b->nf_prev - 1];
        dsvg_pic_job *j = &b->jobs[i];
        memset(j, 0, sizeof(*j));
        j->src_slot = pp->cur_slot;
        j->ref_recon_slot = pp->isP ? s + S * b->rpar[s] : -1;
        b->rpar[s] ^= 1;
        b->has_recon[s] = 1;
        j->recon_slot = s + S * b->rpar[s];
        j->quant = pp->quant;
        j->mvs = (const dsvg_mv *)pp->mvs;
        j->stable_blocks = pp->stable;
        j->out_slot = par * S * F + i;
        j->no_intra_blocks = pp->isP && pp->n_intra == 0;
        j->has_reach = 0;
        j->border_hint = 0;                          /* the whole border */
        b->n_remedied++;
    }
    /* (I and P pictures may be mixed here: one frame step of n jobs on one coding stream) */
    if ((rc = dsvg_code_batch(b->ctx, 1, n, b->jobs))) return rc;
    return dsvg_ctx_sync(b->ctx);
}
int dsv1_batch_recon_all(dsv1_batch *b, int on)
{
    if (!b) return DSVG_ERR_ARG;
    if (b->pending[0] || b->pending[1]) { dsv1_log(1, "dsv1_batch_recon_all with batches in flight"); return DSVG_ERR_ARG; }
    b->keep_all = on != 0;
    return DSVG_OK;
}
long dsv1_batch_dropped_recons(const dsv1_batch *b, long *remedied)
{
    if (!b) return 0;
    if (remedied) *remedied = b->n_remedied;
    return b->n_dropped;
}

static int batch_submit_impl(dsv1_batch *b, const void *yuv, int yuv_on_device, DSV_BUF *abr_out, int nf, int inplace_ok)
{
    int S, F, nblk, with_pyr, s, t, k, rc, npairs = 0, par, nremedy = 0;
    size_t fb;
    const DSV_ENCODER *e0;
    const uint8_t *dyuv = (const uint8_t *)yuv;
    pic_t *pics;

    if (!b || !yuv) return DSVG_ERR_ARG;
    if (hp_on < 0) hp_on = getenv("DSV1_HOST_PROF") != NULL;
    HP_BEGIN();
    S = b->nstreams; F = b->F; nblk = b->nblk; fb = b->g.frame_bytes;
    if (nf <= 0) nf = F;
    if (nf > F || (nf < F && S != 1)) { dsv1_log(1, "a short batch needs a single stream"); return DSVG_ERR_ARG; }
    b->nf_cur = nf;
    e0 = &b->enc[0];
    with_pyr = e0->gop != DSV_GOP_INTRA;
    par = b->parity;
    if (b->pending[par]) { dsv1_log(1, "batch submitted twice without collect"); return DSVG_ERR_ARG; }
    pics = b->pics + (size_t)par * S * F;
    if (!yuv_on_device) {
        /* host frames go through the context's double-buffered ingest (copy stream of its own): nothing here waits
         * for the device, and a clip announced with dsv1_batch_stage() is already on its way */
        if (!b->nstaged) {
            if ((rc = stage_n(b, yuv, nf))) return rc;
        } else if (b->staged_host[0] != yuv) {
            dsv1_log(1, "a different clip was staged for this submit");
            return DSVG_ERR_ARG;
        }
        dyuv = (const uint8_t *)b->staged_dev[0];
        b->staged_host[0] = b->staged_host[1]; b->staged_dev[0] = b->staged_dev[1];
        b->nstaged--;
    }
    /* 1. source-only preparation for every frame: bordered layout, pyramid, mean luma */
    for (s = 0; s < S; s++)
        for (t = 0; t < nf; t++) b->slots_cur[s * F + t] = slot_of(b, s, b->gcount + (unsigned)t);
    {
        /* A clip in the CALLER's device memory stays where it is as far as chroma goes (dsvg_load_frames_map_ex): the forward
         * transform and the motion search read it there, only luma is copied (bordered + pyramid).  Each stream's last frame
         * of the call is copied whole: the first frame of the next call may search against it when this clip is gone. */
        unsigned char *inpl = NULL;
        if (inplace_ok && yuv_on_device && (inpl = (unsigned char *)malloc((size_t)S * nf))) {
            for (s = 0; s < S; s++)
                for (t = 0; t < nf; t++) inpl[s * nf + t] = (unsigned char)(t + 1 < nf);
        }
        rc = dsvg_load_frames_map_ex(b->ctx, S * nf, b->slots_cur, dyuv, fb, with_pyr, inpl);     /* nf < F only with S == 1 */
        free(inpl);
        if (rc) return rc;
    }
    if (with_pyr && e0->do_scd)
        if ((rc = dsvg_get_luma_sums(b->ctx, 0, b->rows * S, b->luma))) return rc;

    HP_MARK(HP_LOAD);
    /* 2. per stream, in coding order: GOP / scene-change decisions; collect ME pairs */
    for (s = 0; s < S; s++) {
        DSV_ENCODER *e = &b->enc[s];
        if (!e->stability) {
            e->stability = (struct DSV_STAB_ACC *)dsv_alloc((int)(sizeof(*e->stability) * nblk));
            e->stable_blocks = (unsigned char *)dsv_alloc(nblk);
        }
        if (e->pyramid_levels == 0) e->pyramid_levels = b->g.pyramid_levels;
        for (t = 0; t < nf; t++) {
            pic_t *pc = &pics[s * F + t];
            pc->fnum = e->next_fnum++;
            pc->cur_slot = slot_of(b, s, b->gcount + (unsigned)t);
            pc->ref_slot = slot_of(b, s, b->gcount + (unsigned)t + (unsigned)b->rows - 1u);
            pc->out_slot = par * S * F + t * S + s;
            pc->gop_start = 0; pc->forced_intra = 0;
            if (e->force_metadata || (DSV_FNUM)(e->prev_gop + (DSV_FNUM)e->gop) <= pc->fnum) {
                pc->gop_start = 1;
                e->prev_gop = pc->fnum;
                e->force_metadata = 0;
            }
            if (e->gop == DSV_GOP_INTRA) {
                pc->is_ref = 0; pc->has_ref = 0;
            } else {
                pc->is_ref = 1;
                pc->has_ref = !pc->gop_start;
                if (e->do_scd) {
                    const int al = (int)b->luma[pc->cur_slot] / (b->small_w * b->small_h);
                    if (abs(e->prev_avg_luma - al) > e->scene_change_delta) { pc->has_ref = 0; pc->forced_intra = 1; }
                    e->prev_avg_luma = al;
                }
            }
            if (pc->has_ref && t == 0 && b->border_skipped[s]) {
                /* the batch before promised a GOP start here (border_hint) and the caller changed the frame numbering
                 * in between (dsv1_batch_set_fnum): give the reference its whole border after all */
                if ((rc = dsvg_extend_recon(b->ctx, b->chains ? b->carry_cur : s + S * b->rpar[s]))) return rc;
            }
            if (t == 0) b->border_skipped[s] = 0;
            if (t == 0 && b->recon_dropped[s]) {
                /* ... and a reconstruction it dropped: coded once more below, kept this time */
                if (pc->has_ref) b->out_slots[nremedy++] = s;
                b->recon_dropped[s] = 0;
            }
            if (pc->has_ref) {
                b->slots_cur[npairs] = pc->cur_slot;
                b->slots_ref[npairs] = pc->ref_slot;
                b->pair_pic[npairs] = s * F + t;
                npairs++;
            }
        }
    }
    if (nremedy && (rc = remedy_dropped(b, nremedy, par))) return rc;
    HP_MARK(HP_DECIDE);
    /* 3. motion estimation for every inter candidate of the batch in one go */
    if (npairs) {
        if ((rc = dsvg_analyse(b->ctx, npairs, b->slots_cur, b->slots_ref, (dsvg_mv *)b->mv_tmp))) return rc;
        for (k = 0; k < npairs; k++)
            memcpy(pics[b->pair_pic[k]].mvs, b->mv_tmp + (size_t)k * nblk, (size_t)nblk * sizeof(DSV_MV));
    }
    HP_MARK(HP_ANALYSE);
    /* 4. per stream, in coding order: forced intra, stability + motion side info -> packet prefix */
    {
        side_ctx sc_;
        sc_.b = b; sc_.pics = pics;
        dsv1_par_for(S, side_stream, &sc_);
    }
    HP_MARK(HP_SIDEINFO);
    /* 5. residual coding, frame step by frame step across all streams */
    {
        const int abr = e0->rc_mode != DSV_RATE_CONTROL_CRF;
        const int devrc = abr && b->abr_dev;                    /* rate control on the device: enqueued like a CRF call */
        const int serial = abr && !devrc;
        side_ctx sc_;
        sc_.b = b; sc_.pics = pics; sc_.rc = DSVG_OK; sc_.nf = nf;
        if (serial && !abr_out) return DSVG_ERR_ARG;
        if (abr && b->chains) return DSVG_ERR_ARG;
        if (abr) {                                              /* ABR: the length of every packet prefix feeds the rate control */
            /* the pictures are the parallel items (a picture's prefix depends on nothing but its own side information): the GPU
             * waits for this -- with two 4K streams of 30 frames a loop over streams took 5 ms on two threads */
            if (S == 1 || nf == F) dsv1_par_for(S * nf, prefix_picture, &sc_);
            else dsv1_par_for(S, prefix_stream, &sc_);
            if (sc_.rc) { dsv1_log(1, "out of memory while writing the packet prefixes"); return sc_.rc; }
        }
        if (b->chains) {
            if ((rc = code_chains(b, pics, nf, par))) return rc;
        } else
        for (t = 0; t < nf; t++) {
            for (s = 0; s < S; s++) {
                pic_t *pc = &pics[s * F + t];
                dsvg_pic_job *j = &b->jobs[(serial ? 0 : t * S) + s];
                pc->quant = devrc ? 0 : pick_quant(&b->enc[s], pc->isP, pc->forced_intra);
                if (devrc) { dsvg_rc_job *q = &b->rcjobs[t * S + s]; q->rc_slot = s; q->prefix_len = pc->prefix_len; q->forced_intra = pc->forced_intra; }
                j->src_slot = pc->cur_slot;
                /* two reconstruction slots per stream, used alternately: a P picture's prediction is written straight
                 * into the slot its reconstruction will live in (the reference sits in the other one), so the inverse
                 * transform only touches the tiles that carry a residual (dsvg_code_batch) */
                j->ref_recon_slot = pc->isP ? s + S * b->rpar[s] : -1;
                {
                    /* who predicts from this picture?  The stream's next one -- when it has a reference at all.  Nobody: no reconstruction
                     * (struct dsv1_batch, recon_dropped; the reference builds it and never looks at it, dsv_encoder.c:665-700) */
                    int dead = 0;
                    if (pc->is_ref && !serial && !b->keep_all) {
                        const DSV_ENCODER *e = &b->enc[s];
                        if (t + 1 < nf) dead = !pics[s * F + t + 1].has_ref;
                        else dead = !abr && (e->force_metadata || (DSV_FNUM)(e->prev_gop + (DSV_FNUM)e->gop) <= e->next_fnum);
                    }
                    if (t + 1 == nf) b->recon_dropped[s] = (unsigned char)dead;
                    b->n_dropped += dead;
                    if (pc->is_ref && !dead) {
                        b->rpar[s] ^= 1;
                        b->has_recon[s] = 1;
                        j->recon_slot = s + S * b->rpar[s];
                    } else j->recon_slot = -1;
                }
                j->quant = pc->quant;
                j->mvs = (const dsvg_mv *)pc->mvs;
                j->stable_blocks = pc->stable;
                j->out_slot = pc->out_slot;
                j->no_intra_blocks = pc->isP && pc->n_intra == 0;
                j->has_reach = pc->isP;
                memcpy(j->mv_reach, pc->reach, sizeof pc->reach);
                /* who predicts from this reconstruction: the next picture of the stream -- in this call, or (last frame of
                 * the batch) the first of the next one, which is known not to when it starts a GOP (dsv_encoder.c:702-708) */
                j->border_hint = !serial && (t + 1 < nf || b->enc[s].force_metadata ||
                                             (DSV_FNUM)(b->enc[s].prev_gop + (DSV_FNUM)b->enc[s].gop) <= b->enc[s].next_fnum);
                if (t + 1 == nf) b->border_skipped[s] = (unsigned char)(j->border_hint && j->recon_slot >= 0);
            }
            if (serial) {
                if ((rc = dsvg_code_pictures(b->ctx, S, b->jobs))) return rc;
                for (s = 0; s < S; s++) b->out_slots[s] = pics[s * F + t].out_slot;
                if ((rc = dsvg_fetch_pictures(b->ctx, S, b->out_slots, b->outs))) return rc;
                for (s = 0; s < S; s++)
                    if ((rc = assemble(b, s, &pics[s * F + t], &b->outs[s], &abr_out[s], &b->sc0, NULL))) return rc;
            }
        }
        if (devrc) {
            if (!b->rc_seeded) {                                /* the device takes over the streams' rate-control state */
                for (s = 0; s < S; s++) rc_load(&b->rc_dev[s], &b->enc[s]);
                if ((rc = dsvg_rc_upload(b->ctx, 0, S, b->rc_dev))) return rc;
                b->rc_seeded = 1;
            } else {
                /* the caller may have changed a stream's parameters since (bitrate, quality bounds, max_q_step, the nudge: the reference reads
                 * them per frame, dsv_encoder.c:84-165): the pictures of THIS batch and later ones are coded with the new values -- the device's
                 * parameter fields are rewritten behind the batches already enqueued, the state fields stay the device's (advisor round 4) */
                int s0 = -1, s1 = -1;
                for (s = 0; s < S; s++) {
                    dsvg_rc_state now;
                    rc_load(&now, &b->enc[s]);
                    if (rc_par_differs(&now, &b->rc_dev[s])) { rc_par_set(&b->rc_dev[s], &now); if (s0 < 0) s0 = s; s1 = s; }
                }
                if (s0 >= 0 && (rc = dsvg_rc_set_params(b->ctx, s0, s1 - s0 + 1, b->rc_dev + s0))) return rc;
            }
            memcpy(b->rc_par + (size_t)par * S, b->rc_dev, (size_t)S * sizeof(dsvg_rc_state));     /* what this batch's replay uses */
            if ((rc = dsvg_code_batch_rc(b->ctx, nf, S, b->jobs, b->rcjobs))) return rc;
        } else
        if (!serial && !b->chains && (rc = dsvg_code_batch(b->ctx, nf, S, b->jobs))) return rc;   /* whole batch, one upload */
        HP_MARK(HP_ENQUEUE);
        /* the bits of the packet prefixes: nobody waits for them before the packets are assembled, and the GPU is busy now */
#ifdef AB_SYNC_PREFIX                                                 /* (A/B: tools/ab/variant.sh syncprefix host_dsv1_enc "" -DAB_SYNC_PREFIX) */
        if (b->chains) dsv1_par_for(nf, prefix_picture, &sc_);
        else if (!abr) dsv1_par_for(S, prefix_stream, &sc_);
        if (0)
#endif
        if (!abr) {
            /* round 6: in the BACKGROUND -- idle workers write them while this thread waits for the GPU in the next calls (the fetch of the batch
             * before, the load and motion search of the batch after); dsv1_batch_collect of THIS batch joins.  Synchronous, the loop sat between the
             * coding enqueue and the fetch on every step: 1.9 ms with 12 workers, 11.7 ms of a 43 ms step on 4 cores (profiles/r06_cpu_starved.txt) */
            b->bg_sc[par] = sc_;
            b->bg_on[par] = 1;
            if (b->chains) dsv1_par_bg_begin(nf, prefix_picture, &b->bg_sc[par]);      /* (one stream: the pictures are the independent items) */
            else dsv1_par_bg_begin(S, prefix_stream, &b->bg_sc[par]);      /* (one loop at a time: this joins the loop of the batch before, if it still runs) */
            b->bg_on[par ^ 1] = 0;
        }
        if (sc_.rc) { dsv1_log(1, "out of memory while writing the packet prefixes"); return sc_.rc; }
        b->pending[par] = serial ? 2 : 1;               /* 2 = already assembled */
        b->nf_pending[par] = nf;
    }
    HP_MARK(HP_PREFIX);
    hp_batches++;
    b->gcount += (unsigned)nf;
    b->parity ^= 1;
    b->nf_prev = nf;
    return DSVG_OK;
}

static int stage_n(dsv1_batch *b, const void *yuv_host, int nf)
{
    int rc;
    if (!b || !yuv_host) return DSVG_ERR_ARG;
    if (b->nstaged == 2) { dsv1_log(1, "two clips are staged already"); return DSVG_ERR_ARG; }
    if ((rc = dsvg_ingest_begin(b->ctx, yuv_host, b->g.frame_bytes * (size_t)b->nstreams * (size_t)nf, &b->staged_dev[b->nstaged]))) return rc;
    b->staged_host[b->nstaged++] = yuv_host;
    return DSVG_OK;
}
int dsv1_batch_stage(dsv1_batch *b, const void *yuv_host) { return stage_n(b, yuv_host, b ? b->F : 0); }

int dsv1_batch_submit(dsv1_batch *b, const void *yuv, int yuv_on_device, DSV_BUF *out)
{
    /* a device clip's chroma stays where it is (the coding kernels read it until the batch is collected) only when the caller
     * said it holds the clip that long: DSV1_CLIP_HELD.  A plain device clip is copied whole -- the call is done with it when
     * it returns, as it was before round 3 (advisor, round 3) */
    return batch_submit_impl(b, yuv, yuv_on_device != 0, out, 0, yuv_on_device == DSV1_CLIP_HELD);
}

/* Collect the OLDEST submitted batch: one gathered device-to-host copy, then packet assembly in
 * stream order.  The packets of stream s are appended to out[s]. */
int dsv1_batch_collect(dsv1_batch *b, DSV_BUF *out)
{
    int par, S, F, k, rc, nf;
    pic_t *pics;
    if (!b || !out) return DSVG_ERR_ARG;
    S = b->nstreams; F = b->F;
    par = b->pending[b->parity] ? b->parity : (b->parity ^ 1);   /* oldest first */
    if (!b->pending[par]) { dsv1_log(1, "nothing to collect"); return DSVG_ERR_ARG; }
    pics = b->pics + (size_t)par * S * F;
    nf = b->nf_pending[par];
    if (b->bg_on[par]) {
        /* the batch's packet prefixes (background loop started by its submit): whatever is left is finished here with this thread's help */
        HP_BEGIN();
        dsv1_par_bg_end();
        b->bg_on[par] = 0;
        HP_MARK(HP_JOIN);
    }
    if (b->bg_sc[par].rc) {                             /* (set by a worker of the batch's loop, whoever joined it) */
        const int rc_ = b->bg_sc[par].rc;
        b->bg_sc[par].rc = 0;
        dsv1_log(1, "out of memory while writing the packet prefixes");
        b->pending[par] = 0;
        return rc_;
    }
    if (b->pending[par] == 1) {
        HP_BEGIN();
        /* a short batch (nf < F) has a single stream: its pictures are the first nf entries */
        for (k = 0; k < S * nf; k++) b->out_slots[k] = pics[k].out_slot;
        {
            /* the copy comes in pieces of whole streams; the packets of a piece are assembled while the next is on the link */
            asm_ctx ac;
            ac.b = b; ac.pics = pics; ac.out = out; ac.rc = DSVG_OK; ac.nf = nf; ac.s0 = 0;
            ac.par = b->abr_dev && b->rc_seeded ? b->rc_par + (size_t)par * S : NULL;
            if ((rc = dsvg_fetch_pictures_cb(b->ctx, S * nf, b->out_slots, b->outs, S >= 16 ? 4 : 1, nf, asm_piece, &ac))) return rc;
            if (ac.rc) return ac.rc;
        }
        HP_MARK(HP_FETCH);
    }
    b->pending[par] = 0;
    return DSVG_OK;
}

int dsv1_batch_encode(dsv1_batch *b, const void *yuv, int yuv_on_device, DSV_BUF *out)
{
    int rc;
    if (!b || !out) return DSVG_ERR_ARG;
    if (b->pending[0] || b->pending[1]) { dsv1_log(1, "dsv1_batch_encode with batches in flight"); return DSVG_ERR_ARG; }
    if ((rc = batch_submit_impl(b, yuv, yuv_on_device, out, 0, 1))) return rc;
    return dsv1_batch_collect(b, out);
}

/* the same for the first nf frames of a single stream's batch (the tail of a clip) */
static int batch_encode_n(dsv1_batch *b, const void *yuv, int nf, DSV_BUF *out)
{
    int rc;
    if (b->pending[0] || b->pending[1]) { dsv1_log(1, "batch encode with batches in flight"); return DSVG_ERR_ARG; }
    if ((rc = batch_submit_impl(b, yuv, 0, out, nf, 1))) return rc;
    return dsv1_batch_collect(b, out);
}

int dsv1_batch_eos(dsv1_batch *b, int stream, DSV_BUF *out)
{
    DSV_BUF tmp;
    int rc;
    if (!b || stream < 0 || stream >= b->nstreams || !out) return DSVG_ERR_ARG;
    dsv_enc_end_of_stream(&b->enc[stream], &tmp);
    rc = dsv1_buf_append(out, tmp.data, tmp.len);
    dsv_buf_free(&tmp);
    return rc ? DSVG_ERR_ARG : DSVG_OK;
}

/* Join independently encoded closed GOPs into one stream.  A serial encode differs from the
 * concatenation only in the prev_link of picture/EOS packets (metadata packets carry 0): each is
 * the length of the previous picture packet (set_link_offsets dsv_encoder.c:171-192). */
int dsv1_concat_gops(const DSV_BUF *gops, int ngops, DSV_BUF *out)
{
    unsigned prev = 0;
    int g;
    uint8_t eos[DSV_PACKET_HDR_SIZE];
    bitw w;
    out->data = NULL; out->len = 0;
    for (g = 0; g < ngops; g++) {
        unsigned o = 0;
        const unsigned start = out->len;
        if (dsv1_buf_append(out, gops[g].data, gops[g].len)) return DSVG_ERR_ARG;
        while (o + DSV_PACKET_HDR_SIZE <= gops[g].len) {
            uint8_t *pk = out->data + start + o;
            const unsigned next = get_be32(pk + DSV_PACKET_NEXT_OFFSET);
            const int type = pk[DSV_PACKET_TYPE_OFFSET];
            if (memcmp(pk, "DSV1", 4)) return DSVG_ERR_ARG;
            if (type == DSV_PT_EOS) {                 /* drop per-GOP EOS packets */
                out->len = start + o;
                break;
            }
            if (type & DSV_PT_PIC) {
                put_be32(pk + DSV_PACKET_PREV_OFFSET, prev);
                prev = next;
            }
            if (next == 0) break;
            o += next;
        }
    }
    memset(eos, 0, sizeof(eos));
    bw_init(&w, eos);
    write_pkt_hdr(&w, DSV_PT_EOS);
    put_be32(eos + DSV_PACKET_PREV_OFFSET, prev);
    return dsv1_buf_append(out, eos, sizeof(eos)) ? DSVG_ERR_ARG : DSVG_OK;
}

/* =================================================================================================
 * drop-in frame-at-a-time API (dsv_encoder.h:112-121)
 * ================================================================================================= */
/* dsv1_enc_set_strict_packets (verdict round 5): the reference's packet contract as an API call instead of an environment variable.  The public
 * struct has no spare field (its layout is the reference's), so the wish is noted per encoder ADDRESS until the session is created by the first
 * dsv_enc; dsv_enc_init / dsv_enc_free forget the address. */
#define STRICT_SLOTS 64
static struct { const DSV_ENCODER *enc; int on; } strict_tab[STRICT_SLOTS];
static pthread_mutex_t strict_mu = PTHREAD_MUTEX_INITIALIZER;
static int strict_take(const DSV_ENCODER *enc, int remove_only)
{
    int i, on = -1;
    pthread_mutex_lock(&strict_mu);
    for (i = 0; i < STRICT_SLOTS; i++)
        if (strict_tab[i].enc == enc) { on = remove_only ? -1 : strict_tab[i].on; strict_tab[i].enc = NULL; }
    pthread_mutex_unlock(&strict_mu);
    return on;
}
int dsv1_enc_set_strict_packets(DSV_ENCODER *enc, int on)
{
    int i, slot = -1;
    if (!enc) return DSVG_ERR_ARG;
    if (enc->ref) { dsv1_log(1, "dsv1_enc_set_strict_packets after the first dsv_enc call: the session exists already"); return DSVG_ERR_ARG; }
    pthread_mutex_lock(&strict_mu);
    for (i = 0; i < STRICT_SLOTS; i++) {
        if (strict_tab[i].enc == enc) { slot = i; break; }
        if (!strict_tab[i].enc && slot < 0) slot = i;
    }
    if (slot >= 0) { strict_tab[slot].enc = enc; strict_tab[slot].on = on != 0; }
    pthread_mutex_unlock(&strict_mu);
    if (slot < 0) { dsv1_log(1, "dsv1_enc_set_strict_packets: more than %d encoders waiting for their first frame", STRICT_SLOTS); return DSVG_ERR_NOMEM; }
    return DSVG_OK;
}

void dsv_enc_init(DSV_ENCODER *enc)                        /* dsv_encoder.c:696-722 */
{
    strict_take(enc, 1);
    memset(enc, 0, sizeof(*enc));
    enc->prev_gop = (DSV_FNUM)-1;
    enc->quality = DSV_QUALITY_PERCENT(85);
    enc->gop = 24;
    enc->rc_mode = DSV_RATE_CONTROL_CRF;
    enc->bitrate = INT_MAX;
    enc->max_q_step = DSV_MAX_QUALITY / 200;
    enc->min_quality = DSV_QUALITY_PERCENT(1);
    enc->max_quality = DSV_QUALITY_PERCENT(95);
    enc->min_I_frame_quality = DSV_QUALITY_PERCENT(5);
    enc->rc_high_motion_nudge = 1;
    enc->intra_pct_thresh = 50;
    enc->stable_refresh = 14;
    enc->scene_change_delta = 4;
    enc->do_scd = 1;
}

void dsv_enc_start(DSV_ENCODER *enc)                       /* dsv_encoder.c:724-734 */
{
    enc->quality = CLAMPI(enc->quality, 0, DSV_MAX_QUALITY);
    if (enc->rc_mode != DSV_RATE_CONTROL_CRF) {
        enc->rc_quant = (unsigned)enc->quality;
        enc->avg_P_frame_q = enc->quality * 4 / 5;
    }
    enc->force_metadata = 1;
}

void dsv_enc_set_metadata(DSV_ENCODER *enc, DSV_META *md) { memcpy(&enc->vidmeta, md, sizeof(DSV_META)); }
void dsv_enc_force_metadata(DSV_ENCODER *enc) { enc->force_metadata = 1; }

/* Session behind the frame-at-a-time API.  CRF streams are PIPELINED and GOP-PARALLEL: dsv_enc copies the frame into a
 * pinned host batch (the CLI reuses its picture buffer, dsv_main.c:506-520), and every F frames -- a lookahead of several GOPs,
 * DSV1_ENC_LOOKAHEAD frames, by default 16 GOPs within 768 MB of pinned memory per half -- one batch goes to the device with
 * dsv1_batch_submit while the batch before it is collected.  The batch runs in chain mode (dsv1_stream_open): analysis of all
 * its frames, the encoder's serial state machine (GOP starts, scene changes, forced-intra pictures, stability) replayed on
 * the host, then the chains of pictures between I pictures coded side by side -- byte for byte the frame-serial stream, for
 * any CRF configuration.  The finished packets wait in a backlog that later calls hand out (at most two buffers per call, as
 * the reference does: metadata + picture, dsv_encoder.c:804-810).  A caller gets 0 buffers while the lookahead fills --
 * dsv_main.c:521-531 loops over whatever count comes back -- and dsv_enc_end_of_stream returns the rest of the backlog in
 * front of the EOS packet, in one buffer: the bytes that reach the file are those of the frame-synchronous encoder.  Changes
 * the caller makes to the encoder's public fields (quality, rate-control parameters, force_metadata) take effect with the frame
 * of the next call, as in the reference: the frames gathered so far are submitted as a short batch first.  ABR needs every
 * packet's size before the next quantiser: its pictures are coded one after the other, but 32 frames (DSV1_ENC_LOOKAHEAD) are
 * gathered and ANALYSED together first, and the group's packets come out of the call that completes it
 * (DSV1_ENC_PIPELINE=0: one frame per call, CRF and ABR).
 * A device error ends the session: every later dsv_enc returns 0 buffers (and logs), nothing is written out of bounds. */
typedef struct {
    dsv1_batch *b;
    int pipelined, F, fill, cur, inflight, failed;
    int gathered;               /* ABR: frames are gathered F at a time (analysis of the whole group at once, coding frame by frame) */
    uint8_t *pin[2];
    void *dev;                  /* device buffer the frames of the batch being gathered are uploaded into, one by one (dsvg_ingest_open / _part) */
    size_t fb;
    DSV_BUF backlog;            /* finished packets not handed out yet */
    unsigned off;               /* first byte of the backlog not handed out */
    /* what the frames gathered so far will be coded with: a change of the caller's (rate-control parameters, quality, a forced
     * metadata packet) takes effect with the FRAME it precedes, as in the reference -- the frames gathered before it go to the
     * device as a short batch first */
    dsvg_rc_state par_cur;
    int quality_cur, fm_seen;
} enc_sess;

static void sess_free(enc_sess *ss)
{
    int i;
    if (!ss) return;
    if (ss->b) {
        if (ss->b->ctx) { dsvg_ctx_sync(ss->b->ctx); for (i = 0; i < 2; i++) if (ss->pin[i]) dsvg_host_free(ss->b->ctx, ss->pin[i]); }
        ss->b->enc = NULL;                              /* the caller owns the DSV_ENCODER */
        dsv1_batch_close(ss->b);
    }
    dsv_buf_free(&ss->backlog);
    free(ss);
}

void dsv_enc_free(DSV_ENCODER *enc)
{
    strict_take(enc, 1);
    if (enc->ref) {
        enc_sess *ss = (enc_sess *)enc->ref;
        if ((ss->pipelined || ss->gathered) && !ss->failed && (ss->fill > 0 || ss->inflight > 0 || ss->backlog.len > ss->off))
            dsv1_log(1, "dsv_enc_free without dsv_enc_end_of_stream: %d buffered frame(s) and %d batch(es) in flight are dropped", ss->fill, ss->inflight);
        sess_free(ss);
        enc->ref = NULL;
    }
    if (enc->stability) { dsv_free(enc->stability); enc->stability = NULL; }
    if (enc->stable_blocks) { dsv_free(enc->stable_blocks); enc->stable_blocks = NULL; }
}

/* collect the oldest batch in flight into the backlog */
static int sess_collect(enc_sess *ss)
{
    DSV_BUF tmp = {NULL, 0};
    int rc = dsv1_batch_collect(ss->b, &tmp);
    if (!rc) ss->inflight--;                            /* (a failed collect leaves the batch pending: the session is marked failed) */
    if (!rc && tmp.len) rc = dsv1_buf_append(&ss->backlog, tmp.data, tmp.len) ? DSVG_ERR_ARG : DSVG_OK;
    dsv_buf_free(&tmp);
    return rc;
}

/* submit the frames gathered so far (a short batch at the end of the stream) */
static int sess_submit(enc_sess *ss)
{
    int rc;
    /* the frames are on the device already (or on their way: the load waits for the last part on the device) */
    if ((rc = batch_submit_impl(ss->b, ss->dev, 1, NULL, ss->fill, 0))) return rc;      /* (the ingest buffer is reused two batches on: chroma is copied) */
    ss->dev = NULL;
    ss->inflight++;
    ss->cur ^= 1;
    ss->fill = 0;
    return DSVG_OK;
}

/* hand out up to `max` whole packets from the backlog */
static int sess_pop(enc_sess *ss, DSV_BUF *bufs, int max)
{
    int n = 0;
    while (n < max && ss->off + DSV_PACKET_HDR_SIZE <= ss->backlog.len) {
        const unsigned len = get_be32(ss->backlog.data + ss->off + DSV_PACKET_NEXT_OFFSET);
        if (!len || ss->off + len > ss->backlog.len) break;
        /* a metadata packet travels with the picture that follows it (the reference returns both from one call) */
        if (n == max - 1 && ss->backlog.data[ss->off + DSV_PACKET_TYPE_OFFSET] == DSV_PT_META) break;
        dsv_mk_buf(&bufs[n], (int)len);
        memcpy(bufs[n].data, ss->backlog.data + ss->off, len);
        ss->off += len;
        n++;
    }
    if (ss->off == ss->backlog.len && ss->backlog.len) { ss->backlog.len = 0; ss->off = 0; }   /* keep the allocation */
    return n;
}

/* the way out of dsv_enc: what the caller finds in force_metadata now is what a later call compares with */
static int sess_ret(DSV_ENCODER *enc, enc_sess *ss, DSV_BUF *bufs)
{
    ss->fm_seen = enc->force_metadata;
    return sess_pop(ss, bufs, 2);
}

/* the frames gathered so far go to the device as a short batch (end of stream, a flush call, or a change of the caller's to
 * the encoder's parameters in mid-group); the batches in flight stay in flight */
static void sess_push_partial(enc_sess *ss)
{
    if (ss && ss->pipelined && !ss->failed && ss->fill > 0) {
        int rc = DSVG_OK;
        if (ss->inflight == 2) rc = sess_collect(ss);
        if (!rc) rc = sess_submit(ss);
        if (rc) { dsv1_log(1, "GPU encode failed: %s", dsvg_last_error()); ss->failed = 1; ss->fill = 0; }
    }
    if (ss && !ss->pipelined && !ss->failed && ss->fill > 0) {
        /* ABR on the frame-serial host path: the frames gathered since the last complete group */
        DSV_BUF acc = {NULL, 0};
        int rc;
        if (ss->gathered && ss->dev) {
            rc = batch_submit_impl(ss->b, ss->dev, 1, &acc, ss->fill, 0);
            ss->dev = NULL;
            if (!rc) rc = dsv1_batch_collect(ss->b, &acc);
        } else rc = batch_encode_n(ss->b, ss->pin[0], ss->fill, &acc);
        if (!rc && acc.len) rc = dsv1_buf_append(&ss->backlog, acc.data, acc.len) ? DSVG_ERR_ARG : DSVG_OK;
        dsv_buf_free(&acc);
        ss->fill = 0;
        if (rc) { dsv1_log(1, "GPU encode failed: %s", dsvg_last_error()); ss->failed = 1; }
    }
}

/* everything the session still holds -- frames waiting for a full batch, batches in flight -- coded and collected into the
 * backlog (end of stream, or the caller's flush calls dsv_enc(enc, NULL, bufs)) */
static void sess_flush(enc_sess *ss)
{
    sess_push_partial(ss);
    if (ss && ss->pipelined && !ss->failed) {
        int rc = DSVG_OK;
        while (!rc && ss->inflight > 0) rc = sess_collect(ss);          /* oldest first */
        if (rc) { dsv1_log(1, "GPU encode failed at end of stream: %s", dsvg_last_error()); ss->failed = 1; ss->fill = 0; }
    }
}

void dsv_enc_end_of_stream(DSV_ENCODER *enc, DSV_BUF *bufs)   /* dsv_encoder.c:766-778 */
{
    bitw w;
    enc_sess *ss = (enc_sess *)enc->ref;
    uint8_t eos[DSV_PACKET_HDR_SIZE];
    unsigned rest = 0;
    /* (a caller that drained the session with dsv_enc(enc, NULL, bufs) calls finds nothing left here: bufs[0] is then the EOS
     * packet alone, exactly the reference's contract) */
    sess_flush(ss);
    if (ss) rest = ss->backlog.len - ss->off;
    memset(eos, 0, sizeof(eos));
    bw_init(&w, eos);
    write_pkt_hdr(&w, DSV_PT_EOS);
    link_packet(enc, eos, sizeof(eos), 1);
    dsv_mk_buf(&bufs[0], (int)(rest + sizeof(eos)));
    if (rest) {
        memcpy(bufs[0].data, ss->backlog.data + ss->off, rest);
        ss->backlog.len = 0; ss->off = 0;
    }
    memcpy(bufs[0].data + rest, eos, sizeof(eos));
}

/* frame -> pinned batch memory, the rows of its three planes shared out over the worker pool (a 1080p frame is 3 MB: one
 * thread copies it in ~0.3 ms, which would cap a single stream at ~3000 frames/s) */
typedef struct { const DSV_FRAME *f; uint8_t *dst; int rows, parts; } copy_ctx;
static void copy_part(void *ctx, int i, int tid)
{
    const copy_ctx *c = (const copy_ctx *)ctx;
    int r0 = (int)((long)c->rows * i / c->parts), r1 = (int)((long)c->rows * (i + 1) / c->parts), p, base = 0;
    uint8_t *o = c->dst;
    (void)tid;
    for (p = 0; p < 3; p++) {
        const DSV_PLANE *pl = &c->f->planes[p];
        int y0 = r0 - base, y1 = r1 - base, y;
        if (y0 < 0) y0 = 0;
        if (y1 > pl->h) y1 = pl->h;
        if (y0 < y1) {
            if (pl->stride == pl->w) memcpy(o + (size_t)y0 * pl->w, pl->data + (size_t)y0 * pl->stride, (size_t)(y1 - y0) * pl->w);
            else for (y = y0; y < y1; y++) memcpy(o + (size_t)y * pl->w, pl->data + (size_t)y * pl->stride, (size_t)pl->w);
        }
        base += pl->h;
        o += (size_t)pl->w * pl->h;
    }
}

int dsv_enc(DSV_ENCODER *enc, DSV_FRAME *frame, DSV_BUF *bufs)
{
    enc_sess *ss;
    int c, rc;

    if (!bufs) { dsv1_log(1, "null buffer list passed to encoder!"); return 0; }
    if (!frame && !enc->ref) return 0;                   /* a flush call before the first frame: nothing is owed */
    if (!enc->ref) {
        const char *e = getenv("DSV1_ENC_PIPELINE"), *la = getenv("DSV1_ENC_LOOKAHEAD");
        /* frame-synchronous session (one frame in, its packets out: the reference's contract): asked for by dsv1_enc_set_strict_packets or DSV1_ENC_PIPELINE=0 */
        const int sync_mode = strict_take(enc, 0) == 1 || (e && atoi(e) == 0);
        int chains = 0;
        ss = (enc_sess *)calloc(1, sizeof(*ss));
        if (!ss) { dsv1_log(1, "out of memory"); dsv_frame_ref_dec(frame); return 0; }
        {
            /* ABR streams are pipelined like CRF ones since round 4: their rate control runs on the device (k_rc), so a group of
             * frames is enqueued whole and the next group's upload and analysis overlap it (DSV1_ABR_SERIAL=1: the gathered,
             * frame-by-frame path of round 3) */
            const char *as = getenv("DSV1_ABR_SERIAL");
            const int abr_dev = enc->rc_mode != DSV_RATE_CONTROL_CRF && !(as && atoi(as) != 0);
            ss->pipelined = (enc->rc_mode == DSV_RATE_CONTROL_CRF || abr_dev) && !sync_mode;
        }
        ss->F = 1;
        if (ss->pipelined && enc->rc_mode != DSV_RATE_CONTROL_CRF) {
            /* ABR: groups of 32 frames (DSV1_ENC_LOOKAHEAD, within 256 MB of pinned memory per half), one stream, no chains */
            const DSV_META *m = &enc->vidmeta;
            const int hs = (m->subsamp >> 2) & 3, vs = m->subsamp & 3;
            const size_t fbytes = (size_t)m->width * m->height + 2 * (size_t)((m->width + (1 << hs) - 1) >> hs) * (size_t)((m->height + (1 << vs) - 1) >> vs);
            long want = la ? atol(la) : 32L, cap = (long)(((size_t)256 << 20) / (fbytes ? fbytes : 1));
            if (want > cap) want = cap;
            if (want > 256) want = 256;
            if (want < 2) want = 2;
            ss->F = (int)want;
        } else
        if (ss->pipelined) {
            /* lookahead: 16 GOPs (intra-only streams: 64 pictures), at least 8 frames, within 768 MB of pinned memory per half */
            const DSV_META *m = &enc->vidmeta;
            const int hs = (m->subsamp >> 2) & 3, vs = m->subsamp & 3;
            const size_t fbytes = (size_t)m->width * m->height + 2 * (size_t)((m->width + (1 << hs) - 1) >> hs) * (size_t)((m->height + (1 << vs) - 1) >> vs);
            const int g = enc->gop > 0 ? (int)enc->gop : 1;
            long want = la ? atol(la) : (enc->gop > 0 ? 16L * g : 64L), cap = (long)(((size_t)768 << 20) / (fbytes ? fbytes : 1));
            if (want > cap) want = cap;
            if (want > 1024) want = 1024;
            if (want < 8) want = 8;
            ss->F = (int)want;
            chains = enc->gop > 0 ? (ss->F + g - 1) / g + 1 : ss->F;
            if (chains > 64) chains = 64;
        } else if (enc->rc_mode != DSV_RATE_CONTROL_CRF && !sync_mode) {
            /* ABR (the reference CLI's default): every packet's size feeds the next quantiser, so the pictures are coded one after
             * the other -- but padding, pyramid, motion search and the GOP / scene-change decisions depend on source pixels only:
             * frames are gathered 32 at a time (DSV1_ENC_LOOKAHEAD, within 256 MB) and analysed in one go, as dsv1_batch_encode
             * does for an ABR batch; the packets of a group come out of the call that completes it, byte for byte the
             * frame-synchronous encoder's (dsv_encoder.c:84-132,728-764) */
            const DSV_META *m = &enc->vidmeta;
            const int hs = (m->subsamp >> 2) & 3, vs = m->subsamp & 3;
            const size_t fbytes = (size_t)m->width * m->height + 2 * (size_t)((m->width + (1 << hs) - 1) >> hs) * (size_t)((m->height + (1 << vs) - 1) >> vs);
            long want = la ? atol(la) : 32L, cap = (long)(((size_t)256 << 20) / (fbytes ? fbytes : 1));
            if (want > cap) want = cap;
            if (want > 256) want = 256;
            if (want < 1) want = 1;
            ss->F = (int)want;
            ss->gathered = ss->F > 1;
        }
        if ((rc = batch_open_on(&ss->b, enc, 0, dsv1_device, 1, ss->F, chains))) {
            dsv1_log(1, "GPU session could not be opened: %s", dsvg_last_error());
            free(ss);
            dsv_frame_ref_dec(frame);
            return 0;
        }
        ss->fb = ss->b->g.frame_bytes;
        for (c = 0; c < (ss->pipelined ? 2 : 1); c++)
            if (dsvg_host_alloc(ss->b->ctx, (void **)&ss->pin[c], ss->fb * (size_t)ss->F)) {
                dsv1_log(1, "pinned frame buffer: %s", dsvg_last_error());
                sess_free(ss);
                dsv_frame_ref_dec(frame);
                return 0;
            }
        enc->ref = ss;
    }
    ss = (enc_sess *)enc->ref;
    if (!frame) {
        /* FLUSH CALL (an extension: the reference would dereference the null frame): code and collect everything the session
         * still holds, then hand out the packets owed like any other call -- at most two per call (metadata + picture), one
         * packet per DSV_BUF; 0 = drained.  After that dsv_enc_end_of_stream returns exactly one EOS packet. */
        sess_flush(ss);
        return sess_ret(enc, ss, bufs);
    }
    if (ss->failed || ss->fill >= ss->F) {
        /* an earlier device error ended the session (never copy past the pinned batch) */
        if (!ss->failed) { ss->failed = 1; ss->fill = 0; }
        dsv1_log(1, "GPU session failed earlier: frame dropped");
        dsv_frame_ref_dec(frame);
        return 0;
    }
    if (ss->pipelined || ss->gathered) {
        /* the reference reads the encoder's public fields when it codes a frame (dsv_encoder.c:84-165,794-803): what the caller changed
         * since the last call applies from THIS frame on -- the frames gathered before it are submitted as a short batch first */
        dsvg_rc_state now;
        rc_load(&now, enc);
        if (ss->fill > 0 && (rc_par_differs(&now, &ss->par_cur) || enc->quality != ss->quality_cur || (enc->force_metadata && !ss->fm_seen))) {
            /* the gathered frames are coded with what was in force when they arrived: the old values go back into the public fields for
             * the duration of the short batch's submit (which reads them there), then the caller's new ones return */
            const int newq = enc->quality, forced = enc->force_metadata && !ss->fm_seen;
            rc_par_to_enc(enc, &ss->par_cur);
            enc->quality = ss->quality_cur;
            if (forced) enc->force_metadata = 0;
            sess_push_partial(ss);
            rc_par_to_enc(enc, &now);
            enc->quality = newq;
            if (forced) enc->force_metadata = 1;
            if (ss->failed) { dsv_frame_ref_dec(frame); return sess_ret(enc, ss, bufs); }
        }
        rc_load(&ss->par_cur, enc);
        ss->quality_cur = enc->quality;
    }
    /* the frame is copied now: the caller may reuse its pixel memory as soon as this returns */
    {
        copy_ctx cc;
        cc.f = frame; cc.dst = ss->pin[ss->cur] + (size_t)ss->fill * ss->fb;
        cc.rows = frame->planes[0].h + frame->planes[1].h + frame->planes[2].h;
        cc.parts = ss->fb >= ((size_t)1 << 20) ? 8 : 1;
        if (cc.parts > 1) dsv1_par_for(cc.parts, copy_part, &cc);
        else copy_part(&cc, 0, 0);
    }
    dsv_frame_ref_dec(frame);                           /* the encoder owns the frame (dsv_encoder.c:38-40) */
    if (ss->pipelined || ss->gathered) {
        /* its upload starts now, under the caller's reading of the next frame -- not in one burst when the batch is full */
        rc = DSVG_OK;
        if (!ss->dev) rc = dsvg_ingest_open(ss->b->ctx, ss->fb * (size_t)ss->F, &ss->dev);
        if (!rc) rc = dsvg_ingest_part(ss->b->ctx, ss->dev, (size_t)ss->fill * ss->fb, ss->pin[ss->cur] + (size_t)ss->fill * ss->fb, ss->fb);
        if (rc) { dsv1_log(1, "frame upload failed: %s", dsvg_last_error()); ss->failed = 1; ss->fill = 0; return sess_ret(enc, ss, bufs); }
    }
    ss->fill++;
    if (!ss->pipelined) {
        DSV_BUF acc = {NULL, 0};
        if (ss->fill < ss->F) return sess_ret(enc, ss, bufs);     /* (ABR: the group is not complete yet) */
        ss->fill = 0;
        if (ss->gathered) {                                     /* the group's frames are on the device already (or on their way) */
            rc = batch_submit_impl(ss->b, ss->dev, 1, &acc, ss->F, 0);
            ss->dev = NULL;
            if (!rc) rc = dsv1_batch_collect(ss->b, &acc);
        } else rc = dsv1_batch_encode(ss->b, ss->pin[0], 0, &acc);
        if (!rc && acc.len) rc = dsv1_buf_append(&ss->backlog, acc.data, acc.len) ? DSVG_ERR_ARG : DSVG_OK;
        dsv_buf_free(&acc);
        if (rc) { dsv1_log(1, "GPU encode failed: %s", dsvg_last_error()); ss->failed = 1; return 0; }
        return sess_ret(enc, ss, bufs);
    }
    if (ss->fill == ss->F) {
        /* the pinned half about to be refilled next was used by the batch before last: it is collected first */
        rc = DSVG_OK;
        if (ss->inflight == 2) rc = sess_collect(ss);
        if (!rc) rc = sess_submit(ss);
        if (!rc && ss->inflight == 2) rc = sess_collect(ss);
        if (rc) {
            /* the session is over: no later call may copy into the batch or touch the device again */
            dsv1_log(1, "GPU encode failed: %s", dsvg_last_error());
            ss->failed = 1; ss->fill = 0;
            return sess_ret(enc, ss, bufs);
        }
    }
    return sess_ret(enc, ss, bufs);
}
