/*
 * orc.h -- ORACLE (test infrastructure, NOT product code).
 *
 * Scalar, re-entrant C restatement of the per-frame hot path of DSV1
 * (LMP88959/Digital-Subband-Video-1 @ 2024_10_08).  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP path
 * against a CPU implementation.  Nothing in the product library may link or call it.
 *
 * Parity status: PINNED.  Every function here is checked byte-for-byte against the real
 * reference compiled from /root/reference (oracle/_ref/libdsv1ref.so, see Makefile) by
 * tests/test_oracle_vs_ref.py, and against the committed fixtures in tests/golden/ that
 * were generated from that same reference build (tools/make_goldens.py).
 *
 * Struct layouts below are ABI-identical to the reference's public types
 * (dsv.h:86-150,181-198; dsv_internal.h:39-49) so that the same ctypes mirrors drive the
 * reference, the oracle and the product.
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BORDER 64          /* DSV_FRAME_BORDER, dsv_internal.h:37 */
#define ORC_MAX_PYR 5          /* DSV_MAX_PYRAMID_LEVELS, dsv_encoder.h:35 */
#define ORC_MAXLVL 3           /* DSV_MAXLVL, dsv_internal.h:88 */

#define ORC_RSHIFT_UP(x, s) (((x) + (1 << (s)) - 1) >> (s))   /* DSV_ROUND_SHIFT dsv.h:62 */
#define ORC_HSHIFT(fmt) (((fmt) >> 2) & 3)                    /* dsv.h:83 */
#define ORC_VSHIFT(fmt) ((fmt) & 3)                           /* dsv.h:84 */

typedef struct {            /* == DSV_META dsv.h:86-95 */
    int width, height, subsamp;
    int fps_num, fps_den, aspect_num, aspect_den;
} orc_meta;

typedef struct {            /* == DSV_PLANE dsv.h:97-104 */
    uint8_t *data;
    int len, format, stride, w, h, hs, vs;
} orc_plane;

typedef struct {            /* == DSV_COEFS dsv.h:107-112 */
    int32_t *data;
    int width, height;
} orc_coefs;

typedef struct {            /* == DSV_FRAME dsv.h:114-126 */
    uint8_t *alloc;
    orc_plane planes[3];
    int refcount, format, width, height, border;
} orc_frame;

typedef struct {            /* == DSV_MV dsv.h:137-150 */
    union { struct { int16_t x, y; } mv; int32_t all; } u;
    uint8_t mode, submask, lo_var, lo_tex, high_detail;
} orc_mv;

typedef struct {            /* == DSV_PARAMS dsv.h:181-193 */
    orc_meta *vidmeta;
    int is_ref, has_ref, blk_w, blk_h, nblocks_h, nblocks_v;
} orc_params;

typedef struct {            /* == DSV_STABILITY dsv_internal.h:39-44 */
    orc_params *params;
    unsigned char *stable_blocks;
    unsigned char cur_plane, isP;
} orc_stability;

typedef struct {            /* == DSV_BS dsv_internal.h:46-49 */
    uint8_t *start;
    unsigned pos;           /* bit position */
} orc_bs;

typedef struct {            /* == DSV_HME dsv_encoder.h:124-130 */
    orc_params *params;
    orc_frame *src[ORC_MAX_PYR + 1];
    orc_frame *ref[ORC_MAX_PYR + 1];
    orc_mv *mvf[ORC_MAX_PYR + 1];
    int levels;
} orc_hme;

/* ---- bit writer / reader (bs.c) ---------------------------------------------------- */
void     orc_bs_init(orc_bs *bs, uint8_t *buf);
void     orc_bs_align(orc_bs *bs);
void     orc_bs_put_bits(orc_bs *bs, unsigned n, unsigned v);
unsigned orc_bs_get_bits(orc_bs *bs, unsigned n);
void     orc_bs_put_ueg(orc_bs *bs, unsigned v);
unsigned orc_bs_get_ueg(orc_bs *bs);
void     orc_bs_put_seg(orc_bs *bs, int v);
int      orc_bs_get_seg(orc_bs *bs);
void     orc_bs_put_neg(orc_bs *bs, int v);
int      orc_bs_get_neg(orc_bs *bs);
void     orc_bs_append(orc_bs *bs, const uint8_t *data, int len);
static inline unsigned orc_bs_bytepos(const orc_bs *bs) { return bs->pos >> 3; }

typedef struct { orc_bs bs; int nz; } orc_zbrle;      /* bs.c:222-267 */
void orc_rle_init(orc_zbrle *r, uint8_t *buf);
void orc_rle_put(orc_zbrle *r, int bit);
int  orc_rle_get(orc_zbrle *r);
int  orc_rle_finish_write(orc_zbrle *r);              /* returns byte length */

/* ---- subband transform (sbt.c) ------------------------------------------------------ */
void orc_fwd_sbt(const orc_plane *src, orc_coefs *dst, int isP);
void orc_inv_sbt(orc_plane *dst, orc_coefs *src, int q, int isP, int c);

/* ---- quantiser + coefficient coder (hzcc.c) ----------------------------------------- */
int  orc_get_quant(int q, int isP, int level);
int  orc_lb2(unsigned n);
void orc_encode_plane(orc_bs *bs, orc_coefs *src, int q, const orc_stability *stab);
void orc_decode_plane(uint8_t *in, unsigned len, orc_coefs *dst, int q, const orc_stability *stab);

/* ---- frames (frame.c) ---------------------------------------------------------------- */
orc_frame *orc_frame_new(int format, int w, int h, int border);
void orc_frame_free(orc_frame *f);
void orc_frame_wrap_planar(orc_frame *f, int format, uint8_t *data, int w, int h); /* no alloc */
void orc_frame_copy(orc_frame *dst, const orc_frame *src);
void orc_frame_extend(orc_frame *f);
void orc_frame_extend_luma(orc_frame *f);
void orc_frame_ds2x_luma(orc_frame *dst, const orc_frame *src);
int  orc_frame_avg_luma(const orc_frame *f);
void orc_coefs_new(orc_coefs c[3], int format, int w, int h);   /* one calloc, c[0].data owns */

/* ---- motion compensation (bmc.c) ------------------------------------------------------ */
void orc_sub_pred(const orc_mv *mv, const orc_params *p, orc_frame *dif, orc_frame *inp, const orc_frame *ref);
void orc_add_pred(const orc_mv *mv, const orc_params *p, orc_frame *dif, orc_frame *out, const orc_frame *ref);
void orc_frame_add(orc_frame *dst, const orc_frame *src);

/* ---- hierarchical motion estimation (hme.c) ------------------------------------------- */
int  orc_hme_run(orc_hme *h);      /* fills h->mvf[0..levels] (calloc'd; caller frees), returns intra % */
void orc_mv_pred(const orc_mv *vecs, const orc_params *p, int x, int y, int *px, int *py); /* dsv.c:200 */

/* ---- session layer (dsv_encoder.c / dsv_decoder.c / dsv_main.c mapping) ---------------- */
typedef struct {
    orc_meta meta;
    int quality;            /* 0..2047, already converted (dsv_main.c:46-50) */
    int gop, do_scd, rc_mode /* 0 = CRF, 1 = ABR (library enum, dsv_encoder.h:32-33) */;
    int rc_high_motion_nudge;
    unsigned bitrate;
    int max_q_step, min_quality, max_quality, min_I_frame_quality;
    int intra_pct_thresh, scene_change_delta;
    unsigned stable_refresh;
    int pyramid_levels;
} orc_enc_cfg;

typedef struct orc_encoder orc_encoder;
typedef struct orc_decoder orc_decoder;

/* fills cfg exactly as the reference CLI would for these flags (dsv_main.c:423-489):
 * qp_pct = -qp, rc_mode_cli = -rc_mode (0 ABR / 1 CRF), kbps = -kbps (0 = auto), stabref 0 = auto */
void orc_cfg_from_cli(orc_enc_cfg *cfg, int w, int h, int subsamp, int qp_pct, int gop,
                      int rc_mode_cli, int kbps, int scd, int ipct, int pyrlevels, int stabref);

orc_encoder *orc_enc_open(const orc_enc_cfg *cfg);
void orc_enc_set_next_fnum(orc_encoder *e, unsigned fnum);
/* a caller's change of the public fields between two frames (quality, bitrate, quality bounds, max_q_step, nudge) / dsv_enc_force_metadata */
void orc_enc_set_params(orc_encoder *e, const orc_enc_cfg *cfg);
void orc_enc_force_metadata(orc_encoder *e);
/* encode one planar frame; appends 1 or 2 packets to *out (realloc'd), returns bytes appended.
 * if recon != NULL receives the encoder's reconstruction (planar, tightly packed) */
size_t orc_enc_frame(orc_encoder *e, const uint8_t *yuv, uint8_t **out, size_t *outlen, size_t *outcap,
                     uint8_t *recon);
size_t orc_enc_eos(orc_encoder *e, uint8_t **out, size_t *outlen, size_t *outcap);
void orc_enc_close(orc_encoder *e);
/* introspection for tests: last frame's motion field / stable flags / quant */
const orc_mv *orc_enc_last_mvs(const orc_encoder *e, int *nblk);
const unsigned char *orc_enc_last_stable(const orc_encoder *e, int *nblk);

orc_decoder *orc_dec_open(void);
/* decode one packet; returns 0 ok(frame written to yuv_out, tightly packed), 2 EOS, 3 meta, 1 error */
int  orc_dec_packet(orc_decoder *d, const uint8_t *pkt, size_t len, uint8_t *yuv_out, unsigned *fnum);
void orc_dec_get_meta(const orc_decoder *d, orc_meta *m);
void orc_dec_close(orc_decoder *d);

/* Coverage of the decision points SURVEY.md Appendix F lists (round 4, tests/test_appendix_f_coverage.py): the oracle counts how
 * often each branch of the level-0 motion search (hme.c:544-721) and of the encoder's per-picture decisions (dsv_encoder.c:
 * 236-252,330-408,538-554) is taken while it encodes -- process-wide counters, test infrastructure only. */
enum {
    ORC_COV_HP_SKIPPED,        /* best <= bw*bh: no half-pel search (hme.c:551,587-591) */
    ORC_COV_HP_KEPT_FULLPEL,   /* searched, no half-pel candidate better (m == -1) */
    ORC_COV_HP_REFINED,        /* a half-pel candidate won */
    ORC_COV_NB0_PLAIN, ORC_COV_NB1_PLAIN, ORC_COV_NB2_PLAIN, ORC_COV_NB3_PLAIN,   /* 0..3 qualifying neighbours, high_detail = 0 (hme.c:621-648) */
    ORC_COV_NB0_HD, ORC_COV_NB1_HD, ORC_COV_NB2_HD, ORC_COV_NB3_HD,               /* ... high_detail = 1 */
    ORC_COV_INTRA_ZEROVAR, ORC_COV_INTRA_REFVAR, ORC_COV_INTRA_FLATSRC, ORC_COV_INTRA_AVG, ORC_COV_INTRA_BADSAD, ORC_COV_INTRA_CHROMA,   /* the six tests of hme.c:652-682, first one that fires */
    ORC_COV_INTRA_NONE,        /* no test fired: inter */
    ORC_COV_VETO_TAKEN,        /* block_intra_test sent the block back to inter (hme.c:685-687) */
    ORC_COV_VETO_NOT_TAKEN,
    ORC_COV_LOWTEX_ALL_INTRA,  /* src_tex <= 1: no quadrant vote, submask 0xF (hme.c:691-692) */
    ORC_COV_QUAD_VOTE,         /* the four quadrants voted */
    ORC_COV_SUBMASK0,          /* + m: final submask m of a block that reached the vote (0 = every quadrant preferred inter: the block stays inter) */
    ORC_COV_LO_TEX = ORC_COV_SUBMASK0 + 16, ORC_COV_LO_VAR, ORC_COV_LO_NEITHER,
    ORC_COV_FORCED_INTRA_IPCT, /* intra_pct > ipct turned a P picture into an I picture (dsv_encoder.c:248-252) */
    ORC_COV_FORCED_INTRA_SCENE,/* scene change (dsv_encoder.c:546-551) */
    ORC_COV_P_KEPT,            /* a P picture stayed a P picture */
    ORC_COV_STAB_REFRESH,      /* refresh_ctr reached stable_refresh: accumulators cleared (dsv_encoder.c:345-348) */
    ORC_COV_STAB_RESET_LO,     /* lo_tex / lo_var block: accumulators set to 0x3fff (:388-391) */
    ORC_COV_STABLE_BY_HD, ORC_COV_STABLE_BY_AVG, ORC_COV_UNSTABLE_INTER, ORC_COV_INTRA_BLOCK_FLAG, ORC_COV_STABLE_I, ORC_COV_UNSTABLE_I,
    ORC_COV_N
};
extern unsigned long long orc_cov[ORC_COV_N];
void orc_cov_reset(void);
int  orc_cov_read(unsigned long long *out, int n);      /* copies min(n, ORC_COV_N) counters, returns ORC_COV_N */

/* (the synthetic clip generator lives in tools/clipgen/clipgen.c: it is input data, not part of the checker) */

#ifdef __cplusplus
}
#endif
#endif
