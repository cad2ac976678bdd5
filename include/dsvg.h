/*
 * dsvg.h -- C ABI of the MI355X-native DSV1 hot path (libdsv1_mi355x.so).
 *
 * Plain C, plain pointers and sizes: this is what a host written in C (the reference's own
 * language) binds.  Two seams are exported:
 *
 *   1. OPERATOR LEVEL (dsvg_op_*): one twin per function of the reference's operator API
 *      (dsv_internal.h:94-109, dsv.h:169, dsv_encoder.h:132).  Same argument meaning, host
 *      pointers in and out, results byte-identical; each call stages through HBM and runs the
 *      HIP kernels.  This is the seam the parity tests drive.
 *
 *   2. PIPELINE LEVEL (dsvg_ctx_* / dsvg_load_frames / dsvg_analyse / dsvg_code_pictures /
 *      dsvg_fetch_pictures): device-resident frames, batched over many closed GOPs, used by the
 *      session layer (dsv_enc / dsv_enc_gops in dsv1_api.h) and by bench.py.
 *
 * All functions return DSVG_OK (0) or a negative DSVG_ERR_* code; nothing here falls back to a
 * CPU implementation -- if no HIP device is usable the call fails.
 *
 * Struct layouts are ABI-identical to the reference's DSV_PLANE / DSV_COEFS / DSV_FRAME /
 * DSV_MV / DSV_PARAMS / DSV_STABILITY / DSV_BS / DSV_HME (dsv.h:86-198, dsv_internal.h:39-49,
 * dsv_encoder.h:124-130) so existing callers can pass their own structs.
 */
#ifndef DSVG_H
#define DSVG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSVG_OK               0
#define DSVG_ERR_HIP         (-1)   /* a HIP runtime call or kernel failed (see dsvg_last_error) */
#define DSVG_ERR_ARG         (-2)   /* bad argument */
#define DSVG_ERR_UNSUPPORTED (-3)   /* geometry outside what the kernels handle */
#define DSVG_ERR_OVERFLOW    (-4)   /* packed plane exceeded its output bound */
#define DSVG_ERR_NODEVICE    (-5)   /* no usable MI355X/HIP device */
#define DSVG_ERR_NOMEM       (-7)   /* a host allocation failed (the object being built is unwound) */
#define DSVG_ERR_RC          (-6)   /* device-resident rate control: the host's replay of a picture disagrees with what the device chose */

#define DSVG_FRAME_BORDER 64        /* DSV_FRAME_BORDER dsv_internal.h:37 */
#define DSVG_MAX_PYRAMID  5         /* DSV_MAX_PYRAMID_LEVELS dsv_encoder.h:35 */

typedef struct { int width, height, subsamp, fps_num, fps_den, aspect_num, aspect_den; } dsvg_meta; /* DSV_META */
typedef struct { uint8_t *data; int len, format, stride, w, h, hs, vs; } dsvg_plane;             /* DSV_PLANE */
typedef struct { int32_t *data; int width, height; } dsvg_coefs;                                 /* DSV_COEFS */
typedef struct { uint8_t *alloc; dsvg_plane planes[3]; int refcount, format, width, height, border; } dsvg_frame; /* DSV_FRAME */
typedef struct {                                                                                 /* DSV_MV */
    union { struct { int16_t x, y; } mv; int32_t all; } u;
    uint8_t mode, submask, lo_var, lo_tex, high_detail;
} dsvg_mv;
typedef struct { dsvg_meta *vidmeta; int is_ref, has_ref, blk_w, blk_h, nblocks_h, nblocks_v; } dsvg_params;     /* DSV_PARAMS */
typedef struct { dsvg_params *params; unsigned char *stable_blocks; unsigned char cur_plane, isP; } dsvg_stability; /* DSV_STABILITY */
typedef struct { uint8_t *start; unsigned pos; } dsvg_bs;                                        /* DSV_BS */
typedef struct {                                                                                 /* DSV_HME */
    dsvg_params *params;
    dsvg_frame *src[DSVG_MAX_PYRAMID + 1];
    dsvg_frame *ref[DSVG_MAX_PYRAMID + 1];
    dsvg_mv *mvf[DSVG_MAX_PYRAMID + 1];
    int levels;
} dsvg_hme;

/* ---------------------------------------------------------------------------------------------
 * library
 * ------------------------------------------------------------------------------------------- */
const char *dsvg_last_error(void);          /* thread-local description of the last failure */
int dsvg_device_count(void);                /* number of HIP devices (0 = none) */
int dsvg_set_device(int device);            /* device used by the operator-level calls (default 0) */
/* the host <-> device link as the pipeline's own copies use it: hipHostMalloc'd memory first touched by the calling thread, `reps` asynchronous
 * copies of `bytes` each way on a stream of their own, timed by HIP events.  gbs[0] = host -> device, gbs[1] = device -> host, GB/s.  A
 * diagnostic (bench.py's `link`, shard.py's choice of the NUMA node to run on): call it from a short-lived process of the affinity in question. */
int dsvg_link_probe(int device, size_t bytes, int reps, double gbs[2]);

/* ---------------------------------------------------------------------------------------------
 * 1. operator level -- twins of the reference operator API
 * ------------------------------------------------------------------------------------------- */
/* dsv_fwd_sbt  sbt.c:630   */ int dsvg_op_fwd_sbt(const dsvg_plane *src, dsvg_coefs *dst, int isP);
/* dsv_inv_sbt  sbt.c:654   */ int dsvg_op_inv_sbt(dsvg_plane *dst, dsvg_coefs *src, int q, int isP, int c);
/* dsv_encode_plane hzcc.c:449 */ int dsvg_op_encode_plane(dsvg_bs *bs, dsvg_coefs *src, int q, const dsvg_stability *stab);
/* dsv_decode_plane hzcc.c:479 */ int dsvg_op_decode_plane(uint8_t *in, unsigned len, dsvg_coefs *dst, int q, const dsvg_stability *stab);
/* dsv_sub_pred bmc.c:318   */ int dsvg_op_sub_pred(const dsvg_mv *mv, const dsvg_params *p, dsvg_frame *dif, dsvg_frame *inp, const dsvg_frame *ref);
/* dsv_add_pred bmc.c:333   */ int dsvg_op_add_pred(const dsvg_mv *mv, const dsvg_params *p, dsvg_frame *dif, dsvg_frame *out, const dsvg_frame *ref);
/* dsv_frame_add bmc.c:304  */ int dsvg_op_frame_add(dsvg_frame *dst, const dsvg_frame *src);
/* dsv_hme      hme.c:730   */ int dsvg_op_hme(dsvg_hme *hme, int *intra_pct); /* mvf[0..levels] from dsv_alloc: the caller frees them with dsv_free, as dsv_encoder.c:239-244 does */
/* the allocator behind memory the operator calls hand to their caller (the motion fields above).  Default (or NULLs): this library's dsv_alloc /
 * dsv_free.  A build that keeps the reference's own dsv.c passes ITS pair, so that its dsv_free -- which steps back over a 16-byte header when
 * DSV_MEMORY_STATS is on, dsv.c:41-66 -- gets blocks its dsv_alloc made (oracle/opswap_shim.c does). */
void dsvg_set_allocator(void *(*alloc_fn)(int), void (*free_fn)(void *));
/* dsv_extend_frame frame.c:263 */ int dsvg_op_extend_frame(dsvg_frame *f);
/* dsv_extend_frame_luma frame.c:297 */ int dsvg_op_extend_frame_luma(dsvg_frame *f);
/* dsv_ds2x_frame_luma frame.c:240 */ int dsvg_op_ds2x_frame_luma(dsvg_frame *dst, const dsvg_frame *src);
/* dsv_frame_avg_luma frame.c:223 */ int dsvg_op_frame_avg_luma(const dsvg_frame *f, int *avg);
/* dsv_get_quant hzcc.c:77, dsv_lb2 hzcc.c:437: host scalars */
int dsvg_get_quant(int q, int isP, int level);
int dsvg_lb2(unsigned n);

/* ---------------------------------------------------------------------------------------------
 * 2. pipeline level -- device-resident, batched
 * ------------------------------------------------------------------------------------------- */
typedef struct dsvg_ctx dsvg_ctx;

typedef struct {
    int width, height, subsamp;
    int blk_w, blk_h, nblocks_h, nblocks_v;   /* dsv_encoder.c:556-595 */
    int pyramid_levels;                       /* dsv_encoder.c:602-613 (after auto) */
    size_t frame_bytes;                       /* tightly packed planar input frame */
    size_t plane_out_cap[3];                  /* capacity of one packed plane payload */
    size_t frame_alloc_bytes;                 /* one frame in the reference layout, borders included (dsvg_download_recon_raw) */
} dsvg_geom;

/* n_src_slots source frames (padded + pyramid) and n_recon_slots reconstructions stay resident;
 * max_jobs = widest batch handed to dsvg_code_pictures in one call (sizes the per-step work
 * buffers); out_slots = number of coded pictures whose packed planes stay resident until fetched
 * (>= max_jobs; a whole GOP batch = streams x frames can be enqueued without a host sync) and the
 * largest number of frame pairs one dsvg_analyse call may carry. */
int dsvg_ctx_create(dsvg_ctx **out, int device, int width, int height, int subsamp,
                    int pyramid_levels, int n_src_slots, int n_recon_slots, int max_jobs, int out_slots);
/* the same with the block size given (0, 0 = the encoder's rule for the frame size, dsv_encoder.c:557-592): decoders take it
 * from their stream (dsv_decoder.c:335-360); multiples of 4 in 16..64 */
int dsvg_ctx_create_blk(dsvg_ctx **out, int device, int width, int height, int subsamp,
                        int pyramid_levels, int n_src_slots, int n_recon_slots, int max_jobs, int out_slots, int blk_w, int blk_h);
void dsvg_ctx_destroy(dsvg_ctx *ctx);
int dsvg_ctx_geom(const dsvg_ctx *ctx, dsvg_geom *g);
int dsvg_ctx_sync(dsvg_ctx *ctx);
/* Coding streams: dsvg_code_batch shares the pictures of every frame step out over n HIP streams (default 2, or
 * DSV1_CODE_STREAMS), so one group's small latency-bound kernels run under another group's large ones.  n >= 1 sets
 * it (takes effect at the next batch), n = 0 only queries; returns the previous value. */
int dsvg_ctx_code_streams(dsvg_ctx *ctx, int n);
/* how many of the context's four streams (coding, analysis, second coding, fetch) were placed on hardware queues of
 * their own by the probe at creation (4 = all apart; 0 = probe switched off with DSV1_NO_STREAM_PROBE) */
int dsvg_ctx_streams_apart(const dsvg_ctx *ctx);
/* where a throughput context's host-to-device copy stream sits: 0 = a hardware queue of its own (2 / 3: as a stream of the lowest / highest
 * priority, whose queues the runtime keeps apart from the plain streams' four), 1 = it shares the analysis stream's queue, -1 = wherever the
 * runtime put it (small contexts, probe off) */
int dsvg_ctx_copy_queue(const dsvg_ctx *ctx);
/* The first coding hipStream_t (operator-style callers: dsvg_download_recon and the host-output dsvg_pack_recons run on it).  ORDERING: work put on
 * the handle after this call runs behind everything the context has enqueued so far, including a device-output packing pass that dsvg_pack_recons
 * placed on the second coding stream (round 5): the call itself makes the first stream wait for that pass.  A handle fetched BEFORE a later
 * dsvg_pack_recons says nothing about that later pass -- ask again after it (cheap), or use dsvg_ctx_join.  NULL on error. */
void *dsvg_ctx_stream(dsvg_ctx *ctx);
/* make `stream` (a hipStream_t of the caller's on the context's device; NULL or the first coding stream itself: only the join below) wait for
 * everything the context has enqueued so far on its first coding stream and for a pending device-output packing pass: what a consumer of
 * dsvg_pack_recons(.., out_on_device=1) calls before it reads yuv_out in stream order, without a whole-context dsvg_ctx_sync. */
int dsvg_ctx_join(dsvg_ctx *ctx, void *stream);
/* Sparse P pictures: tiles of the fused inverse transform (128x64 pixels) counted since the previous call --
 * out[0], out[1] = tiles that took the general path (luma, chroma), out[2], out[3] = tiles found empty (no detail symbol,
 * LL3 zero: reconstruction = prediction, nothing computed).  Counting is off until a call with enable != 0 and stops
 * again with enable = 0 (one atomic per tile).  Syncs the context and clears the counters.  bench.py uses it to price
 * the kernel at the bytes it really moved. */
int dsvg_ctx_tile_stats(dsvg_ctx *ctx, unsigned long long out[4], int enable);
/* the same plus what the symbol fetches of the sparse inverse amount to (round 4: bench.py prices the kernels at ALL the bytes they
 * must move): out[4] = 8x8 luma patches of computed tiles whose flag was up (their 96 bytes of level-1 symbols were fetched by
 * k_inv_p_tile), out[5] = chroma patches with a flag (k_inv_patch_c fetched their 126 bytes of symbols of the three levels),
 * out[6] = chroma patches without a flag that were still rewritten (non-zero LL3 residual, or prediction not in place: 64 bytes
 * in, 64 out), out[7] = bytes of reconstruction border written by the fused inverse kernels (round 6; tallied on the host from the jobs' extents). */
int dsvg_ctx_tile_stats2(dsvg_ctx *ctx, unsigned long long out[8], int enable);

/* device memory helpers for callers that keep the raw clip in HBM (bench.py) */
int dsvg_dev_alloc(dsvg_ctx *ctx, void **dptr, size_t bytes);
int dsvg_dev_free(dsvg_ctx *ctx, void *dptr);
int dsvg_dev_upload(dsvg_ctx *ctx, void *dptr, const void *src, size_t bytes);
int dsvg_dev_download(dsvg_ctx *ctx, void *dst, const void *dptr, size_t bytes);   /* synchronous */
/* Host-resident input (dsv_main.c:394-421 reads each frame from the .yuv into host memory).  dsvg_host_alloc gives
 * pinned memory; dsvg_ingest_begin starts the upload of `bytes` of packed frames on a copy stream of its own into one
 * of two device buffers owned by the context and returns that buffer: a following dsvg_load_frames_map on it waits for
 * the copy on the device, not on the host.  Asynchronous for pinned memory (the caller keeps yuv_host unchanged until
 * that load has run: after a dsvg_get_luma_sums / dsvg_analyse / dsvg_ctx_sync that follows it); pageable memory works
 * and is staged by the runtime inside the call.  At most two ingests may be outstanding. */
int dsvg_host_alloc(dsvg_ctx *ctx, void **hptr, size_t bytes);
int dsvg_host_free(dsvg_ctx *ctx, void *hptr);
/* the same without a context (a buffer that outlives the context it was allocated through: the decoder's pinned frame pool) */
int dsvg_host_free_on(int device, void *hptr);
/* the raw allocation of reconstruction slot `recon_slot` -- the reference's frame layout, borders as the device left them, dsvg_geom.frame_alloc_bytes
 * bytes -- into host memory (pinned: one asynchronous copy) behind the decoding work on the coding stream; waits for the copy, for nothing else */
int dsvg_download_recon_frame(dsvg_ctx *ctx, int recon_slot, void *raw_out, size_t bytes);
int dsvg_ingest_begin(dsvg_ctx *ctx, const void *yuv_host, size_t bytes, void **dptr);
/* The same for a clip that arrives piece by piece (dsv_enc takes a frame per call): dsvg_ingest_open reserves the next of the
 * two buffers for `bytes` bytes and returns it; dsvg_ingest_part queues the upload of one piece (host memory, pinned for an
 * asynchronous copy) to dptr + offset as soon as the caller has it.  A load from the buffer waits, on the device, for the
 * parts queued before it. */
int dsvg_ingest_open(dsvg_ctx *ctx, size_t bytes, void **dptr);
int dsvg_ingest_part(dsvg_ctx *ctx, void *dptr, size_t offset, const void *host, size_t bytes);

/* Tightly packed planar frames -> resident source slots [first_slot, first_slot+n):
 * copy into the bordered reference layout, replicate borders (dsv_clone_frame/dsv_extend_frame),
 * build the luma pyramid (mk_pyramid dsv_encoder.c:194-217) and the smallest level's mean luma
 * (check_scene_change dsv_encoder.c:538-554).  yuv may be a host or a device pointer.  Async. */
int dsvg_load_frames(dsvg_ctx *ctx, int first_slot, int n, const void *yuv, int yuv_on_device, int with_pyramid);
/* same, frame i read from yuv + i*frame_pitch bytes (device pointer only) */
int dsvg_load_frames_strided(dsvg_ctx *ctx, int first_slot, int n, const void *yuv_dev, size_t frame_pitch, int with_pyramid);
/* same, frame i (at yuv_dev + i*frame_pitch) goes to source slot slots[i]: one launch set for a whole batch */
int dsvg_load_frames_map(dsvg_ctx *ctx, int n, const int *slots, const void *yuv_dev, size_t frame_pitch, int with_pyramid);
/* same with chroma_in_place (one flag per frame, or NULL): a flagged frame's chroma planes are not copied -- the forward
 * transform and the motion search's chroma test read them from the caller's clip; only its luma gets the bordered copy and
 * the pyramid (what a frame needs as a motion search REFERENCE; frame.c:122-164,240-327, hme.c:667-681).  The clip must stay
 * unchanged until the pictures coded from those slots have been fetched AND the frame that follows each of them in its stream
 * has been analysed.  Needs even chroma planes a multiple of 8 wide and a 16-byte aligned clip; otherwise (or with
 * DSV1_NO_CHROMA_IN_PLACE) the frames are copied whole.  dsvg_ctx_chroma_in_place_frames: how many frames took it (tests).
 * Round 5, luma in place: where the geometry allows (luma a multiple of 16 wide and 4 high, two or more pyramid levels, a picture
 * wider and taller than twice the ring below + 64) a flagged frame's LUMA stays in the clip as well -- same contract, nothing
 * new for the caller.  The forward transforms read it there, and so does the level-0 motion search for every block that cannot
 * leave the picture; of the bordered copy only a ring is written (a block + twice the search's reach in from each edge, plus
 * the border), which blocks near an edge read.  The pyramid levels are built as ever.  DSV1_NO_LUMA_IN_PLACE switches it
 * off; dsvg_ctx_luma_in_place_frames counts the frames that took it. */
int dsvg_load_frames_map_ex(dsvg_ctx *ctx, int n, const int *slots, const void *yuv_dev, size_t frame_pitch, int with_pyramid,
                            const unsigned char *chroma_in_place);
long dsvg_ctx_chroma_in_place_frames(const dsvg_ctx *ctx);
long dsvg_ctx_luma_in_place_frames(const dsvg_ctx *ctx);
int dsvg_get_luma_sums(dsvg_ctx *ctx, int first_slot, int n, unsigned *sums_out);   /* raw sums, syncs */
int dsvg_get_avg_luma(dsvg_ctx *ctx, int first_slot, int n, int *avg_out);           /* syncs */

/* Hierarchical motion estimation for npairs (current, reference) source-slot pairs
 * (motion_est dsv_encoder.c:219-254 -> dsv_hme).  mvs_out: npairs * nblocks dsvg_mv on the host. Syncs. */
int dsvg_analyse(dsvg_ctx *ctx, int npairs, const int *cur_slots, const int *ref_slots, dsvg_mv *mvs_out);

typedef struct {
    int src_slot;            /* source frame to code */
    int ref_recon_slot;      /* reconstruction used for prediction, -1 for an I picture */
    int recon_slot;          /* where to keep this picture's reconstruction, -1 = do not keep.  May equal ref_recon_slot
                              * (updated in place through a separate prediction frame); a DIFFERENT slot is faster: the
                              * prediction is then written straight into it and only tiles with a residual are touched again */
    int quant;               /* frame quantiser (quality2quant dsv_encoder.c:165) */
    const dsvg_mv *mvs;      /* host, nblocks entries (P pictures) */
    const unsigned char *stable_blocks; /* host, nblocks entries (encode_stable_blocks output) */
    int out_slot;            /* where the packed planes wait for dsvg_fetch_pictures */
    int no_intra_blocks;     /* P pictures: 1 = the caller knows that no block of mvs has mode != 0 (saves a scan) */
    int has_reach;           /* P pictures: 1 = mv_reach holds min(mv.x >> 1), max(mv.x >> 1), min(mv.y >> 1), max(mv.y >> 1) over the
                              * inter blocks of mvs, each taken with 0 (saves a pass over the vectors: see border_hint) */
    short mv_reach[4];
    int border_hint;         /* 1 = no picture coded by a LATER call predicts from this reconstruction (the caller knows the
                              * next picture of the stream: it is in this call, or it starts a GOP).  The replicated border
                              * (dsv_extend_frame) is then written only as far as the motion vectors of the pictures of this
                              * call that predict from it reach -- possibly not at all.  0 = unknown: if the slot is not
                              * rewritten within the call, the whole 64-pixel border is written */
} dsvg_pic_job;

typedef struct {
    int32_t dc[3];           /* unquantised DC (coefficient [0]) of each plane */
    uint32_t nruns[3];       /* number of (run,value) pairs */
    uint32_t nbytes[3];      /* payload length in bytes (bit count rounded up) */
    const uint8_t *payload[3]; /* host (pinned) pointers, valid until the next dsvg_fetch_pictures */
    int32_t rc_quant;        /* pictures coded by dsvg_code_batch_rc: the frame quantiser the device's rate control chose ... */
    uint32_t rc_pkt_len;     /* ... and the bytes of the picture packet as it computed them (0 / 0 for other pictures) */
} dsvg_pic_out;

/* Residual coding of njobs pictures in one batch: frame copy, dsv_sub_pred, then per plane
 * dsv_fwd_sbt / dsv_encode_plane (quantise+dequantise+pack) / dsv_inv_sbt, dsv_frame_add and the
 * extended reconstruction (encode_one_frame dsv_encoder.c:657-674, encode_picture :518-526).
 * Enqueues only; results are collected by dsvg_fetch_pictures (which syncs). */
int dsvg_code_pictures(dsvg_ctx *ctx, int njobs, const dsvg_pic_job *jobs);
/* nsteps consecutive frame steps of njobs pictures each (jobs[step*njobs + j]) in one call: one upload of
 * all tables, kernel chains back to back; step k+1 may predict from the reconstructions of step k.  The out
 * slots of the call must form one contiguous block.  With several coding streams (dsvg_ctx_code_streams) the jobs of a
 * step are shared out by position; a call in which a reconstruction slot would be written and read (or written twice) by
 * jobs of different shares is run on one stream instead -- always correct, fastest when stream s keeps position s. */
int dsvg_code_batch(dsvg_ctx *ctx, int nsteps, int njobs, const dsvg_pic_job *jobs);
/* Device-resident average-bitrate rate control (round 4; quality2quant dsv_encoder.c:70-168 + the statistics of dsv_enc
 * :816-848 as include/dsvg_rc.h, run by k_rc on the device): the same as dsvg_code_batch for streams whose every quantiser
 * depends on the size of the packet before -- jobs[].quant is ignored, rc[step*njobs + j] says which stream's state
 * (rc_slot, < max(n_recon_slots, max_jobs)) picture j of a step belongs to, how many bytes of its packet precede the quantiser
 * field (prefix_len: header + side information, known before the picture is coded) and quality2quant's forced_intra.  A stream
 * keeps its position j from step to step.  The state lives on the device from call to call: dsvg_rc_upload seeds it (once, or when
 * the caller changed the stream's parameters), dsvg_rc_download reads it back (syncs).  dsvg_pic_out.rc_quant / rc_pkt_len return
 * what the device chose and computed for every picture. */
typedef struct { int rc_slot; int prefix_len; int forced_intra; } dsvg_rc_job;
struct dsvg_rc_state;
int dsvg_code_batch_rc(dsvg_ctx *ctx, int nsteps, int njobs, const dsvg_pic_job *jobs, const dsvg_rc_job *rc);
int dsvg_rc_upload(dsvg_ctx *ctx, int first_slot, int n, const struct dsvg_rc_state *states);
int dsvg_rc_download(dsvg_ctx *ctx, int first_slot, int n, struct dsvg_rc_state *states);
/* the PARAMETER fields of the streams' state only (bitrate, frame rate, nudge, max_q_step, quality bounds: everything from
 * dsvg_rc_state.bitrate on) -- what a caller may change between calls (the reference reads them per frame, dsv_encoder.c:84-165);
 * ordered on the coding stream: pictures enqueued before keep the old parameters, pictures enqueued after see the new ones */
int dsvg_rc_set_params(dsvg_ctx *ctx, int first_slot, int n, const struct dsvg_rc_state *states);
int dsvg_fetch_pictures(dsvg_ctx *ctx, int n, const int *out_slots, dsvg_pic_out *outs);
/* The same with the device-to-host copy cut into `nchunks` pieces that end on multiples of `align` pictures: as soon as a
 * piece has arrived, cb(arg, first, count) is called for its pictures (outs[first .. first+count) are valid then) while the
 * later pieces are still being copied -- the caller's packet assembly overlaps the link. */
typedef void (*dsvg_fetch_cb)(void *arg, int first, int count);
int dsvg_fetch_pictures_cb(dsvg_ctx *ctx, int n, const int *out_slots, dsvg_pic_out *outs, int nchunks, int align, dsvg_fetch_cb cb, void *arg);
int dsvg_download_recon(dsvg_ctx *ctx, int recon_slot, uint8_t *yuv_out);            /* syncs */
/* the first `bytes` bytes of the slot's whole frame allocation in the reference layout (dsv_mk_frame frame.c:63-120:
 * Y,U,V back to back, 64-px replicated borders): what the next picture's motion compensation reads.  Syncs. */
int dsvg_download_recon_raw(dsvg_ctx *ctx, int recon_slot, uint8_t *raw_out, size_t bytes);
/* The encoder writes a reconstruction's border only as far as the pictures that predict from it read it (dsv_pic_job.border_hint,
 * dsvg_code_batch).  dsvg_recon_border: the extents the slot's last encoder job wrote -- pixels left, right, rows above, below
 * the picture, luma [0..3] and chroma [4..7] (columns are written in units of 16, rows in units of 8, at most 64).
 * dsvg_download_recon_asis: the allocation as it is (dsvg_download_recon_raw completes the border first).
 * dsvg_extend_recon: complete the border of a slot now (asynchronous, coding stream). */
int dsvg_recon_border(dsvg_ctx *ctx, int recon_slot, short ext_out[8]);
int dsvg_download_recon_asis(dsvg_ctx *ctx, int recon_slot, uint8_t *raw_out, size_t bytes);
int dsvg_extend_recon(dsvg_ctx *ctx, int recon_slot);
/* a frame allocation in the reference layout (as dsvg_download_recon_raw returns it) into a reconstruction slot: how a decoder
 * carries its reference picture over when the stream changes its block size and a new context is needed.  Syncs. */
int dsvg_upload_recon_raw(dsvg_ctx *ctx, int recon_slot, const uint8_t *raw, size_t bytes);
/* n reconstruction slots -> tightly packed planar frames, frame i at yuv_out + i*out_pitch.  Device output: enqueued
 * without a sync: on the first coding stream, or -- n >= 4, decoder contexts -- on the second one beside the next call's entropy decoding.
 * Before reading it: dsvg_ctx_sync, or order the consumer behind the pass with dsvg_ctx_join(ctx, consumer_stream) / a dsvg_ctx_stream()
 * handle fetched AFTER this call.  Host output: copied back and synchronised. */
int dsvg_pack_recons(dsvg_ctx *ctx, int n, const int *recon_slots, void *yuv_out, size_t out_pitch, int out_on_device);

/* Decoder side: coefficient (run,value) pairs parsed on the host are scattered + dequantised,
 * inverse transformed and motion compensated on the device (dsv_dec dsv_decoder.c:379-436). */
typedef struct {
    int ref_recon_slot;      /* -1 for an I picture */
    int recon_slot;          /* output picture slot */
    int quant;
    const dsvg_mv *mvs;
    const unsigned char *stable_blocks;
    const uint8_t *plane_data[3];   /* host: bytes following the 32-bit plane length */
    uint32_t plane_len[3];
} dsvg_dec_job;
int dsvg_decode_pictures(dsvg_ctx *ctx, int njobs, const dsvg_dec_job *jobs);
/* P pictures are decoded through int16 symbol planes (the encoder's fused inverse).  A picture those cannot hold exactly -- a
 * symbol beyond int16 (not producible from 8-bit video, but parsable), or a cell shared by two scan regions that must keep
 * the earlier region's value (hzcc.c:295-435) -- is detected on the device and its call decoded again from int32 coefficients
 * before anything reads or predicts from the result (the next decode call, dsvg_ctx_sync, the downloads, a host-side pack).
 * dsvg_ctx_decoder_redone: how many calls took that second pass (tests). */
long dsvg_ctx_decoder_redone(const dsvg_ctx *ctx);

/* Kernel timing hook for bench.py: every launch of the kernels selected by `kernel_mask` (bit i =
 * kernel id i, names from dsvg_prof_kernel_name) is bracketed by HIP events on the pipeline stream.
 * dsvg_prof_get returns the summed event time (ms), launch count and the ALGORITHMIC bytes those
 * launches moved (compulsory traffic: each input read once, each output written once; the per-sample
 * figures are listed in DESIGN.md) since the last reset. */
/* two marks on the first coding stream and the GPU-side time between them (HIP events): dsvg_ctx_mark(ctx, 0) when a timed region
 * starts, dsvg_ctx_mark(ctx, 1) behind its last enqueued coding work; dsvg_ctx_mark_ms after a sync.  bench.py prints it beside its
 * host clock so that the headline can be corroborated from the device side. */
int dsvg_ctx_mark(dsvg_ctx *ctx, int which);
int dsvg_ctx_mark_ms(dsvg_ctx *ctx, float *ms);
/* Where a step's time goes, from the pipeline's own streams (round 6): dsvg_ctx_timeline(ctx, 1) starts collecting timing marks at the boundaries of
 * clip upload / frame load + pyramid / motion search / table uploads / coding (both coding streams) / fetch (one HIP event per mark, ten per batch;
 * 0 stops and clears).  dsvg_ctx_timeline_get (context synchronised) sums them: out[0] = coding phases seen, [1] = device ms first mark .. last mark,
 * [2] upload, [3] load, [4] motion search, [5] table uploads, [6] coding on the first stream, [7] on the second, [8] fetch (gather + copies) --
 * device ms summed over the phases --, [9] = ms during which none of load / search / tables / coding was in flight (the chip waits for the host),
 * [10] = the same inside [first coding start, last coding end], [11] = ms of coding overlapped by a load / search phase. */
int dsvg_ctx_timeline(dsvg_ctx *ctx, int on);
int dsvg_ctx_timeline_get(dsvg_ctx *ctx, double out[12]);
/* the host's side of dsvg_fetch_pictures(_cb) per call since the last reset: out[0] = ms waiting for the coding calls behind the slots, [1] = ms for
 * the plane summaries (a small device-to-host round trip), [2] = ms for gather + the copy in pieces + the callback's work, [3] = bytes copied, [4] = calls */
int dsvg_ctx_fetch_prof(dsvg_ctx *ctx, double out[5], int reset);
int dsvg_prof_kernels(void);
const char *dsvg_prof_kernel_name(int kid);
int dsvg_prof_enable(dsvg_ctx *ctx, unsigned long long kernel_mask);
int dsvg_prof_reset(dsvg_ctx *ctx);
int dsvg_prof_get(dsvg_ctx *ctx, int kid, double *ms, long *launches, double *alg_bytes);

#ifdef __cplusplus
}
#endif
#endif
