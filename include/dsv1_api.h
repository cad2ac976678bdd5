/*
 * dsv1_api.h -- session-level C API of libdsv1_mi355x.so: the drop-in surface behind dsv_main.c.
 *
 * Every symbol the reference CLI binds (dsv_main.c:440-557,652-717) is exported with the same name,
 * argument meaning, ownership and return codes, and the structs callers poke (DSV_ENCODER,
 * DSV_DECODER, DSV_META, DSV_FRAME, DSV_BUF) keep the reference's field order and sizes
 * (dsv.h:86-198, dsv_encoder.h:58-110, dsv_decoder.h:27-43).  The per-frame arithmetic runs in the HIP
 * kernels behind include/dsvg.h; this layer (plain C) keeps GOP / scene-change / rate-control /
 * stability logic, side-info coding and packet framing on the host.
 *
 * Extensions (names of our own): dsv1_batch_* encodes many independent closed GOPs per call with
 * frames resident in HBM -- the throughput path used by bench.py and by GOP sharding across GPUs.
 */
#ifndef DSV1_API_H
#define DSV1_API_H

#include <limits.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "dsvg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* packet types (dsv.h:34-43) */
#define DSV_PT_META 0x00
#define DSV_PT_PIC  0x04
#define DSV_PT_EOS  0x10
#define DSV_PACKET_HDR_SIZE 14
#define DSV_PACKET_TYPE_OFFSET 5
#define DSV_PACKET_PREV_OFFSET 6
#define DSV_PACKET_NEXT_OFFSET 10

#define DSV_SUBSAMP_444 0x0
#define DSV_SUBSAMP_422 0x4
#define DSV_SUBSAMP_420 0x5
#define DSV_SUBSAMP_411 0x8

#define DSV_MAX_QUALITY 2047
#define DSV_QUALITY_PERCENT(p) (DSV_MAX_QUALITY * (p) / 100)
#define DSV_GOP_INTRA 0
#define DSV_GOP_INF INT_MAX
#define DSV_ENC_NUM_BUFS 0x03
#define DSV_ENC_FINISHED 0x04
#define DSV_RATE_CONTROL_CRF 0
#define DSV_RATE_CONTROL_ABR 1
#define DSV_MAX_PYRAMID_LEVELS 5

typedef uint32_t DSV_FNUM;
typedef dsvg_meta   DSV_META;
typedef dsvg_plane  DSV_PLANE;
typedef dsvg_frame  DSV_FRAME;
typedef dsvg_mv     DSV_MV;
typedef dsvg_params DSV_PARAMS;
typedef struct { unsigned char *data; unsigned len; } DSV_BUF;

typedef struct {                 /* same fields, order and sizes as dsv_encoder.h:58-110 */
    int quality;
    int gop;
    int do_scd;
    int rc_mode;
    int rc_high_motion_nudge;
    unsigned bitrate;
    int max_q_step;
    int min_quality;
    int max_quality;
    int min_I_frame_quality;
    int intra_pct_thresh;
    int scene_change_delta;
    unsigned stable_refresh;
    int pyramid_levels;
    /* internal state (kept in the public struct by the reference, so kept here) */
    unsigned rc_quant;
    unsigned bpf_total;
    unsigned bpf_reset;
    int bpf_avg;
    int total_P_frame_q;
    int avg_P_frame_q;
    int last_P_frame_over;
    int back_into_range;
    DSV_FNUM next_fnum;
    void *ref;                   /* reference: DSV_ENCDATA*; here: opaque device session handle */
    DSV_META vidmeta;
    int prev_link;
    int force_metadata;
    struct DSV_STAB_ACC { signed x : 16; signed y : 16; } *stability;
    unsigned refresh_ctr;
    unsigned char *stable_blocks;
    DSV_FNUM prev_gop;
    int prev_avg_luma;
} DSV_ENCODER;

typedef struct {                 /* dsv_decoder.h:35-43 */
    DSV_META vidmeta;
    void *ref;                   /* reference: DSV_IMAGE*; here: opaque device session handle */
    int draw_info;               /* accepted and ignored: debug overlays are out of scope */
    int got_metadata;
} DSV_DECODER;

#define DSV_DEC_OK        0
#define DSV_DEC_ERROR     1
#define DSV_DEC_EOS       2
#define DSV_DEC_GOT_META  3
#define DSV_DEC_NEED_NEXT 4

/* ---- encoder (dsv_encoder.h:112-121) ---- */
void dsv_enc_init(DSV_ENCODER *enc);
void dsv_enc_free(DSV_ENCODER *enc);
void dsv_enc_set_metadata(DSV_ENCODER *enc, DSV_META *md);
void dsv_enc_force_metadata(DSV_ENCODER *enc);
void dsv_enc_start(DSV_ENCODER *enc);
/* dsv_enc takes ownership of frame (its pixels are copied before the call returns).  CRF streams are coded with a LOOKAHEAD
 * (DSV1_ENC_LOOKAHEAD frames, default 16 GOPs): the call returns 0 buffers while the lookahead fills, then up to two packets
 * per call (metadata + picture, as the reference), always whole packets in stream order; dsv_enc_end_of_stream returns every
 * packet still owed followed by the EOS packet in bufs[0] (ONE buffer holding several packets: a caller that needs one
 * packet per DSV_BUF splits it on the packets' next-link words, or sets DSV1_ENC_PIPELINE=0 for the frame-synchronous
 * behaviour: one picture per call, same bytes).  Changes to the encoder's public fields between calls (quality, bitrate,
 * min_ / max_quality, min_I_frame_quality, max_q_step, rc_high_motion_nudge, dsv_enc_force_metadata) apply from the frame of
 * the NEXT call on, exactly as in the reference (it reads them when it codes a frame, dsv_encoder.c:84-165,794-803): the frames
 * gathered before the change are sent to the device as a short batch first (round 5).  dsv_enc_free without
 * dsv_enc_end_of_stream drops the frames still buffered (and logs it).  ABR streams (their pictures are coded one after the
 * other: every packet's size feeds the next quantiser) gather 32 frames (DSV1_ENC_LOOKAHEAD) for a common analysis pass: the same
 * contract with a shorter lookahead, the same bytes as the frame-synchronous encoder.
 * THE REFERENCE'S PACKET CONTRACT (dsv_encoder.c:766-810: one packet per DSV_BUF, at most two per dsv_enc call, exactly one EOS
 * packet from dsv_enc_end_of_stream) for a library caller that cannot split a buffer: before dsv_enc_end_of_stream, call
 *     while ((n = dsv_enc(enc, NULL, bufs)) > 0) { ...write bufs[0 .. n-1]... }
 * -- a FLUSH CALL (frame == NULL, an extension: the reference would crash on it).  The first one codes and collects everything
 * the session still holds; each returns up to two whole packets, one per DSV_BUF, in stream order; 0 = drained.
 * dsv_enc_end_of_stream then returns the 14-byte EOS packet alone.
 * OR ask for the reference's contract outright: dsv1_enc_set_strict_packets(enc, 1) after dsv_enc_init / dsv_enc_start and BEFORE the first
 * dsv_enc (an extension, round 6; an error once the session exists): the session then runs frame-synchronously -- every dsv_enc call codes its
 * frame and returns that frame's packets (metadata + picture, one per DSV_BUF), dsv_enc_end_of_stream returns the EOS packet alone, exactly
 * dsv_encoder.c:766-810 -- at the frame-at-a-time rate instead of the lookahead's. */
int  dsv1_enc_set_strict_packets(DSV_ENCODER *enc, int on);
int  dsv_enc(DSV_ENCODER *enc, DSV_FRAME *frame, DSV_BUF *bufs);
void dsv_enc_end_of_stream(DSV_ENCODER *enc, DSV_BUF *bufs);

/* ---- decoder (dsv_decoder.h:51-59) ---- */
int  dsv_dec(DSV_DECODER *d, DSV_BUF *buf, DSV_FRAME **out, DSV_FNUM *fn); /* frees buf */
DSV_META *dsv_get_metadata(DSV_DECODER *d);
void dsv_dec_free(DSV_DECODER *d);

/* ---- frames, buffers, allocator, logging, file helpers (dsv.h:160-247, util.h:31-34) ---- */
DSV_FRAME *dsv_mk_frame(int format, int width, int height, int border);
DSV_FRAME *dsv_load_planar_frame(int format, void *data, int width, int height);
DSV_FRAME *dsv_frame_ref_inc(DSV_FRAME *frame);
void dsv_frame_ref_dec(DSV_FRAME *frame);
void dsv_mk_buf(DSV_BUF *buf, int size);
void dsv_buf_free(DSV_BUF *buf);
void *dsv_alloc(int size);
/* Extension: while an encoder batch / session is open, dsv_free keeps blocks of 256 KB .. 64 MB (a batch's packet buffers) for the next
 * dsv_alloc instead of returning their pages to the system -- at most DSV1_RECYCLE_MAX_MB (default 1024) megabytes, DSV1_NO_RECYCLE=1:
 * never.  The parked blocks go back to the system when the last batch / session closes (and a block freed after that is freed for good);
 * dsv1_release_parked gives them back at once, dsv1_parked_bytes says how much is parked.  dsv1_recycle_hold(+1 / -1) is the count
 * behind that (a caller that wants parking without an open batch may hold it itself). */
void dsv1_release_parked(void);
size_t dsv1_parked_bytes(void);
int dsv1_recycle_hold(int delta);
/* host phases of dsv1_batch_submit / _collect (round 6): dsv1_host_prof_enable(1) clears and starts the sums, dsv1_host_prof_get returns the
 * number of phases and fills ms per submitted batch for the first n of them (dsv1_host_prof_name(k) says what phase k is); process-wide. */
/* threads (the calling one included) the session layer's parallel loops use in this process: DSV1_HOST_THREADS, else by the cores the process may
 * run on -- affinity mask, the container's CPU quota, the node's ranks (csrc/host/dsv1_util.c: dsv1_host_threads_rule) */
int dsv1_host_threads(void);
void dsv1_host_prof_enable(int on);
int dsv1_host_prof_get(double *ms_per_batch, int n, long *batches);
const char *dsv1_host_prof_name(int k);
void dsv1_debug_fail_alloc_at(int n);        /* tests: the n-th host allocation of the next dsv1_batch_open / dsv1_stream_open fails (0 = off) */
void dsv_free(void *ptr);
void dsv_memory_report(void);
void dsv_set_log_level(int level);
int  dsv_get_log_level(void);
int  dsv_yuv_write(FILE *out, int fno, DSV_PLANE *planes);
int  dsv_yuv_read(FILE *in, int fno, uint8_t *o, int w, int h, int subsamp);
void dsv_movec_pred(DSV_MV *vecs, DSV_PARAMS *p, int x, int y, int *px, int *py);
unsigned estimate_bitrate(int quality, int gop, DSV_META *md);
void conv444to422(DSV_PLANE *srcf, DSV_PLANE *dstf);
void conv422to420(DSV_PLANE *srcf, DSV_PLANE *dstf);

/* ---- extensions: device selection + batched closed-GOP encoding ---- */
void dsv1_set_device(int device);            /* HIP device used by sessions opened afterwards */

typedef struct dsv1_batch dsv1_batch;
/* cfg: a DSV_ENCODER filled like dsv_main.c:463-489 would (dsv_enc_init + fields + vidmeta);
 * nstreams independent streams, frames_per_call frames each per call. */
int  dsv1_batch_open(dsv1_batch **out, const DSV_ENCODER *cfg, int device, int nstreams, int frames_per_call);
/* ONE stream, GOP-parallel (the always-exact scheme): every call takes frames_per_call CONSECUTIVE frames of
 * the stream (yuv: [frame]; a last, shorter call is not supported here -- dsv_enc does that) and codes the chains of pictures
 * between I pictures side by side, max_chains at a time, after replaying the encoder's serial decisions (GOP starts,
 * scene changes, forced-intra pictures, stability flags: all functions of source pixels) on the host.  The stream equals
 * the frame-serial encoder's for every CRF configuration (dsv_encoder.c:345-399,538-552,624-653); ABR is refused.  Used
 * with dsv1_batch_encode / submit / collect / eos / close like a batch of one stream. */
int  dsv1_stream_open(dsv1_batch **out, const DSV_ENCODER *cfg, int device, int frames_per_call, int max_chains);
void dsv1_batch_close(dsv1_batch *b);
void dsv1_batch_set_fnum(dsv1_batch *b, int stream, DSV_FNUM next_fnum);
/* Round 5: reference pictures nobody predicts from are coded without their reconstruction (no inverse transform; the packets are the
 * same -- the reference encoder builds that picture and never reads it, dsv_encoder.c:665-708).  Known exactly inside a call; for a call's
 * last picture when the next frame number starts a GOP, and if dsv1_batch_set_fnum then turns that GOP start into a P picture the next
 * submit codes the dropped picture again with its reconstruction kept (remedied) before it goes on.  Returns how many reconstructions
 * were dropped so far.  DSV1_RECON_ALL=1: every reference picture is reconstructed. */
long dsv1_batch_dropped_recons(const dsv1_batch *b, long *remedied);
/* the same switch as a call (between batches: nothing in flight): on != 0 reconstructs every reference picture from the next submit on */
int  dsv1_batch_recon_all(dsv1_batch *b, int on);
/* stream s's encoder struct (the batch owns it).  Its public parameter fields -- quality, bitrate, min_ / max_quality,
 * min_I_frame_quality, max_q_step, rc_high_motion_nudge; dsv_enc_force_metadata -- may be changed between submits, as a caller of
 * the reference changes them between dsv_enc calls; geometry, GOP structure and rate-control mode may not. */
DSV_ENCODER *dsv1_batch_encoder(dsv1_batch *b, int stream);
/* Encode frames_per_call frames of every stream.  yuv: [stream][frame] tightly packed planar frames,
 * host (yuv_on_device = 0) or device memory (1).  The call has finished with the clip when it returns (a device clip's chroma
 * planes are read in place by the coding kernels, dsvg_load_frames_map_ex, only luma is copied -- and the call returns after
 * the batch has been collected).  For each stream s the packets are appended to out[s] (a growing buffer the
 * caller owns: data = NULL / len = 0 to start; freed with dsv_free).  Returns 0 or a DSVG_ERR_*. */
int  dsv1_batch_encode(dsv1_batch *b, const void *yuv, int yuv_on_device, DSV_BUF *out);
/* Pipelined form (CRF): submit enqueues a batch and returns while its residual coding still runs on
 * the device; collect fetches + assembles the OLDEST submitted batch.  At most two batches may be in
 * flight, so the steady state is  submit(i+1); collect(i);  -- the analysis of batch i+1 (frame load,
 * pyramid, motion estimation: source pixels only) then overlaps the residual coding of batch i on a
 * second HIP stream, and the host packet assembly overlaps both.  ABR streams pipeline the same way since round 4: every
 * quantiser still follows from the size of the packet before, but that chain runs on the device (k_rc, include/dsvg_rc.h) and
 * the streams' rate-control state stays there from call to call; the session layer replays it on the host when it assembles the
 * packets (the encoder structs stay in step) and fails the batch (DSVG_ERR_RC) if the two ever disagree.  The rate-control
 * PARAMETERS (bitrate, quality bounds, max_q_step, rc_high_motion_nudge) are read from every stream's encoder struct
 * (dsv1_batch_encoder) at each submit: a change applies to every picture of the batches submitted after it -- the reference
 * reads them per frame (dsv_encoder.c:84-165), a batch is the unit here; batches already in flight keep what they were
 * submitted with, and so does their host-side replay.  DSV1_ABR_SERIAL=1: rounds 1-3's path -- submit codes frame by frame, assembles into out, and collect
 * only releases the slot. */
/* yuv_on_device for dsv1_batch_submit: 0 = host memory, 1 = device memory, copied whole -- submit has finished with the clip
 * when it returns; DSV1_CLIP_HELD = device memory that the caller keeps UNCHANGED until dsv1_batch_collect of this batch has
 * returned: chroma is then read in place by the coding kernels and only luma is copied (what bench.py times: 21 % less traffic
 * in the load stage).  Opt-in since round 4: a caller that reuses its clip right after submit gets correct streams with 1. */
#define DSV1_CLIP_HELD 2
int  dsv1_batch_submit(dsv1_batch *b, const void *yuv, int yuv_on_device, DSV_BUF *out);
int  dsv1_batch_collect(dsv1_batch *b, DSV_BUF *out);
/* Host-resident input (the .yuv reader of dsv_main.c:394-421): announce the host clip of a coming submit.  Its
 * upload is queued at once on a copy stream and runs under whatever the device is doing; up to two clips may be
 * staged, and dsv1_batch_submit(b, yuv_host, 0, ...) consumes the oldest one (it must be given the same pointer; a
 * submit with nothing staged starts the upload itself).  Asynchronous when yuv_host is pinned (dsvg_host_alloc): the
 * clip must stay unchanged until the collect of that batch returned.  Steady state, uploads back to back:
 *   stage(i+2); submit(i+1); collect(i). */
int  dsv1_batch_stage(dsv1_batch *b, const void *yuv_host);
/* append the end-of-stream packet of stream s */
int  dsv1_batch_eos(dsv1_batch *b, int stream, DSV_BUF *out);
/* Concatenate per-GOP streams (each encoded independently with fnum seeded to its position) into one
 * .dsv: rewrites the prev_link of every picture/EOS packet exactly as a serial encode would
 * (set_link_offsets dsv_encoder.c:171-192) and appends an EOS.  Returns dsv_alloc'd buffer. */
int  dsv1_concat_gops(const DSV_BUF *gops, int ngops, DSV_BUF *out);
void *dsv1_batch_ctx(dsv1_batch *b);          /* the dsvg_ctx* (profiling hooks) */
/* reconstruction slot that holds `stream`'s current reference picture (the encoder's recon_frame, dsv_encoder.c:663-674),
 * for dsvg_download_recon / dsvg_download_recon_raw; -1 if the stream has none yet */
int   dsv1_batch_recon_slot(const dsv1_batch *b, int stream);

/* ---- extension: batched decoding (dsv_dec decodes one picture per call, dsv_decoder.c:286-472) ----
 * nstreams independent streams of one geometry; every call takes ONE packet per stream (packets[s]: not freed, not
 * modified) and decodes all picture packets among them as one device batch.  status[s] = DSV_DEC_OK (a frame was
 * written), DSV_DEC_GOT_META, DSV_DEC_EOS or DSV_DEC_ERROR; fnum[s] = frame number (-1 if none).  The decoded frame
 * of stream s is written tightly packed planar (Y, U, V) at yuv_out + s*out_pitch (out_pitch 0 = frame size), in
 * device memory (asynchronous: dsvg_ctx_sync(dsv1_decbatch_ctx(d)) before reading) or host memory (synchronised).
 * Reference pictures stay resident on the device.  Returns 0 or a DSVG_ERR_*. */
typedef struct dsv1_decbatch dsv1_decbatch;
int  dsv1_decbatch_open(dsv1_decbatch **out, int device, const DSV_META *meta, int nstreams);
int  dsv1_decbatch_decode(dsv1_decbatch *d, const DSV_BUF *packets, void *yuv_out, size_t out_pitch, int out_on_device,
                          int *status, DSV_FNUM *fnum);
void dsv1_decbatch_close(dsv1_decbatch *d);
void *dsv1_decbatch_ctx(dsv1_decbatch *d);      /* (ask again after every decode call: the batch builds a new context when its streams
                                                  * announce another block size while none of them holds a reference picture) */

#ifdef __cplusplus
}
#endif
#endif
