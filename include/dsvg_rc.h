/* dsvg_rc.h -- DSV1 average-bitrate rate control as ONE piece of integer code that the C session layer and the device
 * both compile (round 4: device-resident ABR).
 *
 * Replaces (reference, file:line):
 *   quality2quant            dsv_encoder.c:70-168   -> dsvg_rc_pick
 *   the statistics of dsv_enc dsv_encoder.c:816-848 -> dsvg_rc_after
 *   the size of a picture packet (encode_picture dsv_encoder.c:518-536 + dsv_encode_plane hzcc.c:449-476 framing)
 *                                                   -> dsvg_rc_packet_len
 * The arithmetic types follow the reference's declarations (dsv_encoder.h:58-104: rc_quant / bpf_total / bpf_reset / bitrate
 * unsigned, the rest int) because its mixed signed / unsigned divisions are part of the behaviour.
 *
 * The host includes this header as plain C (static inline); dsvg_dev.hpp defines DSVG_RC_FN as __host__ __device__ before it
 * includes it, so k_rc (k_rc.hip) runs the very same statements after each picture's k_hz_scan and writes the NEXT picture's
 * quantiser tables on the device: a whole batch of an ABR stream is enqueued without a host round trip per picture.  The host
 * replays the same code when it assembles the packets and refuses the batch if the device chose differently. */
#ifndef DSVG_RC_H
#define DSVG_RC_H

#ifndef DSVG_RC_FN
#define DSVG_RC_FN static inline
#endif

#define DSVG_RC_MAX_QUALITY 2047                                   /* DSV_MAX_QUALITY dsv.h:157 ((1 << 11) - 1) */
#define DSVG_RC_PERCENT(pct) (DSVG_RC_MAX_QUALITY * (pct) / 100)   /* DSV_QUALITY_PERCENT dsv.h:158 */
#define DSVG_RC_BPF_RESET 256                                      /* DSV_BPF_RESET dsv_encoder.h:89 */
#define DSVG_RC_CLAMP(v, lo, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))

typedef struct dsvg_rc_state {
    /* state (DSV_ENCODER "used internally") */
    unsigned rc_quant, bpf_total, bpf_reset;
    int bpf_avg, total_P_frame_q, avg_P_frame_q, last_P_frame_over, back_into_range;
    /* parameters (DSV_ENCODER user fields + the metadata's frame rate) */
    unsigned bitrate;
    int fps_num, fps_den;
    int rc_high_motion_nudge, max_q_step, min_quality, max_quality, min_I_frame_quality;
} dsvg_rc_state;

/* quality2quant for an ABR stream: the next picture's quality (kept in rc_quant) -> the frame quantiser (dsv_encoder.c:165) */
DSVG_RC_FN int dsvg_rc_pick(dsvg_rc_state *e, int isP, int forced_intra)
{
    int q = (int)e->rc_quant;
    int fps = (e->fps_num << 5) / e->fps_den, need, bpf, dir, delta, nudged = 0, cap, low_p, minq, ad;
    if (fps == 0) fps = 1;
    need = (int)(((e->bitrate << 5) / (unsigned)fps) >> 3);
    bpf = e->bpf_avg ? e->bpf_avg : need;
    dir = (bpf - need) > 0 ? -1 : 1;
    ad = bpf - need;
    if (ad < 0) ad = -ad;
    delta = (ad << 9) / need;
    if (dir == 1) delta *= 2;
    if (e->rc_high_motion_nudge) {
        if (isP && e->last_P_frame_over) { delta = (delta + 1) * 2; dir = -1; nudged = 1; }
        else if (e->back_into_range)     { delta = (delta + 1) * 2; dir = 1;  nudged = 1; }
    }
    delta = (q * delta) >> 9;
    e->max_q_step = DSVG_RC_CLAMP(e->max_q_step, 1, DSVG_RC_MAX_QUALITY);
    cap = nudged ? e->max_q_step * 16 : e->max_q_step;
    if (delta > cap) delta = cap;
    q += delta * dir;
    low_p = e->avg_P_frame_q - DSVG_RC_PERCENT(4);
    low_p = DSVG_RC_CLAMP(low_p, e->min_quality, e->max_quality);
    minq = isP ? low_p : e->min_I_frame_quality;
    if (forced_intra) {
        if (q < DSVG_RC_PERCENT(60)) q += DSVG_RC_PERCENT(15);
        else if (q < DSVG_RC_PERCENT(70)) q += DSVG_RC_PERCENT(8);
        else if (q < DSVG_RC_PERCENT(75)) q += DSVG_RC_PERCENT(3);
        q = DSVG_RC_CLAMP(q, 0, e->max_quality - DSVG_RC_PERCENT(5));
    }
    q = DSVG_RC_CLAMP(q, minq, e->max_quality);
    q = DSVG_RC_CLAMP(q, 0, DSVG_RC_MAX_QUALITY);
    e->rc_quant = (unsigned)q;
    return DSVG_RC_MAX_QUALITY - ((DSVG_RC_MAX_QUALITY - 5) * q / DSVG_RC_MAX_QUALITY);
}

/* the rate-control statistics after a picture packet of pkt_len bytes (dsv_enc, dsv_encoder.c:816-848) */
DSVG_RC_FN void dsvg_rc_after(dsvg_rc_state *e, int isP, unsigned pkt_len)
{
    e->bpf_total += pkt_len;
    e->bpf_reset++;
    if (isP) {
        unsigned fps, need;
        int under, over;
        e->total_P_frame_q += (int)e->rc_quant;
        e->avg_P_frame_q = (int)((unsigned)e->total_P_frame_q / e->bpf_reset);
        fps = (unsigned)(e->fps_num << 5) / (unsigned)e->fps_den;
        if (fps == 0) fps = 1;
        need = ((e->bitrate << 5) / fps) >> 3;
        under = pkt_len < (need * 3 / 4);
        need = need * 7 / 8;
        over = pkt_len > need;
        e->back_into_range = (e->last_P_frame_over && under);
        e->last_P_frame_over = over;
    } else {
        e->last_P_frame_over = 0;
        e->back_into_range = 0;
    }
    e->bpf_avg = (int)(e->bpf_total / e->bpf_reset);
    if (e->bpf_reset >= DSVG_RC_BPF_RESET) {
        e->bpf_total = (unsigned)e->bpf_avg;
        e->total_P_frame_q = (int)((unsigned)e->total_P_frame_q / e->bpf_reset);
        e->bpf_reset = 1;
    }
}

/* bits of the signed interleaved exp-Golomb code of v (dsv_bs_put_seg bs.c:147-157: UEG of |v|, then a sign bit unless 0) */
DSVG_RC_FN unsigned dsvg_rc_seg_bits(int v)
{
    const unsigned m = (v < 0 ? (unsigned)-v : (unsigned)v) + 1u;
    unsigned k = 0;
    while ((m >> (k + 1)) != 0) k++;                               /* floor(log2(|v| + 1)) */
    return 2u * k + 1u + (m > 1u ? 1u : 0u);
}

/* bytes of a picture packet: the prefix (header, frame number, block sizes, stability and motion side information: byte
 * aligned, known before the picture is coded), the 11-bit frame quantiser, then per plane [32-bit length][SEG(DC)][align]
 * [32-bit run count][payload bytes][0x55] (hzcc.c:449-476), each plane starting on a byte boundary */
DSVG_RC_FN unsigned dsvg_rc_packet_len(unsigned prefix_len, const int dc[3], const unsigned nbytes[3])
{
    unsigned len = prefix_len + 2u;
    int p;
    for (p = 0; p < 3; p++) len += 4u + ((dsvg_rc_seg_bits(dc[p]) + 7u) >> 3) + 4u + nbytes[p] + 1u;
    return len;
}

#endif
