/* opswap_shim.c -- TEST INFRASTRUCTURE (never linked into the product).
 *
 * The reference's OWN session layer on the product's operators (verdict round 5, "operator seam proven only through twins"): this file
 * defines every function of the reference's operator API that its session files call -- dsv_internal.h:94-109 (dsv_get_quant, dsv_fwd_sbt,
 * dsv_inv_sbt, dsv_encode_plane, dsv_decode_plane, dsv_lb2, dsv_sub_pred, dsv_add_pred), dsv_encoder.h:132 (dsv_hme), bmc.c:304
 * (dsv_frame_add) and the on-path half of frame.c (dsv_extend_frame, dsv_extend_frame_luma, dsv_ds2x_frame_luma, dsv_frame_avg_luma:
 * dsv.h:170-176) -- as a forward to the product's twin dsvg_op_* (include/dsvg.h).  oracle/Makefile (target `ref`) compiles it together with
 * the reference's dsv_main.c dsv_encoder.c dsv_decoder.c dsv.c bs.c util.c frame.c WHERE THEY LIE under /root/reference -- sbt.c hzcc.c bmc.c
 * hme.c are NOT compiled; frame.c is compiled with its four on-path functions renamed out of the way by -D -- and links the result against
 * libdsv1_mi355x.so: oracle/_ref/dsv1_opswap.  tests/test_gpu_dropin_cli.py runs it against the reference CLI's own bytes.
 * This is the binding INTEGRATION.md section 2 describes for a maintainer who swaps operators one at a time. */
#include <stdio.h>
#include <stdlib.h>
#include "dsv.h"
#include "dsv_internal.h"
#include "dsv_encoder.h"
#include "dsvg.h"

/* the twins take the reference's structs as they are: same fields, order and sizes */
_Static_assert(sizeof(DSV_PLANE) == sizeof(dsvg_plane) && sizeof(DSV_COEFS) == sizeof(dsvg_coefs) && sizeof(DSV_FRAME) == sizeof(dsvg_frame), "frame structs");
_Static_assert(sizeof(DSV_MV) == sizeof(dsvg_mv) && sizeof(DSV_PARAMS) == sizeof(dsvg_params) && sizeof(DSV_META) == sizeof(dsvg_meta), "parameter structs");
_Static_assert(sizeof(DSV_STABILITY) == sizeof(dsvg_stability) && sizeof(DSV_BS) == sizeof(dsvg_bs) && sizeof(DSV_HME) == sizeof(dsvg_hme), "operator structs");

static void ck(int rc, const char *what)
{
    if (rc) {
        fprintf(stderr, "[opswap] %s failed rc=%d: %s\n", what, rc, dsvg_last_error());
        abort();                 /* the reference's operators cannot fail: there is nobody to tell */
    }
}
/* the motion fields dsv_hme hands back are released by the reference's dsv_free (dsv_encoder.c:239-244), which steps back over
 * dsv_alloc's 16-byte header (dsv.c:41-66): they must come from the reference's dsv_alloc */
__attribute__((constructor)) static void opswap_init(void) { dsvg_set_allocator(dsv_alloc, dsv_free); }

int dsv_get_quant(int q, int isP, int level) { return dsvg_get_quant(q, isP, level); }                    /* hzcc.c:77-92 */
int dsv_lb2(unsigned n) { return dsvg_lb2(n); }                                                            /* hzcc.c:437-447 */
void dsv_fwd_sbt(DSV_PLANE *src, DSV_COEFS *dst, int isP) { ck(dsvg_op_fwd_sbt((const dsvg_plane *)src, (dsvg_coefs *)dst, isP), "dsv_fwd_sbt"); }
void dsv_inv_sbt(DSV_PLANE *dst, DSV_COEFS *src, int q, int isP, int c) { ck(dsvg_op_inv_sbt((dsvg_plane *)dst, (dsvg_coefs *)src, q, isP, c), "dsv_inv_sbt"); }
void dsv_encode_plane(DSV_BS *bs, DSV_COEFS *src, int q, DSV_STABILITY *stab) { ck(dsvg_op_encode_plane((dsvg_bs *)bs, (dsvg_coefs *)src, q, (const dsvg_stability *)stab), "dsv_encode_plane"); }
void dsv_decode_plane(uint8_t *in, unsigned s, DSV_COEFS *dst, int q, DSV_STABILITY *stab) { ck(dsvg_op_decode_plane(in, s, (dsvg_coefs *)dst, q, (const dsvg_stability *)stab), "dsv_decode_plane"); }
void dsv_sub_pred(DSV_MV *vecs, DSV_PARAMS *p, DSV_FRAME *dif, DSV_FRAME *inp, DSV_FRAME *ref)
{
    ck(dsvg_op_sub_pred((const dsvg_mv *)vecs, (const dsvg_params *)p, (dsvg_frame *)dif, (dsvg_frame *)inp, (const dsvg_frame *)ref), "dsv_sub_pred");
}
void dsv_add_pred(DSV_MV *vecs, DSV_PARAMS *p, DSV_FRAME *dif, DSV_FRAME *out, DSV_FRAME *ref)
{
    ck(dsvg_op_add_pred((const dsvg_mv *)vecs, (const dsvg_params *)p, (dsvg_frame *)dif, (dsvg_frame *)out, (const dsvg_frame *)ref), "dsv_add_pred");
}
void dsv_frame_add(DSV_FRAME *dst, DSV_FRAME *src) { ck(dsvg_op_frame_add((dsvg_frame *)dst, (const dsvg_frame *)src), "dsv_frame_add"); }
int dsv_hme(DSV_HME *hme)
{
    int pct = 0;
    ck(dsvg_op_hme((dsvg_hme *)hme, &pct), "dsv_hme");
    return pct;
}
DSV_FRAME *dsv_extend_frame(DSV_FRAME *frame) { ck(dsvg_op_extend_frame((dsvg_frame *)frame), "dsv_extend_frame"); return frame; }          /* frame.c:263 */
DSV_FRAME *dsv_extend_frame_luma(DSV_FRAME *frame) { ck(dsvg_op_extend_frame_luma((dsvg_frame *)frame), "dsv_extend_frame_luma"); return frame; }
void dsv_ds2x_frame_luma(DSV_FRAME *dst, DSV_FRAME *src) { ck(dsvg_op_ds2x_frame_luma((dsvg_frame *)dst, (const dsvg_frame *)src), "dsv_ds2x_frame_luma"); }
int dsv_frame_avg_luma(DSV_FRAME *frame)
{
    int avg = 0;
    ck(dsvg_op_frame_avg_luma((const dsvg_frame *)frame, &avg), "dsv_frame_avg_luma");
    return avg;
}
