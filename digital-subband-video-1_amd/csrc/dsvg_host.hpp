// dsvg_host.hpp -- host-side helpers shared by the C-ABI implementation files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/dsvg.h"
#include "dsvg_dev.hpp"
#include "dsvg_kernels.hpp"

void dsvg_set_error(const char *fmt, ...);

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            dsvg_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DSVG_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define GUARD_BYTES (256 * 1024)     // zeroed guard in front of / behind every frame slab

static inline int rsu(int x, int s) { return (x + (1 << s) - 1) >> s; }
static inline int fmt_hs(int fmt) { return (fmt >> 2) & 3; }
static inline int fmt_vs(int fmt) { return fmt & 3; }

// reference frame layout (dsv_mk_frame frame.c:63-120), always with the 64-px border
void make_frame_layout(FrameLayout &L, int fmt, int w, int h);
// layout of an existing host frame (op-level calls): offsets / strides taken from the struct
int  layout_from_host(FrameLayout &L, const dsvg_frame *f, const uint8_t **base, size_t *bytes);
void make_coef_layout(CoefLayout &C, int fmt, int w, int h);
int  lb2u(unsigned n);
int  get_quant(int q, int isP, int level);
void make_sbt_geo(SbtGeo &g, int W, int H, int pw, int ph, int pstride, size_t poff, size_t coff, size_t s3off, size_t s1off, size_t s5off = 0);
// HZCC scan geometry + quantisers of one plane (hzcc.c:30-57,77-92,186-212)
void make_hz_plane(HzPlane &hp, int w, int h, int q, int isP, int cur_plane, int nbh, int nbv);
void make_hqp(int hqp[16], int q, int isP);
void block_geometry(int w, int h, int *bw, int *bh, int *nbh, int *nbv);     // dsv_encoder.c:556-595
int  auto_pyramid_levels(int w, int h, int nbh, int nbv);                    // dsv_encoder.c:602-613

// device slab with zeroed guards
struct Slab {
    uint8_t *raw = nullptr;      // allocation
    uint8_t *p = nullptr;        // usable start (raw + GUARD_BYTES)
    size_t bytes = 0;
    int alloc(size_t n, bool zero = true);
    void release();
};

// per-kernel timing with HIP events on the pipeline stream (bench.py roofline leg).  Only kernels whose
// bit is set in `mask` are bracketed, so the timed region pays for the events of one kernel only.
// One id per kernel SYMBOL, named exactly as rocprofv3 prints it (up to the parameter list), so a bench.py
// roofline entry can be held against the committed --kernel-trace --stats summary line by line.
#define DSVG_KERNEL_IDS(X) \
    X(KID_UNPACK, "k_unpack") X(KID_EXTEND, "k_extend") X(KID_EXTEND16, "k_extend16") X(KID_DS2X, "k_ds2x") X(KID_LUMA_SUM, "k_luma_sum") X(KID_PACK16, "void k_pack_n<true>") X(KID_PACK, "void k_pack_n<false>") \
    X(KID_HME_LEVEL, "void k_hme_level<false>") X(KID_HME_LEVEL0, "void k_hme_level<true>") X(KID_HME_DETAIL, "k_hme_detail") X(KID_HME_CSUM, "k_hme_csum") \
    X(KID_MC, "k_mc") \
    X(KID_FWD_HAAR_PIX, "void k_fwd_haar_pix<false>") X(KID_FWD_HAAR_PIX_Q, "void k_fwd_haar_pix<true>") X(KID_FWD_MC_PIX_Y, "void k_fwd_mc_pix<0>") X(KID_FWD_MC_PIX_C, "void k_fwd_mc_pix<1>") \
    X(KID_FWD_MC_FAST_Y, "void k_fwd_mc_fast<0>") X(KID_FWD_MC_FAST_C, "void k_fwd_mc_fast<1>") \
    X(KID_FWD_B4T, "void k_fwd_b4t<false>") X(KID_FWD_B4T_Q, "void k_fwd_b4t<true>") \
    X(KID_FWD_HAAR_MID2, "void k_fwd_haar_mid<2, false>") X(KID_FWD_HAAR_MID2_Q, "void k_fwd_haar_mid<2, true>") X(KID_FWD_HAAR_MID4, "void k_fwd_haar_mid<4, false>") \
    X(KID_FWD_TAIL, "k_fwd_tail") \
    X(KID_HZ_QUANT, "void k_hz_quant<false>") X(KID_HZ_QUANT_LL, "void k_hz_quant<true>") X(KID_HZ_COLLECT, "k_hz_collect") X(KID_TAIL_Q, "k_tail_q") X(KID_INV_P_TILE_F, "void k_inv_p_tile<true>") X(KID_INV_P_TILE, "void k_inv_p_tile<false>") X(KID_HZ_COLLECT_LIST, "k_hz_collect_list") X(KID_HZ_EMIT_LIST, "k_hz_emit_list") X(KID_HZ_SCAN, "k_hz_scan") \
    X(KID_HZ_EMIT, "k_hz_emit") X(KID_HZ_PARSE, "k_hz_parse") X(KID_HZ_CODES, "k_hz_codes") X(KID_HZ_POSITIONS, "k_hz_positions") X(KID_HZ_SCATTER, "k_hz_scatter_lv") \
    X(KID_INV_TAIL, "k_inv_tail") X(KID_INV_PATCH_C, "k_inv_patch_c") \
    X(KID_INV_TILE_54_F, "void k_inv_haar_tile<true, 2, false>") X(KID_INV_TILE_54, "void k_inv_haar_tile<false, 2, false>") X(KID_INV_TILE_54_ALL, "k_inv_tile54_all") \
    X(KID_INV_TILE_PIX_SYM_F, "void k_inv_haar_tile<true, 0, true>") X(KID_INV_TILE_PIX_SYM, "void k_inv_haar_tile<false, 0, true>") \
    X(KID_INV_TILE_PIX_F, "void k_inv_haar_tile<true, 0, false>") X(KID_INV_TILE_PIX, "void k_inv_haar_tile<false, 0, false>") \
    X(KID_INV_TILE_S1_F, "void k_inv_haar_tile<true, 1, false>") X(KID_INV_TILE_S1, "void k_inv_haar_tile<false, 1, false>") \
    X(KID_INV_TILE_S1_SYM_F, "void k_inv_haar_tile<true, 1, true>") X(KID_INV_TILE_S1_SYM, "void k_inv_haar_tile<false, 1, true>") \
    X(KID_INV_B4T, "void k_inv_b4t<false>") X(KID_INV_B4T_SYM, "void k_inv_b4t<true>")
enum {
#define X(id, name) id,
    DSVG_KERNEL_IDS(X)
#undef X
    KID_N
};
const char *kid_name(int kid);
struct Prof {
    unsigned long long mask = 0;
    struct Rec { int kid; hipEvent_t a, b; double bytes; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double ms[KID_N] = {0}, bytes[KID_N] = {0};
    long launches[KID_N] = {0};
    bool open = false;
    hipEvent_t get();
    inline bool want(int kid) const { return (mask >> kid) & 1ull; }
    void begin(hipStream_t st, int kid, double alg_bytes);
    void end(hipStream_t st);
    void collect();
    void reset();
};
