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
enum {
    KID_UNPACK = 0, KID_EXTEND, KID_DS2X, KID_LUMA_SUM,
    KID_HME_LEVEL, KID_HME_LEVEL0, KID_HME_DETAIL,
    KID_MC,
    KID_FWD_HAAR_PIX, KID_FWD_B4T, KID_FWD_HAAR_S1, KID_FWD_TAIL,
    KID_HZ_QUANT, KID_HZ_COLLECT, KID_HZ_SCAN, KID_HZ_EMIT, KID_HZ_SCATTER,
    KID_INV_TAIL, KID_INV_HAAR_TILE, KID_INV_B4T,
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
