// dsvg_dev.hpp -- shared device/host structures of the MI355X DSV1 hot path (gfx950 only).
//
// Data layout in HBM
//   * Pixel frames use EXACTLY the reference layout of dsv_mk_frame (frame.c:63-120): one
//     allocation per frame, planes Y,U,V back to back, 64-px replicated border on every side,
//     row stride round16(w+128).  Motion search / compensation address up to one pixel beyond
//     the border; identical linear layout => identical bytes.  Frames of one
//     kind live in one slab, frame i at base + i*frame_bytes (+ a zeroed guard in front).
//   * Coefficients use the reference's Mallat layout (dsv_mk_coefs frame.c:29-61): int32,
//     row-major w x h per plane, three planes back to back, sub-bands in quadrants.
//   * The top of the pyramid (levels >= 6, a 60x34 band at 1080p) is transformed inside LDS by one
//     workgroup; levels 4-5 run in multi-workgroup "mid" kernels.  The LL3 / LL5 bands travel between
//     the stages through small compact scratch planes (s3, s5), the LL1 band of intra pictures through s1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define DSVG_RSU(x, s) (((x) + (1 << (s)) - 1) >> (s))   // DSV_ROUND_SHIFT dsv.h:62

// Pointers that a kernel reads out of a table in memory (JobDev) are GENERIC pointers to the compiler: it emits flat_load /
// flat_store with a 64-bit address computed per access.  Cast to the global address space where it matters
// (`auto p = dsvg_global(jb.sym)`): global_load / global_store then take the wave-uniform base in SGPRs and a 32-bit lane
// offset (no per-access address arithmetic), and they do not occupy the LDS/flat queue.
#define DSVG_GLOBAL __attribute__((address_space(1)))
template <typename T> static __device__ __forceinline__ DSVG_GLOBAL T *dsvg_global(T *p) { return (DSVG_GLOBAL T *)p; }
// element `idx` of a global array with the byte offset formed in 32 bits (idx * sizeof(T) < 4 GiB is the caller's fact): the
// compiler can then keep the SGPR base + 32-bit lane offset form instead of widening the index
template <typename T> static __device__ __forceinline__ T dsvg_at(const DSVG_GLOBAL T *base, unsigned idx)
{
    return *reinterpret_cast<const DSVG_GLOBAL T *>(reinterpret_cast<const DSVG_GLOBAL char *>(base) + idx * (unsigned)sizeof(T));
}
// 8- and 16-byte accesses through such pointers (the HIP vector classes cannot be copied out of an address space)
typedef unsigned dsvg_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned dsvg_u32x4 __attribute__((ext_vector_type(4)));
template <typename P> static __device__ __forceinline__ uint2 dsvg_ld2(P p) { const dsvg_u32x2 v = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x2 *>(p); return make_uint2(v.x, v.y); }
template <typename P> static __device__ __forceinline__ void dsvg_st2(P p, uint2 v) { *reinterpret_cast<DSVG_GLOBAL dsvg_u32x2 *>(p) = dsvg_u32x2{v.x, v.y}; }
typedef unsigned dsvg_u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));      // 16 / 12 bytes at a dword-aligned address
typedef unsigned dsvg_u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));
template <typename P> static __device__ __forceinline__ uint4 dsvg_ld4(P p) { const dsvg_u32x4 v = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4 *>(p); return make_uint4(v.x, v.y, v.z, v.w); }
template <typename P> static __device__ __forceinline__ void dsvg_st4(P p, uint4 v) { *reinterpret_cast<DSVG_GLOBAL dsvg_u32x4 *>(p) = dsvg_u32x4{v.x, v.y, v.z, v.w}; }

struct DMV {                 // == DSV_MV (dsv.h:137-150), 12 bytes
    int16_t x, y;
    uint8_t mode, submask, lo_var, lo_tex, high_detail, pad[3];
};
static_assert(sizeof(DMV) == 12, "DMV must match DSV_MV");

struct FrameLayout {         // one geometry of the reference frame layout
    int w[3], h[3], stride[3];
    int hs, vs;              // chroma shifts
    size_t off[3];           // byte offset of pixel (0,0) of each plane from the frame base
    size_t bytes;            // allocation size of one frame (sum of plane lens)
    size_t pitch;            // distance between consecutive frames in a slab (bytes, padded)
};

struct CoefLayout {
    int w[3], h[3];          // coefficient plane dims (chroma rounded up to even, frame.c:41-42)
    int lvls[3];             // number of transform levels (sbt.c:617-628)
    int w3[3], h3[3];        // dims of the LL3 band = HZCC "LL" region = LDS tail region
    int w1[3], h1[3];        // dims of the LL1 band
    size_t off[3];           // offset of each plane in ints
    size_t total;            // ints per job
    size_t s3off[3], s3total;
    size_t s1off[3], s1total;
    int w5[3], h5[3];        // dims of the LL5 band (what the LDS tail works on)
    size_t s5off[3], s5total;
};

// HZCC scan geometry of one plane (hzcc.c:30-48,137-293): ten regions in scan order
struct HzRegion {
    int x0, y0, sw, sh;      // rectangle in the coefficient plane
    int base;                // scan index of its first cell
    int level;               // -1 "LL", 0..2
    int qp, qp_h;            // quantiser (level 2: shifts)
    int dbx, dby;            // 14-bit fixed point block steps (hzcc.c:196-197)
};
struct HzPlane {
    HzRegion r[10];
    int nscan;               // total scan cells (sum of sw*sh)
    int nchunks;             // ceil(nscan / HZ_CHUNK)
    int w, h;                // coefficient plane dims
    int nbh;                 // nblocks_h
    int s_w[3], s_h[3];      // region size at level l (for overlap tests)
    int pad;
};
#define HZ_CHUNK 2048        // scan cells per workgroup in the quantise / emit kernels

struct HzChunkSum {          // written by hz_quant per chunk
    int nnz;                 // non-zero count
    int first_pos, last_pos; // scan positions of first / last non-zero (-1 if none)
    int last_val;            // value of the last non-zero
    unsigned bits_inner;     // bits of all symbols except the chunk's first one
    // filled by hz_scan:
    unsigned long long bit_off; // bit offset of the chunk's first symbol in the plane payload
    int prev_pos, prev_val;  // last non-zero before this chunk (-1 / 0 if none)
    int nz_base;             // index of the chunk's first non-zero in the whole plane
    int packed;              // the chunk's list holds one word per entry: (symbol << 16) | position in the chunk (collect_round_pk); else nzpos / nzval
};
struct HzPlaneSum {          // written by hz_scan per plane
    unsigned long long total_bits;
    unsigned nruns;
    int dc;
    int overflow;
    int last_chunk;          // last chunk holding a non-zero (-1 if the plane is empty)
    int rc;                  // device-resident rate control (k_rc, written after k_hz_scan): plane 0 = the frame quantiser the picture was
                             // coded with, plane 1 = the bytes of its packet as the device computed them, plane 2 = 0 (was padding)
};
static_assert(sizeof(HzPlaneSum) == 32, "HzPlaneSum travels to the host as 32-byte records");

// decoder: k_hz_parse -> k_hz_codes hand-over for one 128-bit chunk of a plane payload
struct HzParseChunk {
    unsigned long long m0, m1;   // bit 127-i of {m0,m1}: a code ends with chunk bit i
    int cbase;                   // index (in the alternating chain) of the first code that ends in this chunk
    int prev_end;                // payload bit after the code before that one
};

struct JobDev {              // everything a kernel needs to find one picture job's buffers
    const uint8_t *src;      // source frame (bordered, extended)
    const uint8_t *srcp[3];  // pixel (0,0) of each SOURCE plane and its row stride: the bordered frame above, or -- chroma of frames loaded
    int srcs[3];             // "in place" (dsvg_load_frames_map_ex) -- the caller's packed planar clip (stride = plane width: no border, never read outside the picture)
    const uint8_t *ref;      // reference reconstruction (bordered, extended) or nullptr
    uint8_t *recon;          // where the extended reconstruction is kept, or nullptr
    uint8_t *xf;             // work frame: residual in, reconstruction out
    uint8_t *pred;           // prediction frame ("dif" of dsv_sub_pred)
    int32_t *coef;           // 3 coefficient planes
    int32_t *s3, *s1, *s5;   // LL3 / LL1 / LL5 scratch
    const DMV *mvs;          // device motion field
    const uint8_t *stable;   // device stable_blocks
    int32_t *nzpos, *nzval;  // per-plane compact non-zero lists (chunk-local slots)
    HzChunkSum *chunks;      // per plane chunk summaries (3 * max chunks)
    HzPlaneSum *psum;        // 3 entries
    uint8_t *bits;           // packed payload, 3 planes at bits_off[c]
    size_t bits_off[3], bits_cap[3];
    size_t nz_off[3];        // offsets of each plane in nzpos/nzval (ints)
    size_t hz_coef_off[3];   // offsets of each plane in coef (ints)
    int chunk_off[3];
    int dec_cnt[3];          // decoder: number of (position,value) pairs per plane (written by k_hz_parse)
    int dec_runs[3], dec_len[3], dec_dc[3];   // decoder: run count / byte length / DC of the plane header (host)
    long long dec_bitpos[3]; // decoder: bit offset of the first code inside the uploaded payload
    uint8_t *nzf;            // encoder P pictures: flag byte per 4 scan positions (set by the forward transform where a symbol is non-zero, cleared by k_hz_collect); null = none
    HzParseChunk *dec_meta[3];   // decoder: per 128-bit payload chunk, what k_hz_parse found (>= dec_len/16 + 2 entries)
    long long dec_s0[3];         // decoder: first bit of the alternating code chain (k_hz_parse)
    int dec_npass[3], dec_ncode[3], dec_first_bad[3];   // decoder: passes done / codes seen (k_hz_parse), first entry past the data (k_hz_codes)
    int16_t *sym;            // fused quantiser: quantised symbol of every detail scan cell, indexed nz_off[c] + scan position
    // Sparse P pictures of the encoder (nzf != null): the symbol planes are ZERO between pictures.  The forward transform
    // stores only non-zero symbols (+ nzf, cflag, pflag), the inverse transform skips tiles whose patches carry no flag and
    // whose LL3 values are zero (reconstruction == prediction), and k_hz_collect -- the last reader -- takes symbols and
    // flags down again.
    uint8_t *pflag;          // per plane (offset s3off[c]) and 8x8-pixel patch: 1 = the patch has a non-zero detail symbol (levels 1-3); written for every patch of every picture
    uint8_t *cflag;          // per scan chunk (chunk_off[c] + chunk): 1 = the chunk holds a non-zero detail symbol; cleared by k_hz_collect
    unsigned *stat;          // null, or [4][64] diagnostic counters (64 shards each: one address takes ~90 atomics/us): inverse tiles on the general path {luma, chroma}, on the zero path {luma, chroma}
    int dec_sym[3];          // decoder, P pictures: 1 = the plane's detail entries are scattered as int16 symbols into the (zero-kept) symbol plane and the
                             // fused inverse dequantises them (planes without shared scan cells); 0 = dequantised int32 coefficients
    int *dec_flag;           // decoder: one word per job, set by the scatter when the sparse symbol path cannot represent the picture exactly (bit 0: a symbol
                             // beyond int16; bit 1: a cell shared by two scan regions keeps the EARLIER region's value, hzcc.c:295-435) -- the host then
                             // decodes the call again on the int32 coefficient path (dsvg_decode_pictures)
    int dec_lim[16];         // decoder, sparse path: largest |dequantised value| a P picture of 8-bit video can hold at transform level k ([0]: the DC) --
                             // twice the forward transform's gain bound (tests/test_symbol_range.py); beyond it the packed int16 inverse is not exact
    int pf_off[3];           // offset of each plane in pflag (= CoefLayout.s3off)
    int fused;               // 1: forward transform already quantised the detail bands (P pictures)
    int32_t *llsym;          // encoder, llq: quantised symbols of the LL region (scan cells below hz[c].r[1].base), int32, plane c at ll_off[c] + scan position;
    int ll_off[3];           //   written by k_fwd_haar_mid<4,.,true> (levels 4, 5) and k_tail_q (levels >= 6), read by k_hz_collect*
    int llq;                 // 1: the LL region is quantised where it is produced (no k_hz_quant<true>): k_hz_collect* compacts its chunks too
    short ext[8];            // border of the reconstruction that will be read: pixels left, right, rows above, below -- luma [0..3], chroma [4..7] (k_extend16)
    HzPlane hz[3];
    int hqp[16];             // luma smoothing bound per level (sbt.c:677-696), index = level
    int isP;
    int quant;
};

// ---- everything a picture's frame quantiser determines in its job record: the HZCC region quantisers of the three planes
// (hzcc.c:50-57 fix_quant, :77-92 dsv_get_quant, :190-205) and the smoothing bounds of the luma inverse (sbt.c:677-696).
// ONE implementation for the host's fill_job (make_hz_plane / make_hqp) and for k_rc, which rewrites these fields of the
// NEXT picture's record on the device once the rate control knows its quantiser.
static __host__ __device__ inline int dsvg_lb2u(unsigned n)                     // dsv_lb2 hzcc.c:437-447
{
    unsigned i = 1;
    int l = 0;
    while (i < n) { i <<= 1; l++; }
    return l;
}
static __host__ __device__ inline int dsvg_level_quant(int q, int isP, int level)   // dsv_get_quant hzcc.c:77-92
{
    if (isP) q = q * 3 / 2;
    if (level == 1) q = q * 2 / 3;
    else if (level == 2) q = q * 3 / 2;
    return q < 16 ? 16 : q;
}
static __host__ __device__ inline void dsvg_plane_set_quant(HzPlane &hp, int q, int isP, int cur_plane)
{
    if (cur_plane > 0 && q > 512) q = 512;                     // fix_quant hzcc.c:50-57
    hp.r[0].qp = dsvg_level_quant(q, isP, 0); hp.r[0].qp_h = 0;
    for (int l = 0; l < 3; l++) {
        int qp = dsvg_level_quant(q, isP, l), qp_h = 0;
        if (l == 2) {
            qp = dsvg_lb2u((unsigned)qp);
            qp_h = qp - (isP ? 1 : 3);                         // DSV_QP_P / DSV_QP_I
            qp_h = qp_h < 1 ? 1 : (qp_h > 24 ? 24 : qp_h);
        }
        for (int s = 1; s < 4; s++) { hp.r[3 * l + s].qp = qp; hp.r[3 * l + s].qp_h = qp_h; }
    }
}
static __host__ __device__ inline void dsvg_set_hqp(int *hqp, int q, int isP)   // sbt.c:677-696
{
    const int llq = dsvg_level_quant(q, isP, 0) / 2;
    for (int i = 0; i < 16; i++) {
        int v;
        if (i > 3 || i == 0) v = llq;
        else {
            v = dsvg_level_quant(q, isP, 3 - i);
            if (i == 1) {
                v = dsvg_lb2u((unsigned)v);
                v -= isP ? 1 : 3;
                v = v < 1 ? 1 : (v > 24 ? 24 : v);
                v = (1 << v) >> 1;
            }
            v /= 2;
        }
        hqp[i] = v;
    }
}
static __host__ __device__ inline void dsvg_job_set_quant(JobDev &jb, int quant)
{
    for (int p = 0; p < 3; p++) dsvg_plane_set_quant(jb.hz[p], quant, jb.isP, p);
    dsvg_set_hqp(jb.hqp, quant, jb.isP);
    jb.quant = quant;
}

// rate control shared with the C session layer (include/dsvg_rc.h): the same statements on both sides
#define DSVG_RC_FN static __host__ __device__ inline
#include "../../include/dsvg_rc.h"
struct RcJobDev {            // per device job of a rate-controlled call (dsvg_code_batch_rc)
    int slot;                // rate-control state of the job's stream: rc_state[slot]
    int prefix_len;          // bytes of the packet in front of the quantiser field
    int forced_intra;        // quality2quant's argument
    int next;                // device job (absolute index) of the stream's next picture in this call, -1 = none
};

struct SbtGeo {              // per-plane constants for the transform kernels
    int W, H;                // coefficient dims
    int ph, pstride;         // pixel rows available / row stride of the pixel plane
    size_t poff;             // byte offset of pixel (0,0) in a frame
    size_t coff, s3off, s1off, s5off; // int offsets of the plane in coef / s3 / s1 / s5
    int lvls;                // total levels
    int w3, h3, w1, h1, w5, h5;
    int pw;                  // pixel plane width (recon store guard)
    int l1a;                 // 1: the level-1 scan regions start and advance in multiples of 4 cells (8-byte symbol loads of the fast inverse body)
};

// round-half-away-from-zero divisions (sbt.c:63-88: v < 0 ? -((h - v) >> k) : (v + h) >> k), branch-free:
// (v + h + (v >> 31)) >> k gives the same value for every int (checked exhaustively over +-2*10^5, tools note in DESIGN)
static __device__ __forceinline__ int d_rdiv2(int v) { return (v + 1 + (v >> 31)) >> 1; }
static __device__ __forceinline__ int d_rdiv4(int v) { return (v + 2 + (v >> 31)) >> 2; }
static __device__ __forceinline__ int d_rdiv8(int v) { return (v + 4 + (v >> 31)) >> 3; }
// C's truncating x / 4.  SMALL: |x| < 2^30 is known (transform levels 1..3 of 8-bit video: < 2^20) -- bits 31:30 are
// then 11 for negative x and 00 otherwise, one v_bfe_u32 instead of shift + mask
template <bool SMALL>
static __device__ __forceinline__ int d_div4(int x)
{
    return SMALL ? (x + (int)__builtin_amdgcn_ubfe((unsigned)x, 30u, 2u)) >> 2 : (x + ((x >> 31) & 3)) >> 2;
}
static __device__ __forceinline__ int d_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static __device__ __forceinline__ int d_sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// C.3.1.1 LL scaling (sbt.c:20-21): C integer division truncates toward zero -- load-bearing
static __device__ __forceinline__ int d_ll_down(int x) { return x * 4 / 5; }
static __device__ __forceinline__ int d_ll_up(int x) { return x * 5 / 4; }
template <bool SMALL> static __device__ __forceinline__ int d_ll_up_t(int x) { return d_div4<SMALL>(x * 5); }

// XCD-aware work mapping: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so
// workgroup `lin` takes the logical item (lin % 8) * ceil(total / 8) + lin / 8 -- every XCD then walks ONE contiguous
// range of items and neighbouring blocks share their halo lines in the same L2.  Launch 8 * ceil(total / 8)
// workgroups; the caller drops logical ids >= total.
static __device__ __forceinline__ int d_xcd_remap(int lin, int total)
{
    const int per = (total + 7) >> 3;
    return (lin & 7) * per + (lin >> 3);
}
static inline int xcd_grid(int total) { return 8 * ((total + 7) / 8); }
// A logical gx x gy x gz grid launched as xcd_grid(gx * gy * gz) workgroups in one dimension: the workgroup's logical
// (x, y, z), x fastest -- an XCD then walks whole rows of neighbouring tiles, whose shared cache lines (halo rows, lines
// that straddle two tiles) are fetched into its L2 once instead of once per XCD.  false: a padding workgroup.
struct Blk3 { int x, y, z; };
// The logical grid as the launcher hands it over: its extents and their reciprocals, min(floor(2^32 / d), 2^32 - 1) -- every wave of every
// workgroup splits its index by gx and gy, and as integer divisions that was two ~25-instruction sequences (float reciprocal, two correction
// steps each) at the head of each wave's dependency chain; with the reciprocal the estimate is the quotient or one below it
struct XcdGrid { int gx, gy, gz; unsigned igx, igy; };
static inline XcdGrid mk_xcd_grid(int gx, int gy, int gz)
{
    auto inv = [](int d) { const unsigned long long q = 0x100000000ull / (unsigned long long)(d > 0 ? d : 1); return (unsigned)(q > 0xffffffffull ? 0xffffffffull : q); };
    XcdGrid g;
    g.gx = gx; g.gy = gy; g.gz = gz; g.igx = inv(gx); g.igy = inv(gy);
    return g;
}
static __device__ __forceinline__ unsigned d_udiv_r(unsigned n, unsigned d, unsigned inv, unsigned &rem)
{
    unsigned q = __umulhi(n, inv);
    rem = n - q * d;
    if (rem >= d) { q++; rem -= d; }
    return q;
}
static __device__ __forceinline__ bool d_xcd_blk3(const XcdGrid &G, Blk3 &b, bool plain = false)
{
    const int total = G.gx * G.gy * G.gz, item = plain ? (int)blockIdx.x : d_xcd_remap((int)blockIdx.x, total);   // plain: the hardware's round robin (A/B)
    if (item >= total) return false;
    unsigned x, y;
    const unsigned r = d_udiv_r((unsigned)item, (unsigned)G.gx, G.igx, x);
    b.z = (int)d_udiv_r(r, (unsigned)G.gy, G.igy, y);
    b.x = (int)x; b.y = (int)y;
    return true;
}
// plane and job of a grid layer (npl planes per job: 1 or 2 in every launch of the encoder)
static __device__ __forceinline__ void d_job_plane(int z, int npl, int c0, int &job, int &c)
{
    if (npl == 1) { job = z; c = c0; }
    else if (npl == 2) { job = z >> 1; c = c0 + (z & 1); }
    else { job = z / npl; c = c0 + z % npl; }
}

// ---- diagnostic builds only (make EXTRA=-DDSVG_CLOCK_PROBE, tools/ab/clock_probe.sh): the shader clock a kernel runs at.
// Thread 0 of one workgroup in 128 stamps s_memtime (shader clock) and s_memrealtime (100 MHz) on entry and exit and adds the two
// differences to its kernel's slot; clock = sum(dt) / sum(dr) x 100 MHz.  The library prints the slots when a context is
// destroyed.  Nothing of this exists in the shipped build.
#ifdef DSVG_CLOCK_PROBE
#define DSVG_CLK_SLOTS 16
static __device__ unsigned long long dsvg_clk_acc[DSVG_CLK_SLOTS][3];
struct ClkStamp { unsigned long long t, r; };
static __device__ __forceinline__ ClkStamp d_clk_begin()
{
    ClkStamp s;
    s.t = __builtin_amdgcn_s_memtime(); s.r = __builtin_amdgcn_s_memrealtime();
    return s;
}
static __device__ __forceinline__ void d_clk_end(int slot, const ClkStamp &s)
{
    const unsigned long long t = __builtin_amdgcn_s_memtime() - s.t, r = __builtin_amdgcn_s_memrealtime() - s.r;
    if (threadIdx.x == 0 && (blockIdx.x & 127) == 5) {           // one workgroup in 128: every workgroup's atomics on one line slowed the step 5x
        atomicAdd(&dsvg_clk_acc[slot][0], t); atomicAdd(&dsvg_clk_acc[slot][1], r); atomicAdd(&dsvg_clk_acc[slot][2], 1ull);
    }
}
#define DSVG_CLK_BEGIN() const ClkStamp clk_stamp_ = d_clk_begin()
#define DSVG_CLK_END(slot) d_clk_end(slot, clk_stamp_)
// host side, one per kernel file: prints and clears that file's slots
#define DSVG_CLK_DUMP_FN(fn, ...) extern "C" void fn() { \
    static const char *names[] = {__VA_ARGS__}; unsigned long long h[DSVG_CLK_SLOTS][3]; \
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dsvg_clk_acc), sizeof(h)) != hipSuccess) return; \
    for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]); i++) if (h[i][1]) \
        { if (h[i][2]) fprintf(stderr, "[clock probe] %-28s %8.1f MHz  (%llu workgroups, %.2f us each)\n", names[i], 100.0 * (double)h[i][0] / (double)h[i][1], h[i][2], 0.01 * (double)h[i][1] / (double)h[i][2]); \
          else fprintf(stderr, "[clock probe] %-34s %9.0f cycles per block\n", names[i], (double)h[i][0] / (double)h[i][1]); } \
    memset(h, 0, sizeof(h)); (void)hipMemcpyToSymbol(HIP_SYMBOL(dsvg_clk_acc), h, sizeof(h)); }
#else
#define DSVG_CLK_BEGIN() do { } while (0)
#define DSVG_CLK_END(slot) do { } while (0)
#endif

// ---- HZCC quantiser arithmetic shared by the transform-fused path (k_sbt.hip) and k_hzcc.hip -----------
#define HZ_MINQ 16
static __device__ __forceinline__ int hzq_lo(int v, int q)            // quant hzcc.c:94-112
{
    int m = (v < 0 ? -v : v) << 1;
    if (m <= q) return 0;
    m = (m + 1) / (q << 1);
    return v < 0 ? -m : m;
}
static __device__ __forceinline__ int hzdq_lo(int v, int q)           // dequant hzcc.c:121-128
{
    return v < 0 ? -((-v * (q << 1) + q) >> 1) : (v * (q << 1) + q) >> 1;
}
static __device__ __forceinline__ int hzq_hi(int v, int sh) { return v < 0 ? -((-v) >> sh) : v >> sh; }
static __device__ __forceinline__ int hzdq_hi(int v, int sh) { return (int)((unsigned)v << sh); }
// quantiser for cell (x,y) of region r (tmq4pos hzcc.c:64-74, highest level hzcc.c:221-224)
template <typename SP>
static __device__ __forceinline__ int hz_cell_tq(const HzRegion &r, SP stable, int nbh, int x, int y)
{
    if (r.level < 0) return r.qp;
    const int flag = stable[((y * r.dby) >> 14) * nbh + ((x * r.dbx) >> 14)];
    if (r.level == 2) return flag ? r.qp_h : r.qp;
    const int t = (flag & 2) ? r.qp >> 2 : (flag ? r.qp >> 1 : r.qp);
    return t < HZ_MINQ ? HZ_MINQ : t;
}
static __device__ __forceinline__ int hz_quant_any(const HzRegion &r, int v, int tq) { return r.level == 2 ? hzq_hi(v, tq) : hzq_lo(v, tq); }
static __device__ __forceinline__ int hz_dequant_any(const HzRegion &r, int v, int tq) { return r.level == 2 ? hzdq_hi(v, tq) : hzdq_lo(v, tq); }
