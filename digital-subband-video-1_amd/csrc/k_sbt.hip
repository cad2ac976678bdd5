// k_sbt.hip -- subband transform kernels for gfx950 (MI355X).  HBM-bound integer work: no MFMA.
//
// Replaces the reference's sbt.c: dsv_fwd_sbt sbt.c:630-651 (p2sbc :576, fwd :268, fwd_b4t_* :91-251)
// and dsv_inv_sbt sbt.c:654-714 (inv :438, inv_simple :352, inv_b4t_* :129-265, sbc2int :595).
//
// Decomposition (per plane, batched over picture jobs in grid.z):
//   forward  P : k_fwd_mc_fast    (encoder) one thread = one 8x8 pixel patch: motion compensation (bmc.c) + Haar levels 1..3
//                                 + the quantiser of their detail bands in registers, symbols stored sparsely; the common
//                                 patch (residual rows, level 1 and the half-pel filters that fit 16 bits on two samples
//                                 per instruction, v_pk_*_i16).  k_fwd_mc_pix: the general body (picture edges, intra blocks, shared scan cells),
//                                 on the strips of the grid that can hold such patches.  k_fwd_haar_pix: from a residual
//                                 frame (operator calls, block sizes that are not multiples of 8)
//            I : k_fwd_b4t        level 1 biorthogonal (1,3,3,1): row pass + column pass per thread on a
//                                 10x10 neighbourhood, LL1 -> s1;  k_fwd_haar_mid<2>: levels 2..3 from s1
//            all: k_fwd_haar_mid<4> levels 4..5 (LL3 -> LL5; encoder: with the LL quantiser);  k_fwd_tail: levels 6..top
//                                 inside LDS by one workgroup;  k_tail_q (encoder): forward tail + LL quantiser + inverse tail
//   inverse all: k_inv_tail       levels top..6 inside LDS, LL5 -> s5;  k_inv_haar_tile<.,2>: levels 5,4 -> s3
//            P : k_inv_p_tile     (sparse pictures, luma) levels 3,2,1 on a 128x64 pixel tile through LDS from the symbol
//                                 planes: the fast body, with edge variants for the last tile column / row;
//                k_inv_patch_c    (sparse pictures, chroma: no smoothing filter) one thread = one 8x8 patch, no LDS;
//                k_inv_haar_tile  the general tile body (halo 2/1/0 cells, any edge, coefficient or symbol input), fused
//                                 with sbc2int, the prediction add (dsv_frame_add bmc.c:304) and the store into the
//                                 reconstruction frame
//            I : k_inv_haar_tile<..TO_S1> levels 3,2 -> s1, then k_inv_b4t (column pass through LDS,
//                                 then row pass) fused with sbc2int
// Every formula keeps the C semantics of the reference: truncating *4/5, *5/4 and /4, rounding
// helpers round half away from zero, mirrored left/top and clamped right/bottom B4T edges.
#include <type_traits>
#include "dsvg_dev.hpp"
#include <algorithm>
#include "dsvg_kernels.hpp"
#include "dsvg_host.hpp"

// --------------------------------------------------------------------------------------------
// helpers
// --------------------------------------------------------------------------------------------
template <int M>
static __device__ __forceinline__ void store_row(int32_t *p, const int (&v)[M], int n)
{
    if (n >= M) {
        if (M == 4 && (((uintptr_t)p) & 15) == 0) {
            *reinterpret_cast<int4 *>(p) = make_int4(v[0], v[1 % M], v[2 % M], v[3 % M]);
            return;
        }
        if (M == 2 && (((uintptr_t)p) & 7) == 0) {
            *reinterpret_cast<int2 *>(p) = make_int2(v[0], v[1 % M]);
            return;
        }
#pragma unroll
        for (int i = 0; i < M; i++) p[i] = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < M; i++)
            if (i < n) p[i] = v[i];
    }
}

// ---- quantisation fused into the forward transform (P pictures) -------------------------------------------
// Every detail coefficient is quantised + dequantised the moment it is produced (hzcc.c:172-184 arithmetic):
// the dequantised value goes to the coefficient plane (what the inverse transform reads) and the quantised
// symbol (fits 16 bits for levels 1..3 of an 8-bit residual) to a symbol plane indexed by SCAN position, which
// k_hz_quant<SYM> then only has to compact.  A coefficient that two scan regions cover physically
// belongs to the later region's sub-band, so its producer writes both symbols.
struct QCtx {
    const HzPlane *hp;
    const DSVG_GLOBAL uint8_t *stable;      // global address space: see dsvg_global (dsvg_dev.hpp)
    DSVG_GLOBAL int16_t *sym;
    bool any_ov;
    // P pictures of the encoder (SPARSE mode, nzf != null): the symbol plane is zero between pictures and only non-zero
    // symbols are stored, each with one flag byte per four scan positions (nzf) and one per 2048-cell scan chunk (cfl), so
    // that k_hz_collect touches only what holds data; nz_any gathers "this thread stored something" for the patch flag
    // the inverse transform keys its zero-tile path on.  null: every symbol is stored (I pictures).
    DSVG_GLOBAL uint8_t *nzf = nullptr, *cfl = nullptr;
    mutable int nz_any = 0;
    __device__ __forceinline__ void put_sparse(int pos, int v) const
    {
        sym[pos] = (int16_t)v;
        nzf[pos >> 2] = 1;
        cfl[pos >> 11] = 1;                 // HZ_CHUNK = 2048
    }
};
static_assert(HZ_CHUNK == 2048, "cfl index");
// wave-uniform constants of one scan level: its LH/HL/HH regions differ only in origin and scan base
struct QLevel {
    int qp;                // levels 0,1: quantiser max(qp >> class, 16), class {none, stable, flag&2} (tmq4pos hzcc.c:64-74)
    int sh0, sh1;          // level 2: shift by flag class {none, any} (hzcc.c:221-224)
    int dbx, dby, sw, base0, base1, base2;
};
template <int HZL>
static __device__ __forceinline__ QLevel q_level(const HzPlane &hp)
{
    const HzRegion &r = hp.r[1 + 3 * HZL];
    QLevel L;
    L.qp = r.qp; L.sh0 = r.qp; L.sh1 = r.qp_h;
    L.dbx = r.dbx; L.dby = r.dby; L.sw = r.sw;
    L.base0 = r.base; L.base1 = hp.r[2 + 3 * HZL].base; L.base2 = hp.r[3 + 3 * HZL].base;
    return L;
}
// flag class of each cell of an MxM patch: one byte load when the whole patch sits in one block (the usual case)
template <int HZL, int M>
static __device__ __forceinline__ void q_flags(const QCtx &q, const QLevel &L, int cx0, int cy0, int wo, int ho, int (&cls)[M][M])
{
    const int nbh = q.hp->nbh;
    const int bx0 = (cx0 * L.dbx) >> 14, bx1 = ((cx0 + M - 1) * L.dbx) >> 14;
    const int by0 = (cy0 * L.dby) >> 14, by1 = ((cy0 + M - 1) * L.dby) >> 14;
    if (bx0 == bx1 && by0 == by1) {
        const int f = q.stable[by0 * nbh + bx0];
        const int k = HZL == 2 ? (f != 0) : ((f & 2) ? 2 : (f != 0));
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i < M; i++) cls[j][i] = k;
    } else {
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int i = 0; i < M; i++) {
                int f = 0;
                if (cx0 + i < wo && cy0 + j < ho) f = q.stable[(((cy0 + j) * L.dby) >> 14) * nbh + (((cx0 + i) * L.dbx) >> 14)];
                cls[j][i] = HZL == 2 ? (f != 0) : ((f & 2) ? 2 : (f != 0));
            }
    }
}
// quantise + dequantise one coefficient with quantiser qq (levels 0,1; rc ~ 1/(2 qq)) or shift qq (level 2):
// returns the dequantised value, sym = the symbol
template <int HZL>
static __device__ __forceinline__ int q_coef(int qq, float rc, int v, int &sym)
{
    if (HZL == 2) {
        const int t = (v < 0 ? -v : v) >> qq;
        sym = v < 0 ? -t : t;
        return (int)((unsigned)sym << qq);
    }
    const int m = (v < 0 ? -v : v) << 1;
    const int n = m + 1, d = qq << 1;                    // (m + 1) / (2q), exact: float estimate, then +-1
    int t = (int)((float)n * rc);
    const int rem = n - t * d;
    t += (rem >= d) - (rem < 0);
    const int dq = (t * d + qq) >> 1;
    const bool z = t == 0;                               // covers m <= q (q >= 16) and the q < m < 2q - 1 gap
    sym = z ? 0 : (v < 0 ? -t : t);
    return z ? 0 : (v < 0 ? -dq : dq);
}
// the earlier (level HZL-1) scan pass over a shared cell: writes its symbol, returns what the later pass sees
static __device__ int q_chain(const QCtx &q, int l, int gx, int gy, int val)
{
    const HzPlane &hp = *q.hp;
    if (!(gx < 2 * hp.s_w[l - 1] && gy < 2 * hp.s_h[l - 1])) return val;
    const int rx = gx >= hp.s_w[l - 1], ry = gy >= hp.s_h[l - 1];
    if (!(rx + ry)) return val;
    const HzRegion &e = hp.r[1 + 3 * (l - 1) + (rx + 2 * ry) - 1];
    const int ex = gx - e.x0, ey = gy - e.y0;
    const int etq = hz_cell_tq(e, q.stable, hp.nbh, ex, ey);
    const int ev = hz_quant_any(e, val, etq);
    if (q.nzf) { if (ev) q.put_sparse(e.base + ey * e.sw + ex, ev); }      // (not read by the inverse: no patch flag)
    else q.sym[e.base + ey * e.sw + ex] = (int16_t)ev;
    return ev ? hz_dequant_any(e, ev, etq) : 0;
}
template <int M>
static __device__ __forceinline__ void store_sym_row(DSVG_GLOBAL int16_t *p, const int (&v)[M], int n)
{
    if (n >= M) {
        if (M == 4 && (((uintptr_t)p) & 7) == 0) {
            dsvg_st2(p, make_uint2((v[0] & 0xffff) | ((unsigned)v[1 % M] << 16), (v[2 % M] & 0xffff) | ((unsigned)v[3 % M] << 16)));
            return;
        }
        if (M == 2 && (((uintptr_t)p) & 3) == 0) {
            *reinterpret_cast<DSVG_GLOBAL unsigned *>(p) = (v[0] & 0xffff) | ((unsigned)v[1 % M] << 16);
            return;
        }
#pragma unroll
        for (int i = 0; i < M; i++) p[i] = (int16_t)v[i];
    } else {
#pragma unroll
        for (int i = 0; i < M; i++)
            if (i < n) p[i] = (int16_t)v[i];
    }
}

// One forward Haar level on an NxN register patch whose top-left input sample is cell
// (2*cx0, 2*cy0) of the level's ws x hs input region.  Missing right/bottom samples are mirrored
// (equivalent to the edge formulas sbt.c:309-346); sub-bands that do not exist are not stored.
// LL region of the entropy coder (every band of levels >= 4, hzcc.c:161-186): one quantiser for all of it.  LLQ variants of
// the producers quantise each detail where it appears: the dequantised value goes where the coefficient would have gone
// (the inverse reads it), the symbol into the job's LL symbol plane at its scan position y * sw + x.
struct LLQ {
    int qp, sw;
    int32_t *sym;
    __device__ __forceinline__ int put(int x, int y, int v) const
    {
        const int s = hzq_lo(v, qp);
        sym[y * sw + x] = s;
        return s ? hzdq_lo(s, qp) : 0;
    }
};
template <int N, bool Q = false>
static __device__ __forceinline__ void haar_fwd_patch(const int (&in)[N][N], int (&out)[N / 2][N / 2],
                                                      int cx0, int cy0, int ws, int hs, int W,
                                                      int wo, int ho, int32_t *__restrict__ coef, bool scaled, const LLQ *lq = nullptr)
{
    constexpr int M = N / 2;
    const int nR = min(M, max(0, (ws >> 1) - cx0));   // cells of this row that have a right sample
    const int nC = min(M, max(0, wo - cx0));          // cells of this row that exist
#pragma unroll
    for (int j = 0; j < M; j++) {
        const int cy = cy0 + j;
        const bool hasB = 2 * cy + 1 < hs;
        int lh[M], hl[M], hh[M];
#pragma unroll
        for (int i = 0; i < M; i++) {
            const bool hasR = 2 * (cx0 + i) + 1 < ws;
            const int a = in[2 * j][2 * i];
            const int b = hasR ? in[2 * j][2 * i + 1] : a;
            const int c = hasB ? in[2 * j + 1][2 * i] : a;
            const int d = hasB ? (hasR ? in[2 * j + 1][2 * i + 1] : c) : b;
            const int ll = a + b + c + d;
            out[j][i] = scaled ? d_ll_down(ll) : ll;
            lh[i] = a - b + c - d;
            hl[i] = a + b - c - d;
            hh[i] = a - b - c + d;
        }
        if (2 * cy < hs) {
            if (Q) {
#pragma unroll
                for (int i = 0; i < M; i++) {
                    if (i < nR) lh[i] = lq->put(wo + cx0 + i, cy, lh[i]);
                    if (hasB && i < nC) hl[i] = lq->put(cx0 + i, ho + cy, hl[i]);
                    if (hasB && i < nR) hh[i] = lq->put(wo + cx0 + i, ho + cy, hh[i]);
                }
            }
            store_row<M>(coef + (size_t)cy * W + wo + cx0, lh, nR);
            if (hasB) {
                store_row<M>(coef + (size_t)(ho + cy) * W + cx0, hl, nC);
                store_row<M>(coef + (size_t)(ho + cy) * W + wo + cx0, hh, nR);
            }
        }
    }
}

// the same level with the quantiser of scan level HZL applied to the three detail bands.  Split into the per-patch
// set-up (HaarQ) and the work on one pair of input rows, so that a producer can hand rows over as they appear.
template <int N, int HZL>
struct HaarQ {
    static constexpr int M = N / 2;
    QLevel L;
    int nR, nC, cx0, cy0, ws, hs, wo, ho, bx0, bx1;
    int frow[M];
    bool chx, chy;
    __device__ __forceinline__ void init(const QCtx &q, int cx0_, int cy0_, int ws_, int hs_, int wo_, int ho_)
    {
        cx0 = cx0_; cy0 = cy0_; ws = ws_; hs = hs_; wo = wo_; ho = ho_;
        nR = min(M, max(0, (ws >> 1) - cx0));
        nC = min(M, max(0, wo - cx0));
        L = q_level<HZL>(*q.hp);
        bx0 = (cx0 * L.dbx) >> 14; bx1 = ((cx0 + M - 1) * L.dbx) >> 14;
        // the usual case, all cells of a row in one block: that row's flag byte is requested now, with the producer's
        // own loads, not in the middle of its arithmetic
        if (bx0 == bx1) {
            const int nbh = q.hp->nbh;
#pragma unroll
            for (int j = 0; j < M; j++) frow[j] = cy0 + j < ho ? q.stable[(((cy0 + j) * L.dby) >> 14) * nbh + bx0] : 0;
        }
        chx = HZL >= 1 && q.any_ov && cx0 == 0; chy = HZL >= 1 && q.any_ov && cy0 == 0;
    }
    // input rows 2j and 2j+1 of the patch -> LL row j (out) + the quantised detail symbols of cell row cy0 + j
    __device__ __forceinline__ void rows(const QCtx &q, int j, const int (&r0)[N], const int (&r1)[N], int (&out)[M], bool scaled) const
    {
        const int cy = cy0 + j;
        const bool hasB = 2 * cy + 1 < hs;
        const bool rowok = 2 * cy < hs;
        if constexpr (HZL == 2) {
            // Sparse P pictures, level 1 (shift quantiser, hzcc.c:221-224): a detail is a signed sum of the cell's four
            // samples, so |detail| <= 4 max|sample|, and it quantises to zero when that stays below 2^shift.  Most
            // residual rows of a well predicted picture are that small: one min3/max3 pass over the two rows decides, and
            // the row pair then only yields its LL values -- no details, no quantiser, nothing to store.  (The cells two
            // scan regions share are quantised by the earlier region's rule too: those patches take the full path.)
            if (q.nzf != nullptr && !(chx || chy)) {
                int mx = r0[0], mn = r0[0];
#pragma unroll
                for (int i = 1; i < N; i += 2) {
                    mx = max(mx, max(r0[i], i + 1 < N ? r0[i + 1] : r0[i]));
                    mn = min(mn, min(r0[i], i + 1 < N ? r0[i + 1] : r0[i]));
                }
#pragma unroll
                for (int i = 0; i < N; i += 2) {
                    mx = max(mx, max(r1[i], r1[i + 1]));
                    mn = min(mn, min(r1[i], r1[i + 1]));
                }
                if (4 * max(mx, -mn) < (1 << min(L.sh0, L.sh1))) {
#pragma unroll
                    for (int i = 0; i < M; i++) {
                        const bool hasR = 2 * (cx0 + i) + 1 < ws;
                        const int a = r0[2 * i];
                        const int b = hasR ? r0[2 * i + 1] : a;
                        const int c = hasB ? r1[2 * i] : a;
                        const int d = hasB ? (hasR ? r1[2 * i + 1] : c) : b;
                        const int ll = a + b + c + d;
                        out[i] = scaled ? d_ll_down(ll) : ll;
                    }
                    return;
                }
            }
        }
        int slh[M], shl[M], shh[M];
        // flag class of the cells of this row (one byte load when they sit in one block, the usual case); looked up per
        // row so that no table of the whole patch stays alive across the producer of the rows
        int cls[M];
        {
            const int nbh = q.hp->nbh, by = (cy * L.dby) >> 14;
            if (bx0 == bx1) {
                const int f = frow[j];
                const int k = HZL == 2 ? (f != 0) : ((f & 2) ? 2 : (f != 0));
#pragma unroll
                for (int i = 0; i < M; i++) cls[i] = k;
            } else {
#pragma unroll
                for (int i = 0; i < M; i++) {
                    int f = 0;
                    if (cx0 + i < wo && cy < ho) f = q.stable[by * nbh + (((cx0 + i) * L.dbx) >> 14)];
                    cls[i] = HZL == 2 ? (f != 0) : ((f & 2) ? 2 : (f != 0));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < M; i++) {
            const bool hasR = 2 * (cx0 + i) + 1 < ws;
            const int a = r0[2 * i];
            const int b = hasR ? r0[2 * i + 1] : a;
            const int c = hasB ? r1[2 * i] : a;
            const int d = hasB ? (hasR ? r1[2 * i + 1] : c) : b;
            const int ll = a + b + c + d;
            out[i] = scaled ? d_ll_down(ll) : ll;
            int vlh = a - b + c - d, vhl = a + b - c - d, vhh = a - b - c + d;
            if (HZL >= 1) {                             // shared cells sit on the first column / row of the bands
                if (i == 0 && chx && rowok && nR > 0) {
                    vlh = q_chain(q, HZL, wo, cy, vlh);
                    if (hasB) vhh = q_chain(q, HZL, wo, ho + cy, vhh);
                }
                if (j == 0 && chy && hasB) {
                    if (i < nC) vhl = q_chain(q, HZL, cx0 + i, ho, vhl);
                    if (i < nR && !(i == 0 && chx)) vhh = q_chain(q, HZL, wo + cx0 + i, ho, vhh);
                }
            }
            const int k = cls[i];
            const int qq = HZL == 2 ? (k ? L.sh1 : L.sh0) : max(L.qp >> k, HZ_MINQ);
            const float rc = HZL == 2 ? 0.f : __builtin_amdgcn_rcpf((float)(qq << 1));
            (void)q_coef<HZL>(qq, rc, vlh, slh[i]);
            (void)q_coef<HZL>(qq, rc, vhl, shl[i]);
            (void)q_coef<HZL>(qq, rc, vhh, shh[i]);
        }
        // only the symbols leave the chip: the inverse transform (k_inv_haar_tile<.,0,SYM>) dequantises them again
        if (rowok) {
            const int pr = cy * L.sw + cx0;                               // scan position = base + cy * sw + cx
            if (q.nzf) {
                // sparse: a P picture has a few thousand non-zeros among millions of cells -- one test per row, then
                // single stores (rare, and nothing waits for them)
                int any = 0;
#pragma unroll
                for (int i = 0; i < M; i++) any |= (i < nR ? slh[i] : 0) | ((hasB && i < nC) ? shl[i] : 0) | ((hasB && i < nR) ? shh[i] : 0);
                if (any) {
                    q.nz_any = 1;
#pragma unroll
                    for (int i = 0; i < M; i++) {
                        if (i < nR && slh[i] != 0) q.put_sparse(L.base0 + pr + i, slh[i]);
                        if (hasB && i < nC && shl[i] != 0) q.put_sparse(L.base1 + pr + i, shl[i]);
                        if (hasB && i < nR && shh[i] != 0) q.put_sparse(L.base2 + pr + i, shh[i]);
                    }
                }
            } else {
                store_sym_row<M>(q.sym + L.base0 + pr, slh, nR);
                if (hasB) {
                    store_sym_row<M>(q.sym + L.base1 + pr, shl, nC);
                    store_sym_row<M>(q.sym + L.base2 + pr, shh, nR);
                }
            }
        }
    }
};
template <int N, int HZL>
static __device__ __forceinline__ void haar_fwd_patch_q(const int (&in)[N][N], int (&out)[N / 2][N / 2],
                                                        int cx0, int cy0, int ws, int hs, int W,
                                                        int wo, int ho, int32_t *__restrict__ coef, bool scaled,
                                                        const QCtx &q)
{
    (void)W; (void)coef;
    HaarQ<N, HZL> hq;
    hq.init(q, cx0, cy0, ws, hs, wo, ho);
#pragma unroll
    for (int j = 0; j < N / 2; j++) hq.rows(q, j, in[2 * j], in[2 * j + 1], out[j], scaled);
}

// --------------------------------------------------------------------------------------------
// forward, P pictures: levels 1..3 straight from the 8-bit residual plane
// --------------------------------------------------------------------------------------------
template <bool Q>
__global__ __launch_bounds__(256) void k_fwd_haar_pix(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl,
                                                      int from_src)
{
    int job, c;
    d_job_plane((int)blockIdx.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const int I = blockIdx.x * 64 + threadIdx.x, J = blockIdx.y * 4 + threadIdx.y;
    if (I >= g.w3 || J >= g.h3) return;
    const JobDev &jb = jobs[job];
    const uint8_t *px = from_src ? jb.srcp[c] : jb.xf + g.poff;
    const int pxs = from_src ? jb.srcs[c] : g.pstride;
    int32_t *coef = jb.coef + g.coff;

    int a[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = 8 * J + r;
        uint2 v = make_uint2(0x80808080u, 0x80808080u);         // rows >= ph stay zero (p2sbc skips them)
        if (y < g.ph) v = *reinterpret_cast<const uint2 *>(px + (size_t)y * pxs + 8 * I);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            a[r][i] = (int)((v.x >> (8 * i)) & 0xff) - 128;
            a[r][i + 4] = (int)((v.y >> (8 * i)) & 0xff) - 128;
        }
    }
    const int W = g.W, H = g.H;
    const int wo1 = DSVG_RSU(W, 1), ho1 = DSVG_RSU(H, 1), wo2 = DSVG_RSU(W, 2), ho2 = DSVG_RSU(H, 2);
    int l1[4][4], l2[2][2], l3[1][1];
    QCtx q;
    if (Q) {
        const HzPlane &hp = jb.hz[c];
        q.hp = &hp; q.stable = dsvg_global(jb.stable);
        q.sym = dsvg_global(jb.sym + jb.nz_off[c]);
        q.any_ov = (2 * hp.s_w[0] > hp.s_w[1]) || (2 * hp.s_h[0] > hp.s_h[1]) ||
                   (2 * hp.s_w[1] > hp.s_w[2]) || (2 * hp.s_h[1] > hp.s_h[2]);
        q.nzf = jb.nzf ? dsvg_global(jb.nzf + (jb.nz_off[c] >> 2)) : nullptr;
        q.cfl = jb.nzf ? dsvg_global(jb.cflag + jb.chunk_off[c]) : nullptr;
    }
    // transform level 1,2,3 <-> scan level 2,1,0
    if (Q) {
        haar_fwd_patch_q<8, 2>(a, l1, 4 * I, 4 * J, W, H, W, wo1, ho1, coef, false, q);
        haar_fwd_patch_q<4, 1>(l1, l2, 2 * I, 2 * J, wo1, ho1, W, wo2, ho2, coef, true, q);
        haar_fwd_patch_q<2, 0>(l2, l3, I, J, wo2, ho2, W, g.w3, g.h3, coef, true, q);
    } else {
        haar_fwd_patch<8>(a, l1, 4 * I, 4 * J, W, H, W, wo1, ho1, coef, false);      // LVL_TEST: P level 1 unscaled
        haar_fwd_patch<4>(l1, l2, 2 * I, 2 * J, wo1, ho1, W, wo2, ho2, coef, true);
        haar_fwd_patch<2>(l2, l3, I, J, wo2, ho2, W, g.w3, g.h3, coef, true);
    }
    jb.s3[g.s3off + (size_t)J * g.w3 + I] = l3[0][0];
    if (Q && jb.nzf) jb.pflag[g.s3off + (size_t)J * g.w3 + I] = (uint8_t)q.nz_any;     // every patch, every picture: never stale
}

// --------------------------------------------------------------------------------------------
// forward, P pictures of the encoder, with the motion compensation done in place of the residual load:
// dsv_sub_pred (compensate bmc.c:204-302 + subf bmc.c:43-55) fused into levels 1..3 + quantiser.
// Every thread predicts its own 8x8 patch (it lies inside one block when the block sizes of the plane are multiples
// of 8 -- the launcher checks), subtracts it from the source, keeps the residual in registers and writes only the
// prediction (the inverse transform adds it back): the residual frame never exists, and k_mc's dependent round trips
// (vector -> reference -> source -> two stores) ride along with the transform arithmetic of the other waves.
// One code path for all four half-pel phases, so blocks with different vectors in one wave do not diverge:
//   luma   H = xh ? 9(b+c)-(a+d) : 16 b on 11 rows, V = yh ? 9(H1+H2)-(H0+H3) : 16 H1, sat8((V + 128) >> 8)
//          == hpelL bmc.c:124-174 in every phase: (16 t + 128) >> 8 == (t + 8) >> 4 and (256 p + 128) >> 8 == p;
//   chroma (a + B + C + D + 2) >> 2 with B/C/D falling back to a / b / c when a phase is off == hpel bmc.c:58-110.
// Intra blocks (mode != 0: block means, bmc.c:176-189,262-299) keep the two-kernel route: k_mc runs for them alone
// beforehand and their patches read its residual as k_fwd_haar_pix does.
// --------------------------------------------------------------------------------------------
struct __attribute__((aligned(4))) U4A4 { unsigned x, y, z, w; };     // 16 bytes at a dword-aligned address

typedef short s16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ s16x2 pk2(int a, int b) { return s16x2{(short)a, (short)b}; }
static __device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
static __device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
static __device__ __forceinline__ s16x2 pk_rdiv2(s16x2 v) { return (v + (short)1 + (v >> 15)) >> 1; }
static __device__ __forceinline__ s16x2 pk_rdiv4(s16x2 v) { return (v + (short)2 + (v >> 15)) >> 2; }
static __device__ __forceinline__ s16x2 pk_div4(s16x2 v) { return (v + ((v >> 15) & (short)3)) >> 2; }
// Eight 8-bit samples of a row as four registers of two int16 (v_pk_*_i16 works on two samples per instruction):
// e0 = (p0, p2), e1 = (p4, p6), o0 = (p1, p3), o1 = (p5, p7) -- a sample and its right neighbour sit in the same half of
// an e / o pair, which is what the horizontal steps (half-pel filters, Haar level 1) combine.
struct PkRow { s16x2 e0, e1, o0, o1; };
static __device__ __forceinline__ s16x2 pk_u(unsigned v) { return __builtin_bit_cast(s16x2, v); }
static __device__ __forceinline__ unsigned pk_b(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
#define PK_EVEN(hi_, lo_) pk_u((lo_) & 0x00ff00ffu)                               /* bytes (0, 2) of lo_ */
#define PK_ODD(hi_, lo_) pk_u(__builtin_amdgcn_perm(0u, (lo_), 0x0c030c01u))      /* bytes (1, 3) of lo_ */
#define PK_E2(hi_, lo_) pk_u(__builtin_amdgcn_perm((hi_), (lo_), 0x0c040c02u))    /* bytes (2, 4): 4 = byte 0 of hi_ */
#define PK_O2(hi_, lo_) pk_u(__builtin_amdgcn_perm((hi_), (lo_), 0x0c050c03u))    /* bytes (3, 5) */
static __device__ __forceinline__ PkRow pk_unpack8(unsigned lo, unsigned hi)
{
    return PkRow{PK_EVEN(0u, lo), PK_EVEN(0u, hi), PK_ODD(0u, lo), PK_ODD(0u, hi)};
}
static __device__ __forceinline__ void pk_bytes8(const PkRow &p, unsigned &lo, unsigned &hi)       // samples 0 .. 255
{
    lo = (pk_b(p.o0) << 8) | pk_b(p.e0);
    hi = (pk_b(p.o1) << 8) | pk_b(p.e1);
}
// luma, waves with horizontal half-pel phases only (mc_luma_patch<true, false> on two samples per instruction):
// t = (9 (b + c) - (a + d) + 8) >> 4 lies in -32 .. 287, the products in -510 .. 4598
template <typename EMITPK>
static __device__ __forceinline__ void mc_luma_patch_hpk(const DSVG_GLOBAL uint8_t *gr, int stride, bool xh, EMITPK emit_pk)
{
    const unsigned shb = (unsigned)(((uintptr_t)gr) & 3);
    const DSVG_GLOBAL uint8_t *ga = gr - shb;
    U4A4 rw[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { const dsvg_u32x4a4 t = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(ga + (r + 1) * stride); rw[r] = U4A4{t.x, t.y, t.z, t.w}; }
    __builtin_amdgcn_sched_barrier(0);
    const s16x2 z = s16x2{0, 0}, m255 = s16x2{255, 255};
    auto H = [&](s16x2 a, s16x2 b, s16x2 c, s16x2 d) {
        const s16x2 t = ((b + c) * (short)9 + (short)8 - (a + d)) >> 4;
        const s16x2 u = pk_min(pk_max(t, z), m255);
        return xh ? u : b;
    };
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const unsigned lo = __builtin_amdgcn_alignbyte(rw[r].y, rw[r].x, shb), mi = __builtin_amdgcn_alignbyte(rw[r].z, rw[r].y, shb),
                       hi = __builtin_amdgcn_alignbyte(rw[r].w, rw[r].z, shb);
        // A[m] = (byte m, byte m + 2) of the row's window; sample i reads bytes i .. i + 3
        const s16x2 A0 = PK_EVEN(0u, lo), A1 = PK_ODD(0u, lo), A2 = PK_E2(mi, lo), A3 = PK_O2(mi, lo), A4 = PK_EVEN(0u, mi), A5 = PK_ODD(0u, mi),
                    A6 = PK_E2(hi, mi), A7 = PK_O2(hi, mi), A8 = PK_EVEN(0u, hi);
        emit_pk(r, PkRow{H(A0, A1, A2, A3), H(A4, A5, A6, A7), H(A1, A2, A3, A4), H(A5, A6, A7, A8)});
    }
}

#define MCB(lo, mi, hi, m) ((int)((((m) < 4 ? (lo) : ((m) < 8 ? (mi) : (hi))) >> (8 * ((m) & 3))) & 0xff))
static __device__ __forceinline__ void mc_pack8(const int (&pv)[8], unsigned &plo, unsigned &phi)
{
    plo = (unsigned)pv[0] | ((unsigned)pv[1] << 8) | ((unsigned)pv[2] << 16) | ((unsigned)pv[3] << 24);
    phi = (unsigned)pv[4] | ((unsigned)pv[5] << 8) | ((unsigned)pv[6] << 16) | ((unsigned)pv[7] << 24);
}
// Luma prediction of one 8x8 patch, gr = &reference(wx-1, wy-1).  AX / AY: some lane of the wave has a horizontal /
// vertical half-pel phase (wave-uniform, so whole stages drop out for waves that do not need them); xh / yh: this lane's.
struct NoEmitInts {};
// emit_i (optional): takes the eight samples of a row as ints instead of packed bytes
template <bool AX, bool AY, typename EMIT, typename EMITI = NoEmitInts>
static __device__ __forceinline__ void mc_luma_patch(const DSVG_GLOBAL uint8_t *gr, int stride, bool xh, bool yh, EMIT emit, EMITI emit_i = EMITI{})
{
    auto out8 = [&](int r, const int (&pv)[8]) {
        if constexpr (std::is_same_v<EMITI, NoEmitInts>) {
            unsigned pl, ph_;
            mc_pack8(pv, pl, ph_);
            emit(r, pl, ph_);
        } else {
            emit_i(r, pv);
        }
    };
    const unsigned shb = (unsigned)(((uintptr_t)gr) & 3);
    const DSVG_GLOBAL uint8_t *ga = gr - shb;
    if (!AY) {
        // rows wy .. wy+7 only: copy, or the horizontal filter rounded on its own ((t + 8) >> 4)
        U4A4 rw[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { const dsvg_u32x4a4 t = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(ga + (r + 1) * stride); rw[r] = U4A4{t.x, t.y, t.z, t.w}; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned lo = __builtin_amdgcn_alignbyte(rw[r].y, rw[r].x, shb), mi = __builtin_amdgcn_alignbyte(rw[r].z, rw[r].y, shb),
                           hi = __builtin_amdgcn_alignbyte(rw[r].w, rw[r].z, shb);
            if (!AX) {
                emit(r, __builtin_amdgcn_alignbyte(mi, lo, 1u), __builtin_amdgcn_alignbyte(hi, mi, 1u));              // bytes 1..8
            } else {
                int bq[11], pv[8];
#pragma unroll
                for (int m = 0; m < 11; m++) bq[m] = MCB(lo, mi, hi, m);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int t = (9 * (bq[i + 1] + bq[i + 2]) - (bq[i] + bq[i + 3]) + 8) >> 4;
                    pv[i] = xh ? d_sat8(t) : bq[i + 1];
                }
                out8(r, pv);
            }
        }
    } else {
        // all phases in one body: H = xh ? 9(b+c)-(a+d) : 16 b on rows wy-1 .. wy+9, V = yh ? 9(H1+H2)-(H0+H3) : 16 H1
        U4A4 rw[11];
#pragma unroll
        for (int k = 0; k < 11; k++) { const dsvg_u32x4a4 t = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(ga + k * stride); rw[k] = U4A4{t.x, t.y, t.z, t.w}; }
        __builtin_amdgcn_sched_barrier(0);
        int Hq[4][8];
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const unsigned lo = __builtin_amdgcn_alignbyte(rw[k].y, rw[k].x, shb), mi = __builtin_amdgcn_alignbyte(rw[k].z, rw[k].y, shb),
                           hi = __builtin_amdgcn_alignbyte(rw[k].w, rw[k].z, shb);
            int bq[11];
#pragma unroll
            for (int m = 0; m < 11; m++) bq[m] = MCB(lo, mi, hi, m);
#pragma unroll
            for (int i = 0; i < 8; i++) Hq[k & 3][i] = (AX && xh) ? 9 * (bq[i + 1] + bq[i + 2]) - (bq[i] + bq[i + 3]) : 16 * bq[i + 1];
            if (k >= 3) {
                const int r = k - 3;
                int pv[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int h0 = Hq[r & 3][i], h1 = Hq[(r + 1) & 3][i], h2 = Hq[(r + 2) & 3][i], h3 = Hq[(r + 3) & 3][i];
                    const int v = yh ? 9 * (h1 + h2) - (h0 + h3) : 16 * h1;
                    pv[i] = d_sat8((v + 128) >> 8);
                }
                out8(r, pv);
            }
        }
    }
}
// luma, waves with a vertical half-pel phase (mc_luma_patch<AX, true>): the horizontal stage H = 9 (b + c) - (a + d) or 16 b
// (-510 .. 4590) on two samples per instruction, four rows of it kept as int16 pairs; the vertical stage pairs up
// u = h1 + h2, w = h0 + h3 (lanes without a vertical phase: u = w = 2 h1, 9 u - w = 16 h1) in 16 bits and finishes
// 9 u - w in 32 (17-bit sums)
template <bool AX, typename EMITPK>
static __device__ __forceinline__ void mc_luma_patch_vpk(const DSVG_GLOBAL uint8_t *gr, int stride, bool xh, bool yh, EMITPK emit_pk)
{
    const unsigned shb = (unsigned)(((uintptr_t)gr) & 3);
    const DSVG_GLOBAL uint8_t *ga = gr - shb;
    U4A4 rw[11];
#pragma unroll
    for (int k = 0; k < 11; k++) { const dsvg_u32x4a4 t = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(ga + k * stride); rw[k] = U4A4{t.x, t.y, t.z, t.w}; }
    __builtin_amdgcn_sched_barrier(0);
    auto Hrow = [&](s16x2 a, s16x2 b, s16x2 c, s16x2 d) {
        const s16x2 g = b << 4;
        if (!AX) return g;
        const s16x2 f = (b + c) * (short)9 - (a + d);
        return xh ? f : g;
    };
    auto V = [&](s16x2 a0, s16x2 a1, s16x2 a2, s16x2 a3) {
        const s16x2 u = a1 + (yh ? a2 : a1), w = (yh ? a0 : a1) + (yh ? a3 : a1);
        const int vx = (9 * (int)u.x - (int)w.x + 128) >> 8, vy = (9 * (int)u.y - (int)w.y + 128) >> 8;
        return pk2(d_sat8(vx), d_sat8(vy));
    };
    PkRow Hq[4];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const unsigned lo = __builtin_amdgcn_alignbyte(rw[k].y, rw[k].x, shb), mi = __builtin_amdgcn_alignbyte(rw[k].z, rw[k].y, shb),
                       hi = __builtin_amdgcn_alignbyte(rw[k].w, rw[k].z, shb);
        const s16x2 A0 = PK_EVEN(0u, lo), A1 = PK_ODD(0u, lo), A2 = PK_E2(mi, lo), A3 = PK_O2(mi, lo), A4 = PK_EVEN(0u, mi), A5 = PK_ODD(0u, mi),
                    A6 = PK_E2(hi, mi), A7 = PK_O2(hi, mi), A8 = PK_EVEN(0u, hi);
        Hq[k & 3] = PkRow{Hrow(A0, A1, A2, A3), Hrow(A4, A5, A6, A7), Hrow(A1, A2, A3, A4), Hrow(A5, A6, A7, A8)};
        if (k >= 3) {
            const int r = k - 3;
            const PkRow &h0 = Hq[r & 3], &h1 = Hq[(r + 1) & 3], &h2 = Hq[(r + 2) & 3], &h3 = Hq[(r + 3) & 3];
            emit_pk(r, PkRow{V(h0.e0, h1.e0, h2.e0, h3.e0), V(h0.e1, h1.e1, h2.e1, h3.e1), V(h0.o0, h1.o0, h2.o0, h3.o0), V(h0.o1, h1.o1, h2.o1, h3.o1)});
        }
    }
}
// Chroma prediction of one 8x8 patch, gr = &reference(wx-1, wy-1).  ANY: some lane of the wave has a half-pel phase.
// Bytes wx .. wx+8 of a row: 9 + misalignment <= 12, three dwords.
struct __attribute__((aligned(4))) U3A4 { unsigned x, y, z; };
template <bool ANY, typename EMIT>
static __device__ __forceinline__ void mc_chroma_patch(const DSVG_GLOBAL uint8_t *gr, int stride, bool xh, bool yh, EMIT emit)
{
    const DSVG_GLOBAL uint8_t *g1 = gr + 1;                               // reference (wx, wy-1)
    const unsigned shb = (unsigned)(((uintptr_t)g1) & 3);
    const DSVG_GLOBAL uint8_t *ga = g1 - shb;
    constexpr int NR = ANY ? 9 : 8;
    U3A4 rw[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) { const dsvg_u32x3a4 t = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x3a4 *>(ga + (k + 1) * stride); rw[k] = U3A4{t.x, t.y, t.z}; }
    __builtin_amdgcn_sched_barrier(0);
    unsigned qlo = 0, qhi = 0, qx = 0;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        // lo / hi = bytes wx .. wx+7, x = byte wx+8 (in its low byte)
        const unsigned lo = __builtin_amdgcn_alignbyte(rw[k].y, rw[k].x, shb), hi = __builtin_amdgcn_alignbyte(rw[k].z, rw[k].y, shb),
                       x8 = rw[k].z >> (8 * shb);
        if (!ANY) {
            emit(k, lo, hi);
        } else if (k >= 1) {
            int pv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int pa = MCB(qlo, qhi, qx, i), pb = MCB(qlo, qhi, qx, i + 1), pc = MCB(lo, hi, x8, i), pd = MCB(lo, hi, x8, i + 1);
                const int B = xh ? pb : pa, C = yh ? pc : pa, D = xh ? (yh ? pd : pb) : (yh ? pc : pa);
                pv[i] = (pa + B + C + D + 2) >> 2;
            }
            unsigned pl, ph_;
            mc_pack8(pv, pl, ph_);
            emit(k - 1, pl, ph_);
        }
        qlo = lo; qhi = hi; qx = x8;
    }
}

// chroma, waves with a half-pel phase, two samples per instruction: h = a + B per row (B = the right neighbour when the lane's
// vector has a horizontal phase), prediction = (h_above + (vertical phase ? h_below : h_above) + 2) >> 2 == mc_chroma_patch<true>
template <typename EMITPK>
static __device__ __forceinline__ void mc_chroma_patch_pk(const DSVG_GLOBAL uint8_t *gr, int stride, bool xh, bool yh, EMITPK emit_pk)
{
    const DSVG_GLOBAL uint8_t *g1 = gr + 1;                               // reference (wx, wy-1)
    const unsigned shb = (unsigned)(((uintptr_t)g1) & 3);
    const DSVG_GLOBAL uint8_t *ga = g1 - shb;
    U3A4 rw[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { const dsvg_u32x3a4 t = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x3a4 *>(ga + (k + 1) * stride); rw[k] = U3A4{t.x, t.y, t.z}; }
    __builtin_amdgcn_sched_barrier(0);
    const s16x2 two = s16x2{2, 2};
    PkRow hp;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const unsigned lo = __builtin_amdgcn_alignbyte(rw[k].y, rw[k].x, shb), hi = __builtin_amdgcn_alignbyte(rw[k].z, rw[k].y, shb),
                       x8 = rw[k].z >> (8 * shb);
        const s16x2 A0 = PK_EVEN(0u, lo), A1 = PK_ODD(0u, lo), A2 = PK_E2(hi, lo), A4 = PK_EVEN(0u, hi), A5 = PK_ODD(0u, hi), A6 = PK_E2(x8, hi);
        const PkRow h = PkRow{A0 + (xh ? A1 : A0), A4 + (xh ? A5 : A4), A1 + (xh ? A2 : A1), A5 + (xh ? A6 : A5)};
        if (k >= 1)
            emit_pk(k - 1, PkRow{(hp.e0 + (yh ? h.e0 : hp.e0) + two) >> 2, (hp.e1 + (yh ? h.e1 : hp.e1) + two) >> 2,
                                 (hp.o0 + (yh ? h.o0 : hp.o0) + two) >> 2, (hp.o1 + (yh ? h.o1 : hp.o1) + two) >> 2});
        hp = h;
    }
}

// ---- the common patch of a sparse P picture: inter block, whole 8x8 patch inside the picture, every cell of each level in
// ONE block of the stability map, no cell shared between scan regions.  Same arithmetic as the general body of
// k_fwd_mc_pix below (compensate + subf + fwd + hzcc quantiser), with every "does this sample / cell / band exist" fact a
// constant: ~600 instructions per patch instead of ~1900.
struct FwdFastQ {
    int sh1[4];                 // level 1: shift of each cell row (hzcc.c:221-224)
    int q2[2], q3;              // levels 2, 3: quantiser of each cell row (tmq4pos hzcc.c:64-74)
    float rc2[2], rc3;
};
// trunc(v / 2^sh) == sign(v) * (|v| >> sh): the level-1 symbol
static __device__ __forceinline__ int sym_shift(int v, int sh) { return (v + ((v >> 31) & ((1 << sh) - 1))) >> sh; }
// the four level-1 symbols of one cell row and band of a patch are neighbours in the symbol plane (scan positions pos .. pos + 3): ONE 8-byte
// store and one flag byte of each kind when the group is aligned (band widths and scan bases that are multiples of 4: every broadcast size),
// instead of a 2-byte store and two flag stores per non-zero symbol -- with every patch of a picture flagged (the dense-residual shape) the
// lean kernel spent its time issuing up to 189 stores per patch.  Zeros may be stored: the plane is zero between pictures, the bytes in
// memory are the same either way.  (DSV1 A/B: -DFWD_FAST_NO_ROW4)
static __device__ __forceinline__ void put_row4(const QCtx &q, int pos, s16x2 ab, s16x2 cd)      // the same with the symbols as int16 pairs (cells 0, 1 / 2, 3)
{
    const unsigned x = __builtin_bit_cast(unsigned, ab), y = __builtin_bit_cast(unsigned, cd);
    if (!(x | y)) return;
#ifndef FWD_FAST_NO_ROW4
    if ((pos & 3) == 0) {
        typedef unsigned row4_t __attribute__((ext_vector_type(2)));
        row4_t v;
        v.x = x; v.y = y;
        *reinterpret_cast<DSVG_GLOBAL row4_t *>(q.sym + pos) = v;
        q.nzf[pos >> 2] = 1;
        q.cfl[pos >> 11] = 1;
        return;
    }
#endif
    if (ab.x) q.put_sparse(pos, ab.x);
    if (ab.y) q.put_sparse(pos + 1, ab.y);
    if (cd.x) q.put_sparse(pos + 2, cd.x);
    if (cd.y) q.put_sparse(pos + 3, cd.y);
}
static __device__ __forceinline__ void put_row4(const QCtx &q, int pos, int a, int b, int c, int d)
{
    if (!(a | b | c | d)) return;
#ifndef FWD_FAST_NO_ROW4
    if ((pos & 3) == 0) {
        typedef unsigned row4_t __attribute__((ext_vector_type(2)));
        row4_t v;
        v.x = ((unsigned)a & 0xffffu) | ((unsigned)b << 16);
        v.y = ((unsigned)c & 0xffffu) | ((unsigned)d << 16);
        *reinterpret_cast<DSVG_GLOBAL row4_t *>(q.sym + pos) = v;
        q.nzf[pos >> 2] = 1;
        q.cfl[pos >> 11] = 1;
        return;
    }
#endif
    if (a) q.put_sparse(pos, a);
    if (b) q.put_sparse(pos + 1, b);
    if (c) q.put_sparse(pos + 2, c);
    if (d) q.put_sparse(pos + 3, d);
}

// level 1 of one pair of residual rows (cell row cy1 of the plane, cells cx1 .. cx1 + 3) -> LL row + sparse symbols
static __device__ __forceinline__ void fwd_fast_rows1(const QCtx &q, const QLevel &L1, int sh1, int cx1, int cy1,
                                                      const int (&r0)[8], const int (&r1)[8], int (&out)[4])
{
    int mx = max(r0[0], r0[1]), mn = min(r0[0], r0[1]);
#pragma unroll
    for (int i = 2; i < 8; i += 2) { mx = max(mx, max(r0[i], r0[i + 1])); mn = min(mn, min(r0[i], r0[i + 1])); }
#pragma unroll
    for (int i = 0; i < 8; i += 2) { mx = max(mx, max(r1[i], r1[i + 1])); mn = min(mn, min(r1[i], r1[i + 1])); }
    if (4 * max(mx, -mn) < (1 << sh1)) {                // |detail| <= 4 max|sample| < 2^shift: every symbol is zero
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = (r0[2 * i] + r0[2 * i + 1]) + (r1[2 * i] + r1[2 * i + 1]);
        return;
    }
    int slh[4], shl[4], shh[4], any = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s0 = r0[2 * i] + r0[2 * i + 1], s1 = r1[2 * i] + r1[2 * i + 1];
        const int d0 = r0[2 * i] - r0[2 * i + 1], d1 = r1[2 * i] - r1[2 * i + 1];
        out[i] = s0 + s1;                               // level 1 of a P picture is unscaled (LVL_TEST sbt.c:22)
        slh[i] = sym_shift(d0 + d1, sh1); shl[i] = sym_shift(s0 - s1, sh1); shh[i] = sym_shift(d0 - d1, sh1);
        any |= slh[i] | shl[i] | shh[i];
    }
    if (any) {
        q.nz_any = 1;
        const int pr = cy1 * L1.sw + cx1;
        put_row4(q, L1.base0 + pr, slh[0], slh[1], slh[2], slh[3]);
        put_row4(q, L1.base1 + pr, shl[0], shl[1], shl[2], shl[3]);
        put_row4(q, L1.base2 + pr, shh[0], shh[1], shh[2], shh[3]);
    }
}
// the same on packed rows: 16 v_pk instructions for the four cells; the all-zero test is exact (a symbol is zero iff
// |detail| < 2^shift; details are at most 1020)
static __device__ __forceinline__ void fwd_fast_rows1_pk(const QCtx &q, const QLevel &L1, int sh1, int cx1, int cy1,
                                                         const PkRow &r0, const PkRow &r1, int (&out)[4])
{
    const s16x2 s0a = r0.e0 + r0.o0, s0b = r0.e1 + r0.o1, d0a = r0.e0 - r0.o0, d0b = r0.e1 - r0.o1;
    const s16x2 s1a = r1.e0 + r1.o0, s1b = r1.e1 + r1.o1, d1a = r1.e0 - r1.o0, d1b = r1.e1 - r1.o1;
    const s16x2 lla = s0a + s1a, llb = s0b + s1b;              // cells (0, 1), (2, 3); level 1 of a P picture is unscaled (LVL_TEST sbt.c:22)
    const s16x2 lha = d0a + d1a, lhb = d0b + d1b, hla = s0a - s1a, hlb = s0b - s1b, hha = d0a - d1a, hhb = d0b - d1b;
    out[0] = lla.x; out[1] = lla.y; out[2] = llb.x; out[3] = llb.y;
    const s16x2 mx = pk_max(pk_max(pk_max(lha, lhb), pk_max(hla, hlb)), pk_max(hha, hhb));
    const s16x2 mn = pk_min(pk_min(pk_min(lha, lhb), pk_min(hla, hlb)), pk_min(hha, hhb));
    const s16x2 am = pk_max(mx, s16x2{0, 0} - mn);
    if ((pk_b(am) & ~(((1u << min(sh1, 15)) - 1u) * 0x00010001u)) == 0) return;
    q.nz_any = 1;
    const int pr = cy1 * L1.sw + cx1;
    // (sh1 <= 10 here: with a larger shift every symbol is zero and the test above has returned; the symbols of a cell pair by packed
    // instructions -- trunc(v / 2^sh) as in sym_shift -- are the two halves of the dword that is stored)
    const short ms = (short)((1 << sh1) - 1), ss = (short)sh1;
    const s16x2 mm = {ms, ms}, sv = {ss, ss}, s15 = {15, 15};
    auto pks = [&](s16x2 v) { return (v + ((v >> s15) & mm)) >> sv; };
    put_row4(q, L1.base0 + pr, pks(lha), pks(lhb));
    put_row4(q, L1.base1 + pr, pks(hla), pks(hlb));
    put_row4(q, L1.base2 + pr, pks(hha), pks(hhb));
}
// one scaled level (transform levels 2, 3: scan levels 1, 0) on an NxN patch of LL values -> LL patch + sparse symbols
template <int N, int HZL>
static __device__ __forceinline__ void fwd_fast_level(const QCtx &q, const QLevel &L, const int (&qqv)[N / 2], const float (&rcv)[N / 2], int cx0, int cy0,
                                                      const int (&in)[N][N], int (&out)[N / 2][N / 2])
{
#pragma unroll
    for (int j = 0; j < N / 2; j++)
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            const int qq = qqv[j];
            const float rc = rcv[j];
            const int a = in[2 * j][2 * i], b = in[2 * j][2 * i + 1], c = in[2 * j + 1][2 * i], d = in[2 * j + 1][2 * i + 1];
            const int s0 = a + b, s1 = c + d, d0 = a - b, d1 = c - d;
            out[j][i] = d_ll_down(s0 + s1);
            // a symbol is floor((2 |v| + 1) / 2q): zero iff |v| < q -- one test per cell in front of the three divisions
            const int vlh = d0 + d1, vhl = s0 - s1, vhh = d0 - d1;
#ifndef FWD_FAST_NO_QTEST
            if (max(max(max(vlh, vhl), vhh), -min(min(vlh, vhl), vhh)) < qq) continue;
#endif
            int slh, shl, shh;
            (void)q_coef<HZL>(qq, rc, vlh, slh);
            (void)q_coef<HZL>(qq, rc, vhl, shl);
            (void)q_coef<HZL>(qq, rc, vhh, shh);
            if (slh | shl | shh) {
                q.nz_any = 1;
                const int pr = (cy0 + j) * L.sw + cx0 + i;
                if (slh) q.put_sparse(L.base0 + pr, slh);
                if (shl) q.put_sparse(L.base1 + pr, shl);
                if (shh) q.put_sparse(L.base2 + pr, shh);
            }
        }
}

// which patches take the lean kernel: sparse P picture, inter block, the whole 8x8 patch inside the picture, no cell shared
// between scan regions (they sit in the first row / column of patches), and ONE stability flag per level -- the cells of
// the patch lie in one block of the map except where the fixed-point block steps of hzcc.c:196-197 round across an edge
// (FWD_FAST_INTRA: the lean kernel also takes the patches of INTRA blocks that meet the other conditions -- their residual is in the work
// frame already, written by k_mc with the block means as prediction; 0 = they go to the general kernel, over the whole grid, as before)
#ifndef FWD_FAST_INTRA
#define FWD_FAST_INTRA 1
#endif
struct FwdFastSel { bool ok, intra; QLevel L1, L2, L3; int i1[4], i2[2], i3; };
// what a patch's selection and quantiser need of the job table (HzPlane of its plane): read in ONE batch by the callers that care
// (k_fwd_mc_fast: the compiler otherwise fetches each field where it is first used, behind the short-circuit tests below -- a dozen
// dependent scalar-memory round trips in front of the patch's first pixel load)
struct FwdFastHp { QLevel L1, L2, L3; int nbh; bool ovx, ovy, sparse; };
static __device__ __forceinline__ FwdFastHp fwd_fast_hp(const JobDev &jb, const HzPlane &hp)
{
    FwdFastHp H;
    const int sw0 = hp.s_w[0], sw1 = hp.s_w[1], sw2 = hp.s_w[2], sh0 = hp.s_h[0], sh1 = hp.s_h[1], sh2 = hp.s_h[2];
    H.L1 = q_level<2>(hp); H.L2 = q_level<1>(hp); H.L3 = q_level<0>(hp);
    H.nbh = hp.nbh;
    // cells shared between scan regions (hzcc.c:30-48 rounds the region sizes up at every level): column 0 of the LH / HH
    // bands of a level whose width is odd, row 0 of its HL / HH bands when its height is odd -- first column / row of patches
    H.ovx = (int)(2 * sw0 > sw1) | (int)(2 * sw1 > sw2);
    H.ovy = (int)(2 * sh0 > sh1) | (int)(2 * sh1 > sh2);
    H.sparse = jb.nzf != nullptr;
    return H;
}
static __device__ __forceinline__ FwdFastSel fwd_fast_sel(const FwdFastHp &H, int mode, int I, int J, int x0, int y0, int pw, int ph)
{
    // The map's block rows are hp.nbv / region height apart in coefficient space, not blk_h: a picture whose height is not
    // a multiple of the block height makes them drift against the real block grid, so a patch often spans two map rows.
    // One flag per CELL ROW of each level covers that; only a patch that spans two map COLUMNS goes to the general kernel.
    FwdFastSel S;
    S.ok = false; S.intra = mode != 0;
    if (H.sparse && (FWD_FAST_INTRA || mode == 0) && x0 + 8 <= pw && y0 + 8 <= ph && !(H.ovx && I == 0) && !(H.ovy && J == 0)) {
        S.L1 = H.L1; S.L2 = H.L2; S.L3 = H.L3;
        const int nbh = H.nbh;
        const int b1x = (4 * I * S.L1.dbx) >> 14, b2x = (2 * I * S.L2.dbx) >> 14;
        S.ok = b1x == ((4 * I + 3) * S.L1.dbx) >> 14 && b2x == ((2 * I + 1) * S.L2.dbx) >> 14;
#pragma unroll
        for (int j = 0; j < 4; j++) S.i1[j] = (((4 * J + j) * S.L1.dby) >> 14) * nbh + b1x;
#pragma unroll
        for (int j = 0; j < 2; j++) S.i2[j] = (((2 * J + j) * S.L2.dby) >> 14) * nbh + b2x;
        S.i3 = ((J * S.L3.dby) >> 14) * nbh + ((I * S.L3.dbx) >> 14);
    }
    return S;
}
static __device__ __forceinline__ FwdFastSel fwd_fast_sel(const JobDev &jb, const HzPlane &hp, bool any_ov, int mode, int I, int J,
                                                          int x0, int y0, int pw, int ph)
{
    (void)any_ov;
    return fwd_fast_sel(fwd_fast_hp(jb, hp), mode, I, J, x0, y0, pw, ph);
}

template <int CH>
static __device__ __forceinline__ void fwd_mc_fast_body(const JobDev *__restrict__ jobs, const SbtGeo3 &G, const McGeo &MG, int c0, int npl,
                                                        const DMV *__restrict__ mvs0, XcdGrid XG, int plain)
{
    // one-dimensional launch in XCD order (d_xcd_blk3): the reference rows above and below a workgroup's 32 pixel rows and the
    // lines its rows share with the workgroup beside it are in the same L2 as the neighbour that reads them too
    Blk3 B;
    if (!d_xcd_blk3(XG, B, plain != 0)) return;
    int job, c;
    d_job_plane(B.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const int I = B.x * 64 + threadIdx.x, J = B.y * 4 + threadIdx.y;
    if (I >= g.w3 || J >= g.h3) return;
    const JobDev &jb = jobs[job];
    const int sh = CH ? MG.hs : 0, sv = CH ? MG.vs : 0;
    const int bw = MG.blk_w >> sh, bh = MG.blk_h >> sv;
    const int pw = MG.w[c], ph = g.ph, stride = g.pstride;
    const int x0 = 8 * I, y0 = 8 * J;
    const int bi = (int)(((float)I + 0.5f) * __builtin_amdgcn_rcpf((float)(bw >> 3)));
    const int bj = (int)(((float)J + 0.5f) * __builtin_amdgcn_rcpf((float)(bh >> 3)));
    const int nblk = MG.nbh * MG.nbv, blk = bj * MG.nbh + bi;
    // the job table's part of this patch, requested in one go (FwdFastHp) and held where it is: the pins keep the compiler from moving
    // each fetch down to its first use
    const HzPlane &hp = jb.hz[c];
    FwdFastHp H = fwd_fast_hp(jb, hp);
    const DMV *mvtab = jb.mvs;
    const uint8_t *stable_p = jb.stable, *srcp_p = jb.srcp[c];
    int sstride = jb.srcs[c];
#define FWD_PIN(x) asm volatile("" : "+s"(x))
#ifndef FWD_FAST_NO_PIN
    FWD_PIN(H.L1.sh0); FWD_PIN(H.L1.sh1); FWD_PIN(H.L1.dbx); FWD_PIN(H.L1.dby); FWD_PIN(H.L2.qp); FWD_PIN(H.L2.dbx); FWD_PIN(H.L2.dby);
    FWD_PIN(H.L3.qp); FWD_PIN(H.L3.dbx); FWD_PIN(H.L3.dby); FWD_PIN(H.nbh); FWD_PIN(mvtab); FWD_PIN(stable_p); FWD_PIN(srcp_p); FWD_PIN(sstride);
#endif
    DMV mv;                                              // (x, y, mode: global loads -- through the generic pointer they were flat loads, which the flag loads behind them cannot overtake)
    {
        const auto mq = reinterpret_cast<const DSVG_GLOBAL char *>(dsvg_global(mvs0 ? mvs0 + (size_t)job * nblk : mvtab)) + (unsigned)blk * (unsigned)sizeof(DMV);
        const unsigned xy = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(mq);
        mv.x = (int16_t)(xy & 0xffffu); mv.y = (int16_t)(xy >> 16); mv.mode = *reinterpret_cast<const DSVG_GLOBAL uint8_t *>(mq + 4);
    }
    QCtx q;
    q.hp = &hp; q.stable = dsvg_global(stable_p);
    q.sym = dsvg_global(jb.sym + jb.nz_off[c]);
    q.any_ov = H.ovx | H.ovy;
    q.nzf = H.sparse ? dsvg_global(jb.nzf + (jb.nz_off[c] >> 2)) : nullptr;
    q.cfl = H.sparse ? dsvg_global(jb.cflag + jb.chunk_off[c]) : nullptr;
    const FwdFastSel S = fwd_fast_sel(H, mv.mode, I, J, x0, y0, pw, ph);
    if (!S.ok) return;                                   // k_fwd_mc_pix takes these
    const QLevel &L1 = S.L1, &L2 = S.L2, &L3 = S.L3;
    // (the source rows are requested before the stability flags: whatever the compiler then waits for on the flags' account does not hold them back)
    const auto sp = dsvg_global(srcp_p);
    uint2 sw[8];
    if (S.intra) {                                      // an intra block's patch: the residual rows, from the work frame (k_mc wrote them and the prediction)
        const auto px = dsvg_global(static_cast<const uint8_t *>(jb.xf + g.poff));
#pragma unroll
        for (int r = 0; r < 8; r++) sw[r] = dsvg_ld2(px + (unsigned)((y0 + r) * stride + x0));
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) sw[r] = dsvg_ld2(sp + (unsigned)((y0 + r) * sstride + x0));
    }
    int f1[4], f2[2];
#pragma unroll
    for (int j = 0; j < 4; j++) f1[j] = q.stable[S.i1[j]];
#pragma unroll
    for (int j = 0; j < 2; j++) f2[j] = q.stable[S.i2[j]];
    const int f3 = q.stable[S.i3];
    // (the stability flags stay raw until a level needs them: turned into shifts / quantisers here, they were waited for before the
    // reference rows below could be requested -- a second memory round trip behind the source rows')
    FwdFastQ fq;
    const int dx = mv.x >> sh, dy = mv.y >> sv;
    const int xb = bi * bw, yb = bj * bh;
    const int wx = d_clamp(xb + (dx >> 1), -DSVG_BORDER, pw - bw + DSVG_BORDER - 1) + (x0 - xb);
    const int wy = d_clamp(yb + (dy >> 1), -DSVG_BORDER, ph - bh + DSVG_BORDER - 1) + (y0 - yb);
    const bool xh = !S.intra && (dx & 1), yh = !S.intra && (dy & 1);
    const auto gr = dsvg_global(static_cast<const uint8_t *>(jb.ref + g.poff)) + ((wy - 1) * stride + (wx - 1));
    const bool any_x = __ballot(xh) != 0ull, any_y = __ballot(yh) != 0ull;     // over the lanes on this path that predict
    const auto pp = dsvg_global(jb.pred + g.poff);
#ifndef FWD_FAST_NO_PK
    // residual rows and level 1 on two samples per instruction; the half-pel paths of the bench's kind (luma: horizontal
    // phases; chroma: any) hand their prediction over in that form already
    int l1[4][4];
    PkRow rp[2];
    const s16x2 cmin = s16x2{-128, -128}, cmax = s16x2{127, 127};
    auto emit_core = [&](int r, const PkRow &P, unsigned plo, unsigned phi) {
        const PkRow Sr = pk_unpack8(sw[r].x, sw[r].y);
        PkRow &R = rp[r & 1];
        R.e0 = pk_min(pk_max(Sr.e0 - P.e0, cmin), cmax); R.e1 = pk_min(pk_max(Sr.e1 - P.e1, cmin), cmax);       // subf bmc.c:43-55 + p2sbc sbt.c:576
        R.o0 = pk_min(pk_max(Sr.o0 - P.o0, cmin), cmax); R.o1 = pk_min(pk_max(Sr.o1 - P.o1, cmin), cmax);
        dsvg_st2(pp + (unsigned)((y0 + r) * stride + x0), make_uint2(plo, phi));
        if (r & 1) fwd_fast_rows1_pk(q, L1, f1[r >> 1] ? L1.sh1 : L1.sh0, 4 * I, 4 * J + (r >> 1), rp[0], rp[1], l1[r >> 1]);
    };
    auto emit = [&](int r, unsigned plo, unsigned phi) { emit_core(r, pk_unpack8(plo, phi), plo, phi); };
    auto emit_pk = [&](int r, const PkRow &P) {
        unsigned plo, phi;
        pk_bytes8(P, plo, phi);
        emit_core(r, P, plo, phi);
    };
    if (S.intra) {
        // p2sbc of the stored residual: byte - 128 (sbt.c:576) -- the rows the inter path forms from source and prediction
        const s16x2 c128 = s16x2{128, 128};
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const PkRow X = pk_unpack8(sw[r].x, sw[r].y);
            PkRow &R = rp[r & 1];
            R.e0 = X.e0 - c128; R.e1 = X.e1 - c128; R.o0 = X.o0 - c128; R.o1 = X.o1 - c128;
            if (r & 1) fwd_fast_rows1_pk(q, L1, f1[r >> 1] ? L1.sh1 : L1.sh0, 4 * I, 4 * J + (r >> 1), rp[0], rp[1], l1[r >> 1]);
        }
    } else
    // (the vertical filter's 17-bit sums stay in 32-bit lanes: its samples arrive as ints and are paired up here -- never
    // through packed bytes: the compiler turns sat8(x >> 8) pairs followed by a byte merge into v_ashr_pk_u8_i32 and takes
    // bits 31:16 of its result for zero, which gfx950 leaves as they were; see the ISA check in the Makefile)
#ifdef FWD_FAST_NO_VPK
    auto emit_i = [&](int r, const int (&pv)[8]) { emit_pk(r, PkRow{pk2(pv[0], pv[2]), pk2(pv[4], pv[6]), pk2(pv[1], pv[3]), pk2(pv[5], pv[7])}); };
#endif
    if (CH == 0) {
#ifdef FWD_FAST_NO_VPK
        if (any_y) mc_luma_patch<true, true>(gr, stride, xh, yh, emit, emit_i);
#else
        if (any_y && any_x) mc_luma_patch_vpk<true>(gr, stride, xh, yh, emit_pk);
        else if (any_y) mc_luma_patch_vpk<false>(gr, stride, xh, yh, emit_pk);
#endif
        else if (any_x) mc_luma_patch_hpk(gr, stride, xh, emit_pk);
        else mc_luma_patch<false, false>(gr, stride, xh, yh, emit);
    } else {
        if (any_x || any_y) mc_chroma_patch_pk(gr, stride, xh, yh, emit_pk);
        else mc_chroma_patch<false>(gr, stride, xh, yh, emit);
    }
#else
    int l1[4][4], ra[2][8];
    auto emit = [&](int r, unsigned plo, unsigned phi) {
        int (&row)[8] = ra[r & 1];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int sv_ = (int)(((i < 4 ? sw[r].x : sw[r].y) >> (8 * (i & 3))) & 0xff);
            const int pv_ = (int)(((i < 4 ? plo : phi) >> (8 * (i & 3))) & 0xff);
            row[i] = d_clamp(sv_ - pv_, -128, 127);       // subf bmc.c:43-55 + p2sbc sbt.c:576
        }
        dsvg_st2(pp + (unsigned)((y0 + r) * stride + x0), make_uint2(plo, phi));
        if (r & 1) fwd_fast_rows1(q, L1, f1[r >> 1] ? L1.sh1 : L1.sh0, 4 * I, 4 * J + (r >> 1), ra[0], ra[1], l1[r >> 1]);
    };
    if (CH == 0) {
        if (any_y) mc_luma_patch<true, true>(gr, stride, xh, yh, emit);
        else if (any_x) mc_luma_patch<true, false>(gr, stride, xh, yh, emit);
        else mc_luma_patch<false, false>(gr, stride, xh, yh, emit);
    } else {
        if (any_x || any_y) mc_chroma_patch<true>(gr, stride, xh, yh, emit);
        else mc_chroma_patch<false>(gr, stride, xh, yh, emit);
    }
#endif
#pragma unroll
    for (int j = 0; j < 2; j++) {
        fq.q2[j] = max(L2.qp >> ((f2[j] & 2) ? 2 : (f2[j] != 0)), HZ_MINQ);
        fq.rc2[j] = __builtin_amdgcn_rcpf((float)(fq.q2[j] << 1));
    }
    fq.q3 = max(L3.qp >> ((f3 & 2) ? 2 : (f3 != 0)), HZ_MINQ);
    fq.rc3 = __builtin_amdgcn_rcpf((float)(fq.q3 << 1));
    int l2[2][2], l3[1][1];
    fwd_fast_level<4, 1>(q, L2, fq.q2, fq.rc2, 2 * I, 2 * J, l1, l2);
    const int q3v[1] = {fq.q3};
    const float rc3v[1] = {fq.rc3};
    fwd_fast_level<2, 0>(q, L3, q3v, rc3v, I, J, l2, l3);
    dsvg_global(jb.s3 + g.s3off)[(unsigned)(J * g.w3 + I)] = l3[0][0];
    dsvg_global(jb.pflag + g.s3off)[(unsigned)(J * g.w3 + I)] = (uint8_t)q.nz_any;
}
// Occupancy per plane kind, measured (160-GOP step): luma 4 waves per SIMD 2.60 ms, 5 (what its 96 VGPRs would give) 2.65,
// 3: 2.98; chroma at the 5 its registers give 1.31, 4: 1.31-1.39, 6 (spills): 1.72
#ifndef FAST_WPE_L
#define FAST_WPE_L 4
#endif
#ifndef FAST_WPE_C
#define FAST_WPE_C 0
#endif
#if FAST_WPE_L > 0
#define FAST_WPE_L_ATTR __attribute__((amdgpu_waves_per_eu(FAST_WPE_L, FAST_WPE_L)))
#else
#define FAST_WPE_L_ATTR
#endif
#if FAST_WPE_C > 0
#define FAST_WPE_C_ATTR __attribute__((amdgpu_waves_per_eu(FAST_WPE_C, FAST_WPE_C)))
#else
#define FAST_WPE_C_ATTR
#endif
template <int CH>
__global__ void k_fwd_mc_fast(const JobDev *__restrict__ jobs, SbtGeo3 G, McGeo MG, int c0, int npl, const DMV *__restrict__ mvs0, XcdGrid XG, int plain);
template <>
__global__ __launch_bounds__(256) FAST_WPE_L_ATTR void k_fwd_mc_fast<0>(const JobDev *__restrict__ jobs, SbtGeo3 G, McGeo MG, int c0, int npl,
                                                                        const DMV *__restrict__ mvs0, XcdGrid XG, int plain)
{
    DSVG_CLK_BEGIN();
    fwd_mc_fast_body<0>(jobs, G, MG, c0, npl, mvs0, XG, plain);
    DSVG_CLK_END(1);
}
template <>
__global__ __launch_bounds__(256) FAST_WPE_C_ATTR void k_fwd_mc_fast<1>(const JobDev *__restrict__ jobs, SbtGeo3 G, McGeo MG, int c0, int npl,
                                                                        const DMV *__restrict__ mvs0, XcdGrid XG, int plain)
{
    DSVG_CLK_BEGIN();
    fwd_mc_fast_body<1>(jobs, G, MG, c0, npl, mvs0, XG, plain);
    DSVG_CLK_END(2);
}

// ---- decoder: prediction only (compensate bmc.c:204-302 without subf), the lean kernel's motion compensation by itself ----
// One thread per 8x8 patch that lies wholly inside the picture, in an INTER block left of block column ex0 and above block row
// ey0 (the blocks from there on hold ragged patches in some plane; they and the intra blocks are k_mc's, by list: the two
// kernels share the predicate).  Same row emitters as k_fwd_mc_fast: one body for all half-pel phases, stages dropping out
// wave-uniformly; the prediction goes straight to JobDev.pred.
template <int CH>
__global__ __launch_bounds__(256) void k_mc_patch(const JobDev *__restrict__ jobs, SbtGeo3 G, McGeo MG, int c0, int npl, const DMV *__restrict__ mvs0,
                                                  XcdGrid XG, int ex0, int ey0)
{
    Blk3 B;
    if (!d_xcd_blk3(XG, B)) return;
    int job, c;
    d_job_plane(B.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const int I = B.x * 64 + threadIdx.x, J = B.y * 4 + threadIdx.y;
    if (I >= g.w3 || J >= g.h3) return;
    const JobDev &jb = jobs[job];
    const int sh = CH ? MG.hs : 0, sv = CH ? MG.vs : 0;
    const int bw = MG.blk_w >> sh, bh = MG.blk_h >> sv;
    const int pw = MG.w[c], ph = g.ph, stride = g.pstride;
    const int x0 = 8 * I, y0 = 8 * J;
    if (x0 + 8 > pw || y0 + 8 > ph) return;                 // a ragged patch: its block is on k_mc's list
    const int bi = (int)(((float)I + 0.5f) * __builtin_amdgcn_rcpf((float)(bw >> 3)));
    const int bj = (int)(((float)J + 0.5f) * __builtin_amdgcn_rcpf((float)(bh >> 3)));
    if (bi >= ex0 || bj >= ey0) return;
    const int nblk = MG.nbh * MG.nbv, blk = bj * MG.nbh + bi;
    const DMV mv = mvs0 ? mvs0[(size_t)job * nblk + blk] : jb.mvs[blk];
    if (mv.mode != 0) return;                               // intra blocks: k_mc (block means, bmc.c:176-189)
    const int dx = mv.x >> sh, dy = mv.y >> sv;
    const int xb = bi * bw, yb = bj * bh;
    const int wx = d_clamp(xb + (dx >> 1), -DSVG_BORDER, pw - bw + DSVG_BORDER - 1) + (x0 - xb);
    const int wy = d_clamp(yb + (dy >> 1), -DSVG_BORDER, ph - bh + DSVG_BORDER - 1) + (y0 - yb);
    const bool xh = dx & 1, yh = dy & 1;
    const auto gr = dsvg_global(static_cast<const uint8_t *>(jb.ref + g.poff)) + ((wy - 1) * stride + (wx - 1));
    const bool any_x = __ballot(xh) != 0ull, any_y = __ballot(yh) != 0ull;
    const auto pp = dsvg_global(jb.pred + g.poff);
    auto emit = [&](int r, unsigned plo, unsigned phi) { dsvg_st2(pp + (unsigned)((y0 + r) * stride + x0), make_uint2(plo, phi)); };
    auto emit_pk = [&](int r, const PkRow &P) {
        unsigned plo, phi;
        pk_bytes8(P, plo, phi);
        dsvg_st2(pp + (unsigned)((y0 + r) * stride + x0), make_uint2(plo, phi));
    };
    if (CH == 0) {
        if (any_y && any_x) mc_luma_patch_vpk<true>(gr, stride, xh, yh, emit_pk);
        else if (any_y) mc_luma_patch_vpk<false>(gr, stride, xh, yh, emit_pk);
        else if (any_x) mc_luma_patch_hpk(gr, stride, xh, emit_pk);
        else mc_luma_patch<false, false>(gr, stride, xh, yh, emit);
    } else {
        if (any_x || any_y) mc_chroma_patch_pk(gr, stride, xh, yh, emit_pk);
        else mc_chroma_patch<false>(gr, stride, xh, yh, emit);
    }
}

// four waves per SIMD (128 VGPRs, a few dwords spilled) measured against three without spills: luma the same, chroma 4 % faster
#ifndef MC_WPE
#define MC_WPE 4
#endif
#if MC_WPE > 0
#define MC_WPE_ATTR __attribute__((amdgpu_waves_per_eu(MC_WPE)))
#else
#define MC_WPE_ATTR
#endif
template <int CH>
__global__ __launch_bounds__(256) MC_WPE_ATTR void k_fwd_mc_pix(const JobDev *__restrict__ jobs, SbtGeo3 G, McGeo MG, int c0, int npl,
                                                    const DMV *__restrict__ mvs0, int skip_fast, int bxofs, int byofs, int bxstep, int bystep)
{
    int job, c;
    d_job_plane((int)blockIdx.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    // (strips of the grid: block columns bxofs + k * bxstep, block rows byofs + k * bystep)
    const int I = (bxofs + (int)blockIdx.x * bxstep) * 64 + threadIdx.x, J = (byofs + (int)blockIdx.y * bystep) * 4 + threadIdx.y;
    if (I >= g.w3 || J >= g.h3) return;
    const JobDev &jb = jobs[job];
    const int sh = CH ? MG.hs : 0, sv = CH ? MG.vs : 0;
    const int bw = MG.blk_w >> sh, bh = MG.blk_h >> sv;
    const int pw = MG.w[c], ph = g.ph, stride = g.pstride;
    const int x0 = 8 * I, y0 = 8 * J;
    // block of the patch: block sizes are multiples of 8, so (I + 0.5) / (bw / 8) is never near an integer
    const int bi = (int)(((float)I + 0.5f) * __builtin_amdgcn_rcpf((float)(bw >> 3)));
    const int bj = (int)(((float)J + 0.5f) * __builtin_amdgcn_rcpf((float)(bh >> 3)));
    const int nblk = MG.nbh * MG.nbv, blk = bj * MG.nbh + bi;
    const DMV mv = mvs0 ? mvs0[(size_t)job * nblk + blk] : jb.mvs[blk];
    int32_t *coef = jb.coef + g.coff;
    const int W = g.W, H = g.H;
    const int wo1 = DSVG_RSU(W, 1), ho1 = DSVG_RSU(H, 1), wo2 = DSVG_RSU(W, 2), ho2 = DSVG_RSU(H, 2);
    QCtx q;
    const HzPlane &hp = jb.hz[c];
    q.hp = &hp; q.stable = dsvg_global(jb.stable);
    q.sym = dsvg_global(jb.sym + jb.nz_off[c]);
    q.any_ov = (2 * hp.s_w[0] > hp.s_w[1]) || (2 * hp.s_h[0] > hp.s_h[1]) ||
               (2 * hp.s_w[1] > hp.s_w[2]) || (2 * hp.s_h[1] > hp.s_h[2]);
    q.nzf = jb.nzf ? dsvg_global(jb.nzf + (jb.nz_off[c] >> 2)) : nullptr;
    q.cfl = jb.nzf ? dsvg_global(jb.cflag + jb.chunk_off[c]) : nullptr;
    // the common patches (inter block, inside the picture, one stability flag per level) were coded by k_fwd_mc_fast
    if (skip_fast && fwd_fast_sel(jb, hp, q.any_ov, mv.mode, I, J, x0, y0, pw, ph).ok) return;
#ifdef AB_COUNT_GENERAL
    if (jb.stat) atomicAdd(jb.stat + 64 * (2 + (c != 0)) + (threadIdx.x & 63), 1u);
#endif
    // transform level 1 consumes the residual rows in pairs as they appear: only two of them are alive at a time
    HaarQ<8, 2> hq1;
    hq1.init(q, 4 * I, 4 * J, W, H, wo1, ho1);
    int l1[4][4], ra[2][8];

    if (mv.mode != 0) {
        const auto px = dsvg_global(static_cast<const uint8_t *>(jb.xf + g.poff));
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int y = y0 + r;
            uint2 v = make_uint2(0x80808080u, 0x80808080u);
            if (y < ph) v = dsvg_ld2(px + (unsigned)(y * stride + x0));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ra[r & 1][i] = (int)((v.x >> (8 * i)) & 0xff) - 128;
                ra[r & 1][i + 4] = (int)((v.y >> (8 * i)) & 0xff) - 128;
            }
            if (r & 1) hq1.rows(q, r >> 1, ra[0], ra[1], l1[r >> 1], false);
        }
    } else {
        // the source rows depend on nothing: requested first
        const auto sp = dsvg_global(jb.srcp[c]);
        const int sstride = jb.srcs[c];
        uint2 sw[8];
#pragma unroll
        for (int r = 0; r < 8; r++) sw[r] = dsvg_ld2(sp + (unsigned)(min(y0 + r, ph - 1) * sstride + x0));
        const int dx = mv.x >> sh, dy = mv.y >> sv;
        const int xb = bi * bw, yb = bj * bh;
        const int wx = d_clamp(xb + (dx >> 1), -DSVG_BORDER, pw - bw + DSVG_BORDER - 1) + (x0 - xb);
        const int wy = d_clamp(yb + (dy >> 1), -DSVG_BORDER, ph - bh + DSVG_BORDER - 1) + (y0 - yb);
        const bool xh = dx & 1, yh = dy & 1;
        const auto gr = dsvg_global(static_cast<const uint8_t *>(jb.ref + g.poff)) + ((wy - 1) * stride + (wx - 1));      // reference (wx-1, wy-1); may lie before the plane's origin (border)
        const bool any_x = __ballot(xh) != 0ull, any_y = __ballot(yh) != 0ull;        // over the lanes that predict
        const auto pp = dsvg_global(jb.pred + g.poff);
        const bool inside = x0 + 8 <= pw && y0 + 8 <= ph;
        // per prediction row: residual clamp(src - pred + 128) - 128 == clamp(src - pred, -128, 127) (subf bmc.c:43-55 +
        // p2sbc sbt.c:576) into the row registers, the prediction itself to its frame
        auto emit = [&](int r, unsigned plo, unsigned phi) {
            int (&row)[8] = ra[r & 1];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int sv_ = (int)(((i < 4 ? sw[r].x : sw[r].y) >> (8 * (i & 3))) & 0xff);
                const int pv_ = (int)(((i < 4 ? plo : phi) >> (8 * (i & 3))) & 0xff);
                row[i] = d_clamp(sv_ - pv_, -128, 127);
            }
            if (inside) {
                dsvg_st2(pp + (unsigned)((y0 + r) * stride + x0), make_uint2(plo, phi));
            } else {
                // patches on the right / bottom edge: rows past the picture are zero (p2sbc skips them); the column
                // right after an odd-width picture holds the replicated source edge in the reference's residual frame
                const int y = y0 + r;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (y >= ph) row[i] = 0;
                    else if (x0 + i < pw) pp[(size_t)y * stride + x0 + i] = (uint8_t)((i < 4 ? plo : phi) >> (8 * (i & 3)));
                    else {
                        const int e = (int)(((i - 1 < 4 ? sw[r].x : sw[r].y) >> (8 * ((i - 1) & 3))) & 0xff);
                        row[i] = (x0 + i == pw && MG.cw_extra[c] && i > 0) ? e - 128 : 0;
                    }
                }
            }
            if (r & 1) hq1.rows(q, r >> 1, ra[0], ra[1], l1[r >> 1], false);
        };
        if (CH == 0) {
            if (any_y) mc_luma_patch<true, true>(gr, stride, xh, yh, emit);
            else if (any_x) mc_luma_patch<true, false>(gr, stride, xh, yh, emit);
            else mc_luma_patch<false, false>(gr, stride, xh, yh, emit);
        } else {
            if (any_x || any_y) mc_chroma_patch<true>(gr, stride, xh, yh, emit);
            else mc_chroma_patch<false>(gr, stride, xh, yh, emit);
        }
    }
    int l2[2][2], l3[1][1];
    haar_fwd_patch_q<4, 1>(l1, l2, 2 * I, 2 * J, wo1, ho1, W, wo2, ho2, coef, true, q);
    haar_fwd_patch_q<2, 0>(l2, l3, I, J, wo2, ho2, W, g.w3, g.h3, coef, true, q);
    dsvg_global(jb.s3 + g.s3off)[(unsigned)(J * g.w3 + I)] = l3[0][0];
    if (jb.nzf) dsvg_global(jb.pflag + g.s3off)[(unsigned)(J * g.w3 + I)] = (uint8_t)q.nz_any;      // every patch, every picture: never stale
}

// --------------------------------------------------------------------------------------------
// forward, I pictures: level 1 = biorthogonal 4-tap, rows then columns (fwd_b4t_2d sbt.c:240-251)
// --------------------------------------------------------------------------------------------
#ifndef FWD_B4T_WPE
#define FWD_B4T_WPE 6             // 89 VGPRs by themselves (5 waves); 77 under this limit, no spills: 0.85 -> 0.79-0.80 ms per 320 I pictures
#endif
#if FWD_B4T_WPE
#define FWD_B4T_ATTR __attribute__((amdgpu_waves_per_eu(FWD_B4T_WPE, FWD_B4T_WPE)))
#else
#define FWD_B4T_ATTR
#endif
template <bool Q>
__global__ __launch_bounds__(256) FWD_B4T_ATTR void k_fwd_b4t(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl,
                                                 int from_src)
{
    int job, c;
    d_job_plane((int)blockIdx.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const int W = g.W, H = g.H;
    const int I = blockIdx.x * 64 + threadIdx.x, J = blockIdx.y * 4 + threadIdx.y;
    if (8 * I >= W || 8 * J >= H) return;
    const JobDev &jb = jobs[job];
    const uint8_t *px = from_src ? jb.srcp[c] : jb.xf + g.poff;
    const int pxs = from_src ? jb.srcs[c] : g.pstride;     // (an in-place chroma plane has no border: the window's edge bytes are fixed up below, never used as read)
    int32_t *coef = jb.coef + g.coff;
    const int hw = W >> 1, hh = H >> 1;

    // Row pass and column pass interleaved: cell row m needs the row-pass outputs of rows 2m .. 2m+3 of the ten (8J-1 .. 8J+8),
    // so a window of four rows is alive at a time instead of all ten (158 -> ~100 VGPRs: three waves per SIMD became five).
    int RL[4][4], RH[4][4];         // row-pass outputs of the last four rows for cells 4I..4I+3 (row r in slot r & 3)
    const int nC = min(4, max(0, hw - 4 * I));
    // Q: the level-1 detail bands (scan level 2, shift quantiser) are quantised here; the DEQUANTISED values go to the
    // coefficient plane (k_inv_b4t reads them) and the symbols to the symbol plane (k_hz_collect compacts them), so the
    // separate quantiser pass over the plane (k_hz_quant<false>: 8 B per coefficient) disappears for I pictures too.
    QCtx q;
    QLevel Lq;
    int cls[4][4];
    bool chx = false, chy = false;
    if (Q) {
        const HzPlane &hp = jb.hz[c];
        q.hp = &hp; q.stable = dsvg_global(jb.stable);
        q.sym = dsvg_global(jb.sym + jb.nz_off[c]);
        q.any_ov = (2 * hp.s_w[0] > hp.s_w[1]) || (2 * hp.s_h[0] > hp.s_h[1]) ||
                   (2 * hp.s_w[1] > hp.s_w[2]) || (2 * hp.s_h[1] > hp.s_h[2]);
        Lq = q_level<2>(hp);
        q_flags<2, 4>(q, Lq, 4 * I, 4 * J, hw, hh, cls);
        chx = q.any_ov && I == 0; chy = q.any_ov && J == 0;
    }
#ifndef FWD_B4T_NO_PK
    if constexpr (Q) {
        // The encoder's body on int16 pairs (cells (0,1) and (2,3) of the patch side by side, v_pk_*_i16): samples are 8 bit,
        // the row pass stays within +-1018 before its rounding, the column pass within +-4072.  Picture edges are byte fix-ups of
        // the row's window (left mirror x[-1] = x[1], right clamp x[W] = x[W-1]; rows mirror / clamp by address); a patch whose
        // eight samples do not all exist (width not a multiple of 8) takes the scalar body below.
        if (W - 1 - 8 * I >= 7) {
            const bool ledge = I == 0, redge = 8 * I + 8 > W - 1;
            const auto pxg = dsvg_global(px);
            s16x2 RL[4][2], RH[4][2];
            s16x2 shv[4][2];
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int h = 0; h < 2; h++)      // 16-bit shifts take four bits: |value| < 2^12, so 15 stands for anything larger
                    shv[m][h] = s16x2{(short)min(cls[m][2 * h] ? Lq.sh1 : Lq.sh0, 15), (short)min(cls[m][2 * h + 1] ? Lq.sh1 : Lq.sh0, 15)};
            const s16x2 c512 = s16x2{512, 512};
            // The ten rows of the patch's window are requested FB4T_AHEAD rows ahead of the row pass (per 320 I pictures: under the branch
            // 0.92-0.94 ms; 2 rows ahead 0.86, 3: 0.82-0.83, 5: 0.92, all ten: 0.85 -- the registers cost waves per SIMD: 78 / 89 / 99 / 101).  (A coefficient row past the picture's last pixel row -- odd heights -- reads that
            // last row and is replaced by sample 0 = byte 128 afterwards: with the loads under that test, and the cell rows' stores and
            // branches between them, each row was fetched, waited for and consumed before the next one was requested.)
#ifndef FB4T_AHEAD
#define FB4T_AHEAD 3
#endif
            uint2 mmv[10];
            unsigned lfv[10], rgv[10];
            auto request = [&](int r) {
                int y = 8 * J - 1 + r;
                y = y < 0 ? 1 : (y > H - 1 ? H - 1 : y);            // top mirror / bottom clamp (sbt.c:171-175,193-195)
                const unsigned ro = (unsigned)(min(y, g.ph - 1) * pxs + 8 * I);
                mmv[r] = dsvg_ld2(pxg + ro);
                // (the dword left of the row's first patch / right of its last one is replaced by the mirror / clamp fix-up below: it is not
                // fetched from outside the plane -- a plane that lives in the caller's packed clip has no border, and the first row of the
                // clip's first frame nothing in front of it)
                lfv[r] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(pxg + (ledge ? ro : ro - 4u));
                rgv[r] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(pxg + (redge ? ro + 4u : ro + 8u));
            };
#pragma unroll
            for (int r = 0; r < FB4T_AHEAD && r < 10; r++) request(r);
#pragma unroll
            for (int r = 0; r < 10; r++) {
                int y = 8 * J - 1 + r;
                y = y < 0 ? 1 : (y > H - 1 ? H - 1 : y);
                if (r + FB4T_AHEAD < 10) request(r + FB4T_AHEAD);
                unsigned lo, mi, hi;                                // bytes x = 8I-1 .. 8I+2, +3 .. +6, +7 ..
                {
                    const uint2 mm = mmv[r];
                    const unsigned lft = lfv[r], rgt = rgv[r];
                    lo = __builtin_amdgcn_alignbyte(mm.x, lft, 3u);
                    mi = __builtin_amdgcn_alignbyte(mm.y, mm.x, 3u);
                    hi = __builtin_amdgcn_alignbyte(rgt, mm.y, 3u);
                    if (ledge) lo = __builtin_amdgcn_perm(0u, lo, 0x03020102u);       // x[-1] = x[1]
                    if (redge) hi = __builtin_amdgcn_perm(0u, hi, 0x03020000u);       // x[W] = x[W-1]
                    if (y >= g.ph) lo = mi = hi = 0x80808080u;
                }
                // A[j] = (byte j, byte j + 2): cell k reads bytes 2k .. 2k+3 as xm, x0, x1, xp
                const s16x2 A0 = PK_EVEN(0u, lo), A1 = PK_ODD(0u, lo), A2 = PK_E2(mi, lo), A3 = PK_O2(mi, lo), A4 = PK_EVEN(0u, mi), A5 = PK_ODD(0u, mi),
                            A6 = PK_E2(hi, mi), A7 = PK_O2(hi, mi);
                // (the -128 of every sample: -512 in the low-pass sum, nothing in the high-pass one)
                RL[r & 3][0] = pk_rdiv2((A1 + A2) * (short)3 - (A0 + A3) - c512); RH[r & 3][0] = pk_rdiv2((A0 - A3) + (A2 - A1) * (short)3);
                RL[r & 3][1] = pk_rdiv2((A5 + A6) * (short)3 - (A4 + A7) - c512); RH[r & 3][1] = pk_rdiv2((A4 - A7) + (A6 - A5) * (short)3);
                if (r >= 3 && ((r - 3) & 1) == 0) {
                    const int m = (r - 3) >> 1;
                    const int cy = 4 * J + m;
                    if (cy < hh) {
                        int ll[4], slh[4], shl[4], shh[4];
                        s16x2 plh[2], phl[2], phh[2];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const s16x2 am = RL[(r - 3) & 3][h], a0 = RL[(r - 2) & 3][h], a1 = RL[(r - 1) & 3][h], ap = RL[r & 3][h];
                            const s16x2 bm = RH[(r - 3) & 3][h], b0 = RH[(r - 2) & 3][h], b1 = RH[(r - 1) & 3][h], bp = RH[r & 3][h];
                            const s16x2 pll = pk_rdiv2((a0 + a1) * (short)3 - (am + ap));
                            phl[h] = pk_rdiv2((am - ap) + (a1 - a0) * (short)3);
                            plh[h] = pk_rdiv2((b0 + b1) * (short)3 - (bm + bp));
                            phh[h] = pk_rdiv2((bm - bp) + (b1 - b0) * (short)3);
                            ll[2 * h] = pll.x; ll[2 * h + 1] = pll.y;
                        }
                        if (chx || (m == 0 && chy)) {
                            // cells two scan regions share (first column / row of the bands): the scalar steps of the body below
                            int lh[4] = {plh[0].x, plh[0].y, plh[1].x, plh[1].y}, hl[4] = {phl[0].x, phl[0].y, phl[1].x, phl[1].y},
                                hhv[4] = {phh[0].x, phh[0].y, phh[1].x, phh[1].y};
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                slh[k] = shl[k] = shh[k] = 0;
                                if (k < nC) {
                                    if (k == 0 && chx) {
                                        lh[k] = q_chain(q, 2, hw, cy, lh[k]);
                                        hhv[k] = q_chain(q, 2, hw, hh + cy, hhv[k]);
                                    }
                                    if (m == 0 && chy) {
                                        hl[k] = q_chain(q, 2, 4 * I + k, hh, hl[k]);
                                        if (!(k == 0 && chx)) hhv[k] = q_chain(q, 2, hw + 4 * I + k, hh, hhv[k]);
                                    }
                                    const int sh = cls[m][k] ? Lq.sh1 : Lq.sh0;
                                    (void)q_coef<2>(sh, 0.f, lh[k], slh[k]);
                                    (void)q_coef<2>(sh, 0.f, hl[k], shl[k]);
                                    (void)q_coef<2>(sh, 0.f, hhv[k], shh[k]);
                                }
                            }
                        } else {
                            // the shift quantiser (hzcc.c:115-130): symbol = sign(v) (|v| >> shift)
                            auto qs = [](s16x2 v, s16x2 sh) { const s16x2 sg = v >> 15, a = (v ^ sg) - sg; return ((a >> sh) ^ sg) - sg; };
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const s16x2 a = qs(plh[h], shv[m][h]), b = qs(phl[h], shv[m][h]), d = qs(phh[h], shv[m][h]);
                                slh[2 * h] = a.x; slh[2 * h + 1] = a.y; shl[2 * h] = b.x; shl[2 * h + 1] = b.y; shh[2 * h] = d.x; shh[2 * h + 1] = d.y;
                            }
                        }
                        const int o = cy * Lq.sw + 4 * I;
                        store_sym_row<4>(q.sym + Lq.base0 + o, slh, nC);
                        store_sym_row<4>(q.sym + Lq.base1 + o, shl, nC);
                        store_sym_row<4>(q.sym + Lq.base2 + o, shh, nC);
                        store_row<4>(jb.s1 + g.s1off + (size_t)cy * g.w1 + 4 * I, ll, nC);
                    }
                }
            }
            return;
        }
    }
#endif
#pragma unroll
    for (int r = 0; r < 10; r++) {
        int y = 8 * J - 1 + r;
        y = y < 0 ? 1 : (y > H - 1 ? H - 1 : y);            // top mirror / bottom clamp (sbt.c:171-175,193-195)
        int v[10];                                          // v[k] = sample at x = 8I-1+k
        if (y < g.ph) {
            const uint8_t *row = px + (size_t)y * pxs + 8 * I;
            const uint2 m = *reinterpret_cast<const uint2 *>(row);
            const unsigned lft = *reinterpret_cast<const unsigned *>(I == 0 ? row : row - 4);                  // (replaced below where it lies outside the plane: not fetched from there)
            const unsigned rgt = *reinterpret_cast<const unsigned *>((8 * I + 8 > W - 1) ? row + 4 : row + 8);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v[1 + i] = (int)((m.x >> (8 * i)) & 0xff) - 128;
                v[5 + i] = (int)((m.y >> (8 * i)) & 0xff) - 128;
            }
            v[0] = (I == 0) ? v[2] : (int)(lft >> 24) - 128;           // left mirror x[-1] = x[1]
            const int last = W - 1 - 8 * I;                            // index of the last valid sample
            int edge = v[8];
            if (last < 7) {                                            // ragged width: pick x[W-1]
#pragma unroll
                for (int i = 0; i < 7; i++) edge = (i == last) ? v[1 + i] : edge;
#pragma unroll
                for (int i = 1; i < 8; i++) v[1 + i] = (i > last) ? edge : v[1 + i];
            }
            v[9] = (8 * I + 8 > W - 1) ? edge : (int)(rgt & 0xff) - 128; // right clamp x[W] = x[W-1]
        } else {
#pragma unroll
            for (int i = 0; i < 10; i++) v[i] = 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xm = v[2 * k], x0 = v[2 * k + 1], x1 = v[2 * k + 2], xp = v[2 * k + 3];
            RL[r & 3][k] = d_rdiv2(3 * x0 + 3 * x1 - xm - xp);
            RH[r & 3][k] = d_rdiv2(xm - 3 * x0 + 3 * x1 - xp);
        }
        if (r >= 3 && ((r - 3) & 1) == 0) {
            const int m = (r - 3) >> 1;
            const int cy = 4 * J + m;
            if (cy < hh) {
                int ll[4], lh[4], hl[4], hhv[4];
                int slh[4], shl[4], shh[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int am = RL[(r - 3) & 3][k], a0 = RL[(r - 2) & 3][k], a1 = RL[(r - 1) & 3][k], ap = RL[r & 3][k];
                    const int bm = RH[(r - 3) & 3][k], b0 = RH[(r - 2) & 3][k], b1 = RH[(r - 1) & 3][k], bp = RH[r & 3][k];
                    ll[k] = d_rdiv2(3 * a0 + 3 * a1 - am - ap);
                    hl[k] = d_rdiv2(am - 3 * a0 + 3 * a1 - ap);
                    lh[k] = d_rdiv2(3 * b0 + 3 * b1 - bm - bp);
                    hhv[k] = d_rdiv2(bm - 3 * b0 + 3 * b1 - bp);
                    if (Q) {
                        slh[k] = shl[k] = shh[k] = 0;
                        if (k < nC) {
                            // cells two scan regions share sit on the first column / row of the bands (as in haar_fwd_patch_q)
                            if (k == 0 && chx) {
                                lh[k] = q_chain(q, 2, hw, cy, lh[k]);
                                hhv[k] = q_chain(q, 2, hw, hh + cy, hhv[k]);
                            }
                            if (m == 0 && chy) {
                                hl[k] = q_chain(q, 2, 4 * I + k, hh, hl[k]);
                                if (!(k == 0 && chx)) hhv[k] = q_chain(q, 2, hw + 4 * I + k, hh, hhv[k]);
                            }
                            const int sh = cls[m][k] ? Lq.sh1 : Lq.sh0;
                            lh[k] = q_coef<2>(sh, 0.f, lh[k], slh[k]);
                            hl[k] = q_coef<2>(sh, 0.f, hl[k], shl[k]);
                            hhv[k] = q_coef<2>(sh, 0.f, hhv[k], shh[k]);
                        }
                    }
                }
                if (Q) {
                    const int o = cy * Lq.sw + 4 * I;
                    store_sym_row<4>(q.sym + Lq.base0 + o, slh, nC);
                    store_sym_row<4>(q.sym + Lq.base1 + o, shl, nC);
                    store_sym_row<4>(q.sym + Lq.base2 + o, shh, nC);
                }
                store_row<4>(jb.s1 + g.s1off + (size_t)cy * g.w1 + 4 * I, ll, nC);
                if (!Q) {               // Q: k_inv_b4t<true> dequantises the symbols itself: the int32 bands are not needed
                    store_row<4>(coef + (size_t)cy * W + hw + 4 * I, lh, nC);
                    store_row<4>(coef + (size_t)(hh + cy) * W + 4 * I, hl, nC);
                    store_row<4>(coef + (size_t)(hh + cy) * W + hw + 4 * I, hhv, nC);
                }
            }
        }
    }
}

// forward: two Haar levels (LV, LV+1) from a compact LL band.  LV = 2: intra pictures, LL1 (s1) -> levels
// 2..3, LL3 -> s3.  LV = 4: every picture, LL3 (s3) -> levels 4..5, LL5 -> s5 (the band the LDS tail takes).
template <int LV, bool Q, bool LQ = false>
__global__ __launch_bounds__(256) void k_fwd_haar_mid(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl)
{
    int job, c;
    d_job_plane((int)blockIdx.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const int W = g.W, H = g.H;
    const int ow = DSVG_RSU(W, LV + 1), oh = DSVG_RSU(H, LV + 1);           // output LL band
    const int I = blockIdx.x * 64 + threadIdx.x, J = blockIdx.y * 4 + threadIdx.y;
    if (I >= ow || J >= oh) return;
    const JobDev &jb = jobs[job];
    int32_t *coef = jb.coef + g.coff;
    const int iw = DSVG_RSU(W, LV - 1), ih = DSVG_RSU(H, LV - 1);           // input LL band
    const int32_t *in = (LV == 2) ? jb.s1 + g.s1off : jb.s3 + g.s3off;
    int32_t *out = (LV == 2) ? jb.s3 + g.s3off : jb.s5 + g.s5off;
    int a[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = 4 * I + i, y = 4 * J + r;
            a[r][i] = (x < iw && y < ih) ? in[(size_t)y * iw + x] : 0;
        }
    const int wo1 = DSVG_RSU(W, LV), ho1 = DSVG_RSU(H, LV);
    int l2[2][2], l3[1][1];
    // tiny planes (<= 16 samples a side) have fewer levels than this kernel covers: the band is 1x1 by then and the
    // absent levels pass it through unchanged (sbt.c:617-628 stops at lvls)
    if (LV > g.lvls) { out[0] = a[0][0]; return; }
    if constexpr (Q) {
        // levels 2,3 of a fused I picture: quantised on the spot, symbols only (scan levels 1, 0)
        static_assert(LV == 2, "only transform levels 1..3 have per-level scan regions");
        const HzPlane &hp = jb.hz[c];
        QCtx q;
        q.hp = &hp; q.stable = dsvg_global(jb.stable);
        q.sym = dsvg_global(jb.sym + jb.nz_off[c]);
        q.any_ov = (2 * hp.s_w[0] > hp.s_w[1]) || (2 * hp.s_h[0] > hp.s_h[1]) ||
                   (2 * hp.s_w[1] > hp.s_w[2]) || (2 * hp.s_h[1] > hp.s_h[2]);
        haar_fwd_patch_q<4, 1>(a, l2, 2 * I, 2 * J, iw, ih, W, wo1, ho1, coef, true, q);
        haar_fwd_patch_q<2, 0>(l2, l3, I, J, wo1, ho1, W, ow, oh, coef, true, q);
        out[(size_t)J * ow + I] = l3[0][0];
        return;
    }
    LLQ lq;
    if (LQ) { lq.qp = jb.hz[c].r[0].qp; lq.sw = jb.hz[c].r[0].sw; lq.sym = jb.llsym + jb.ll_off[c]; }
    haar_fwd_patch<4, LQ>(a, l2, 2 * I, 2 * J, iw, ih, W, wo1, ho1, coef, true, &lq);     // levels >= 2 are always scaled
    if (LV + 1 > g.lvls) { out[0] = l2[0][0]; return; }
    haar_fwd_patch<2, LQ>(l2, l3, I, J, wo1, ho1, W, ow, oh, coef, true, &lq);
    out[(size_t)J * ow + I] = l3[0][0];
}

// --------------------------------------------------------------------------------------------
// LDS tails: levels >= 4 of one plane inside one workgroup
// --------------------------------------------------------------------------------------------
// 256 threads, not 1024: beside another stream's saturating kernel a 16-wave workgroup waits for a whole CU's worth of wave
// slots to come free at once (33 / 26 us per launch in the two-stream timed region against 11 alone on the chip)
#define TAIL_THREADS 256
#define TAIL_MAXC 16            // cells per thread at the first tail level (host checks)
#define TAIL_LV 6               // first level handled inside LDS

__global__ __launch_bounds__(TAIL_THREADS) void k_fwd_tail(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl)
{
    extern __shared__ int T[];
    int job, c;
    d_job_plane((int)blockIdx.x, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const JobDev &jb = jobs[job];
    const int w3 = g.w5, h3 = g.h5, n3 = w3 * h3, W = g.W, H = g.H;      // the band this kernel owns
    const int32_t *s3 = jb.s5 + g.s5off;
    for (int i = threadIdx.x; i < n3; i += TAIL_THREADS) T[i] = s3[i];
    __syncthreads();

    for (int lvl = TAIL_LV; lvl <= g.lvls; lvl++) {
        const int ws = DSVG_RSU(W, lvl - 1), hs = DSVG_RSU(H, lvl - 1);
        const int wo = DSVG_RSU(W, lvl), ho = DSVG_RSU(H, lvl);
        const int ncell = wo * ho;
        int ll[TAIL_MAXC], lh[TAIL_MAXC], hl[TAIL_MAXC], hh[TAIL_MAXC];
#pragma unroll
        for (int k = 0; k < TAIL_MAXC; k++) {
            const int cell = threadIdx.x + k * TAIL_THREADS;
            if (cell < ncell) {
                const int cy = cell / wo, cx = cell - cy * wo;
                const bool hasR = 2 * cx + 1 < ws, hasB = 2 * cy + 1 < hs;
                const int *p = T + 2 * cy * w3 + 2 * cx;
                const int a = p[0];
                const int b = hasR ? p[1] : a;
                const int cc = hasB ? p[w3] : a;
                const int d = hasB ? (hasR ? p[w3 + 1] : cc) : b;
                ll[k] = d_ll_down(a + b + cc + d);          // levels >= 2 are always scaled
                lh[k] = a - b + cc - d;
                hl[k] = a + b - cc - d;
                hh[k] = a - b - cc + d;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TAIL_MAXC; k++) {
            const int cell = threadIdx.x + k * TAIL_THREADS;
            if (cell < ncell) {
                const int cy = cell / wo, cx = cell - cy * wo;
                const bool hasR = 2 * cx + 1 < ws, hasB = 2 * cy + 1 < hs;
                T[cy * w3 + cx] = ll[k];
                if (hasR) T[cy * w3 + wo + cx] = lh[k];
                if (hasB) T[(ho + cy) * w3 + cx] = hl[k];
                if (hasR && hasB) T[(ho + cy) * w3 + wo + cx] = hh[k];
            }
        }
        __syncthreads();
    }
    int32_t *coef = jb.coef + g.coff;
    for (int i = threadIdx.x; i < n3; i += TAIL_THREADS) {
        const int y = i / w3, x = i - y * w3;
        coef[(size_t)y * W + x] = T[i];
    }
}

// smoothing nudge of the filtered inverse (sbt.c:480-527)
static __device__ __forceinline__ int d_nudge(int ll, int lp, int ln, int det, int hqp)
{
    int mx = ll - ln, mn = lp - ll;
    if (mn > mx) { const int t = mn; mn = mx; mx = t; }
    mx = min(mx, 0);
    mn = max(mn, 0);
    if (mx != mn) {
        const int t = d_rdiv4(lp - ln);
        const int n = d_rdiv2(d_clamp(t, mx, mn) - (det << 1));
        det += d_clamp(n, -hqp, hqp);
    }
    return det;
}

__global__ __launch_bounds__(TAIL_THREADS) void k_inv_tail(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl)
{
    extern __shared__ int T[];
    int job, c;
    d_job_plane((int)blockIdx.x, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const JobDev &jb = jobs[job];
    const int w3 = g.w5, h3 = g.h5, n3 = w3 * h3, W = g.W, H = g.H;      // the band this kernel owns
    const int32_t *coef = jb.coef + g.coff;
    const bool filt = (c == 0);
    for (int i = threadIdx.x; i < n3; i += TAIL_THREADS) {
        const int y = i / w3, x = i - y * w3;
        T[i] = coef[(size_t)y * W + x];
    }
    __syncthreads();

    for (int lvl = g.lvls; lvl >= TAIL_LV; lvl--) {
        const int ws = DSVG_RSU(W, lvl - 1), hs = DSVG_RSU(H, lvl - 1);
        const int wo = DSVG_RSU(W, lvl), ho = DSVG_RSU(H, lvl);
        const int wfull = ws & ~1, hfull = hs & ~1;
        const int ncell = wo * ho;
        const int hqp = jb.hqp[lvl];
        int o0[TAIL_MAXC], o1[TAIL_MAXC], o2[TAIL_MAXC], o3[TAIL_MAXC];
#pragma unroll
        for (int k = 0; k < TAIL_MAXC; k++) {
            const int cell = threadIdx.x + k * TAIL_THREADS;
            if (cell < ncell) {
                const int cy = cell / wo, cx = cell - cy * wo;
                const int x = 2 * cx, y = 2 * cy;
                const bool hasR = x + 1 < ws, hasB = y + 1 < hs;
                const int *pLL = T + cy * w3 + cx;
                const int LL = d_ll_up(pLL[0]);
                int LH = hasR ? pLL[wo] : 0;
                int HL = hasB ? pLL[ho * w3] : 0;
                const int HH = (hasR && hasB) ? pLL[ho * w3 + wo] : 0;
                if (filt && hasR && hasB) {
                    if (x > 0 && x < wfull - 1) LH = d_nudge(LL, d_ll_up(pLL[-1]), d_ll_up(pLL[1]), LH, hqp);
                    if (y > 0 && y < hfull - 1) HL = d_nudge(LL, d_ll_up(pLL[-w3]), d_ll_up(pLL[w3]), HL, hqp);
                }
                o0[k] = d_div4<false>(LL + LH + HL + HH);
                o1[k] = d_div4<false>(LL - LH + HL - HH);
                o2[k] = d_div4<false>(LL + LH - HL - HH);
                o3[k] = d_div4<false>(LL - LH - HL + HH);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TAIL_MAXC; k++) {
            const int cell = threadIdx.x + k * TAIL_THREADS;
            if (cell < ncell) {
                const int cy = cell / wo, cx = cell - cy * wo;
                const int x = 2 * cx, y = 2 * cy;
                const bool hasR = x + 1 < ws, hasB = y + 1 < hs;
                int *o = T + y * w3 + x;
                o[0] = o0[k];
                if (hasR) o[1] = o1[k];
                if (hasB) {
                    o[w3] = o2[k];
                    if (hasR) o[w3 + 1] = o3[k];
                }
            }
        }
        __syncthreads();
    }
    int32_t *s5 = jb.s5 + g.s5off;
    for (int i = threadIdx.x; i < n3; i += TAIL_THREADS) s5[i] = T[i];
}

// encoder: forward tail, the LL quantiser on its band, inverse tail -- one kernel, the band never leaves LDS in between
// (k_fwd_tail + k_hz_quant<true>'s share + k_inv_tail).  Cell 0 is the DC, which travels unquantised (hzcc.c:161,457-460).
// Round 4: on the critical path of every frame step of a SMALL batch this kernel was the longest link (46 us per 4K 4:4:4
// picture: config 5) -- for its latency, not its work: the band came in by one dependent load per iteration (32 round trips),
// every cell index cost two 32-bit divisions by a runtime width, the quantiser one more per cell.  Now: the band arrives in
// batches of TQ_LD independent loads per thread, cell -> (row, column) is a multiply-high (exact: cell * width < 2^32), the
// quantiser's division by the region's ONE quantiser is a multiply-high with two correction steps, a level's loop ends with its
// cells, and a launch with few workgroups (NT = 1024: the small batches) puts four times the threads on each band.
#define TQ_LD 8
static __device__ __forceinline__ unsigned tq_div(unsigned n, unsigned d, unsigned inv) // n / d with inv = ceil(2^32 / d) (wraps to 0 for d = 1): exact for n * d < 2^32
{
    return d == 1u ? n : __umulhi(n, inv);
}
template <int NT>
__global__ __launch_bounds__(NT) void k_tail_q(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl)
{
    extern __shared__ int T[];
    constexpr int MAXC = TAIL_MAXC * TAIL_THREADS / NT;          // cells per thread at the first tail level (host checks the total)
    int job, c;
    d_job_plane((int)blockIdx.x, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const JobDev &jb = jobs[job];
    const int w3 = g.w5, h3 = g.h5, n3 = w3 * h3, W = g.W, H = g.H;      // the band this kernel owns
    const auto s5 = dsvg_global(jb.s5 + g.s5off);
    const int tid = threadIdx.x;
    for (int i0 = 0; i0 < n3; i0 += NT * TQ_LD) {                 // batches of independent loads: one round trip per batch
        int v[TQ_LD];
#pragma unroll
        for (int u = 0; u < TQ_LD; u++) { const int i = i0 + tid + NT * u; v[u] = i < n3 ? s5[i] : 0; }
#pragma unroll
        for (int u = 0; u < TQ_LD; u++) { const int i = i0 + tid + NT * u; if (i < n3) T[i] = v[u]; }
    }
    __syncthreads();
    for (int lvl = TAIL_LV; lvl <= g.lvls; lvl++) {
        const int ws = DSVG_RSU(W, lvl - 1), hs = DSVG_RSU(H, lvl - 1);
        const int wo = DSVG_RSU(W, lvl), ho = DSVG_RSU(H, lvl);
        const int ncell = wo * ho;
        const unsigned inv = 0xFFFFFFFFu / (unsigned)wo + 1u;
        int ll[MAXC], lh[MAXC], hl[MAXC], hh[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            if (k * NT >= ncell) break;                            // (workgroup-uniform)
            const int cell = tid + k * NT;
            if (cell < ncell) {
                const int cy = (int)tq_div((unsigned)cell, (unsigned)wo, inv), cx = cell - cy * wo;
                const bool hasR = 2 * cx + 1 < ws, hasB = 2 * cy + 1 < hs;
                const int *p = T + 2 * cy * w3 + 2 * cx;
                const int a = p[0];
                const int b = hasR ? p[1] : a;
                const int cc = hasB ? p[w3] : a;
                const int d = hasB ? (hasR ? p[w3 + 1] : cc) : b;
                ll[k] = d_ll_down(a + b + cc + d);
                lh[k] = a - b + cc - d;
                hl[k] = a + b - cc - d;
                hh[k] = a - b - cc + d;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            if (k * NT >= ncell) break;
            const int cell = tid + k * NT;
            if (cell < ncell) {
                const int cy = (int)tq_div((unsigned)cell, (unsigned)wo, inv), cx = cell - cy * wo;
                const bool hasR = 2 * cx + 1 < ws, hasB = 2 * cy + 1 < hs;
                T[cy * w3 + cx] = ll[k];
                if (hasR) T[cy * w3 + wo + cx] = lh[k];
                if (hasB) T[(ho + cy) * w3 + cx] = hl[k];
                if (hasR && hasB) T[(ho + cy) * w3 + wo + cx] = hh[k];
            }
        }
        __syncthreads();
    }
    {   // the quantiser (quant hzcc.c:94-112 with the region's one quantiser): symbols out, dequantised values stay in LDS
        const HzRegion &r0 = jb.hz[c].r[0];
        const int qp = r0.qp, sw = r0.sw;
        const unsigned q2 = (unsigned)qp << 1, qinv = 0xFFFFFFFFu / q2;             // floor((2^32 - 1) / 2q): the estimate below is never too large
        const unsigned winv = 0xFFFFFFFFu / (unsigned)w3 + 1u;
        const auto ls = dsvg_global(jb.llsym + jb.ll_off[c]);
        for (int i = tid; i < n3; i += NT) {
            const int y = (int)tq_div((unsigned)i, (unsigned)w3, winv), x = i - y * w3;
            if (i == 0) { jb.psum[c].dc = T[0]; ls[0] = 0; continue; }
            const int tv = T[i];
            const unsigned m = (unsigned)(tv < 0 ? -tv : tv) << 1;
            int v = 0;
            if (m > (unsigned)qp) {
                const unsigned n1 = m + 1u;
                unsigned e = __umulhi(n1, qinv), r = n1 - e * q2;                    // e <= n1 / 2q <= e + 2
                if (r >= q2) { e++; r -= q2; }
                if (r >= q2) { e++; r -= q2; }
                v = tv < 0 ? -(int)e : (int)e;
            }
            ls[y * sw + x] = v;
            T[i] = v ? hzdq_lo(v, qp) : 0;
        }
    }
    __syncthreads();
    const bool filt = (c == 0);
    for (int lvl = g.lvls; lvl >= TAIL_LV; lvl--) {
        const int ws = DSVG_RSU(W, lvl - 1), hs = DSVG_RSU(H, lvl - 1);
        const int wo = DSVG_RSU(W, lvl), ho = DSVG_RSU(H, lvl);
        const int wfull = ws & ~1, hfull = hs & ~1;
        const int ncell = wo * ho;
        const int hqp = jb.hqp[lvl];
        const unsigned inv = 0xFFFFFFFFu / (unsigned)wo + 1u;
        int o0[MAXC], o1[MAXC], o2[MAXC], o3[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            if (k * NT >= ncell) break;
            const int cell = tid + k * NT;
            if (cell < ncell) {
                const int cy = (int)tq_div((unsigned)cell, (unsigned)wo, inv), cx = cell - cy * wo;
                const int x = 2 * cx, y = 2 * cy;
                const bool hasR = x + 1 < ws, hasB = y + 1 < hs;
                const int *pLL = T + cy * w3 + cx;
                const int LL = d_ll_up(pLL[0]);
                int LH = hasR ? pLL[wo] : 0;
                int HL = hasB ? pLL[ho * w3] : 0;
                const int HH = (hasR && hasB) ? pLL[ho * w3 + wo] : 0;
                if (filt && hasR && hasB) {
                    if (x > 0 && x < wfull - 1) LH = d_nudge(LL, d_ll_up(pLL[-1]), d_ll_up(pLL[1]), LH, hqp);
                    if (y > 0 && y < hfull - 1) HL = d_nudge(LL, d_ll_up(pLL[-w3]), d_ll_up(pLL[w3]), HL, hqp);
                }
                o0[k] = d_div4<false>(LL + LH + HL + HH);
                o1[k] = d_div4<false>(LL - LH + HL - HH);
                o2[k] = d_div4<false>(LL + LH - HL - HH);
                o3[k] = d_div4<false>(LL - LH - HL + HH);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            if (k * NT >= ncell) break;
            const int cell = tid + k * NT;
            if (cell < ncell) {
                const int cy = (int)tq_div((unsigned)cell, (unsigned)wo, inv), cx = cell - cy * wo;
                const int x = 2 * cx, y = 2 * cy;
                const bool hasR = x + 1 < ws, hasB = y + 1 < hs;
                int *o = T + y * w3 + x;
                o[0] = o0[k];
                if (hasR) o[1] = o1[k];
                if (hasB) {
                    o[w3] = o2[k];
                    if (hasR) o[w3 + 1] = o3[k];
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n3; i += NT) s5[i] = T[i];
}

// --------------------------------------------------------------------------------------------
// inverse levels 3,2,(1) on a tile through LDS
// --------------------------------------------------------------------------------------------
struct LvlGeo {
    int ws, hs, wo, ho, wfull, hfull, hqp;
    bool scaled;
};
static __device__ __forceinline__ LvlGeo mk_lvl(int W, int H, int lvl, int hqp, bool scaled)
{
    LvlGeo L;
    L.ws = DSVG_RSU(W, lvl - 1); L.hs = DSVG_RSU(H, lvl - 1);
    L.wo = DSVG_RSU(W, lvl);     L.ho = DSVG_RSU(H, lvl);
    L.wfull = L.ws & ~1;         L.hfull = L.hs & ~1;
    L.hqp = hqp; L.scaled = scaled;
    return L;
}

// Where the three detail bands of one level come from:
//   Det<false,.>: the int32 coefficient plane (I pictures, the decoder, operator-level calls);
//   Det<true,HZL>: the scan-order SYMBOL plane the fused forward transform left behind (P pictures of the
//                  encoder) -- dequantised here (hzcc.c:121-128,221-224), so the dequantised coefficients
//                  never exist in HBM.  The value of a cell two scan regions share is the later region's
//                  dequantised symbol, which is the region of the band the cell physically sits in.
struct Raw3 { int lh, hl, hh, k, m; };              // the three details of one cell as fetched, its stability flag (raw), m: bit 0 / 1 = the cell has an LH / HL band
template <bool SYM> struct Raw4;                    // the same for four adjacent complete level-1 cells
template <> struct Raw4<false> { int4 lh, hl, hh; };
template <> struct Raw4<true> { uint2 lh, hl, hh; int k0, k1, k2, k3; };
template <bool SYM, int HZL> struct Det;
template <int HZL> struct Det<false, HZL> {
    const DSVG_GLOBAL int32_t *coef; int W, wo, ho;          // (global address space: flat loads would tie the fetches to every scalar-load wait, see dsvg_global)
    __device__ __forceinline__ Raw3 fetch(int cx, int cy, bool hasR, bool hasB) const
    {
        // (a band the cell does not have is fetched at the plane's first coefficient and dropped in finish(): a select on the
        // loaded value here would make the fetch wait for it -- the tile kernels request every level's cells before they use any)
        Raw3 r;
        r.lh = coef[hasR ? (size_t)cy * W + wo + cx : 0];
        r.hl = coef[hasB ? (size_t)(ho + cy) * W + cx : 0];
        r.hh = coef[(hasR && hasB) ? (size_t)(ho + cy) * W + wo + cx : 0];
        r.k = 0; r.m = (hasR ? 1 : 0) | (hasB ? 2 : 0);
        return r;
    }
    __device__ __forceinline__ void finish(const Raw3 &r, int &LH, int &HL, int &HH) const
    {
        LH = (r.m & 1) ? r.lh : 0; HL = (r.m & 2) ? r.hl : 0; HH = r.m == 3 ? r.hh : 0;
    }
    __device__ __forceinline__ Raw4<false> fetch4(int cx0, int cy) const
    {
        Raw4<false> r;
        auto ld = [](const DSVG_GLOBAL int32_t *q) { const uint4 t = dsvg_ld4(q); return make_int4((int)t.x, (int)t.y, (int)t.z, (int)t.w); };
        r.lh = ld(coef + (size_t)cy * W + wo + cx0);
        r.hl = ld(coef + (size_t)(ho + cy) * W + cx0);
        r.hh = ld(coef + (size_t)(ho + cy) * W + wo + cx0);
        return r;
    }
    __device__ __forceinline__ void finish4(const Raw4<false> &r, int (&lhv)[4], int (&hlv)[4], int (&hhv)[4]) const
    {
        lhv[0] = r.lh.x; lhv[1] = r.lh.y; lhv[2] = r.lh.z; lhv[3] = r.lh.w;
        hlv[0] = r.hl.x; hlv[1] = r.hl.y; hlv[2] = r.hl.z; hlv[3] = r.hl.w;
        hhv[0] = r.hh.x; hhv[1] = r.hh.y; hhv[2] = r.hh.z; hhv[3] = r.hh.w;
    }
    __device__ __forceinline__ int lh(int cx, int cy) const { return coef[(size_t)cy * W + wo + cx]; }
    __device__ __forceinline__ int hl(int cx, int cy) const { return coef[(size_t)(ho + cy) * W + cx]; }
    __device__ __forceinline__ void get3(int cx, int cy, bool hasR, bool hasB, int &LH, int &HL, int &HH) const
    {
        LH = hasR ? coef[(size_t)cy * W + wo + cx] : 0;
        HL = hasB ? coef[(size_t)(ho + cy) * W + cx] : 0;
        HH = (hasR && hasB) ? coef[(size_t)(ho + cy) * W + wo + cx] : 0;
    }
};
template <int HZL> struct Det<true, HZL> {
    const DSVG_GLOBAL int16_t *sym; const DSVG_GLOBAL uint8_t *stable; int nbh; QLevel L;
    __device__ __forceinline__ Raw3 fetch(int cx, int cy, bool hasR, bool hasB) const
    {
        Raw3 r;
        const int o = cy * L.sw + cx;
        // RAW flag and symbols: the class is taken and missing bands are dropped in finish() -- done here, either would make every
        // fetch wait for its loads (a missing band is fetched at the level's first symbol)
        r.k = flag(cx, cy); r.m = (hasR ? 1 : 0) | (hasB ? 2 : 0);
        r.lh = (int)sym[L.base0 + (hasR ? o : 0)];
        r.hl = (int)sym[L.base1 + (hasB ? o : 0)];
        r.hh = (int)sym[L.base2 + ((hasR && hasB) ? o : 0)];
        return r;
    }
    __device__ __forceinline__ void finish(const Raw3 &r, int &LH, int &HL, int &HH) const
    {
        const int k = kcls(r.k);
        LH = (r.m & 1) ? dq(r.lh, k) : 0; HL = (r.m & 2) ? dq(r.hl, k) : 0; HH = r.m == 3 ? dq(r.hh, k) : 0;
    }
    __device__ __forceinline__ Raw4<true> fetch4(int cx0, int cy) const
    {
        Raw4<true> r;
        const int o = cy * L.sw + cx0;
        r.lh = dsvg_ld2(sym + L.base0 + o);
        r.hl = dsvg_ld2(sym + L.base1 + o);
        r.hh = dsvg_ld2(sym + L.base2 + o);
        const int by = ((cy * L.dby) >> 14) * nbh;
        const int bx0 = (cx0 * L.dbx) >> 14, bx3 = ((cx0 + 3) * L.dbx) >> 14;
        r.k0 = stable[by + bx0];                  // raw flags; -1: "as cell 0" (a copy of the loaded value would wait for it)
        r.k1 = r.k2 = r.k3 = -1;
        if (bx0 != bx3) {
            r.k1 = stable[by + (((cx0 + 1) * L.dbx) >> 14)];
            r.k2 = stable[by + (((cx0 + 2) * L.dbx) >> 14)];
            r.k3 = stable[by + bx3];
        }
        return r;
    }
    __device__ __forceinline__ void finish4(const Raw4<true> &r, int (&lhv)[4], int (&hlv)[4], int (&hhv)[4]) const
    {
        const int k0 = kcls(r.k0), k1 = r.k1 < 0 ? k0 : kcls(r.k1), k2 = r.k2 < 0 ? k0 : kcls(r.k2), k3 = r.k3 < 0 ? k0 : kcls(r.k3);
        lhv[0] = dq((int16_t)(r.lh.x & 0xffff), k0); lhv[1] = dq((int)r.lh.x >> 16, k1);
        lhv[2] = dq((int16_t)(r.lh.y & 0xffff), k2); lhv[3] = dq((int)r.lh.y >> 16, k3);
        hlv[0] = dq((int16_t)(r.hl.x & 0xffff), k0); hlv[1] = dq((int)r.hl.x >> 16, k1);
        hlv[2] = dq((int16_t)(r.hl.y & 0xffff), k2); hlv[3] = dq((int)r.hl.y >> 16, k3);
        hhv[0] = dq((int16_t)(r.hh.x & 0xffff), k0); hhv[1] = dq((int)r.hh.x >> 16, k1);
        hhv[2] = dq((int16_t)(r.hh.y & 0xffff), k2); hhv[3] = dq((int)r.hh.y >> 16, k3);
    }
    __device__ __forceinline__ int flag(int cx, int cy) const { return stable[((cy * L.dby) >> 14) * nbh + ((cx * L.dbx) >> 14)]; }
    static __device__ __forceinline__ int kcls(int f) { return HZL == 2 ? (f != 0) : ((f & 2) ? 2 : (f != 0)); }
    __device__ __forceinline__ int cls(int cx, int cy) const { return kcls(flag(cx, cy)); }
    __device__ __forceinline__ int dq(int v, int k) const
    {
        if (HZL == 2) return (int)((unsigned)v << (k ? L.sh1 : L.sh0));
        const int q = max(L.qp >> k, HZ_MINQ);
        const int m = ((v < 0 ? -v : v) * (q << 1) + q) >> 1;
        return v ? (v < 0 ? -m : m) : 0;
    }
    __device__ __forceinline__ int lh(int cx, int cy) const { return dq(sym[L.base0 + cy * L.sw + cx], cls(cx, cy)); }
    __device__ __forceinline__ int hl(int cx, int cy) const { return dq(sym[L.base1 + cy * L.sw + cx], cls(cx, cy)); }
    __device__ __forceinline__ void get3(int cx, int cy, bool hasR, bool hasB, int &LH, int &HL, int &HH) const
    {
        const int k = cls(cx, cy), o = cy * L.sw + cx;
        LH = hasR ? dq(sym[L.base0 + o], k) : 0;
        HL = hasB ? dq(sym[L.base1 + o], k) : 0;
        HH = (hasR && hasB) ? dq(sym[L.base2 + o], k) : 0;
    }
};
template <bool SYM, int HZL>
static __device__ __forceinline__ Det<SYM, HZL> mk_det(const JobDev &jb, int c, const int32_t *coef, int W, const LvlGeo &L)
{
    Det<SYM, HZL> D;
    if constexpr (SYM) {
        const HzPlane &hp = jb.hz[c];
        D.sym = dsvg_global(static_cast<const int16_t *>(jb.sym + jb.nz_off[c])); D.stable = dsvg_global(static_cast<const uint8_t *>(jb.stable)); D.nbh = hp.nbh; D.L = q_level<HZL>(hp);
    } else {
        D.coef = dsvg_global(coef); D.W = W; D.wo = L.wo; D.ho = L.ho;
    }
    return D;
}

// the low-pass nudges of a complete cell (sbt.c:463-527)
// NOTE (reference quirk): for an even region the last complete cell still passes the inX/inY test and its
// "next LL" is read ACROSS the band boundary: column wo of the same row (= LH[cy][0]) / row ho of the same
// column (= HL[0][cx]) of the coefficient plane.
template <bool SMALL, typename DET>
static __device__ __forceinline__ void inv_nudge(const int *pA, int astride, int cx, int cy, const LvlGeo &L, const DET &D,
                                                 int LL, int &LH, int &HL)
{
    const int x = 2 * cx, y = 2 * cy;
    if (x > 0 && x < L.wfull - 1) {
        int lp = pA[-1], ln = (cx + 1 < L.wo) ? pA[1] : D.lh(0, cy);
        if (L.scaled) { lp = d_ll_up_t<SMALL>(lp); ln = d_ll_up_t<SMALL>(ln); }
        LH = d_nudge(LL, lp, ln, LH, L.hqp);
    }
    if (y > 0 && y < L.hfull - 1) {
        int lp = pA[-astride], ln = (cy + 1 < L.ho) ? pA[astride] : D.hl(cx, 0);
        if (L.scaled) { lp = d_ll_up_t<SMALL>(lp); ln = d_ll_up_t<SMALL>(ln); }
        HL = d_nudge(LL, lp, ln, HL, L.hqp);
    }
}

// inverse of one cell (cx,cy); A = LDS array of this level's LL values, pA -> this cell's LL
template <bool FILT, bool SMALL, typename DET>
static __device__ __forceinline__ void inv_cell(const int *pA, int astride, int cx, int cy, const LvlGeo &L,
                                                const DET &D, int (&o)[4])
{
    const int x = 2 * cx, y = 2 * cy;
    const bool hasR = x + 1 < L.ws, hasB = y + 1 < L.hs;
    const int LL = L.scaled ? d_ll_up_t<SMALL>(pA[0]) : pA[0];
    int LH, HL, HH;
    D.get3(cx, cy, hasR, hasB, LH, HL, HH);
    if (FILT && hasR && hasB) inv_nudge<SMALL>(pA, astride, cx, cy, L, D, LL, LH, HL);
    o[0] = d_div4<SMALL>(LL + LH + HL + HH);
    o[1] = d_div4<SMALL>(LL - LH + HL - HH);
    o[2] = d_div4<SMALL>(LL + LH - HL - HH);
    o[3] = d_div4<SMALL>(LL - LH - HL + HH);
}

// inverse of one cell whose raw details were fetched earlier (Det::fetch)
template <bool FILT, bool SMALL, typename DET>
static __device__ __forceinline__ void inv_cell_raw(const int *pA, int astride, int cx, int cy, const LvlGeo &L,
                                                    const DET &D, const Raw3 &raw, int (&o)[4])
{
    const bool hasR = 2 * cx + 1 < L.ws, hasB = 2 * cy + 1 < L.hs;
    const int LL = L.scaled ? d_ll_up_t<SMALL>(pA[0]) : pA[0];
    int LH, HL, HH;
    D.finish(raw, LH, HL, HH);
    if (FILT && hasR && hasB) inv_nudge<SMALL>(pA, astride, cx, cy, L, D, LL, LH, HL);
    o[0] = d_div4<SMALL>(LL + LH + HL + HH);
    o[1] = d_div4<SMALL>(LL - LH + HL - HH);
    o[2] = d_div4<SMALL>(LL + LH - HL - HH);
    o[3] = d_div4<SMALL>(LL - LH - HL + HH);
}

// same as inv_cell for a COMPLETE cell whose three details were already fetched (vector loads)
template <bool FILT, bool SMALL, typename DET>
static __device__ __forceinline__ void inv_cell_vals(const int *pA, int astride, int cx, int cy, const LvlGeo &L,
                                                     const DET &D, int LH, int HL, int HH, int (&o)[4])
{
    const int LL = L.scaled ? d_ll_up_t<SMALL>(pA[0]) : pA[0];
    if (FILT) inv_nudge<SMALL>(pA, astride, cx, cy, L, D, LL, LH, HL);
    o[0] = d_div4<SMALL>(LL + LH + HL + HH);
    o[1] = d_div4<SMALL>(LL - LH + HL - HH);
    o[2] = d_div4<SMALL>(LL + LH - HL - HH);
    o[3] = d_div4<SMALL>(LL - LH - HL + HH);
}

// ---- two level-1 cells per instruction (v_pk_*_i16): the encoder's P-picture luma inverse ----------------------
// Level 1 of a P picture is unscaled and works on an 8-bit residual.  Worst-case magnitudes in the ENCODER (symbols
// come from our own forward transform; a dequantised value is at most twice the coefficient it came from): level-1
// details <= 1020, reconstructed LL1 (the level-2 outputs in A1) <= ~7000, so lp - ln <= 14000, mn - mx <= 28000 and
// the output sums <= 10100 -- all inside int16.  The decoder, which must follow the reference on arbitrary streams,
// keeps the 32-bit cells.
static __device__ __forceinline__ s16x2 pk_nudge(s16x2 ll, s16x2 lp, s16x2 ln, s16x2 det, short hqp, s16x2 pm = s16x2{-1, -1})      // d_nudge x 2 (pm: lanes that take it)
{
    const s16x2 a = ll - ln, b = lp - ll, z = {0, 0};
    const s16x2 mx = pk_min(pk_max(a, b), z), mn = pk_max(pk_min(a, b), z);
    const s16x2 t = pk_rdiv4(lp - ln);
    const s16x2 n = pk_rdiv2(pk_min(pk_max(t, mx), mn) - (det << 1));
    const s16x2 h = {hqp, hqp}, nh = {(short)-hqp, (short)-hqp};
    const s16x2 dl = pk_min(pk_max(n, nh), h);
    const s16x2 mask = ((z - (mn - mx)) >> 15) & pm;        // all ones where mx != mn (mx <= 0 <= mn)
    return det + (dl & mask);
}
// pixels of two cells: even/odd output columns e, o (int16 pairs) -> sbc2int (+ prediction), packed back to bytes
static __device__ __forceinline__ unsigned pk_pixels(s16x2 e, s16x2 o, bool has_pred, unsigned predw)
{
    const s16x2 z = {0, 0}, m255 = {255, 255}, c128 = {128, 128};
    e = pk_min(pk_max(e + c128, z), m255);
    o = pk_min(pk_max(o + c128, z), m255);
    if (has_pred) {                                         // dsv_frame_add / addf bmc.c:29-41
        const unsigned pe = predw & 0x00ff00ffu, po = (predw >> 8) & 0x00ff00ffu;
        e = pk_min(pk_max(e + __builtin_bit_cast(s16x2, pe) - c128, z), m255);
        o = pk_min(pk_max(o + __builtin_bit_cast(s16x2, po) - c128, z), m255);
    }
    return __builtin_bit_cast(unsigned, e) | (__builtin_bit_cast(unsigned, o) << 8);
}

// the same with the prediction, in 11 instructions instead of 19: sat8(sat8(v + 128) + p - 128) == sat8(clamp(v, -128, 127) + p);
// v_sat_pk_u8_i16 clamps a pair to bytes, v_perm_b32 interleaves the even and odd columns and splits the prediction bytes
static __device__ __forceinline__ unsigned pk_pixels_pred(s16x2 e, s16x2 o, unsigned predw)
{
    const s16x2 lo = {-128, -128}, hi = {127, 127};
    const unsigned pe = predw & 0x00ff00ffu, po = __builtin_amdgcn_perm(0u, predw, 0x0c030c01u);
    e = pk_min(pk_max(e, lo), hi) + __builtin_bit_cast(s16x2, pe);
    o = pk_min(pk_max(o, lo), hi) + __builtin_bit_cast(s16x2, po);
    unsigned eb, ob;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(eb) : "v"(__builtin_bit_cast(unsigned, e)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ob) : "v"(__builtin_bit_cast(unsigned, o)));
    return __builtin_amdgcn_perm(ob, eb, 0x05010400u);      // bytes e0 o0 e1 o1
}
// truncating x / 4 on a pair: x + 3 for negative x = x - 3 * (x >> 15), one multiply-add
static __device__ __forceinline__ s16x2 pk_div4m(s16x2 v) { return (v + (v >> 15) * (short)-3) >> 2; }

#define IT_TX 16     // level-3 cells per tile in x  (=> 128 px)
#define IT_TY 8      // level-3 cells per tile in y  (=>  64 px)
#define A3W (IT_TX + 4)
#define A3H (IT_TY + 4)
#define A2W (2 * IT_TX + 4)
#define A2H (2 * IT_TY + 4)
#define A1W (4 * IT_TX + 4)
#define A1H (4 * IT_TY + 4)

// ---- the encoder's P pictures, tiles away from the right / bottom edge: the fast body of k_inv_haar_tile<FILT,0,true> ----
// Same arithmetic as the general body below (sbt.c:352-574 + sbc2int + addf), arranged for the common case:
//   * every cell of the tile and its halo is complete and has in-band neighbours on the right and below (the caller
//     checks I0 + IT_TX + 1 < w3 and J0 + IT_TY + 1 < h3), so no edge predicate survives except "cell 0 takes no nudge";
//   * the LL arrays in LDS hold values that are already scaled up (x * 5 / 4, C.3.1.1): each LL value is a neighbour of
//     four cells and its own cell's centre, so scaling it once where it is produced replaces five scalings per cell;
//   * level 1 (unscaled in a P picture) works on int16 pairs straight from LDS (range: see pk_* above) -- two adjacent
//     cells per instruction, neighbours by v_alignbit;
//   * the level-1 symbols (1.5 of the 2 symbol bytes per sample) are only fetched for patches whose flag is up -- a P
//     picture has a few thousand non-zero symbols -- so an empty patch costs prediction in, reconstruction out.
static __device__ __forceinline__ int dq_lo24(int v, int q)                // hzdq_lo (hzcc.c:121-128), |v| < 2^16, q < 2^11
{
    const int a = v < 0 ? -v : v;
    const int m = (int)(__umul24((unsigned)a, (unsigned)(q << 1)) + (unsigned)q) >> 1;
    const int sg = v >> 31;
    return v ? (m ^ sg) - sg : 0;
}
static __device__ __forceinline__ unsigned pk_i16(int a, int b) { return ((unsigned)a & 0xffffu) | ((unsigned)b << 16); }

// pk_nudge for a detail that is zero: the nudge itself
static __device__ __forceinline__ s16x2 pk_nudge0(s16x2 ll, s16x2 lp, s16x2 ln, short hqp, s16x2 pm)
{
    const s16x2 a = ll - ln, b = lp - ll, z = {0, 0};
    const s16x2 mx = pk_min(pk_max(a, b), z), mn = pk_max(pk_min(a, b), z);
    const s16x2 n = pk_rdiv2(pk_min(pk_max(pk_rdiv4(lp - ln), mx), mn));
    const s16x2 h = {hqp, hqp}, nh = {(short)-hqp, (short)-hqp};
#ifdef AB_INVP_NUDGE0_MASK
    return pk_min(pk_max(n, nh), h) & ((mx - mn) >> 15) & pm;          // mx <= 0 <= mn, mn - mx <= 28000
#else
    return pk_min(pk_max(n, nh), h) & pm;          // (d_nudge's "mx == mn: no nudge" needs no mask here: mx <= 0 <= mn, so mx == mn means both are 0, the clamp gives 0 and rdiv2(0) = 0)
#endif
}
// level 1 of one item of inv_p_fast: four adjacent cells (two int16 pairs) of one cell row -> 8 pixels x 2 rows.
// row: the item's first LL1 pair in LDS (one pair of halo on each side, rows above / below at -WP / +WP); ZERO: no detail
// symbol in the item (LH = HL = HH = 0 before the nudge); colnz / rownz: 0 for the plane's first cell column / row
template <bool FILT, bool ZERO>
static __device__ __forceinline__ void inv_l1_item(const unsigned *row, int WP, uint2 dlh, uint2 dhl, uint2 dhh, const s16x2 (&shv)[2], const uint2 (&pv)[2],
                                                   short hq1, int colnz, int rownz, unsigned (&row0)[2], unsigned (&row1)[2])
{
    const unsigned q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
    const short pv_ = (short)(rownz ? -1 : 0);                             // row 0 of the plane has none above
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        const unsigned cw = h2 ? q2 : q1, lw = h2 ? q1 : q0, rw = h2 ? q3 : q2;
        const s16x2 C = __builtin_bit_cast(s16x2, cw);
        s16x2 LH = {0, 0}, HL = {0, 0}, HH = {0, 0};
        if (!ZERO) {
            const unsigned sl = h2 ? dlh.y : dlh.x, sh_ = h2 ? dhl.y : dhl.x, sd = h2 ? dhh.y : dhh.x;
            LH = __builtin_bit_cast(s16x2, sl) << shv[h2]; HL = __builtin_bit_cast(s16x2, sh_) << shv[h2];
            HH = __builtin_bit_cast(s16x2, sd) << shv[h2];
        }
        if (FILT) {
            const s16x2 lp = __builtin_bit_cast(s16x2, __builtin_amdgcn_alignbit(cw, lw, 16u));
            const s16x2 ln = __builtin_bit_cast(s16x2, __builtin_amdgcn_alignbit(rw, cw, 16u));
            const s16x2 up = __builtin_bit_cast(s16x2, row[1 + h2 - WP]), dn = __builtin_bit_cast(s16x2, row[1 + h2 + WP]);
            const s16x2 pmh = s16x2{(short)((h2 | colnz) ? -1 : 0), -1};    // cell 0 of the plane has no left neighbour
            if (ZERO) {
                LH = pk_nudge0(C, lp, ln, hq1, pmh);
                HL = pk_nudge0(C, up, dn, hq1, s16x2{pv_, pv_});
            } else {
                LH = pk_nudge(C, lp, ln, LH, hq1, pmh);
                HL = pk_nudge(C, up, dn, HL, hq1, s16x2{pv_, pv_});
            }
        }
        const s16x2 sA = C + HL, sB = ZERO ? LH : LH + HH, sC = C - HL, sD = ZERO ? LH : LH - HH;
        row0[h2] = pk_pixels_pred(pk_div4m(sA + sB), pk_div4m(sA - sB), h2 ? pv[0].y : pv[0].x);
        row1[h2] = pk_pixels_pred(pk_div4m(sC + sD), pk_div4m(sC - sD), h2 ? pv[1].y : pv[1].x);
    }
}

// ER: the tile is the last of its tile row and ends exactly where the band ends (w3 a multiple of IT_TX, every cell of every
// level complete: W a multiple of 8).  The cells right of it do not exist; what the reference reads as the "next LL" of the
// last cell of a row is the first coefficient of the LH band of that row and level (sbt.c:463-527: the bands lie side by
// side in its buffer) -- the threads that own the non-existent halo cells put exactly those values where the halo values go.
// EB: the tile lies in the last tile row, the band ends inside it or at its end (h3 - J0 <= IT_TY), every cell complete (H a
// multiple of 8).  The same with rows: the "next LL" below a column's last cell is the first HL coefficient of that column
// and level; the cell rows past the band's end are not loaded, computed or stored.
template <bool FILT, bool ER = false, bool EB = false>
static __device__ __forceinline__ void inv_p_fast(const JobDev &jb, const SbtGeo &g, int c, int I0, int J0, int tid,
                                                  int *__restrict__ A3u, int *__restrict__ A2u, unsigned *__restrict__ A1p)
{
    constexpr int W3 = IT_TX + 4, W2 = 2 * IT_TX + 4, WP = 2 * IT_TX + 2;      // row pitches: LL3 values, LL2 values, LL1 pairs
    const HzPlane &hp = jb.hz[c];
    const auto sym = dsvg_global(static_cast<const int16_t *>(jb.sym + jb.nz_off[c]));
    const auto stable = dsvg_global(jb.stable);
    const auto pfl = dsvg_global(static_cast<const uint8_t *>(jb.pflag + g.s3off));
    const auto s3 = dsvg_global(static_cast<const int32_t *>(jb.s3 + g.s3off));
    const int nbh = hp.nbh, w3 = g.w3, h3 = g.h3, stride = g.pstride;
    const QLevel Q3 = q_level<0>(hp), Q2 = q_level<1>(hp), Q1 = q_level<2>(hp);
    const auto pred = dsvg_global(static_cast<const uint8_t *>(jb.pred + g.poff));
    const auto outp = dsvg_global((jb.recon ? jb.recon : jb.xf) + g.poff);

    // ---- phase 0: every global load of the tile.  The flags of this thread's two level-1 items go first: their symbol
    // loads depend on them and are issued as soon as they are back, under the rest of the loads.  Addresses are 32-bit
    // byte offsets from wave-uniform bases (global_load with an SGPR base: no 64-bit address arithmetic per load), index
    // products are 24-bit multiplies, and the divisions by the row lengths 20 / 18 / 34 are multiply-shifts (exact for
    // the item counts of a tile).
    static_assert(IT_TX == 16 && IT_TY == 8, "the multiply-shift constants below are for 20 / 18 / 34 cells per row");
    const unsigned utid = (unsigned)tid;
    auto ldu8 = [](auto base, unsigned off) { return (int)*(base + off); };
    auto lds16 = [](auto base, unsigned idx) { return (int)dsvg_at(base, idx); };
    auto flagidx = [&](const QLevel &Q, unsigned cx, unsigned cy) { return __umul24(__umul24(cy, (unsigned)Q.dby) >> 14, (unsigned)nbh) + (__umul24(cx, (unsigned)Q.dbx) >> 14); };
    // The stability flags fetched in phase 0 stay RAW until their level is computed: turning a flag into its quantiser class (or
    // shift) where it is fetched put an s_waitcnt behind every group of loads -- four memory round trips one after the other in a
    // phase whose point is to have every load of the tile in flight at once
    auto CLS = [](int f) { return (f & 2) ? 2 : (f != 0); };                    // tmq4pos hzcc.c:64-74
    auto SH1 = [&](int f) { return f ? Q1.sh1 : Q1.sh0; };                      // hzcc.c:221-224
    int pf1[2];
    bool v1[2];                                         // the item's cell row exists (EB: the band may end inside the tile)
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const unsigned it = utid + 256u * u, ly = it >> 4, gx = it & 15u;
        v1[u] = !EB || J0 + (int)(ly >> 2) < h3;
        pf1[u] = v1[u] ? ldu8(pfl, __umul24((unsigned)J0 + (ly >> 2), (unsigned)w3) + (unsigned)I0 + gx) : 0;
    }
    int a3v = 0, nzv = 0;
    if (tid < A3H * A3W) {
        const unsigned ly = (utid * 205u) >> 12, lx = utid - ly * W3;
        const int cx = I0 - 2 + (int)lx, cy = J0 - 2 + (int)ly;
        if (ER && cx >= w3) {
            if (cx == w3 && cy >= 0 && (!EB || cy < h3)) {      // "next LL" of the row's last cell: LH3 of column 0, dequantised
                const int f = ldu8(stable, flagidx(Q3, 0u, (unsigned)cy));
                a3v = dq_lo24(lds16(sym, (unsigned)Q3.base0 + __umul24((unsigned)cy, (unsigned)Q3.sw)), max(Q3.qp >> ((f & 2) ? 2 : (f != 0)), HZ_MINQ));
                nzv = ldu8(pfl, __umul24((unsigned)cy, (unsigned)w3));
            }
        } else
        if (EB && cy >= h3) {
            if (cy == h3 && cx >= 0) {                   // below the column's last cell: HL3 of row 0, dequantised
                const int f = ldu8(stable, flagidx(Q3, (unsigned)cx, 0u));
                a3v = dq_lo24(lds16(sym, (unsigned)Q3.base1 + (unsigned)cx), max(Q3.qp >> ((f & 2) ? 2 : (f != 0)), HZ_MINQ));
                nzv = ldu8(pfl, (unsigned)cx);
            }
        } else
        if (cx >= 0 && cy >= 0) {
            const unsigned o = __umul24((unsigned)cy, (unsigned)w3) + (unsigned)cx;
            a3v = dsvg_at(s3, o);
            nzv = ldu8(pfl, o);
        }
    }
    int s3lh = 0, s3hl = 0, s3hh = 0, k3 = 0, lx3 = 0, ly3 = 0;
    bool ok3 = false;
    if (tid < (IT_TY + 2) * (IT_TX + 2)) {
        ly3 = (int)((utid * 57u) >> 10); lx3 = tid - ly3 * (IT_TX + 2);
        const int cx = I0 - 1 + lx3, cy = J0 - 1 + ly3;
        ok3 = cx >= 0 && cy >= 0 && (!EB || cy <= h3) && !(ER && EB && cx >= w3 && cy >= h3);
        // (fetching these symbols only for flagged patches, as level 1 does, was slower: 4.45 -> 4.70 ms per step -- the flag
        // load puts a second round trip in front of the first barrier, and these are 0.47 B/sample, not 1.5)
        if (ER && ok3 && cx >= w3) {
            // the non-existent cell right of the band: its two LL2 outputs are the LH2 values of column 0 of level-2 rows
            // 2cy, 2cy+1 (s3lh / s3hl: the symbols, k3 / s3hh: their quantiser classes)
            s3lh = lds16(sym, (unsigned)Q2.base0 + __umul24(2u * (unsigned)cy, (unsigned)Q2.sw));
            s3hl = lds16(sym, (unsigned)Q2.base0 + __umul24(2u * (unsigned)cy + 1u, (unsigned)Q2.sw));
            const int fa = ldu8(stable, flagidx(Q2, 0u, 2u * (unsigned)cy)), fb = ldu8(stable, flagidx(Q2, 0u, 2u * (unsigned)cy + 1u));
            k3 = fa; s3hh = fb;                       // (raw flags: see CLS below)
        } else
        if (EB && ok3 && cy >= h3) {
            // the non-existent cell row below the band: its LL2 outputs are the HL2 values of row 0 of level-2 columns 2cx, 2cx+1
            s3lh = lds16(sym, (unsigned)Q2.base1 + 2u * (unsigned)cx);
            s3hl = lds16(sym, (unsigned)Q2.base1 + 2u * (unsigned)cx + 1u);
            const int fa = ldu8(stable, flagidx(Q2, 2u * (unsigned)cx, 0u)), fb = ldu8(stable, flagidx(Q2, 2u * (unsigned)cx + 1u, 0u));
            k3 = fa; s3hh = fb;                       // (raw flags: see CLS below)
        } else
        if (ok3) {
            const unsigned o = __umul24((unsigned)cy, (unsigned)Q3.sw) + (unsigned)cx;
            s3lh = lds16(sym, (unsigned)Q3.base0 + o); s3hl = lds16(sym, (unsigned)Q3.base1 + o); s3hh = lds16(sym, (unsigned)Q3.base2 + o);
            const int f = ldu8(stable, flagidx(Q3, (unsigned)cx, (unsigned)cy));
            k3 = f;
        }
    }
    constexpr int N2 = ((2 * IT_TY + 2) * (2 * IT_TX + 2) + 255) / 256;
    int s2lh[N2], s2hl[N2], s2hh[N2], k2[N2];
    bool ok2[N2];
#pragma unroll
    for (int u = 0; u < N2; u++) {
        const unsigned i = utid + 256u * u;
        const unsigned ly = (i * 241u) >> 13, lx = i - ly * (2 * IT_TX + 2);
        const int cx = 2 * I0 - 1 + (int)lx, cy = 2 * J0 - 1 + (int)ly;
        ok2[u] = i < (2 * IT_TY + 2) * (2 * IT_TX + 2) && cx >= 0 && cy >= 0 && (!EB || cy <= 2 * h3) && !(ER && EB && cx >= 2 * w3 && cy >= 2 * h3);
        s2lh[u] = s2hl[u] = s2hh[u] = k2[u] = 0;
        if (ER && ok2[u] && cx >= 2 * w3) {
            // right of the band at level 2: the LL1 halo values are the LH1 values of column 0 of level-1 rows 2cy, 2cy+1
            // (s2lh / s2hl: symbols, k2 / s2hh: their shifts)
            s2lh[u] = lds16(sym, (unsigned)Q1.base0 + __umul24(2u * (unsigned)cy, (unsigned)Q1.sw));
            s2hl[u] = lds16(sym, (unsigned)Q1.base0 + __umul24(2u * (unsigned)cy + 1u, (unsigned)Q1.sw));
            const unsigned bya = __umul24(__umul24(2u * (unsigned)cy, (unsigned)Q1.dby) >> 14, (unsigned)nbh),
                           byb = __umul24(__umul24(2u * (unsigned)cy + 1u, (unsigned)Q1.dby) >> 14, (unsigned)nbh);
            k2[u] = ldu8(stable, bya); s2hh[u] = ldu8(stable, byb);
        } else
        if (EB && ok2[u] && cy >= 2 * h3) {
            // below the band at level 2: the LL1 halo values are the HL1 values of row 0 of level-1 columns 2cx, 2cx+1
            s2lh[u] = lds16(sym, (unsigned)Q1.base1 + 2u * (unsigned)cx);
            s2hl[u] = lds16(sym, (unsigned)Q1.base1 + 2u * (unsigned)cx + 1u);
            k2[u] = ldu8(stable, __umul24(2u * (unsigned)cx, (unsigned)Q1.dbx) >> 14);
            s2hh[u] = ldu8(stable, __umul24(2u * (unsigned)cx + 1u, (unsigned)Q1.dbx) >> 14);
        } else
        if (ok2[u]) {
            const unsigned o = __umul24((unsigned)cy, (unsigned)Q2.sw) + (unsigned)cx;
            s2lh[u] = lds16(sym, (unsigned)Q2.base0 + o); s2hl[u] = lds16(sym, (unsigned)Q2.base1 + o); s2hh[u] = lds16(sym, (unsigned)Q2.base2 + o);
            const int f = ldu8(stable, flagidx(Q2, (unsigned)cx, (unsigned)cy));
            k2[u] = f;
        }
    }
    uint2 pv[2][2];
    unsigned poff[2];                                   // byte offset of each item's first pixel row in the plane
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const unsigned it = utid + 256u * u, ly = it >> 4, gx = it & 15u;
        poff[u] = __umul24(2u * (4u * (unsigned)J0 + ly), (unsigned)stride) + 8u * ((unsigned)I0 + gx);
        pv[u][0] = pv[u][1] = make_uint2(0u, 0u);
        if (v1[u]) {
            pv[u][0] = dsvg_ld2(pred + poff[u]);
            pv[u][1] = dsvg_ld2(pred + poff[u] + (unsigned)stride);
        }
    }
    uint2 d1lh[2], d1hl[2], d1hh[2];
    int fr1[2][4];                                      // the level-1 flags of the item's four cells, raw (see CLS above)
    // (both items' patch flags are looked at HERE: behind the first item's conditional loads the compiler has to assume the worst
    // about what is in flight and would wait for nearly everything before it tests the second flag)
    unsigned long long any1[2] = {__ballot(pf1[0] != 0), __ballot(pf1[1] != 0)};
    asm volatile("" : "+s"(any1[0]), "+s"(any1[1]));
#pragma unroll
    for (int u = 0; u < 2; u++) {
        d1lh[u] = d1hl[u] = d1hh[u] = make_uint2(0u, 0u);
        fr1[u][0] = fr1[u][1] = fr1[u][2] = fr1[u][3] = 0;
        if (pf1[u]) {
            const unsigned it = utid + 256u * u, ly = it >> 4, gx = it & 15u;
            const unsigned cy = 4u * (unsigned)J0 + ly, cx0 = 4u * ((unsigned)I0 + gx);
            const unsigned o = __umul24(cy, (unsigned)Q1.sw) + cx0;
            const auto sb = reinterpret_cast<const DSVG_GLOBAL char *>(sym);
            d1lh[u] = dsvg_ld2(sb + 2u * ((unsigned)Q1.base0 + o));
            d1hl[u] = dsvg_ld2(sb + 2u * ((unsigned)Q1.base1 + o));
            d1hh[u] = dsvg_ld2(sb + 2u * ((unsigned)Q1.base2 + o));
            const unsigned by = __umul24(__umul24(cy, (unsigned)Q1.dby) >> 14, (unsigned)nbh);
            const unsigned bx0 = __umul24(cx0, (unsigned)Q1.dbx) >> 14, bx3 = __umul24(cx0 + 3u, (unsigned)Q1.dbx) >> 14;
            fr1[u][0] = ldu8(stable, by + bx0);
            fr1[u][1] = fr1[u][2] = fr1[u][3] = -1;         // "as cell 0" (a copy of the loaded flag would wait for it here)
            if (bx0 != bx3) {
                fr1[u][1] = ldu8(stable, by + (__umul24(cx0 + 1u, (unsigned)Q1.dbx) >> 14));
                fr1[u][2] = ldu8(stable, by + (__umul24(cx0 + 2u, (unsigned)Q1.dbx) >> 14));
                fr1[u][3] = ldu8(stable, by + bx3);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    if (tid < A3H * A3W) A3u[tid] = d_ll_up_t<true>(a3v);
    if (!__syncthreads_or((nzv | a3v) != 0)) {
        // nothing in reach: every output is zero, the reconstruction is the prediction
        if (tid == 0 && jb.stat) atomicAdd(jb.stat + 64 * (2 + (c != 0)) + (((blockIdx.x >> 3) + 5 * blockIdx.y + 11 * blockIdx.z) & 63), 1u);
        if ((const DSVG_GLOBAL uint8_t *)outp == pred) return;     // written in place by the forward transform (ping-pong slots)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (!v1[u]) continue;
            const auto dst = outp + poff[u];
            dsvg_st2(dst, pv[u][0]);
            dsvg_st2(dst + (unsigned)stride, pv[u][1]);
        }
        return;
    }
    if (jb.stat) {                                      // (wave-uniform; counting is on in bench.py's untimed selection step and in tests only)
        if (tid == 0) atomicAdd(jb.stat + 64 * (c != 0) + (((blockIdx.x >> 3) + 5 * blockIdx.y + 11 * blockIdx.z) & 63), 1u);
        // the 8x8 patches of this tile whose level-1 symbols were fetched (96 bytes each): a patch's flag sits in the four items of
        // its four cell rows -- count it in the first
        const bool top = ((utid >> 4) & 3u) == 0u;
        const unsigned nfl = (unsigned)__popcll(__ballot(top && pf1[0] != 0)) + (unsigned)__popcll(__ballot(top && pf1[1] != 0));
        if ((tid & 63) == 0 && nfl) atomicAdd(jb.stat + 64 * 4 + (((blockIdx.x >> 3) + (tid >> 6)) & 63), nfl);
    }

    // ---- level 3: cells I0-1 .. I0+TX (halo 1) -> LL2 values, scaled up, in A2u
    if (ER && ok3 && I0 - 1 + lx3 >= w3) {
        int *d = A2u + (2 * ly3) * W2 + 2 * lx3;
        d[0] = d_ll_up_t<true>(dq_lo24(s3lh, max(Q2.qp >> CLS(k3), HZ_MINQ))); d[1] = 0;
        d[W2] = d_ll_up_t<true>(dq_lo24(s3hl, max(Q2.qp >> CLS(s3hh), HZ_MINQ))); d[W2 + 1] = 0;
    } else
    if (EB && ok3 && J0 - 1 + ly3 >= h3) {
        int *d = A2u + (2 * ly3) * W2 + 2 * lx3;
        d[0] = d_ll_up_t<true>(dq_lo24(s3lh, max(Q2.qp >> CLS(k3), HZ_MINQ)));
        d[1] = d_ll_up_t<true>(dq_lo24(s3hl, max(Q2.qp >> CLS(s3hh), HZ_MINQ)));
        d[W2] = 0; d[W2 + 1] = 0;
    } else
#ifdef AB_INVP_NO_L3                   // timing probes (wrong pictures): a level reduced to copying its LL value
    if (ok3) {
        const int LL = A3u[(ly3 + 1) * W3 + lx3 + 1];
        int *d = A2u + (2 * ly3) * W2 + 2 * lx3;
        d[0] = d[1] = d[W2] = d[W2 + 1] = LL;
    } else
#endif
    if (ok3) {
        const int *pA = A3u + (ly3 + 1) * W3 + lx3 + 1;
        const int LL = pA[0];
        int LH = 0, HL = 0, HH = 0;
        if (__ballot((s3lh | s3hl | s3hh) != 0)) {           // (a wave without a level-3 symbol: no dequantiser)
            const int q = max(Q3.qp >> CLS(k3), HZ_MINQ);
            LH = dq_lo24(s3lh, q); HL = dq_lo24(s3hl, q); HH = dq_lo24(s3hh, q);
        }
        if (FILT) {
            const int hq = jb.hqp[3];
            if (I0 - 1 + lx3 > 0) LH = d_nudge(LL, pA[-1], pA[1], LH, hq);
            if (J0 - 1 + ly3 > 0) HL = d_nudge(LL, pA[-W3], pA[W3], HL, hq);
        }
        const int sA = LL + HL, sB = LH + HH, sC = LL - HL, sD = LH - HH;
        int *d = A2u + (2 * ly3) * W2 + 2 * lx3;
        d[0] = d_ll_up_t<true>(d_div4<true>(sA + sB)); d[1] = d_ll_up_t<true>(d_div4<true>(sA - sB));
        d[W2] = d_ll_up_t<true>(d_div4<true>(sC + sD)); d[W2 + 1] = d_ll_up_t<true>(d_div4<true>(sC - sD));
    }
    __syncthreads();
    // ---- level 2: cells 2*I0-1 .. 2*I0+2*TX (halo 1) -> LL1 values (level 1 of a P picture is unscaled) as int16 pairs
#pragma unroll
    for (int u = 0; u < N2; u++) {
        if (!ok2[u]) continue;
        const unsigned i = (unsigned)tid + 256u * u;
        const int ly = (int)((i * 241u) >> 13), lx = (int)i - ly * (2 * IT_TX + 2);
        if (ER && 2 * I0 - 1 + lx >= 2 * w3) {
            unsigned *d = A1p + (2 * ly) * WP + lx;
            d[0] = pk_i16((int)((unsigned)s2lh[u] << SH1(k2[u])), 0);
            d[WP] = pk_i16((int)((unsigned)s2hl[u] << SH1(s2hh[u])), 0);
            continue;
        }
        if (EB && 2 * J0 - 1 + ly >= 2 * h3) {
            unsigned *d = A1p + (2 * ly) * WP + lx;
            d[0] = pk_i16((int)((unsigned)s2lh[u] << SH1(k2[u])), (int)((unsigned)s2hl[u] << SH1(s2hh[u])));
            d[WP] = 0;
            continue;
        }
        const int *pA = A2u + (ly + 1) * W2 + lx + 1;
        const int LL = pA[0];
#ifdef AB_INVP_NO_L2
        { unsigned *d = A1p + (2 * ly) * WP + lx; d[0] = d[WP] = pk_i16(LL >> 4, LL >> 4); continue; }
#endif
        int LH = 0, HL = 0, HH = 0;
        if (__ballot((s2lh[u] | s2hl[u] | s2hh[u]) != 0)) {
            const int q = max(Q2.qp >> CLS(k2[u]), HZ_MINQ);
            LH = dq_lo24(s2lh[u], q); HL = dq_lo24(s2hl[u], q); HH = dq_lo24(s2hh[u], q);
        }
        if (FILT) {
            const int hq = jb.hqp[2];
            if (2 * I0 - 1 + lx > 0) LH = d_nudge(LL, pA[-1], pA[1], LH, hq);
            if (2 * J0 - 1 + ly > 0) HL = d_nudge(LL, pA[-W2], pA[W2], HL, hq);
        }
        const int sA = LL + HL, sB = LH + HH, sC = LL - HL, sD = LH - HH;
        unsigned *d = A1p + (2 * ly) * WP + lx;
        d[0] = pk_i16(d_div4<true>(sA + sB), d_div4<true>(sA - sB));
        d[WP] = pk_i16(d_div4<true>(sC + sD), d_div4<true>(sC - sD));
    }
    __syncthreads();
    // ---- level 1 + sbc2int + prediction add: an item = four adjacent cells = 8 pixels x 2 rows, as two int16 pairs.
    // An item none of whose lanes has a flagged patch (the usual one in a sparse picture) takes the body without details.
    const short hq1 = (short)jb.hqp[1];
#ifdef AB_INVP_DUMMY_VALU               // sensitivity probe: N extra packed instructions per thread in a dependent chain
    { unsigned dv_ = pv[0][0].x;
#pragma unroll
      for (int u = 0; u < AB_INVP_DUMMY_VALU; u++) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(dv_) : "v"(pv[0][1].x));
      if (dv_ == 0x12345u) return; }
#endif
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int it = tid + 256 * u, ly = it >> 4, gx = it & 15;
        const unsigned *row = A1p + (ly + 2) * WP + 2 * gx;          // pairs 2gx .. 2gx+3 hold LL1 columns 4gx .. 4gx+7 (halo 2)
        unsigned row0[2], row1[2];
        if (EB && !v1[u]) continue;
        const bool zero1 = any1[u] == 0ull;
        s16x2 shv[2][2];
        auto fcell = [&](int k) { return fr1[u][k] < 0 ? fr1[u][0] : fr1[u][k]; };
        shv[u][0] = s16x2{(short)SH1(fr1[u][0]), (short)SH1(fcell(1))};
        shv[u][1] = s16x2{(short)SH1(fcell(2)), (short)SH1(fcell(3))};
#if defined(AB_INVP_NO_L1)
        row0[0] = pv[u][0].x ^ row[0]; row0[1] = pv[u][0].y; row1[0] = pv[u][1].x; row1[1] = pv[u][1].y;
        if (false)
#elif defined(AB_INVP_NO_L1F)
        if (zero1) inv_l1_item<false, true>(row, WP, d1lh[u], d1hl[u], d1hh[u], shv[u], pv[u], hq1, gx | I0, ly | J0, row0, row1);
        else inv_l1_item<false, false>(row, WP, d1lh[u], d1hl[u], d1hh[u], shv[u], pv[u], hq1, gx | I0, ly | J0, row0, row1);
        if (false)
#endif
        if (zero1) inv_l1_item<FILT, true>(row, WP, d1lh[u], d1hl[u], d1hh[u], shv[u], pv[u], hq1, gx | I0, ly | J0, row0, row1);
        else inv_l1_item<FILT, false>(row, WP, d1lh[u], d1hl[u], d1hh[u], shv[u], pv[u], hq1, gx | I0, ly | J0, row0, row1);
        const auto dst = outp + poff[u];
        dsvg_st2(dst, make_uint2(row0[0], row0[1]));
        dsvg_st2(dst + (unsigned)stride, make_uint2(row1[0], row1[1]));
    }
}


#ifndef INV_P_TILE_WPE
#define INV_P_TILE_WPE 7          // 73 VGPRs by themselves: one over the seven-wave limit (measured: 6 waves 5.89-6.04 ms, 7: 5.70-5.84, 8 -- spills -- 6.8-7.0)
#endif
#if INV_P_TILE_WPE
#define INV_P_TILE_ATTR __attribute__((amdgpu_waves_per_eu(INV_P_TILE_WPE, INV_P_TILE_WPE)))
#else
#define INV_P_TILE_ATTR
#endif
// The fast body as a kernel of its own, for the tiles that take it (launch_inv_sbt: the interior of the tile grid of sparse
// P pictures; k_inv_haar_tile gets the right / bottom strips): register allocation and LDS are then the fast body's, not
// the maximum over the general body as well.
// er_col: the tile column that ends exactly where the band ends and takes the ER body (-1: none) -- in the same launch as the
// interior tiles (as a launch of its own the column's 16 x 160 workgroups took as long as the general kernel's strip did)
template <bool FILT>
__global__ __launch_bounds__(256) INV_P_TILE_ATTR void k_inv_p_tile(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl, int er_col, int eb_row,
                                                                    XcdGrid XG, int plain)
{
    __shared__ int A3[A3H * A3W];
    __shared__ int A2[A2H * A2W];
    __shared__ unsigned A1p[(4 * IT_TY + 4) * (2 * IT_TX + 2)];
    // one-dimensional launch, tiles dealt to the XCDs in contiguous runs (d_xcd_blk3): a tile shares the 128-byte lines its
    // pixel rows straddle (the 64-pixel border shifts them by half a line) and its halo rows of symbols with its neighbours
    Blk3 B;
    if (!d_xcd_blk3(XG, B, plain != 0)) return;
    int job, c;
    d_job_plane(B.z, npl, c0, job, c);
    const int I0 = B.x * IT_TX, J0 = B.y * IT_TY;
    const bool er = B.x == er_col, eb = B.y == eb_row;      // (eb_row: the last tile row, likewise)
    DSVG_CLK_BEGIN();
    if (er && eb) inv_p_fast<FILT, true, true>(jobs[job], G.g[c], c, I0, J0, threadIdx.x, A3, A2, A1p);
    else if (er) inv_p_fast<FILT, true, false>(jobs[job], G.g[c], c, I0, J0, threadIdx.x, A3, A2, A1p);
    else if (eb) inv_p_fast<FILT, false, true>(jobs[job], G.g[c], c, I0, J0, threadIdx.x, A3, A2, A1p);
    else inv_p_fast<FILT, false, false>(jobs[job], G.g[c], c, I0, J0, threadIdx.x, A3, A2, A1p);
    DSVG_CLK_END(0);
}

// MODE 0: levels 3,2,1 from s3 -> pixels (P pictures).  MODE 1: levels 3,2 from s3 -> LL1 in s1 (I pictures,
// whose level 1 is the B4T kernel below).  MODE 2: levels 5,4 from s5 -> LL3 in s3 (every picture; feeds the
// other two modes).  The tile is IT_TX x IT_TY cells of the mode's top level.
template <bool FILT, int MODE, bool SYM>
static __device__ __forceinline__ void inv_haar_tile_body(const JobDev *__restrict__ jobs, const SbtGeo3 &G, int c0, int npl, int bxofs, int byofs,
                                                          int *__restrict__ A3, int *__restrict__ A2, int *__restrict__ A1)
{
    static_assert(!SYM || MODE != 2, "levels >= 4 live in the LL region: int32 coefficients");
    constexpr bool TO_PIX = (MODE == 0);
    constexpr int TOP = (MODE == 2) ? 5 : 3;
    int job, c;
    d_job_plane((int)blockIdx.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const JobDev &jb = jobs[job];
    const int W = g.W, H = g.H;
    const int32_t *coef = jb.coef + g.coff;
    const int inw = DSVG_RSU(W, TOP), inh = DSVG_RSU(H, TOP);
    const int32_t *s3 = (MODE == 2) ? jb.s5 + g.s5off : jb.s3 + g.s3off;       // LL band of level TOP
    // (offsets: a strip of the tile grid.  byofs < 0: the L-shaped rest of the grid beside the bxofs x (-byofs - 1) tiles that
    // k_inv_p_tile takes, blockIdx.x running through the right strip (all rows), then the bottom strip)
    int bx = (int)blockIdx.x + bxofs, by = (int)blockIdx.y + byofs;
    if (byofs < 0) {
        const int fx = bxofs, fy = -byofs - 1, ntx = (inw + IT_TX - 1) / IT_TX, nty = (inh + IT_TY - 1) / IT_TY;
        const int nr = (ntx - fx) * nty;
        int b = (int)blockIdx.x;
        if (b < nr) { by = b / (ntx - fx); bx = fx + b - by * (ntx - fx); }
        else { b -= nr; by = b / fx; bx = b - by * fx; by += fy; }
    }
    const int I0 = bx * IT_TX, J0 = by * IT_TY;
    const int tid = threadIdx.x;
    const bool isP = jb.isP != 0;
    // (k_inv_tile54_all: one grid sized for the largest plane serves all three -- a tile beyond this plane's band has nothing to do)
    if (MODE == 2 && g.lvls >= 5 && (I0 >= inw || J0 >= inh)) return;

    if (MODE == 2 && g.lvls < 5) {
        // tiny planes (<= 16 samples a side): level 5 (and 4) do not exist, the 1x1 band passes through
        if (blockIdx.x | blockIdx.y | tid) return;
        int32_t *s3o = jb.s3 + g.s3off;
        const int ll = s3[0];
        if (g.lvls == 4) {
            const LvlGeo L4 = mk_lvl(W, H, 4, jb.hqp[4], true);
            const auto D4 = mk_det<false, 0>(jb, c, coef, W, L4);
            int o[4];
            inv_cell<FILT, false>(&ll, 1, 0, 0, L4, D4, o);
            const int w3o = DSVG_RSU(W, 3), h3o = DSVG_RSU(H, 3);
            s3o[0] = o[0];
            if (w3o > 1) s3o[1] = o[1];
            if (h3o > 1) s3o[w3o] = o[2];
            if (w3o > 1 && h3o > 1) s3o[w3o + 1] = o[3];
        } else s3o[0] = ll;
        return;
    }
    if constexpr (MODE == 0 && SYM) {
        // sparse P pictures of the encoder, tiles away from the right / bottom edge (cells complete, neighbours in band)
        const QLevel Lq1 = q_level<2>(jb.hz[c]);
        if (jb.nzf != nullptr && jb.ref != nullptr && I0 + IT_TX + 1 < inw && J0 + IT_TY + 1 < inh &&
            ((Lq1.sw | Lq1.base0 | Lq1.base1 | Lq1.base2) & 3) == 0) {
            inv_p_fast<FILT>(jb, g, c, I0, J0, tid, A3, A2, reinterpret_cast<unsigned *>(A1));
            return;
        }
#ifdef AB_FAST_ONLY
        return;
#endif
    }
    // ---- phase 0: EVERY global load of the tile is issued here, before the first barrier, so the three level
    // phases below only wait on LDS: one memory round trip per workgroup instead of one per phase and loop pass.
    static_assert(A3H * A3W <= 256 && (IT_TY + 2) * (IT_TX + 2) <= 256, "one pass per thread");
    constexpr int N2 = ((2 * IT_TY + 2) * (2 * IT_TX + 2) + 255) / 256;          // level TOP-1 passes
    constexpr int N1 = ((4 * IT_TY) * IT_TX + 255) / 256;                        // level 1 passes (MODE 0)
    int a3v = 0;
    if (tid < A3H * A3W) {
        const int ly = tid / A3W, lx = tid - ly * A3W;
        const int cx = I0 - 2 + lx, cy = J0 - 2 + ly;
        if (cx >= 0 && cy >= 0 && cx < inw && cy < inh) a3v = s3[(size_t)cy * inw + cx];
    }
    // Sparse P pictures (MODE 0, SYM, jb.nzf set): when no patch of the tile (halo included) carries a detail symbol and
    // every LL3 value in reach is zero, every level's output is zero (no nudge fires on a flat band: mx == mn == 0) and
    // the reconstruction is the prediction (sbc2int gives 128, addf adds pred - 128).  The "next LL" of a region's last
    // complete cell is a detail of column / row 0 (sbt.c:463-527), so tiles that reach the last column / row also look
    // at those patches.  The flags are requested here with everything else; the decision rides on the first barrier.
    int nzv = 0;
    if constexpr (MODE == 0 && SYM) {
        nzv = 1;
        if (jb.nzf != nullptr) {
            const uint8_t *pfl = jb.pflag + g.s3off;
            nzv = a3v;
            if (tid < A3H * A3W) {
                const int ly = tid / A3W, lx = tid - ly * A3W;
                const int cx = I0 - 2 + lx, cy = J0 - 2 + ly;
                if (cx >= 0 && cy >= 0 && cx < inw && cy < inh) nzv |= pfl[(size_t)cy * inw + cx];
                if (lx == 0 && I0 + IT_TX + 2 >= inw && cy >= 0 && cy < inh) nzv |= pfl[(size_t)cy * inw];
                if (ly == 0 && J0 + IT_TY + 2 >= inh && cx >= 0 && cx < inw) nzv |= pfl[cx];
            }
        }
    }
    const LvlGeo L3 = mk_lvl(W, H, TOP, jb.hqp[TOP], true);
    const auto D3 = mk_det<SYM, 0>(jb, c, coef, W, L3);
    Raw3 q3; bool ok3 = false;
    {
        const int ly = tid / (IT_TX + 2), lx = tid - ly * (IT_TX + 2);
        const int cx = I0 - 1 + lx, cy = J0 - 1 + ly;
        ok3 = tid < (IT_TY + 2) * (IT_TX + 2) && cx >= 0 && cy >= 0 && cx < L3.wo && cy < L3.ho;
        q3.lh = q3.hl = q3.hh = q3.k = q3.m = 0;
        if (ok3) q3 = D3.fetch(cx, cy, 2 * cx + 1 < L3.ws, 2 * cy + 1 < L3.hs);
    }
    const LvlGeo L2 = mk_lvl(W, H, TOP - 1, jb.hqp[TOP - 1], true);
    const auto D2 = mk_det<SYM, 1>(jb, c, coef, W, L2);
    Raw3 q2[N2]; bool ok2[N2];
#pragma unroll
    for (int u = 0; u < N2; u++) {
        const int i = tid + 256 * u;
        const int ly = i / (2 * IT_TX + 2), lx = i - ly * (2 * IT_TX + 2);
        const int cx = 2 * I0 - 1 + lx, cy = 2 * J0 - 1 + ly;
        ok2[u] = i < (2 * IT_TY + 2) * (2 * IT_TX + 2) && cx >= 0 && cy >= 0 && cx < L2.wo && cy < L2.ho;
        q2[u].lh = q2[u].hl = q2[u].hh = q2[u].k = q2[u].m = 0;
        if (ok2[u]) q2[u] = D2.fetch(cx, cy, 2 * cx + 1 < L2.ws, 2 * cy + 1 < L2.hs);
    }
    // level 1 (MODE 0): the vector-loadable groups and their prediction pixels
    const LvlGeo L = mk_lvl(W, H, 1, jb.hqp[1], !isP);        // LVL_TEST: P level 1 unscaled
    const auto D = mk_det<SYM, 2>(jb, c, coef, W, L);
    const uint8_t *pred = (TO_PIX && jb.ref != nullptr) ? jb.pred + g.poff : nullptr;
    Raw4<SYM> q1[N1]; bool fast1[N1]; uint2 pv1[N1][2];
    if constexpr (TO_PIX) {
        bool vec_ok;
        if constexpr (SYM) vec_ok = ((D.L.sw | D.L.base0 | D.L.base1 | D.L.base2) & 3) == 0;
        else vec_ok = (((W | L.wo) & 3) == 0) && ((g.coff & 3) == 0) && ((((uintptr_t)jb.coef) & 15) == 0);
#pragma unroll
        for (int u = 0; u < N1; u++) {
            const int it = tid + 256 * u;
            const int ly = it / IT_TX, gx = it - ly * IT_TX;
            const int cy = 4 * J0 + ly, cx0 = 4 * I0 + 4 * gx;
            fast1[u] = it < (4 * IT_TY) * IT_TX && cy < L.ho && vec_ok && cx0 + 3 < (L.ws >> 1) && 2 * cy + 1 < L.hs;
            if (fast1[u]) q1[u] = D.fetch4(cx0, cy);
            pv1[u][0] = pv1[u][1] = make_uint2(0, 0);
            if (pred && it < (4 * IT_TY) * IT_TX && cy < L.ho) {
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
                    if (2 * cy + rr < g.ph) pv1[u][rr] = *reinterpret_cast<const uint2 *>(pred + (size_t)(2 * cy + rr) * g.pstride + 2 * (4 * I0 + 4 * gx));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    if (tid < A3H * A3W) A3[tid] = a3v;
    if constexpr (MODE == 0 && SYM) {
        if (!__syncthreads_or(nzv != 0)) {
            const uint8_t *predz = jb.ref != nullptr ? jb.pred + g.poff : nullptr;
            uint8_t *outz = (jb.recon ? jb.recon : jb.xf) + g.poff;
            if (tid == 0 && jb.stat) atomicAdd(jb.stat + 64 * (2 + (c != 0)) + (((blockIdx.x >> 3) + 5 * blockIdx.y + 11 * blockIdx.z) & 63), 1u);
            if (predz == outz) return;                           // the prediction was written in place (ping-pong slots)
            const int px0t = 8 * I0, py0t = 8 * J0;               // tile origin in pixels
            for (int u = tid; u < (8 * IT_TY) * (8 * IT_TX / 16); u += 256) {
                const int ry = u / (8 * IT_TX / 16), ux = u - ry * (8 * IT_TX / 16);
                const int y = py0t + ry, x = px0t + 16 * ux;
                if (y >= g.ph || x >= g.pw) continue;
                uint8_t *d = outz + (size_t)y * g.pstride + x;
                const uint8_t *sp = predz ? predz + (size_t)y * g.pstride + x : nullptr;
                if (x + 16 <= g.pw && ((((uintptr_t)d) | (uintptr_t)sp) & 15) == 0) {
                    *reinterpret_cast<uint4 *>(d) = sp ? *reinterpret_cast<const uint4 *>(sp) : make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
                } else {
                    for (int i = 0; i < 16 && x + i < g.pw; i++) d[i] = sp ? sp[i] : (uint8_t)128;
                }
            }
            return;
        }
        if (tid == 0 && jb.stat && jb.nzf) atomicAdd(jb.stat + 64 * (c != 0) + (((blockIdx.x >> 3) + 5 * blockIdx.y + 11 * blockIdx.z) & 63), 1u);
    } else __syncthreads();
    if (ok3) {      // level TOP: cells I0-1 .. I0+TX (halo 1)
        const int ly = tid / (IT_TX + 2), lx = tid - ly * (IT_TX + 2);
        int o[4];
#ifdef AB_NO_L23
        o[0] = o[1] = o[2] = o[3] = A3[(ly + 1) * A3W + (lx + 1)] + q3.lh;
#else
        inv_cell_raw<FILT, MODE != 2>(A3 + (ly + 1) * A3W + (lx + 1), A3W, I0 - 1 + lx, J0 - 1 + ly, L3, D3, q3, o);
#endif
        int *d = A2 + (2 * ly) * A2W + 2 * lx;
        d[0] = o[0]; d[1] = o[1]; d[A2W] = o[2]; d[A2W + 1] = o[3];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < N2; u++) {      // level TOP-1: cells 2*I0-1 .. 2*I0+2*TX (halo 1)
        if (!ok2[u]) continue;
        const int i = tid + 256 * u;
        const int ly = i / (2 * IT_TX + 2), lx = i - ly * (2 * IT_TX + 2);
        int o[4];
#ifdef AB_NO_L23
        o[0] = o[1] = o[2] = o[3] = A2[(ly + 1) * A2W + (lx + 1)] + q2[u].lh;
#else
        inv_cell_raw<FILT, MODE != 2>(A2 + (ly + 1) * A2W + (lx + 1), A2W, 2 * I0 - 1 + lx, 2 * J0 - 1 + ly, L2, D2, q2[u], o);
#endif
        int *d = A1 + (2 * ly) * A1W + 2 * lx;
        d[0] = o[0]; d[1] = o[1]; d[A1W] = o[2]; d[A1W + 1] = o[3];
    }
    __syncthreads();

    if (!TO_PIX) {
        int32_t *s1 = (MODE == 2) ? jb.s3 + g.s3off : jb.s1 + g.s1off;         // LL band of level TOP-2
        const int ow = DSVG_RSU(W, TOP - 2), oh = DSVG_RSU(H, TOP - 2);
        for (int i = tid; i < (4 * IT_TY) * (4 * IT_TX); i += 256) {
            const int ly = i / (4 * IT_TX), lx = i - ly * (4 * IT_TX);
            const int cx = 4 * I0 + lx, cy = 4 * J0 + ly;
            if (cx < ow && cy < oh) s1[(size_t)cy * ow + cx] = A1[(ly + 2) * A1W + lx + 2];
        }
        return;
    }

    // level 1 + sbc2int (+ prediction add) : each work item = 4 adjacent cells = 8 px x 2 rows
    uint8_t *outp = (jb.recon ? jb.recon : jb.xf) + g.poff;
#pragma unroll
    for (int u = 0; u < N1; u++) {
        const int it = tid + 256 * u;
        if (it >= (4 * IT_TY) * IT_TX) continue;
        const int ly = it / IT_TX, gx = it - ly * IT_TX;       // gx: group of 4 cells
        const int cy = 4 * J0 + ly;
        if (cy >= L.ho) continue;
        int r0[8], r1[8];
        const int cx0 = 4 * I0 + 4 * gx;
        const int px0 = 2 * (4 * I0 + 4 * gx);                // pixel x of the group
        bool done = false;
        if constexpr (FILT && SYM) {
            // packed path: four complete interior cells (every cell takes the horizontal nudge with in-band
            // neighbours; the row takes the vertical nudge or, on the first row, none), both output rows inside the
            // plane, 8 whole pixels
            const bool hx = cx0 > 0 && 2 * (cx0 + 3) < L.wfull - 1 && cx0 + 4 < L.wo;
            const bool vy = 2 * cy > 0 && 2 * cy < L.hfull - 1;
            if (fast1[u] && hx && (!vy || cy + 1 < L.ho) && 2 * cy + 1 < g.ph && px0 + 8 <= g.pw) {
                int lhv[4], hlv[4], hhv[4];
                D.finish4(q1[u], lhv, hlv, hhv);
                const int *rc = A1 + (ly + 2) * A1W + (4 * gx + 2);
                const short hq = (short)L.hqp;
                unsigned row0[2], row1[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {             // cells (0,1) then (2,3)
                    const int k = 2 * h2;
                    const s16x2 C = pk2(rc[k], rc[k + 1]);
                    s16x2 LH = pk2(lhv[k], lhv[k + 1]), HL = pk2(hlv[k], hlv[k + 1]);
                    const s16x2 HH = pk2(hhv[k], hhv[k + 1]);
#ifndef AB_NO_L1_NUDGE
                    LH = pk_nudge(C, pk2(rc[k - 1], rc[k]), pk2(rc[k + 1], rc[k + 2]), LH, hq);
                    if (vy) HL = pk_nudge(C, pk2(rc[k - A1W], rc[k + 1 - A1W]), pk2(rc[k + A1W], rc[k + 1 + A1W]), HL, hq);
#endif
                    const s16x2 sA = C + HL, sB = LH + HH, sC = C - HL, sD = LH - HH;
                    row0[h2] = pk_pixels(pk_div4(sA + sB), pk_div4(sA - sB), pred != nullptr, h2 ? pv1[u][0].y : pv1[u][0].x);
                    row1[h2] = pk_pixels(pk_div4(sC + sD), pk_div4(sC - sD), pred != nullptr, h2 ? pv1[u][1].y : pv1[u][1].x);
                }
                uint8_t *dst = outp + (size_t)(2 * cy) * g.pstride + px0;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(row0[0], row0[1]);
                *reinterpret_cast<uint2 *>(dst + g.pstride) = make_uint2(row1[0], row1[1]);
                done = true;
            }
        }
        if (done) continue;
        if (fast1[u]) {     // four complete cells whose details came in as three vector loads
            int lhv[4], hlv[4], hhv[4];
            D.finish4(q1[u], lhv, hlv, hhv);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int o[4];
                inv_cell_vals<FILT, true>(A1 + (ly + 2) * A1W + (4 * gx + k + 2), A1W, cx0 + k, cy, L, D, lhv[k], hlv[k], hhv[k], o);
                r0[2 * k] = o[0]; r0[2 * k + 1] = o[1];
                r1[2 * k] = o[2]; r1[2 * k + 1] = o[3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int lx = 4 * gx + k, cx = 4 * I0 + lx;
                int o[4] = {0, 0, 0, 0};
                if (cx < L.wo) inv_cell<FILT, true>(A1 + (ly + 2) * A1W + (lx + 2), A1W, cx, cy, L, D, o);
                r0[2 * k] = o[0]; r0[2 * k + 1] = o[1];
                r1[2 * k] = o[2]; r1[2 * k + 1] = o[3];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int y = 2 * cy + rr;
            if (y >= g.ph) continue;
            const int *v = rr ? r1 : r0;
            uint8_t *dst = outp + (size_t)y * g.pstride + px0;
            unsigned lo = 0, hi = 0;
            const uint2 pv = pv1[u][rr];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int s = d_sat8(v[i] + 128);
                if (pred) {
                    const int p = (int)(((i < 4 ? pv.x : pv.y) >> (8 * (i & 3))) & 0xff);
                    s = d_sat8(s + p - 128);                   // dsv_frame_add / addf bmc.c:29-41
                }
                if (i < 4) lo |= (unsigned)s << (8 * i); else hi |= (unsigned)s << (8 * (i - 4));
            }
            if (px0 + 8 <= g.pw) {
                *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (px0 + i < g.pw) dst[i] = (uint8_t)(((i < 4 ? lo : hi) >> (8 * (i & 3))) & 0xff);
            }
        }
    }
}

template <bool FILT, int MODE, bool SYM>
__global__ __launch_bounds__(256) void k_inv_haar_tile(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl, int bxofs, int byofs)
{
    __shared__ int A3[A3H * A3W];
    __shared__ int A2[A2H * A2W];
    __shared__ int A1[A1H * A1W];
    inv_haar_tile_body<FILT, MODE, SYM>(jobs, G, c0, npl, bxofs, byofs, A3, A2, A1);
}
// levels 5, 4 (LL5 -> LL3) of ALL THREE planes of every job in one launch (round 4: a launch less on the chain of every frame
// step): blockIdx.z = 3 * job + plane, the grid is the luma band's; luma takes the filtered body (sbt.c:438-574), chroma the plain one
__global__ __launch_bounds__(256) void k_inv_tile54_all(const JobDev *__restrict__ jobs, SbtGeo3 G)
{
    __shared__ int A3[A3H * A3W];
    __shared__ int A2[A2H * A2W];
    __shared__ int A1[A1H * A1W];
    if (blockIdx.z % 3 == 0) inv_haar_tile_body<true, 2, false>(jobs, G, 0, 3, 0, 0, A3, A2, A1);
    else inv_haar_tile_body<false, 2, false>(jobs, G, 0, 3, 0, 0, A3, A2, A1);
}

// --------------------------------------------------------------------------------------------
// inverse levels 3,2,1 of a sparse P picture WITHOUT the smoothing filter (chroma, sbt.c:352-435): a cell's outputs depend
// on nothing but its own LL value and details, so an 8x8-pixel patch is a closed computation -- one thread per patch, no LDS,
// no halo, no barrier (the tile kernel spent its time waiting on four barriers and a tile's worth of loads per workgroup:
// 1.4 ms per step for 680 instructions per wave).  Most patches carry no detail symbol (flag off): every level's output is
// then the scaled LL value, the whole patch gets ONE residual value v = ((LL3 * 5/4 / 4) * 5/4 / 4) / 4, and when that is zero
// -- a small LL3 -- the reconstruction is the prediction that is already in place: nothing is loaded or stored.
// Patches [0, imax) x [0, jmax): whole patches inside the picture; the tile kernel takes the strips beyond.
// --------------------------------------------------------------------------------------------
// jpart (round 4): patch row jpart = jmax - 1 is the plane's last and holds FOUR pixel rows (plane height = 8k + 4: the chroma of 1080 lines).  Then
// levels 1 and 2 are complete for those rows and only level 3's last cell row is the odd one (sbt.c:392-431: LL and LH alone, outputs (LL +- LH) / 4):
// the patch is the same closed computation with half of its rows.  Until now the whole last TILE row went to the general tile kernel for it --
// 6 % of the plane at a quarter of this kernel's rate, and a launch per frame step.  -1: every patch of the launch is whole.
#ifndef INV_PATCH_C_WPE
#define INV_PATCH_C_WPE 0           // 66 VGPRs with the border pass (64 without): seven waves per SIMD.  Pinned at eight (56 VGPRs, six scalars spilled) 2.7 -> 3.3 ms per step
#endif
#if INV_PATCH_C_WPE
#define INV_PATCH_C_ATTR __attribute__((amdgpu_waves_per_eu(INV_PATCH_C_WPE, INV_PATCH_C_WPE)))
#else
#define INV_PATCH_C_ATTR
#endif
__global__ __launch_bounds__(256) INV_PATCH_C_ATTR void k_inv_patch_c(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl, int imax, int jmax,
                                                                      XcdGrid XG, int plain, int jpart, int fb)
{
    Blk3 B;                                                 // one-dimensional launch in XCD order (d_xcd_blk3)
    if (!d_xcd_blk3(XG, B, plain != 0)) return;
    int job, c;
    d_job_plane(B.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const int I = B.x * 64 + threadIdx.x, J = B.y * 4 + threadIdx.y;
    if ((I >= imax || J >= jmax) && !fb) return;          // (fb: the lanes beyond the plane help with its border below)
    // fb (round 5): this launch covers every patch of the chroma planes, the luma plane of these pictures is finished (the luma inverse ran
    // before it on this stream) -- the waves at the planes' edges write the reconstruction's BORDER, what k_extend16 did in a launch of its
    // own behind the inverse transforms (one link of every frame step's chain).  The side borders are a wave's work, not a thread's: the
    // wave (= 64 patches of one patch row) that holds the row's first / last patch shares the rows out over its lanes -- a lane takes a
    // pixel row of the chroma plane or of the luma rows beside it (and, in the plane's first / last patch row, the corner rows above /
    // below): one 8-byte load of the row's edge pixels, then the stores, in jb.ext's columns (units of 16) and rows (units of 8): k_extend16's
    // units.  The rows above / below a plane are each patch's own columns: every lane of the first / last patch row's waves.  The LUMA part
    // -- two thirds of it -- is done first, under the patch body's own loads (the plane is complete); the chroma part after the body, behind
    // a fence (a lane reads rows its neighbours in the wave have just stored).  As one thread's loop over its 24 rows the edge patch kept its
    // wave alive for 24 dependent round trips (k_inv_patch_c 2.0 -> 3.6 ms per step); as one section behind the body 2.0 -> 2.5.  In the luma
    // kernel itself the border stores cost k_inv_p_tile its seventh wave per SIMD (5 to 10 spilled registers whichever way they were written).
    // (offsets left of / above a plane's first pixel are negative: signed 64-bit pointer steps)
    const bool bwl = B.x == 0, bwr = (imax - 1) / 64 == B.x, btop = J == 0, bbot = J == jmax - 1;
    const int bnrow = J == jpart ? 4 : 8;
    const int lane = (int)threadIdx.x;
    const bool top = btop, bot = bbot, wl = bwl, wr = bwr;
    const JobDev &jb = jobs[job];
    auto plane_sides = [&](const SbtGeo &q, const short *ex, int y0, int nr, int wpx) {      // wpx: the plane's width in pixels
        const auto pl = dsvg_global((jb.recon ? jb.recon : jb.xf) + q.poff);
        const long sl = (long)q.pstride;
        const int el = min(DSVG_BORDER, (ex[0] + 15) & ~15), er_ = min(DSVG_BORDER, (ex[1] + 15) & ~15);
        const int et = top ? min(DSVG_BORDER, (ex[2] + 7) & ~7) : 0, eb_ = bot ? min(DSVG_BORDER, (ex[3] + 7) & ~7) : 0;
        const int nit = nr + et + eb_;                       // rows y0 - et .. y0 + nr - 1 + eb_
        for (int side = 0; side < 2; side++) {
            if (side ? !wr : !wl) continue;
            const int n = side ? er_ : el;
            if (!n) continue;
            for (int i = lane; i < nit; i += 64) {
                const int y = i - et, ys = min(max(y, 0), nr - 1);                 // the row this border row copies its edge pixel from
                const uint2 w = dsvg_ld2(pl + ((y0 + ys) * sl + (side ? wpx - 8 : 0)));
                const unsigned v = (side ? (w.y >> 24) : (w.x & 0xffu)) * 0x01010101u;
                const auto d = pl + ((y0 + y) * sl + (side ? (long)wpx : -(long)n));
                for (int k = 0; k < n; k += 16) dsvg_st4(d + k, make_uint4(v, v, v, v));
            }
        }
    };
    auto plane_tb = [&](const SbtGeo &q, const short *ex, int x0, int ncol8, int y0, int nr) {
        // this patch's columns (ncol8 groups of 8 pixels from x0) of the rows above the plane's first / below its last pixel row
        const auto pl = dsvg_global((jb.recon ? jb.recon : jb.xf) + q.poff);
        const long sl = (long)q.pstride;
        const int et = top ? min(DSVG_BORDER, (ex[2] + 7) & ~7) : 0, eb_ = bot ? min(DSVG_BORDER, (ex[3] + 7) & ~7) : 0;
        uint2 wt[4], wb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            wt[k] = wb[k] = make_uint2(0u, 0u);
            if (k < ncol8 && et) wt[k] = dsvg_ld2(pl + (y0 * sl + x0 + 8 * k));
            if (k < ncol8 && eb_) wb[k] = dsvg_ld2(pl + ((y0 + nr - 1) * sl + x0 + 8 * k));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= ncol8) break;
            for (int r = 1; r <= et; r++) dsvg_st2(pl + ((y0 - r) * sl + x0 + 8 * k), wt[k]);
            for (int r = 1; r <= eb_; r++) dsvg_st2(pl + ((y0 + nr - 1 + r) * sl + x0 + 8 * k), wb[k]);
        }
    };
    if (fb && J < jmax && c == c0 && (bwl || bwr || btop || bbot)) {
        const SbtGeo &gy = G.g[0];
        const int hr = gy.pw / g.pw, vr = gy.ph / g.ph;                      // luma pixels per chroma pixel (1, 2 or 4 across; 1 or 2 down)
        if (bwl || bwr) plane_sides(gy, jb.ext, 8 * J * vr, bnrow * vr, 8 * imax * hr);
        if ((btop || bbot) && I < imax) plane_tb(gy, jb.ext, 8 * I * hr, hr, 8 * J * vr, bnrow * vr);
    }
    // (the patch row is wave-uniform -- a wave is one row of patches: the partial row takes a body of its own, the whole patches' body
    // carries none of its tests)
    auto run = [&](auto PART_) {
    constexpr bool part = decltype(PART_)::value;
    constexpr int nrow = part ? 4 : 8;
    const JobDev &jb = jobs[job];
    const unsigned pidx = (unsigned)(J * g.w3 + I);
    const int ll3 = dsvg_at(dsvg_global(static_cast<const int32_t *>(jb.s3 + g.s3off)), pidx);
    const int pf = dsvg_global(static_cast<const uint8_t *>(jb.pflag + g.s3off))[pidx];
    const auto pred = dsvg_global(static_cast<const uint8_t *>(jb.pred + g.poff));
    const auto outp = dsvg_global((jb.recon ? jb.recon : jb.xf) + g.poff);
    const bool inplace = (const DSVG_GLOBAL uint8_t *)outp == pred;
    const unsigned stride = (unsigned)g.pstride, p0 = (unsigned)(8 * J) * stride + 8u * (unsigned)I;
    if (jb.stat) {                                      // (counting on: bench.py's untimed selection step, tests) flagged patches / patches moved
        const int v1_ = d_div4<true>(d_div4<true>(d_ll_up_t<true>(d_div4<true>(d_ll_up_t<true>(ll3)))));
        const unsigned nfl = (unsigned)__popcll(__ballot(pf != 0)), nmv = (unsigned)__popcll(__ballot(pf == 0 && (d_clamp(v1_, -128, 127) != 0 || !inplace)));
        if ((threadIdx.x & 63) == 0) {
            if (nfl) atomicAdd(jb.stat + 64 * 5 + ((blockIdx.x + threadIdx.y) & 63), nfl);
            if (nmv) atomicAdd(jb.stat + 64 * 6 + ((blockIdx.x + threadIdx.y) & 63), nmv);
        }
    }
    if (!pf) {
        const int v3 = d_div4<true>(d_ll_up_t<true>(ll3));
        const int v2 = d_div4<true>(d_ll_up_t<true>(v3));
        const int v1 = d_div4<true>(v2);                        // level 1 of a P picture is unscaled
        const int cv = d_clamp(v1, -128, 127);
        if (cv == 0) {                                          // reconstruction = prediction
            if (inplace) return;
#pragma unroll
            for (int r = 0; r < 8; r++) if (r < nrow) dsvg_st2(outp + (p0 + r * stride), dsvg_ld2(pred + (p0 + r * stride)));
            return;
        }
        const s16x2 cc = s16x2{(short)cv, (short)cv};
        uint2 pv[8];
#pragma unroll
        for (int r = 0; r < 8; r++) pv[r] = r < nrow ? dsvg_ld2(pred + (p0 + r * stride)) : make_uint2(0u, 0u);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r >= nrow) break;
            unsigned o[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned pw_ = h ? pv[r].y : pv[r].x;
                const s16x2 e = cc + __builtin_bit_cast(s16x2, pw_ & 0x00ff00ffu), od = cc + __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, pw_, 0x0c030c01u));
                unsigned eb, ob;
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(eb) : "v"(__builtin_bit_cast(unsigned, e)));
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ob) : "v"(__builtin_bit_cast(unsigned, od)));
                o[h] = __builtin_amdgcn_perm(ob, eb, 0x05010400u);
            }
            dsvg_st2(outp + (p0 + r * stride), make_uint2(o[0], o[1]));
        }
        return;
    }
    // the patch has detail symbols: the three levels in registers, symbols dequantised on the way (rare)
    const HzPlane &hp = jb.hz[c];
    const auto sym = dsvg_global(static_cast<const int16_t *>(jb.sym + jb.nz_off[c]));
    const auto stb = dsvg_global(jb.stable);
    const QLevel Q3 = q_level<0>(hp), Q2 = q_level<1>(hp), Q1 = q_level<2>(hp);
    const unsigned nbh = (unsigned)hp.nbh;
    auto flag = [&](const QLevel &Q, int cx, int cy) { return (int)stb[__umul24(__umul24((unsigned)cy, (unsigned)Q.dby) >> 14, nbh) + (__umul24((unsigned)cx, (unsigned)Q.dbx) >> 14)]; };
    int l2[2][2], l1[4][4];
    {
        const int f = flag(Q3, I, J), q = max(Q3.qp >> ((f & 2) ? 2 : (f != 0)), HZ_MINQ);
        const unsigned o = (unsigned)(J * Q3.sw + I);
        // (part: the band's odd last cell row -- no HL / HH, two outputs; sbt.c:418-431)
        const int LL = d_ll_up_t<true>(ll3), LH = dq_lo24(dsvg_at(sym, (unsigned)Q3.base0 + o), q),
                  HL = part ? 0 : dq_lo24(dsvg_at(sym, (unsigned)Q3.base1 + o), q), HH = part ? 0 : dq_lo24(dsvg_at(sym, (unsigned)Q3.base2 + o), q);
        l2[0][0] = d_div4<true>(LL + LH + HL + HH); l2[0][1] = d_div4<true>(LL - LH + HL - HH);
        l2[1][0] = d_div4<true>(LL + LH - HL - HH); l2[1][1] = d_div4<true>(LL - LH - HL + HH);
    }
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (part && j == 1) { l1[2][2 * i] = l1[2][2 * i + 1] = l1[3][2 * i] = l1[3][2 * i + 1] = 0; continue; }   // (level-2 cell row 2J + 1 does not exist)
            const int cx = 2 * I + i, cy = 2 * J + j;
            const int f = flag(Q2, cx, cy), q = max(Q2.qp >> ((f & 2) ? 2 : (f != 0)), HZ_MINQ);
            const unsigned o = (unsigned)(cy * Q2.sw + cx);
            const int LL = d_ll_up_t<true>(l2[j][i]), LH = dq_lo24(dsvg_at(sym, (unsigned)Q2.base0 + o), q), HL = dq_lo24(dsvg_at(sym, (unsigned)Q2.base1 + o), q),
                      HH = dq_lo24(dsvg_at(sym, (unsigned)Q2.base2 + o), q);
            l1[2 * j][2 * i] = d_div4<true>(LL + LH + HL + HH); l1[2 * j][2 * i + 1] = d_div4<true>(LL - LH + HL - HH);
            l1[2 * j + 1][2 * i] = d_div4<true>(LL + LH - HL - HH); l1[2 * j + 1][2 * i + 1] = d_div4<true>(LL - LH - HL + HH);
        }
    // level 1 + pixels one cell row (two pixel rows) at a time: few values alive (this path sets the kernel's registers)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (part && j >= 2) break;                              // (level-1 cell rows 4J + 2, 4J + 3 do not exist)
        int r0[8], r1[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cx = 4 * I + i, cy = 4 * J + j;
            const int sh = flag(Q1, cx, cy) ? Q1.sh1 : Q1.sh0;
            const unsigned o = (unsigned)(cy * Q1.sw + cx);
            const int LL = l1[j][i], LH = (int)((unsigned)(int)dsvg_at(sym, (unsigned)Q1.base0 + o) << sh), HL = (int)((unsigned)(int)dsvg_at(sym, (unsigned)Q1.base1 + o) << sh),
                      HH = (int)((unsigned)(int)dsvg_at(sym, (unsigned)Q1.base2 + o) << sh);
            r0[2 * i] = d_div4<true>(LL + LH + HL + HH); r0[2 * i + 1] = d_div4<true>(LL - LH + HL - HH);
            r1[2 * i] = d_div4<true>(LL + LH - HL - HH); r1[2 * i + 1] = d_div4<true>(LL - LH - HL + HH);
        }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = 2 * j + rr;
            const uint2 pw2 = dsvg_ld2(pred + (p0 + r * stride));
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int p = (int)(((i < 4 ? pw2.x : pw2.y) >> (8 * (i & 3))) & 0xff);
                const int sv = d_sat8(d_sat8((rr ? r1[i] : r0[i]) + 128) + p - 128);      // sbc2int, then dsv_frame_add / addf bmc.c:29-41
                if (i < 4) lo |= (unsigned)sv << (8 * i); else hi |= (unsigned)sv << (8 * (i - 4));
            }
            dsvg_st2(outp + (p0 + r * stride), make_uint2(lo, hi));
        }
    }
    };
    if (I < imax && J < jmax) { if (J == jpart) run(std::true_type{}); else run(std::false_type{}); }
    if (!fb || J >= jmax) return;
    if (!(bwl || bwr || btop || bbot)) return;              // (wave-uniform: a wave is one patch row)
    __threadfence_block();                                  // (a lane reads rows its neighbours in the wave have just stored)
    if (bwl || bwr) plane_sides(g, jobs[job].ext + 4, 8 * J, bnrow, 8 * imax);
    if ((btop || bbot) && I < imax) plane_tb(g, jobs[job].ext + 4, 8 * I, 1, 8 * J, bnrow);
}

// --------------------------------------------------------------------------------------------
// inverse, I pictures: level 1 biorthogonal, columns then rows (inv_b4t_2d sbt.c:253-265) + sbc2int
// --------------------------------------------------------------------------------------------
// level-1 cells per tile: BT_CX x BT_CY (-> 2 BT_CX x 2 BT_CY pixels).  Round 6: 60 x 32 instead of 32 x 32 -- the column pass has 2 (BT_CX + 2) x BT_CY / BT_SEG
// items for the workgroup's 256 threads: 136 with the square tile (the third and fourth wave idled through the kernel's heavier half), 248 now; the halo
// columns are 3 % instead of 6 % of the loads, and 960 / 480 cells (1080p luma / chroma) and 1920 (4K) are whole numbers of tiles
#ifndef BT_CX
#define BT_CX 60
#endif
#ifndef BT_CY
#define BT_CY 32
#endif
#define BT_VW (BT_CX + 2)       // columns k0-1 .. k0+BT_CX of each half
#ifndef BT_SEG
#define BT_SEG 16               // cell rows per item of the column pass (160 I pictures: cell by cell 1.155 ms, 4: 0.92, 8: 0.81, 16: 0.75; 32 spills)
#endif

// SYM (the encoder's I pictures): the level-1 details come from the int16 symbol planes k_fwd_b4t<true> left behind and are
// dequantised here (shift quantiser hzcc.c:221-224) -- the dequantised int32 bands are then neither written by the forward
// transform nor read back (3 + 3 B/sample less)
#ifndef INV_B4T_WPE
#define INV_B4T_WPE 0
#endif
#if INV_B4T_WPE
#define INV_B4T_ATTR __attribute__((amdgpu_waves_per_eu(INV_B4T_WPE, INV_B4T_WPE)))
#else
#define INV_B4T_ATTR
#endif
template <bool SYM>
__global__ __launch_bounds__(256) INV_B4T_ATTR void k_inv_b4t(const JobDev *__restrict__ jobs, SbtGeo3 G, int c0, int npl, XcdGrid XG, int plain)
{
    static_assert(BT_CX % 4 == 0 && BT_CY % BT_SEG == 0, "phase B works on four cells, phase A on BT_SEG cell rows");
    __shared__ int VL[2 * BT_CY][BT_VW];    // column-pass output, low-horizontal half
    __shared__ int VH[2 * BT_CY][BT_VW];    // column-pass output, high-horizontal half
    Blk3 B;                                 // one-dimensional launch in XCD order (d_xcd_blk3): a tile shares its halo columns / rows and
    if (!d_xcd_blk3(XG, B, plain != 0)) return;     // the lines its rows straddle with the neighbours in the same L2
    int job, c;
    d_job_plane(B.z, npl, c0, job, c);
    const SbtGeo g = G.g[c];
    const JobDev &jb = jobs[job];
    const int W = g.W, H = g.H, hw = W >> 1, hh = H >> 1;
    const int32_t *coef = jb.coef + g.coff;
    const int32_t *s1 = jb.s1 + g.s1off;
    const int k0 = B.x * BT_CX, m0 = B.y * BT_CY;
    const int tid = threadIdx.x;

    // phase A: vertical pass for columns k0-1..k0+BT_CX (clamped) of both halves, rows 2*m0 .. 2*m0+2*BT_CY-1.
    // One item = BT_SEG consecutive cell rows of one column: the BT_SEG + 2 rows it needs (X and Y values, the rows' shift
    // flags) are requested in ONE batch and each serves the three outputs around it -- a third of the loads of a cell-by-cell
    // walk and one memory round trip per item instead of one per cell (the kernel waits on memory, not on the VALUs)
    constexpr int NSEG = BT_CY / BT_SEG, NCOL = 2 * BT_VW;
    for (int i = tid; i < NCOL * NSEG; i += 256) {
        const int seg = i / NCOL, r = i - seg * NCOL;
        const int half = r >= BT_VW, kl = r - half * BT_VW;
        const int mb = m0 + seg * BT_SEG;                      // first output row of the item
        if (mb >= hh) continue;
        const int k = d_clamp(k0 - 1 + kl, 0, hw - 1);
        int X[BT_SEG + 2], Y[BT_SEG + 2];
        if constexpr (SYM) {
            const HzPlane &hp = jb.hz[c];
            const QLevel L1 = q_level<2>(hp);
            const auto sym = dsvg_global(static_cast<const int16_t *>(jb.sym + jb.nz_off[c]));
            const auto stb = dsvg_global(jb.stable);
            const auto s1g = dsvg_global(s1);
            const unsigned bxk = __umul24((unsigned)k, (unsigned)L1.dbx) >> 14, nbh = (unsigned)hp.nbh;
            const unsigned xb = (unsigned)(half ? L1.base0 : 0), yb = (unsigned)(half ? L1.base2 : L1.base1);
            int fl[BT_SEG + 2];
#pragma unroll
            for (int j = 0; j < BT_SEG + 2; j++) {
                const unsigned m = (unsigned)d_clamp(mb - 1 + j, 0, hh - 1);
                const unsigned o = __umul24(m, (unsigned)L1.sw) + (unsigned)k;
                fl[j] = stb[__umul24(__umul24(m, (unsigned)L1.dby) >> 14, nbh) + bxk];
                X[j] = half ? (int)dsvg_at(sym, xb + o) : dsvg_at(s1g, __umul24(m, (unsigned)g.w1) + (unsigned)k);
                Y[j] = dsvg_at(sym, yb + o);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < BT_SEG + 2; j++) {
                const int sh = fl[j] ? L1.sh1 : L1.sh0;       // shift quantiser of the row's block (hzcc.c:221-224)
                if (half) X[j] = (int)((unsigned)X[j] << sh);
                Y[j] = (int)((unsigned)Y[j] << sh);
            }
        } else {
#pragma unroll
            for (int j = 0; j < BT_SEG + 2; j++) {
                const int m = d_clamp(mb - 1 + j, 0, hh - 1);
                X[j] = half ? coef[(size_t)m * W + hw + k] : s1[(size_t)m * g.w1 + k];
                Y[j] = coef[(size_t)(hh + m) * W + (half ? hw : 0) + k];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        int (*V)[BT_VW] = half ? VH : VL;
#pragma unroll
        for (int j = 0; j < BT_SEG; j++) {
            if (mb + j >= hh) break;
            const int ml = seg * BT_SEG + j;
            V[2 * ml][kl] = d_rdiv8(X[j] + 3 * X[j + 1] + Y[j] - 3 * Y[j + 1]);
            V[2 * ml + 1][kl] = d_rdiv8(3 * X[j + 1] + X[j + 2] + 3 * Y[j + 1] - Y[j + 2]);
        }
    }
    __syncthreads();

    // phase B: horizontal pass, 4 cells (8 px) per work item
    uint8_t *outp = (jb.recon ? jb.recon : jb.xf) + g.poff;
    for (int it = tid; it < 2 * BT_CY * (BT_CX / 4); it += 256) {
        const int yl = it / (BT_CX / 4), gx = it - yl * (BT_CX / 4);
        const int y = 2 * m0 + yl;
        if (y >= H || y >= g.ph) continue;
        unsigned lo = 0, hi = 0;
        // the item's six columns of both halves (local columns 4gx .. 4gx + 5: cells 4gx .. 4gx + 3 and one neighbour on each side), read
        // once.  Phase A filled the halo columns with CLAMPED plane columns, so the neighbour of the plane's first / last cell is the cell
        // itself there already (inv_b4t_h's edges, sbt.c:137-165): no edge select here (round 4: each cell read its three columns
        // again behind two selects)
        int vl[6], vh[6];
#pragma unroll
        for (int q = 0; q < 6; q++) { vl[q] = VL[yl][4 * gx + q]; vh[q] = VH[yl][4 * gx + q]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kl = 4 * gx + q, k = k0 + kl;
            int e = 0, o = 0;
            if (k < hw) {
                const int Lp = vl[q], L0 = vl[q + 1], Ln = vl[q + 2];
                const int Hp = vh[q], H0 = vh[q + 1], Hn = vh[q + 2];
                e = d_sat8(d_rdiv8(Lp + 3 * L0 + Hp - 3 * H0) + 128);
                o = d_sat8(d_rdiv8(3 * L0 + Ln + 3 * H0 - Hn) + 128);
            }
            if (q < 2) lo |= ((unsigned)e << (16 * q)) | ((unsigned)o << (16 * q + 8));
            else       hi |= ((unsigned)e << (16 * (q - 2))) | ((unsigned)o << (16 * (q - 2) + 8));
        }
        const int px0 = 2 * (k0 + 4 * gx);
        uint8_t *dst = outp + (size_t)y * g.pstride + px0;
        if (px0 + 8 <= g.pw) {
            *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (px0 + i < g.pw) dst[i] = (uint8_t)(((i < 4 ? lo : hi) >> (8 * (i & 3))) & 0xff);
        }
    }
}

// --------------------------------------------------------------------------------------------
// host launchers
// --------------------------------------------------------------------------------------------
static inline dim3 grid3(int w3, int h3, int nz) { return dim3((w3 + 63) / 64, (h3 + 3) / 4, nz); }
// a logical gx x gy x gz grid for a kernel that decodes it with d_xcd_blk3 (DSV1_NO_XCD_ORDER: the A/B switch -- the
// same kernels with the hardware's round-robin order, i.e. neighbouring tiles on different XCDs)
static inline dim3 tile_grid(int gx, int gy, int gz) { return dim3(xcd_grid(gx * gy * gz)); }
static inline int xcd_plain() { static const int v = getenv("DSV1_NO_XCD_ORDER") != nullptr; return v; }

int sbt_tail_supported(const SbtGeo &g)
{
    const long n5 = (long)g.w5 * g.h5;
    if (n5 * 4 > 160 * 1024 - 256) return 0;
    if (g.lvls < 3) return 0;                       // planes of at most 4 samples a side do not occur (luma >= 32)
    const long c6 = (long)DSVG_RSU(g.W, TAIL_LV) * DSVG_RSU(g.H, TAIL_LV);
    return c6 <= (long)TAIL_MAXC * TAIL_THREADS;
}

// forward transform of planes [c0, c0+npl) of njobs jobs (all P or all I)
#define PB(kid, bytes) do { if (pf) pf->begin(st, kid, bytes); } while (0)
#define PE() do { if (pf) pf->end(st); } while (0)

bool mc_fusable(const McGeo &MG)
{
    return (MG.blk_w >> MG.hs) % 8 == 0 && (MG.blk_h >> MG.vs) % 8 == 0 && MG.blk_w % 8 == 0 && MG.blk_h % 8 == 0;
}

// levels 4..5 (LL3 -> LL5) of planes [c0, c0+npl) of njobs jobs in one launch (the grid is sized for the largest plane; the
// threads of a smaller plane beyond its band leave at once); llq: with the LL quantiser
void launch_fwd_mid4(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, bool llq, Prof *pf)
{
    int w5 = 1, h5 = 1;
    double s3 = 0;
    for (int c = c0; c < c0 + npl; c++) {
        w5 = std::max(w5, G.g[c].w5); h5 = std::max(h5, G.g[c].h5);
        s3 += (double)G.g[c].w3 * G.g[c].h3 * njobs;
    }
    PB(KID_FWD_HAAR_MID4, s3 * 8.0);
    if (llq) hipLaunchKernelGGL((k_fwd_haar_mid<4, false, true>), grid3(w5, h5, njobs * npl), dim3(64, 4), 0, st, jobs, G, c0, npl);
    else     hipLaunchKernelGGL((k_fwd_haar_mid<4, false>), grid3(w5, h5, njobs * npl), dim3(64, 4), 0, st, jobs, G, c0, npl);
    PE();
}

void launch_fwd_sbt(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, int isP,
                    int from_src, Prof *pf, int with_tail, int fused, const McGeo *mc, const DMV *mvs0, int general_whole)
{
    const SbtGeo &g = G.g[c0];
    const int nz = njobs * npl;
    const double smp = (double)g.W * g.H * nz, s3 = (double)g.w3 * g.h3 * nz;
    const bool llq = fused == 2 || fused == 4;       // the LL quantiser in k_fwd_haar_mid<4> (JobDev.llq set by the caller)
    if (isP && fused && mc) {
        // motion compensation inside the transform: reference + source in, prediction + symbols out
        // two launches over the same grid: the lean kernel codes the common patches (few registers, no edge logic: it runs
        // at the HBM rate of its 3 B/sample: reference + source in, prediction out), the general kernel everything else
        // (picture edges, intra blocks, cells shared between scan regions) and returns at once for the common ones
        PB(c0 == 0 ? KID_FWD_MC_FAST_Y : KID_FWD_MC_FAST_C, smp * 3.0);
        const dim3 fg = grid3(g.w3, g.h3, nz);
        if (c0 == 0) hipLaunchKernelGGL((k_fwd_mc_fast<0>), tile_grid(fg.x, fg.y, fg.z), dim3(64, 4), 0, st, jobs, G, *mc, c0, npl, mvs0, mk_xcd_grid((int)fg.x, (int)fg.y, (int)fg.z), xcd_plain());
        else         hipLaunchKernelGGL((k_fwd_mc_fast<1>), tile_grid(fg.x, fg.y, fg.z), dim3(64, 4), 0, st, jobs, G, *mc, c0, npl, mvs0, mk_xcd_grid((int)fg.x, (int)fg.y, (int)fg.z), xcd_plain());
        PE();
        // Where can a patch fail fwd_fast_sel?  Intra blocks (anywhere: the caller knows), otherwise only in the first /
        // last row or column of patches (cells shared between scan regions, a ragged picture edge) -- unless a patch can
        // span two columns of the stability map, which depends on the geometry alone.  The general kernel is launched
        // over the whole grid, over the strips that can hold such patches, or not at all.
        bool whole = general_whole != 0 && !FWD_FAST_INTRA, top = false, bottom = false, left = false, right = false;
        for (int c = c0; c < c0 + npl; c++) {
            const SbtGeo &gc = G.g[c];
            const int sw0 = DSVG_RSU(gc.W, 3), sw1 = DSVG_RSU(gc.W, 2), sw2 = DSVG_RSU(gc.W, 1);
            const int sh0 = DSVG_RSU(gc.H, 3), sh1 = DSVG_RSU(gc.H, 2), sh2 = DSVG_RSU(gc.H, 1);
            left = left || 2 * sw0 > sw1 || 2 * sw1 > sw2;
            top = top || 2 * sh0 > sh1 || 2 * sh1 > sh2;
            right = right || (mc->w[c] & 7);
            bottom = bottom || (gc.ph & 7);
            const int d1 = (mc->nbh << 14) / sw2, d2 = (mc->nbh << 14) / sw1;       // HzRegion.dbx of transform levels 1, 2
            for (int I = 0; I < gc.w3 && !whole; I++)
                whole = ((4 * I * d1) >> 14) != (((4 * I + 3) * d1) >> 14) || ((2 * I * d2) >> 14) != (((2 * I + 1) * d2) >> 14);
        }
        const dim3 full = grid3(g.w3, g.h3, nz);
        auto general = [&](dim3 gr, int bxo, int byo, int bxs, int bys) {
            PB(c0 == 0 ? KID_FWD_MC_PIX_Y : KID_FWD_MC_PIX_C, 0.0);
            if (c0 == 0) hipLaunchKernelGGL((k_fwd_mc_pix<0>), gr, dim3(64, 4), 0, st, jobs, G, *mc, c0, npl, mvs0, 1, bxo, byo, bxs, bys);
            else         hipLaunchKernelGGL((k_fwd_mc_pix<1>), gr, dim3(64, 4), 0, st, jobs, G, *mc, c0, npl, mvs0, 1, bxo, byo, bxs, bys);
            PE();
        };
        if (whole) general(full, 0, 0, 1, 1);
        else {
            // the top and bottom strips in one launch (two block rows: 0 and the last), likewise left and right
            const int fy = (int)full.y, fx = (int)full.x;
            if ((top || bottom) && fy == 1) general(dim3(full.x, 1, nz), 0, 0, 1, 1);
            else if (top && bottom) general(dim3(full.x, 2, nz), 0, 0, 1, fy - 1);
            else if (top) general(dim3(full.x, 1, nz), 0, 0, 1, 1);
            else if (bottom) general(dim3(full.x, 1, nz), 0, fy - 1, 1, 1);
            if ((left || right) && fx == 1) general(dim3(1, full.y, nz), 0, 0, 1, 1);
            else if (left && right) general(dim3(2, full.y, nz), 0, 0, fx - 1, 1);
            else if (left) general(dim3(1, full.y, nz), 0, 0, 1, 1);
            else if (right) general(dim3(1, full.y, nz), fx - 1, 0, 1, 1);
        }
    } else if (isP) {
        PB(fused ? KID_FWD_HAAR_PIX_Q : KID_FWD_HAAR_PIX, smp * (fused ? 3.0 : 5.0));   // 1 B/sample in, 4 B/sample out (details + LL3); fused: 2 B symbols
        if (fused) hipLaunchKernelGGL((k_fwd_haar_pix<true>), grid3(g.w3, g.h3, nz), dim3(64, 4), 0, st, jobs, G, c0, npl, from_src);
        else       hipLaunchKernelGGL((k_fwd_haar_pix<false>), grid3(g.w3, g.h3, nz), dim3(64, 4), 0, st, jobs, G, c0, npl, from_src);
        PE();
    } else {
        PB(fused ? KID_FWD_B4T_Q : KID_FWD_B4T, smp * (fused ? 3.5 : 5.0));   // 1 in, LL1 1 + details 3 out; fused: LL1 1 + symbols 1.5 out
        if (fused) hipLaunchKernelGGL((k_fwd_b4t<true>), grid3((g.W + 7) / 8, (g.H + 7) / 8, nz), dim3(64, 4), 0, st, jobs, G, c0, npl, from_src);
        else       hipLaunchKernelGGL((k_fwd_b4t<false>), grid3((g.W + 7) / 8, (g.H + 7) / 8, nz), dim3(64, 4), 0, st, jobs, G, c0, npl, from_src);
        PE();
        PB(fused ? KID_FWD_HAAR_MID2_Q : KID_FWD_HAAR_MID2, smp * (fused ? 1.4 : 2.0));        // LL1 (1/4) in, levels 2..3 out
        if (fused) hipLaunchKernelGGL((k_fwd_haar_mid<2, true>), grid3(g.w3, g.h3, nz), dim3(64, 4), 0, st, jobs, G, c0, npl);
        else       hipLaunchKernelGGL((k_fwd_haar_mid<2, false>), grid3(g.w3, g.h3, nz), dim3(64, 4), 0, st, jobs, G, c0, npl);
        PE();
    }
    // levels 4..5 (LL3 -> LL5) for every picture type (fused >= 3: the caller launches them once for all planes and jobs
    // of the frame step, launch_fwd_mid4)
    if (fused < 3) launch_fwd_mid4(st, jobs, njobs, G, c0, npl, llq, pf);
    if (with_tail) {
        PB(KID_FWD_TAIL, (double)g.w5 * g.h5 * nz * 8.0);
        hipLaunchKernelGGL(k_fwd_tail, dim3(nz), dim3(TAIL_THREADS), (size_t)g.w5 * g.h5 * 4, st, jobs, G, c0, npl);
        PE();
    }
}

// the LDS tails of planes [c0, c0+npl) of all jobs in ONE launch (dynamic LDS sized for the largest plane)
void launch_sbt_tail(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, int inverse, Prof *pf)
{
    size_t lds = 0;
    double s3 = 0;
    for (int c = c0; c < c0 + npl; c++) {
        lds = std::max(lds, (size_t)G.g[c].w5 * G.g[c].h5 * 4);
        s3 += (double)G.g[c].w5 * G.g[c].h5 * njobs;
    }
    PB(inverse ? KID_INV_TAIL : KID_FWD_TAIL, s3 * 8.0);
    if (inverse) hipLaunchKernelGGL(k_inv_tail, dim3(njobs * npl), dim3(TAIL_THREADS), lds, st, jobs, G, c0, npl);
    else         hipLaunchKernelGGL(k_fwd_tail, dim3(njobs * npl), dim3(TAIL_THREADS), lds, st, jobs, G, c0, npl);
    PE();
}

// encoder: forward tail + LL quantiser + inverse tail of planes [c0, c0+npl) of all jobs in one launch
void launch_tail_q(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, Prof *pf)
{
    size_t lds = 0;
    double s3 = 0;
    for (int c = c0; c < c0 + npl; c++) {
        lds = std::max(lds, (size_t)G.g[c].w5 * G.g[c].h5 * 4);
        s3 += (double)G.g[c].w5 * G.g[c].h5 * njobs;
    }
    PB(KID_TAIL_Q, s3 * 8.0);
    // few workgroups (a small batch: the kernel is a link of a latency-bound chain): 1024 threads per band; many: 256 -- a
    // 16-wave workgroup would wait for a whole CU's worth of free wave slots beside the other coding stream's kernels
    if (njobs * npl <= 96) hipLaunchKernelGGL((k_tail_q<1024>), dim3(njobs * npl), dim3(1024), lds, st, jobs, G, c0, npl);
    else hipLaunchKernelGGL((k_tail_q<TAIL_THREADS>), dim3(njobs * npl), dim3(TAIL_THREADS), lds, st, jobs, G, c0, npl);
    PE();
}

// decoder: the prediction of every inter block left of block column ex0 / above block row ey0 (see k_mc_patch), luma then chroma
void launch_mc_patch(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, const McGeo &MG, const DMV *mvs0, int ex0, int ey0, Prof *pf)
{
    double smp = 0;
    for (int c = 0; c < 3; c++) smp += (double)G.g[c].W * G.g[c].H;
    PB(KID_MC, smp * njobs * 2.0);                       // reference in, prediction out
    {
        const dim3 fg = grid3(G.g[0].w3, G.g[0].h3, njobs);
        hipLaunchKernelGGL((k_mc_patch<0>), tile_grid(fg.x, fg.y, fg.z), dim3(64, 4), 0, st, jobs, G, MG, 0, 1, mvs0, mk_xcd_grid((int)fg.x, (int)fg.y, (int)fg.z), ex0, ey0);
    }
    {
        const dim3 fg = grid3(G.g[1].w3, G.g[1].h3, 2 * njobs);
        hipLaunchKernelGGL((k_mc_patch<1>), tile_grid(fg.x, fg.y, fg.z), dim3(64, 4), 0, st, jobs, G, MG, 1, 2, mvs0, mk_xcd_grid((int)fg.x, (int)fg.y, (int)fg.z), ex0, ey0);
    }
    PE();
}

// levels 5..4 of all three planes of njobs pictures (I and P alike) in one launch; the callers then pass with_tail | 2 to launch_inv_sbt
void launch_inv54_all(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, Prof *pf)
{
    int mw = 0, mh = 0;
    double s3 = 0;
    for (int c = 0; c < 3; c++) {
        mw = std::max(mw, G.g[c].w5); mh = std::max(mh, G.g[c].h5);
        s3 += (double)G.g[c].w3 * G.g[c].h3 * njobs;
    }
    PB(KID_INV_TILE_54_ALL, s3 * 8.0);
    hipLaunchKernelGGL(k_inv_tile54_all, dim3((mw + IT_TX - 1) / IT_TX, (mh + IT_TY - 1) / IT_TY, 3 * njobs), dim3(256), 0, st, jobs, G);
    PE();
}

// Can the patch kernel of these pictures' chroma planes write the reconstruction's border (k_inv_patch_c, fb) -- its own planes' and the luma
// plane's?  Every patch of the chroma planes must be the patch kernel's (no ragged strips for the tile kernel), both chroma planes alike, the
// luma plane an exact multiple of them, and the widths / strides what the 16-byte border stores need.
bool inv_sbt_fuses_border(const SbtGeo3 &G, int insym_c, int patch_kernel_c)
{
    static const bool off = getenv("DSV1_NO_FUSED_BORDER") != nullptr || getenv("DSV1_NO_PATCH_PART") != nullptr;   // (A/B; read once: this sits on the enqueue path)
    if (off || !insym_c || !patch_kernel_c) return false;
    const SbtGeo &g = G.g[1], &gy = G.g[0];
    for (int c = 1; c <= 2; c++) {
        const SbtGeo &q = G.g[c];
        if (q.pw != g.pw || q.ph != g.ph || q.w3 != g.w3 || q.h3 != g.h3 || (q.pw & 15) != 0 || (q.pstride & 15) != 0) return false;
        const int fullc = q.pw / 8, fullr = q.ph / 8;
        const bool part4 = fullr == q.h3 - 1 && fullr >= 1 && (q.ph & 7) == 4 && (q.H & 7) == 4;
        if (fullc < q.w3 || !(fullr >= q.h3 || part4)) return false;
    }
    if (g.pw < 16 || g.ph < 8 || gy.pw % g.pw || gy.ph % g.ph || (gy.pstride & 15) != 0 || (gy.pw & 15) != 0) return false;
    const int hr = gy.pw / g.pw, vr = gy.ph / g.ph;
    return (hr == 1 || hr == 2 || hr == 4) && (vr == 1 || vr == 2);
}

void launch_inv_sbt(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, int isP, Prof *pf, int with_tail,
                    int insym, int patch_kernel, int fuse_border)
{
    const int fb = (fuse_border && isP && c0 == 1 && npl == 2 && inv_sbt_fuses_border(G, insym, patch_kernel)) ? 1 : 0;
    const SbtGeo &g = G.g[c0];
    const int nz = njobs * npl;
    const double smp = (double)g.W * g.H * nz, s3 = (double)g.w3 * g.h3 * nz;
    const bool filt = (c0 == 0);
    if (with_tail & 1) {
        PB(KID_INV_TAIL, (double)g.w5 * g.h5 * nz * 8.0);
        hipLaunchKernelGGL(k_inv_tail, dim3(nz), dim3(TAIL_THREADS), (size_t)g.w5 * g.h5 * 4, st, jobs, G, c0, npl);
        PE();
    }
    if (!(with_tail & 2)) {   // levels 5..4 (LL5 -> LL3) for every picture type (with_tail & 2: launch_inv54_all did them for all planes)
        const dim3 mg((g.w5 + IT_TX - 1) / IT_TX, (g.h5 + IT_TY - 1) / IT_TY, nz);
        PB(filt ? KID_INV_TILE_54_F : KID_INV_TILE_54, s3 * 8.0);
        if (filt) hipLaunchKernelGGL((k_inv_haar_tile<true, 2, false>), mg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
        else      hipLaunchKernelGGL((k_inv_haar_tile<false, 2, false>), mg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
        PE();
    }
    const dim3 tg((g.w3 + IT_TX - 1) / IT_TX, (g.h3 + IT_TY - 1) / IT_TY, nz);
    if (isP) {
        // 4 B/sample coefficients + 1 B prediction in, 1 B out; from the symbol planes (insym): prediction 1 + reconstruction 1 +
        // the level-2/3 symbols 0.47 + flags -- the level-1 symbols (1.5 B/sample) are only fetched for flagged patches
        if (insym && !filt && patch_kernel) {
            // no smoothing filter: patches are closed computations -- the lean kernel takes every whole patch that does not lie
            // in a tile of the last tile row / column with a ragged edge, the tile kernel those strips
            const int fullc = G.g[c0].pw / 8, fullr = G.g[c0].ph / 8;
            // a last patch row of exactly four pixel rows (plane height 8k + 4 -- 1080 lines of 4:2:0 / 4:2:2 chroma: 540) stays with the lean
            // kernel (k_inv_patch_c, jpart); every plane of the launch must have it (they share one grid): 4:2:0 and 4:4:4 do
            static const bool no_part = getenv("DSV1_NO_PATCH_PART") != nullptr;       // (A/B)
            bool part4 = !no_part && fullr == g.h3 - 1 && fullr >= 1;
            for (int c = c0; c < c0 + npl; c++) part4 = part4 && G.g[c].ph == G.g[c0].ph && (G.g[c].ph & 7) == 4 && (G.g[c].H & 7) == 4 && G.g[c].h3 == g.h3;
            const int tcx = fullc >= g.w3 ? (int)tg.x : fullc / IT_TX, tcy = (fullr >= g.h3 || part4) ? (int)tg.y : fullr / IT_TY;   // first tile column / row of the strips
            const int imax = tcx >= (int)tg.x ? g.w3 : tcx * IT_TX, jmax = tcy >= (int)tg.y ? g.h3 : tcy * IT_TY;
            if (imax > 0 && jmax > 0) {
                PB(KID_INV_PATCH_C, 64.0 * imax * jmax * nz * 2.0);          // prediction in, reconstruction out (+ 5 B per patch: LL3, flag)
                const int cgx = (imax + 63) / 64, cgy = (jmax + 3) / 4;
                hipLaunchKernelGGL(k_inv_patch_c, tile_grid(cgx, cgy, nz), dim3(64, 4), 0, st, jobs, G, c0, npl, imax, jmax, mk_xcd_grid(cgx, cgy, nz), xcd_plain(),
                                   part4 ? g.h3 - 1 : -1, fb);
                PE();
            }
            if (tcx < (int)tg.x || tcy < (int)tg.y) {
                PB(KID_INV_TILE_PIX_SYM, (smp - 64.0 * imax * jmax * nz) * 2.5);
                // (right strip and bottom strip in one L-shaped launch)
                const int nrest = ((int)tg.x - tcx) * (int)tg.y + tcx * ((int)tg.y - tcy);
                hipLaunchKernelGGL((k_inv_haar_tile<false, 0, true>), dim3(nrest, 1, nz), dim3(256), 0, st, jobs, G, c0, npl, tcx, -tcy - 1);
                PE();
            }
        } else if (insym) {
            // sparse pictures (patch_kernel: every job has flags and a reference) on a geometry with aligned level-1 rows: the
            // tiles whose cells and halo are complete (not in the last tile column / row, nor within a cell of the band's end)
            // take the fast body as a kernel of its own, the general kernel the right and bottom strips
            const int fx = (patch_kernel && g.l1a && g.w3 >= IT_TX + 2) ? (g.w3 - IT_TX - 2) / IT_TX + 1 : 0;
            const int fy = (patch_kernel && g.l1a && g.h3 >= IT_TY + 2) ? (g.h3 - IT_TY - 2) / IT_TY + 1 : 0;
            static const bool no_er = getenv("DSV1_NO_EDGE_TILES") != nullptr;
            if (fx > 0 && fy > 0) {
                // the last tile column too, when it ends exactly where the band ends and every cell of every level is complete
                const bool er = !no_er && fx == (int)tg.x - 1 && g.w3 == (int)tg.x * IT_TX && (g.W & 7) == 0;
                // ... and the last tile row, when it is the only one left and every cell row is complete
                const bool eb = !no_er && fy == (int)tg.y - 1 && (g.H & 7) == 0;
                const int fxg = er ? fx + 1 : fx, fyg = eb ? fy + 1 : fy;      // tile columns / rows the fast kernel takes
                const double fsmp = 64.0 * std::min(fxg * IT_TX, g.w3) * std::min(fyg * IT_TY, g.h3) * nz;   // samples of the fast tiles
                PB(filt ? KID_INV_P_TILE_F : KID_INV_P_TILE, fsmp * 2.5);
                const dim3 pg = tile_grid(fxg, fyg, nz);
                if (filt) hipLaunchKernelGGL((k_inv_p_tile<true>), pg, dim3(256), 0, st, jobs, G, c0, npl, er ? fx : -1, eb ? fy : -1, mk_xcd_grid(fxg, fyg, nz), xcd_plain());
                else      hipLaunchKernelGGL((k_inv_p_tile<false>), pg, dim3(256), 0, st, jobs, G, c0, npl, er ? fx : -1, eb ? fy : -1, mk_xcd_grid(fxg, fyg, nz), xcd_plain());
                PE();
                const int nrest = ((int)tg.x - fxg) * (int)tg.y + fxg * ((int)tg.y - fyg);      // right strip + bottom strip, one launch
                if (nrest > 0) {
                    PB(filt ? KID_INV_TILE_PIX_SYM_F : KID_INV_TILE_PIX_SYM, (smp - fsmp) * 2.5);
                    if (filt) hipLaunchKernelGGL((k_inv_haar_tile<true, 0, true>), dim3(nrest, 1, nz), dim3(256), 0, st, jobs, G, c0, npl, fxg, -fyg - 1);
                    else      hipLaunchKernelGGL((k_inv_haar_tile<false, 0, true>), dim3(nrest, 1, nz), dim3(256), 0, st, jobs, G, c0, npl, fxg, -fyg - 1);
                    PE();
                }
                return;
            }
            PB(filt ? KID_INV_TILE_PIX_SYM_F : KID_INV_TILE_PIX_SYM, smp * 2.5);
            if (filt) hipLaunchKernelGGL((k_inv_haar_tile<true, 0, true>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
            else      hipLaunchKernelGGL((k_inv_haar_tile<false, 0, true>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
            PE();
        } else {
            PB(filt ? KID_INV_TILE_PIX_F : KID_INV_TILE_PIX, smp * 6.0);
            if (filt) hipLaunchKernelGGL((k_inv_haar_tile<true, 0, false>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
            else      hipLaunchKernelGGL((k_inv_haar_tile<false, 0, false>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
            PE();
        }
    } else {
        PB(insym ? (filt ? KID_INV_TILE_S1_SYM_F : KID_INV_TILE_S1_SYM) : (filt ? KID_INV_TILE_S1_F : KID_INV_TILE_S1), smp * (insym ? 1.4 : 2.0));
        if (insym) {
            if (filt) hipLaunchKernelGGL((k_inv_haar_tile<true, 1, true>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
            else      hipLaunchKernelGGL((k_inv_haar_tile<false, 1, true>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
        } else {
            if (filt) hipLaunchKernelGGL((k_inv_haar_tile<true, 1, false>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
            else      hipLaunchKernelGGL((k_inv_haar_tile<false, 1, false>), tg, dim3(256), 0, st, jobs, G, c0, npl, 0, 0);
        }
        PE();
        const dim3 bg(((g.W >> 1) + BT_CX - 1) / BT_CX, ((g.H >> 1) + BT_CY - 1) / BT_CY, nz);
        PB(insym ? KID_INV_B4T_SYM : KID_INV_B4T, smp * (insym ? 3.5 : 5.0));           // LL1 1 + details 3 (symbols: 1.5) in, 1 out
        if (insym) hipLaunchKernelGGL((k_inv_b4t<true>), tile_grid(bg.x, bg.y, bg.z), dim3(256), 0, st, jobs, G, c0, npl, mk_xcd_grid((int)bg.x, (int)bg.y, (int)bg.z), xcd_plain());
        else       hipLaunchKernelGGL((k_inv_b4t<false>), tile_grid(bg.x, bg.y, bg.z), dim3(256), 0, st, jobs, G, c0, npl, mk_xcd_grid((int)bg.x, (int)bg.y, (int)bg.z), xcd_plain());
        PE();
    }
}

void sbt_set_func_attributes()
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fwd_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_inv_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tail_q<TAIL_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tail_q<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
}

#ifdef DSVG_CLOCK_PROBE
DSVG_CLK_DUMP_FN(dsvg_clk_dump_sbt, "k_inv_p_tile", "k_fwd_mc_fast<0>", "k_fwd_mc_fast<1>")
#endif
