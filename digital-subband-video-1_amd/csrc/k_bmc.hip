// k_bmc.hip -- half-pel block motion compensation for gfx950 (MI355X).  HBM-bound byte work.
//
// Replaces compensate (bmc.c:204-302) with hpelL (bmc.c:124-174) / hpel (bmc.c:58-110) / avgval
// (bmc.c:176-189), fused with subf (bmc.c:43-55): one wave per (block, plane, job) in an XCD-aware order;
// every lane produces 4 adjacent prediction pixels per row pass (one 32-bit coalesced store to the
// prediction frame and, for the encoder, one to the residual frame:  res = clamp(src - pred + 128)).
// Copy / horizontal half-pel blocks read their reference bytes straight from global memory (one
// reference row per output row: nothing to share); vertical / diagonal / intra blocks stage the
// (cw+3)x(ch+3) window in LDS with batched aligned dword loads.
//   luma  : 4-tap (-1,9,9,-1): V / H rounded (+8)>>4, HV = H taps unrounded then V taps, (+128)>>8
//   chroma: bilinear (a+b+1)>>1, (a+b+c+d+2)>>2
//   intra : (sub-)block mean of the co-located reference pixels, truncating division
// Reads reach one pixel beyond the 64-px border exactly like the reference (same frame layout).
#include "dsvg_dev.hpp"
#include "dsvg_kernels.hpp"
#include "dsvg_host.hpp"

#define WPITCH 76                       // LDS row pitch in bytes: 64+3 window + 3 misalignment, +1 spare dword for ROW8
#define WROWS 68
#define MC_NT 64                        // threads per workgroup: ONE wave per block (A/B: 256 -> 3.7, 128 -> 2.7, 64 -> 2.5 ms/step)

static __device__ __forceinline__ int tap4(int m, int a, int b, int p) { return 9 * (a + b) - (m + p); }

// store 4 prediction pixels of row `o` (+ the residual clamp(src - pred + 128), bmc.c:43-55); n = pixels that exist
static __device__ __forceinline__ void mc_store4(uint8_t *pp, uint8_t *xp, const uint8_t *sp, size_t o, const int (&pv)[4], int n,
                                                 int do_sub, unsigned s)
{
    if (n >= 4) {
        const unsigned pk = (unsigned)pv[0] | ((unsigned)pv[1] << 8) | ((unsigned)pv[2] << 16) | ((unsigned)pv[3] << 24);
        *reinterpret_cast<unsigned *>(pp + o) = pk;
        if (do_sub) {
            unsigned r = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                r |= (unsigned)d_sat8((int)((s >> (8 * k)) & 0xff) - pv[k] + 128) << (8 * k);
            *reinterpret_cast<unsigned *>(xp + o) = r;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k < n) {
                pp[o + k] = (uint8_t)pv[k];
                if (do_sub) xp[o + k] = (uint8_t)d_sat8((int)sp[o + k] - pv[k] + 128);
            }
    }
}

__global__ __launch_bounds__(MC_NT) void k_mc(const JobDev *__restrict__ jobs, McGeo G, int do_sub, int njobs, const DMV *__restrict__ mvs0,
                                              const int *__restrict__ list, int nlist)
{
    __shared__ __align__(16) uint8_t win[WROWS * WPITCH];
    __shared__ int s_sum[5];
    // logical order: job, plane, block -- one XCD's L2 sees whole neighbouring block rows of one plane
    const int nblk = G.nbh * G.nbv;
    int job, c, blk;
    if (list) {                                 // only the listed blocks (job * nblk + blk), three planes each
        if ((int)blockIdx.x >= 3 * nlist) return;
        const int fi = list[blockIdx.x / 3];
        c = blockIdx.x % 3; job = fi / nblk; blk = fi - job * nblk;
    } else {
        const int item = d_xcd_remap(blockIdx.x, nblk * 3 * njobs);
        if (item >= nblk * 3 * njobs) return;
        job = item / (3 * nblk); c = (item - job * 3 * nblk) / nblk; blk = item - (job * 3 + c) * nblk;
    }
    const JobDev &jb = jobs[job];
    const int tid = threadIdx.x;
    const int sh = c ? G.hs : 0, sv = c ? G.vs : 0;
    const int bw = G.blk_w >> sh, bh = G.blk_h >> sv;
    const int pw = G.w[c], ph = G.h[c], stride = G.stride[c];
    const int sstride = jb.srcs[c];                     // the source plane has its own stride (in-place chroma: dsvg_load_frames_map_ex)
    const int bi = blk % G.nbh, bj = blk / G.nbh;
    const int x = bi * bw, y = bj * bh;
    if (x >= pw || y >= ph) return;
    const int cw = (x + bw >= pw) ? pw - x : bw;
    const int ch = (y + bh >= ph) ? ph - y : bh;
    // Every workgroup starts with a chain of dependent memory round trips (kernel arguments -> job table -> motion
    // vector -> reference pixels) and lives only a few microseconds, so the chain is kept short: the vector comes
    // from a kernel-argument base when the jobs' vector arrays are contiguous (mvs0), and the source pixels of the
    // first row passes -- which depend on neither -- are requested before the vector is waited for.
    const int lq0 = bw > 32 ? 4 : (bw > 16 ? 3 : 2);
    const int x40 = 4 * (tid & ((1 << lq0) - 1)), rpp0 = MC_NT >> lq0;
    unsigned s_pre[4] = {0u, 0u, 0u, 0u};
    if (do_sub && x40 + 4 <= cw) {
        const uint8_t *sp0 = jb.srcp[c];
#pragma unroll
        for (int u = 0; u < 4; u++)
            s_pre[u] = *reinterpret_cast<const unsigned *>(sp0 + (size_t)(y + min((tid >> lq0) + u * rpp0, ch - 1)) * sstride + x + x40);
    }
    // one vector per block: make its fields wave-uniform (SGPRs) so that the path selection below is scalar branching
    DMV mv = mvs0 ? mvs0[(size_t)job * nblk + blk] : jb.mvs[blk];
    mv.x = (int16_t)__builtin_amdgcn_readfirstlane((int)mv.x);
    mv.y = (int16_t)__builtin_amdgcn_readfirstlane((int)mv.y);
    mv.mode = (uint8_t)__builtin_amdgcn_readfirstlane((int)mv.mode);
    mv.submask = (uint8_t)__builtin_amdgcn_readfirstlane((int)mv.submask);
    const uint8_t *rp = jb.ref + G.off[c];

    int wx, wy, xh = 0, yh = 0;                 // window origin = (wx-1, wy-1)
    if (mv.mode == 0) {
        const int dx = mv.x >> sh, dy = mv.y >> sv;
        wx = d_clamp(x + (dx >> 1), -DSVG_BORDER, pw - bw + DSVG_BORDER - 1);
        wy = d_clamp(y + (dy >> 1), -DSVG_BORDER, ph - bh + DSVG_BORDER - 1);
        xh = dx & 1; yh = dy & 1;
    } else {
        wx = x; wy = y;
    }
    if (mv.mode == 0 && !yh) {
        // Copy and horizontal half-pel blocks use ONE reference row per output row: nothing to share through LDS.
        // Every thread fetches its 8 reference bytes (x4-1 .. x4+6) and its source dword for up to 8 row passes back
        // to back (one memory round trip), then filters, subtracts and stores.
        uint8_t *pp = jb.pred + G.off[c];
        uint8_t *xp = jb.xf + G.off[c];
        const uint8_t *sp = jb.srcp[c];
        const int nq = (cw + 3) >> 2;
        const int lq = bw > 32 ? 4 : (bw > 16 ? 3 : 2);
        const int x4 = 4 * (tid & ((1 << lq) - 1)), rpp = MC_NT >> lq;
        if (x4 >= 4 * nq) return;
        const uint8_t *gr = rp + (long)wy * stride + (wx + x4 - 1);
        const unsigned shb = (unsigned)(((uintptr_t)gr) & 3);
        const unsigned *ga = reinterpret_cast<const unsigned *>(gr - shb);
        const int sdw = stride >> 2;
        const bool full4 = x4 + 4 <= cw;
#pragma unroll 1
        for (int y0 = tid >> lq; y0 < ch; y0 += 4 * rpp) {
            unsigned d0[4], d1[4], d2[4], sv4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int yy = min(y0 + u * rpp, ch - 1);
                const long ro = (long)yy * sdw;
                d0[u] = ga[ro]; d1[u] = ga[ro + 1]; d2[u] = ga[ro + 2];
                sv4[u] = y0 == (tid >> lq) ? s_pre[u]
                                           : ((do_sub && full4) ? *reinterpret_cast<const unsigned *>(sp + (size_t)(y + yy) * sstride + x + x4) : 0u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int yy = y0 + u * rpp;
                if (yy >= ch) break;
                const unsigned lo = __builtin_amdgcn_alignbyte(d1[u], d0[u], shb), hi = __builtin_amdgcn_alignbyte(d2[u], d1[u], shb);
                int pv[4];
#define RB(k) ((int)((((k) < 4 ? lo : hi) >> (8 * ((k) & 3))) & 0xff))          /* reference byte x4 - 1 + k */
                if (!xh) {
#pragma unroll
                    for (int k = 0; k < 4; k++) pv[k] = RB(k + 1);
                } else if (c == 0) {
#pragma unroll
                    for (int k = 0; k < 4; k++) pv[k] = d_sat8((tap4(RB(k), RB(k + 1), RB(k + 2), RB(k + 3)) + 8) >> 4);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) pv[k] = (RB(k + 1) + RB(k + 2) + 1) >> 1;
                }
#undef RB
                mc_store4(pp, xp, sp + (size_t)(y + yy) * sstride - (size_t)(y + yy) * stride, (size_t)(y + yy) * stride + x + x4, pv, cw - x4, do_sub, sv4[u]);
                if (do_sub && G.cw_extra[c] && x + cw == pw && x4 + 4 >= cw)
                    xp[(size_t)(y + yy) * stride + pw] = sp[(size_t)(y + yy) * sstride + pw - 1];
            }
        }
        return;
    }
    if (mv.mode != 0) {
        // Intra blocks that are complete, 16 / 32 / 64 pixels wide and dword aligned (every block of a 1080p or 4K picture but the
        // partial ones at the edges): no window in LDS.  lane = (row of a pass, dword): the co-located reference rows and the source
        // rows of ALL passes are requested back to back, the four quadrant sums are v_sad_u8 against zero + DPP row sums, the
        // prediction of a row piece is its quadrant's mean replicated (or the reference dword itself where the submask keeps the
        // zero-vector prediction, bmc.c:176-189,254-283) and the residual is formed on int16 pairs.  The staged path below spent
        // ~1 500 instructions, two barriers and a byte loop per block on the same thing (content with a third of its blocks intra
        // had k_mc as its largest kernel).
        const int lqi = bw == 64 ? 4 : (bw == 32 ? 3 : (bw == 16 ? 2 : -1));
        const int rppi = lqi >= 0 ? MC_NT >> lqi : 1, npass = lqi >= 0 ? bh / rppi : 99;
        const uint8_t *r0p = rp + (size_t)y * stride + x;
        const uint8_t *s0p = jb.srcp[c] + (size_t)y * sstride + x;
        if (lqi >= 0 && cw == bw && ch == bh && bh == npass * rppi && npass <= 16 && !(ch & 1) && (stride & 3) == 0 &&
            ((unsigned)(uintptr_t)r0p & 3u) == 0 && (!do_sub || ((((unsigned)(uintptr_t)s0p | (unsigned)sstride) & 3u) == 0))) {
            const int x4i = 4 * (tid & ((1 << lqi) - 1)), yr = tid >> lqi;
            const int qwi = bw >> 1, qhi = bh >> 1;
            unsigned rw[16], sw[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                rw[u] = sw[u] = 0u;
                if (u < npass) {                                // (wave-uniform)
                    rw[u] = *reinterpret_cast<const unsigned *>(r0p + (size_t)(yr + u * rppi) * stride + x4i);
                    if (do_sub) sw[u] = *reinterpret_cast<const unsigned *>(s0p + (size_t)(yr + u * rppi) * sstride + x4i);
                }
            }
            unsigned at = 0, ab = 0;                            // this lane's share of its column half: rows above / below the middle
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (u < npass) {
                    const unsigned sm = __builtin_amdgcn_sad_u8(rw[u], 0u, 0u);
                    if (yr + u * rppi < qhi) at += sm; else ab += sm;
                }
            const bool right = x4i >= qwi;
            unsigned qs[4] = {right ? 0u : at, right ? at : 0u, right ? 0u : ab, right ? ab : 0u};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned t = qs[k];
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xf, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x141, 0xf, 0xf, true);
                t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x140, 0xf, 0xf, true);
                qs[k] = (unsigned)__builtin_amdgcn_readlane((int)t, 0) + (unsigned)__builtin_amdgcn_readlane((int)t, 16) +
                        (unsigned)__builtin_amdgcn_readlane((int)t, 32) + (unsigned)__builtin_amdgcn_readlane((int)t, 48);
            }
            const unsigned mfull = (((qs[0] + qs[1] + qs[2] + qs[3]) / (unsigned)(cw * ch)) & 0xffu) * 0x01010101u;
            unsigned mq[4];
#pragma unroll
            for (int k = 0; k < 4; k++) mq[k] = ((qs[k] / (unsigned)(qwi * qhi)) & 0xffu) * 0x01010101u;
            uint8_t *pp = jb.pred + G.off[c];
            uint8_t *xp = jb.xf + G.off[c];
            typedef short v2s __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (u < npass) {
                    const int yy = yr + u * rppi;
                    const int q = (right ? 1 : 0) + (yy >= qhi ? 2 : 0);
                    const unsigned pv = mv.submask == 0xF ? mfull : (((mv.submask >> q) & 1) ? mq[q] : rw[u]);
                    const size_t o = (size_t)(y + yy) * stride + x + x4i;
                    *reinterpret_cast<unsigned *>(pp + o) = pv;
                    if (do_sub) {
                        // clamp(src - pred + 128) on the even and the odd bytes as int16 pairs (bmc.c:43-55)
                        const v2s c128 = {128, 128};
                        const v2s e = __builtin_bit_cast(v2s, sw[u] & 0x00ff00ffu) - __builtin_bit_cast(v2s, pv & 0x00ff00ffu) + c128;
                        const v2s od = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, sw[u], 0x0c030c01u)) -
                                       __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, pv, 0x0c030c01u)) + c128;
                        unsigned eb, ob;
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(eb) : "v"(__builtin_bit_cast(unsigned, e)));
                        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ob) : "v"(__builtin_bit_cast(unsigned, od)));
                        *reinterpret_cast<unsigned *>(xp + o) = __builtin_amdgcn_perm(ob, eb, 0x05010400u);
                        if (G.cw_extra[c] && x + cw == pw && x4i + 4 >= cw)
                            xp[(size_t)(y + yy) * stride + pw] = jb.srcp[c][(size_t)(y + yy) * sstride + pw - 1];
                    }
                }
            return;
        }
    }
    // stage rows wy-1 .. wy+ch+1, columns wx-1 .. wx+cw+1 with aligned dword loads
    const long rowbase = (long)(wy - 1) * stride + (wx - 1);
    const uint8_t *g0 = rp + rowbase;
    const int mis = (int)(((uintptr_t)g0) & 3);           // same for every row (stride % 4 == 0)
    const int ndw = (mis + cw + 3 + 3) >> 2;
    {   // TPR threads share a row (ndw <= TPR), loads in batches of up to 7 issued back to back before their LDS
        // stores: a memory round trip per batch, not per row
        const int ltpr = ndw > 16 ? 5 : 4;
        const int d = tid & ((1 << ltpr) - 1), rb = tid >> ltpr, rpp = MC_NT >> ltpr;
        const int nh = ch + 3;
        if (d < ndw) {
#pragma unroll 1
            for (int rbase = rb; rbase < nh; rbase += 7 * rpp) {
                unsigned v[7];
#pragma unroll
                for (int u = 0; u < 7; u++) {
                    const int r = min(rbase + u * rpp, nh - 1);
                    v[u] = *reinterpret_cast<const unsigned *>(g0 - mis + (long)r * stride + 4 * d);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 7; u++) {
                    const int r = rbase + u * rpp;
                    if (r < nh) *reinterpret_cast<unsigned *>(win + r * WPITCH + 4 * d) = v[u];
                }
            }
        }
    }
    if (tid < 5) s_sum[tid] = 0;
    __syncthreads();
    // win[(r)*WPITCH + mis + k] = ref(wx-1+k, wy-1+r)
    const uint8_t *w0 = win + mis + WPITCH + 1;           // -> ref(wx, wy)

    const int qw = cw / 2, qh = ch / 2;
    int mean_full = 0, mean_q[4] = {0, 0, 0, 0};
    if (mv.mode != 0) {
        int acc[5] = {0, 0, 0, 0, 0};
        for (int p = tid; p < cw * ch; p += MC_NT) {
            const int yy = p / cw, xx = p - yy * cw;
            const int v = w0[yy * WPITCH + xx];
            acc[4] += v;
            if (xx < 2 * qw && yy < 2 * qh) acc[(xx >= qw) + 2 * (yy >= qh)] += v;
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int a = acc[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
            if ((tid & 63) == 0) atomicAdd(&s_sum[k], a);
        }
        __syncthreads();
        mean_full = s_sum[4] / (cw * ch);
        if (qw > 0 && qh > 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) mean_q[k] = s_sum[k] / (qw * qh);
        }
    }

    uint8_t *pp = jb.pred + G.off[c];
    uint8_t *xp = jb.xf + G.off[c];
    const uint8_t *sp = jb.srcp[c];
    const int nq = (cw + 3) >> 2;
    // 8 reference bytes b0..b7 = ref(wx + x4 - 1 .. wx + x4 + 6) of window row rr as two dwords (the window
    // is stored as aligned dwords; `mis` is the same for every row, so one v_alignbyte pair re-aligns it)
#define ROW8(rr, lo, hi)                                                                     \
    do {                                                                                     \
        const unsigned *W_ = reinterpret_cast<const unsigned *>(win + (rr) * WPITCH) + ((mis + x4) >> 2); \
        const unsigned a_ = W_[0], b_ = W_[1], c_ = W_[2];                                   \
        lo = __builtin_amdgcn_alignbyte(b_, a_, (unsigned)((mis + x4) & 3));                  \
        hi = __builtin_amdgcn_alignbyte(c_, b_, (unsigned)((mis + x4) & 3));                  \
    } while (0)
#define BYTE(lo, hi, k) ((int)((((k) < 4 ? (lo) : (hi)) >> (8 * ((k) & 3))) & 0xff))
    // 2^lq threads share a row of 4-pixel groups (lq from the nominal block width), MC_NT >> lq rows per pass
    const int lq = bw > 32 ? 4 : (bw > 16 ? 3 : 2);
    const int x4 = 4 * (tid & ((1 << lq) - 1));
    if (x4 >= 4 * nq) return;
    // the source dword of the NEXT row pass is requested before this pass is computed (the subtraction would
    // otherwise sit behind a full memory round trip in every pass)
    const bool full4 = x4 + 4 <= cw;
    unsigned s_cur = s_pre[0];
    for (int yy = tid >> lq; yy < ch; yy += MC_NT >> lq) {
        unsigned s_next = 0;
        if (do_sub && full4 && yy + (MC_NT >> lq) < ch)
            s_next = *reinterpret_cast<const unsigned *>(sp + (size_t)(y + yy + (MC_NT >> lq)) * sstride + x + x4);
        int pv[4];
        if (mv.mode == 0) {
            if (c == 0) {
                if (!xh && !yh) {
                    unsigned lo, hi; ROW8(yy + 1, lo, hi);
#pragma unroll
                    for (int k = 0; k < 4; k++) pv[k] = BYTE(lo, hi, k + 1);
                } else if (!xh) {
                    unsigned l0, h0, l1, h1, l2, h2, l3, h3;
                    ROW8(yy, l0, h0); ROW8(yy + 1, l1, h1); ROW8(yy + 2, l2, h2); ROW8(yy + 3, l3, h3);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        pv[k] = d_sat8((tap4(BYTE(l0, h0, k + 1), BYTE(l1, h1, k + 1), BYTE(l2, h2, k + 1), BYTE(l3, h3, k + 1)) + 8) >> 4);
                } else if (!yh) {
                    unsigned lo, hi; ROW8(yy + 1, lo, hi);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        pv[k] = d_sat8((tap4(BYTE(lo, hi, k), BYTE(lo, hi, k + 1), BYTE(lo, hi, k + 2), BYTE(lo, hi, k + 3)) + 8) >> 4);
                } else {
                    unsigned l0, h0, l1, h1, l2, h2, l3, h3;
                    ROW8(yy, l0, h0); ROW8(yy + 1, l1, h1); ROW8(yy + 2, l2, h2); ROW8(yy + 3, l3, h3);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int hm = tap4(BYTE(l0, h0, k), BYTE(l0, h0, k + 1), BYTE(l0, h0, k + 2), BYTE(l0, h0, k + 3));
                        const int hc = tap4(BYTE(l1, h1, k), BYTE(l1, h1, k + 1), BYTE(l1, h1, k + 2), BYTE(l1, h1, k + 3));
                        const int hn = tap4(BYTE(l2, h2, k), BYTE(l2, h2, k + 1), BYTE(l2, h2, k + 2), BYTE(l2, h2, k + 3));
                        const int hp = tap4(BYTE(l3, h3, k), BYTE(l3, h3, k + 1), BYTE(l3, h3, k + 2), BYTE(l3, h3, k + 3));
                        pv[k] = d_sat8((tap4(hm, hc, hn, hp) + 128) >> 8);
                    }
                }
            } else {
                unsigned l1, h1, l2, h2;
                ROW8(yy + 1, l1, h1);
                ROW8(yy + 2, l2, h2);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int a = BYTE(l1, h1, k + 1), bq = BYTE(l1, h1, k + 2), cq = BYTE(l2, h2, k + 1), dq = BYTE(l2, h2, k + 2);
                    pv[k] = (!xh && !yh) ? a : (!xh ? (a + cq + 1) >> 1 : (!yh ? (a + bq + 1) >> 1 : (a + bq + cq + dq + 2) >> 2));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xx = x4 + k;
                int v;
                if (mv.submask == 0xF) v = mean_full;
                else if (xx < 2 * qw && yy < 2 * qh) {
                    const int q = (xx >= qw) + 2 * (yy >= qh);
                    v = (mv.submask & (1 << q)) ? mean_q[q] : (int)w0[yy * WPITCH + xx];
                } else v = 0;                    // odd-sized edge blocks: untouched (zeroed) in the reference
                pv[k] = v & 0xff;
            }
        }
        const size_t o = (size_t)(y + yy) * stride + x + x4;
        mc_store4(pp, xp, sp + (size_t)(y + yy) * sstride - (size_t)(y + yy) * stride, o, pv, cw - x4, do_sub, s_cur);
        // odd plane width whose coefficient plane is one wider: the transform reads column pw of the
        // residual frame, which in the reference still holds the replicated source edge (frame.c:199-221)
        if (do_sub && G.cw_extra[c] && x + cw == pw && x4 + 4 >= cw)
            xp[(size_t)(y + yy) * stride + pw] = sp[(size_t)(y + yy) * sstride + pw - 1];
        s_cur = s_next;
    }
}

void launch_mc(hipStream_t st, const JobDev *jobs, int njobs, const McGeo &G, int do_sub, Prof *pf, const DMV *mvs0, const int *list, int nlist)
{
    double smp = 0;
    for (int c = 0; c < 3; c++) smp += (double)G.w[c] * G.h[c];
    if (list) {                              // the blocks of a list only (intra blocks beside k_fwd_mc_pix): a few waves
        if (nlist <= 0) return;
        if (pf) pf->begin(st, KID_MC, 0.0);
        hipLaunchKernelGGL(k_mc, dim3(3 * nlist), dim3(MC_NT), 0, st, jobs, G, do_sub, njobs, mvs0, list, nlist);
        if (pf) pf->end(st);
        return;
    }
    if (pf) pf->begin(st, KID_MC, smp * njobs * (do_sub ? 4.0 : 2.0));   // ref + src in, pred + residual out
    hipLaunchKernelGGL(k_mc, dim3(xcd_grid(G.nbh * G.nbv * 3 * njobs)), dim3(MC_NT), 0, st, jobs, G, do_sub, njobs, mvs0, nullptr, 0);
    if (pf) pf->end(st);
}
