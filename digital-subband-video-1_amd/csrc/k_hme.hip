// k_hme.hip -- hierarchical motion estimation + level-0 mode decision for gfx950 (MI355X).
//
// Replaces dsv_hme / refine_level (hme.c:378-741).  One launch per pyramid level (coarse -> fine),
// one 256-thread workgroup per visited block and frame pair.  The source block and each reference
// window are staged in LDS with aligned dword loads; SADs are per-thread partial sums folded with
// wave shuffles (ds_swizzle/DPP on gfx950) and a small LDS stage; every decision that depends on
// candidate ORDER (first minimum wins, hme.c:503-506,531-534,572-575; last candidate is the
// fallback, hme.c:482-509) is taken by one lane in the reference's order.
// Level 0 adds: the 32x32 half-pel lattice of the centred 16x16 reference patch (hpel hme.c:350-376)
// and the 8-point half-pel search on the 14x14 window (hme.c:551-591), the block statistics with
// 32-bit unsigned wrap-around (hme.c:181-300), the intra tests (hme.c:652-682), the
// representability veto (hme.c:147-179) and the 4-quadrant vote (hme.c:89-134, 689-716).
// high_detail needs the left/top/top-left neighbours' final flags (hme.c:621-648) and is
// resolved by the second tiny kernel k_hme_detail.
#include "dsvg_dev.hpp"
#include "dsvg_kernels.hpp"
#include "dsvg_host.hpp"

#define WIN 14
#define LAT 32
#define SP 64              // pitch of the source block in LDS
#define RP 72              // pitch of reference windows in LDS
#define RROWS 67

static __device__ __forceinline__ int tap4(int m, int a, int b, int p) { return 9 * (a + b) - (m + p); }

// stage rows [oy,oy+nh) x cols [ox,ox+nw) of a plane into LDS (pitch P); returns the byte shift
// `mis` such that dst[r*P + mis + k] == plane(ox+k, oy+r)
static __device__ __forceinline__ int load_win(uint8_t *dst, int P, const uint8_t *plane, int stride,
                                               int ox, int oy, int nw, int nh)
{
    const uint8_t *g0 = plane + (long)oy * stride + ox;
    const int mis = (int)(((uintptr_t)g0) & 3);
    const int ndw = (mis + nw + 3) >> 2;
    for (int i = threadIdx.x; i < nh * ndw; i += 256) {
        const int r = i / ndw, d = i - r * ndw;
        *reinterpret_cast<unsigned *>(dst + r * P + 4 * d) =
            *reinterpret_cast<const unsigned *>(g0 - mis + (long)r * stride + 4 * d);
    }
    return mis;
}

// sum over the workgroup, result visible to all threads (s_red: 4 words, s_out: 1 word)
static __device__ __forceinline__ unsigned block_sum(unsigned v, unsigned *s_red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

static __device__ __forceinline__ int frame_invalid(int fw, int fh, int x, int y, int w, int h)   // invalid_block
{
    const int b = DSVG_BORDER;
    return x < -b || y < -b || x + w > fw + b || y + h > fh + b;
}

struct HmeShared {
    uint8_t src[64 * SP];
    __attribute__((aligned(16))) uint8_t ref[RROWS * RP];
    __attribute__((aligned(16))) uint8_t patch[20 * 24];
    uint8_t lat[LAT * LAT];
    __attribute__((aligned(16))) uint8_t swin[WIN * 24];
    uint8_t rwin[WIN * WIN];
    unsigned red[4];
    unsigned part[4][9];
    int cand[8];
    int ncand, pick, best, bestk;
    int mvx, mvy;
    int any;
};

static __device__ const int FP_X[9] = {0, 1, -1, 0, 0, -1, 1, -1, 1};
static __device__ const int FP_Y[9] = {0, 0, 0, 1, -1, -1, -1, 1, 1};
static __device__ const int HP_X[8] = {1, -1, 0, 0, -1, 1, -1, 1};
static __device__ const int HP_Y[8] = {0, 0, 1, -1, -1, -1, 1, 1};

// gradient / moment sums of a w x h byte block held in LDS (pitch P)
static __device__ __forceinline__ void stat_partial(const uint8_t *p, int P, int w, int h,
                                                    unsigned &gh, unsigned &gv, unsigned &s1, unsigned &s2)
{
    gh = gv = s1 = s2 = 0;
    for (int i = threadIdx.x; i < w * h; i += 256) {
        const int y = i / w, x = i - y * w;
        const int px = p[y * P + x];
        if (x + 1 < w) gh += (unsigned)abs(px - (int)p[y * P + x + 1]);
        if (y > 0) gv += (unsigned)abs(px - (int)p[(y - 1) * P + x]);
        s1 += (unsigned)px;
        s2 += (unsigned)(px * px);
    }
}

template <bool LEVEL0>
__global__ __launch_bounds__(256) void k_hme_level(HmeArgs A, int level)
{
    __shared__ HmeShared S;
    const int tid = threadIdx.x;
    const int pair = blockIdx.y;
    const int step = 1 << level;
    const int nvx = (A.nxb + step - 1) / step;
    const int vi = blockIdx.x % nvx, vj = blockIdx.x / nvx;
    const int i = vi * step, j = vj * step;
    const FrameLayout &L = A.L[level];
    const int fw = L.w[0], fh = L.h[0], stride = L.stride[0];
    const int BW = A.blk_w, BH = A.blk_h;
    const int bx = (i * BW) >> level, by = (j * BH) >> level;
    if (bx >= fw || by >= fh) return;                       // stays a zero inter vector (hme.c:441-444)
    const int cur = A.cur_slots[pair], rf = A.ref_slots[pair];
    const uint8_t *sp = A.slab[level] + (size_t)cur * L.pitch + L.off[0];
    const uint8_t *rp = A.slab[level] + (size_t)rf * L.pitch + L.off[0];
    const int bw = min(max(fw - bx, 0), BW), bh = min(max(fh - by, 0), BH);
    DMV *mf = A.mvf + ((size_t)pair * (A.levels + 1) + level) * A.nblk;
    const DMV *parent = level < A.levels ? A.mvf + ((size_t)pair * (A.levels + 1) + level + 1) * A.nblk : nullptr;

    // source block -> LDS
    for (int q = tid; q < bh * ((bw + 3) >> 2); q += 256) {
        const int r = q / ((bw + 3) >> 2), d = q - r * ((bw + 3) >> 2);
        *reinterpret_cast<unsigned *>(S.src + r * SP + 4 * d) =
            *reinterpret_cast<const unsigned *>(sp + (size_t)(by + r) * stride + bx + 4 * d);
    }
    if (tid == 0) {
        int n = 0;
        S.cand[n++] = 0;
        if (parent) {
            const unsigned pmask = ~(unsigned)((step << 1) - 1);
            const int pi = (int)((unsigned)i & pmask), pj = (int)((unsigned)j & pmask);
            const int ox[5] = {0, -2, 2, 0, 0}, oy[5] = {0, 0, 0, -2, 2};
            for (int m = 0; m < 5; m++) {
                const int x = pi + ox[m] * step, y = pj + oy[m] * step;
                if (x < 0 || x >= A.nxb || y < 0 || y >= A.nyb) continue;
                const DMV pv = parent[x + y * A.nxb];
                const int all = (int)(((unsigned)(uint16_t)pv.x) | ((unsigned)(uint16_t)pv.y << 16));
                if (!all) continue;
                bool dup = false;
                for (int k = 0; k < n; k++) dup |= (S.cand[k] == all);
                if (!dup) S.cand[n++] = all;
            }
        }
        S.ncand = n;
        S.pick = n - 1;
        S.best = 0x7fffffff;
    }
    __syncthreads();
    const int n = S.ncand;

    if (n > 1) {
        for (int k = 0; k < n; k++) {
            const int all = S.cand[k];
            const int dx = ((int)(int16_t)(all & 0xffff)) >> level, dy = ((int)(int16_t)((unsigned)all >> 16)) >> level;
            if (frame_invalid(fw, fh, bx, by, bw, bh)) continue;
            if (frame_invalid(fw, fh, bx + dx, by + dy, bw, bh)) continue;
            __syncthreads();
            const int mis = load_win(S.ref, RP, rp, stride, bx + dx, by + dy, bw, bh);
            __syncthreads();
            unsigned acc = 0;
            for (int q = tid; q < bw * bh; q += 256) {
                const int y = q / bw, x = q - y * bw;
                acc += (unsigned)abs((int)S.src[y * SP + x] - (int)S.ref[y * RP + mis + x]);
            }
            const int sc = (int)block_sum(acc, S.red);
            if (tid == 0 && S.best > sc) { S.best = sc; S.pick = k; }
        }
        __syncthreads();
    }
    int dx, dy;
    {
        const int all = S.cand[S.pick];
        dx = ((int)(int16_t)(all & 0xffff)) >> level;
        dy = ((int)(int16_t)((unsigned)all >> 16)) >> level;
        dx = d_clamp(dx, -bw - bx, fw - bx);
        dy = d_clamp(dy, -bh - by, fh - by);
    }
    __syncthreads();
    // 9-point +-1 search around (dx,dy): window (bw+2)x(bh+2) at (bx+dx-1, by+dy-1)
    {
        const int mis = load_win(S.ref, RP, rp, stride, bx + dx - 1, by + dy - 1, bw + 2, bh + 2);
        __syncthreads();
        unsigned acc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) acc[k] = 0;
        for (int q = tid; q < bw * bh; q += 256) {
            const int y = q / bw, x = q - y * bw;
            const int s = S.src[y * SP + x];
            const uint8_t *r = S.ref + (y + 1) * RP + mis + x + 1;
#pragma unroll
            for (int k = 0; k < 9; k++) acc[k] += (unsigned)abs(s - (int)r[FP_Y[k] * RP + FP_X[k]]);
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            unsigned v = acc[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if ((tid & 63) == 0) S.part[tid >> 6][k] = v;
        }
        __syncthreads();
        if (tid == 0) {
            int best = 0x7fffffff, m = 0;
            for (int k = 0; k < 9; k++) {
                const int sc = (int)(S.part[0][k] + S.part[1][k] + S.part[2][k] + S.part[3][k]);
                if (best > sc) { best = sc; m = k; }
            }
            S.best = best;
            S.bestk = m;
        }
        __syncthreads();
    }
    dx += FP_X[S.bestk];
    dy += FP_Y[S.bestk];
    int mvx = (int)(int16_t)(dx << level), mvy = (int)(int16_t)(dy << level);
    DMV out;
    out.x = (int16_t)mvx; out.y = (int16_t)mvy;
    out.mode = 0; out.submask = 0; out.lo_var = 0; out.lo_tex = 0; out.high_detail = 0;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    if (!LEVEL0) {
        if (tid == 0) mf[i + j * A.nxb] = out;
        return;
    }

    // ------------------------------------------------------------------ level 0 only
    int best = S.best;
    const unsigned yarea = (unsigned)(bw * bh), yareasq = yarea * yarea;
    const int wx = bx + ((bw >> 1) - WIN / 2), wy = by + ((bh >> 1) - WIN / 2);
    const int smis = load_win(S.swin, 24, sp, stride, wx, wy, WIN, WIN);
    bool have_hp = false;
    if (best > BW * BH) {
        // 16x16 patch at (wx+mvx-1, wy+mvy-1) plus the filter margins: rows -1..18, cols -1..17
        const int pmis = load_win(S.patch, 24, rp, stride, wx + mvx - 2, wy + mvy - 2, 19, 20);
        __syncthreads();
        {   // one lattice cell (j,i) per thread: F, H, V, D
            const int lj = tid >> 4, li = tid & 15;
            const uint8_t *p = S.patch + (lj + 1) * 24 + pmis + li + 1;       // -> patch sample (li, lj)
            const int F = p[0];
            const int H = d_sat8((tap4(p[-1], p[0], p[1], p[2]) + 8) >> 4);
            const int V = d_sat8((tap4(p[-24], p[0], p[24], p[48]) + 8) >> 4);
            const int hm = tap4(p[-24 - 1], p[-24], p[-24 + 1], p[-24 + 2]);
            const int h0 = tap4(p[-1], p[0], p[1], p[2]);
            const int h1 = tap4(p[24 - 1], p[24], p[24 + 1], p[24 + 2]);
            const int h2 = tap4(p[48 - 1], p[48], p[48 + 1], p[48 + 2]);
            const int D = d_sat8((tap4(hm, h0, h1, h2) + 128) >> 8);
            uint8_t *e = S.lat + (2 * lj) * LAT + 2 * li;
            e[0] = (uint8_t)F; e[1] = (uint8_t)H; e[LAT] = (uint8_t)V; e[LAT + 1] = (uint8_t)D;
        }
        __syncthreads();
        unsigned acc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = 0;
        if (tid < WIN * WIN) {
            const int y = tid / WIN, x = tid - y * WIN;
            const int s = S.swin[y * 24 + smis + x];
            const uint8_t *c = S.lat + 2 + 2 * LAT + 2 * x + y * 2 * LAT;
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] = (unsigned)abs(s - (int)c[HP_X[k] + HP_Y[k] * LAT]);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned v = acc[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if ((tid & 63) == 0) S.part[tid >> 6][k] = v;
        }
        __syncthreads();
        if (tid == 0) {
            int best_hp = (int)((unsigned)(best * (WIN * WIN)) / yarea);
            int hm = -1;
            for (int k = 0; k < 8; k++) {
                const int sc = (int)(S.part[0][k] + S.part[1][k] + S.part[2][k] + S.part[3][k]);
                if (best_hp > sc) { best_hp = sc; hm = k; }
            }
            S.bestk = hm;
            S.best = hm >= 0 ? (int)((unsigned)best_hp * yarea / (WIN * WIN)) : best;
        }
        __syncthreads();
        const int hm = S.bestk;
        best = S.best;
        mvx = (int)(int16_t)(mvx << 1);
        mvy = (int)(int16_t)(mvy << 1);
        if (hm >= 0) {
            mvx = (int)(int16_t)(mvx + HP_X[hm]);
            mvy = (int)(int16_t)(mvy + HP_Y[hm]);
            if (tid < WIN * WIN) {
                const int y = tid / WIN, x = tid - y * WIN;
                S.rwin[tid] = S.lat[2 + 2 * LAT + HP_X[hm] + HP_Y[hm] * LAT + 2 * x + y * 2 * LAT];
            }
            have_hp = true;
        }
    } else {
        mvx = (int)(int16_t)(mvx << 1);
        mvy = (int)(int16_t)(mvy << 1);
    }
    __syncthreads();
    if (!have_hp) {
        const int rm = load_win(S.patch, 24, rp, stride, wx + (mvx >> 1), wy + (mvy >> 1), WIN, WIN);
        __syncthreads();
        if (tid < WIN * WIN) {
            const int y = tid / WIN, x = tid - y * WIN;
            S.rwin[tid] = S.patch[y * 24 + rm + x];
        }
    }
    __syncthreads();

    // ---- statistics
    unsigned gh, gv, s1, s2;
    stat_partial(S.src, SP, bw, bh, gh, gv, s1, s2);
    gh = block_sum(gh, S.red); gv = block_sum(gv, S.red); s1 = block_sum(s1, S.red); s2 = block_sum(s2, S.red);
    const unsigned luma_tex = ((gh + gv) / 2) / yarea;
    const unsigned luma_var = s2 - (s1 * s1) / yarea;

    stat_partial(S.swin + smis, 24, WIN, WIN, gh, gv, s1, s2);
    gh = block_sum(gh, S.red); gv = block_sum(gv, S.red); s1 = block_sum(s1, S.red); s2 = block_sum(s2, S.red);
    const int src_tex = (int)(((gh + gv) / 2) / (WIN * WIN));
    const int src_avg = (int)(s1 / (WIN * WIN));
    const int src_var = (int)(s2 - (s1 * s1) / (WIN * WIN));

    stat_partial(S.rwin, WIN, WIN, WIN, gh, gv, s1, s2);
    gh = block_sum(gh, S.red); gv = block_sum(gv, S.red); s1 = block_sum(s1, S.red); s2 = block_sum(s2, S.red);
    const int ref_tex = (int)(((gh + gv) / 2) / (WIN * WIN));
    const int ref_avg = (int)(s1 / (WIN * WIN));
    const int ref_var = (int)(s2 - (s1 * s1) / (WIN * WIN));

    // zero-motion reference block -> LDS
    __syncthreads();
    const int zmis = load_win(S.ref, RP, rp, stride, bx, by, bw, bh);
    __syncthreads();
    const uint8_t *zref = S.ref + zmis;
    unsigned zs1 = 0, zs2 = 0;
    for (int q = tid; q < bw * bh; q += 256) {
        const int y = q / bw, x = q - y * bw;
        const unsigned px = zref[y * RP + x];
        zs1 += px; zs2 += px * px;
    }
    zs1 = block_sum(zs1, S.red); zs2 = block_sum(zs2, S.red);
    const unsigned zvar = zs2 - (zs1 * zs1) / yarea;

    out.x = (int16_t)mvx; out.y = (int16_t)mvy;
    out.lo_tex = (luma_tex <= 2);
    out.lo_var = (luma_var < yareasq);

    bool want_intra = false;
    if (src_tex < 2 && zvar > luma_var * 2) want_intra = true;
    else if (ref_var > src_var * 2) want_intra = true;
    else if (src_tex == 0 && ref_tex != 0) want_intra = true;
    else if (abs(src_avg - ref_avg) > 8) want_intra = true;
    else if (luma_tex <= 10 && (unsigned)best > yareasq / 16) want_intra = true;
    else {
        // chroma variance test (c_maxvar hme.c:269-300) straight from HBM (level-0 frames carry chroma)
        const FrameLayout &L0 = A.L[0];
        const int cbx = i * (BW >> L0.hs), cby = j * (BH >> L0.vs);
        const int cbw = bw >> L0.hs, cbh = bh >> L0.vs;
        unsigned mv_[2] = {0, 0};
        for (int side = 0; side < 2; side++) {
            const uint8_t *fb = A.slab[0] + (size_t)(side ? rf : cur) * L0.pitch;
            unsigned best_v = 0;
            for (int pl = 1; pl <= 2; pl++) {
                const uint8_t *cp = fb + L0.off[pl] + (long)cby * L0.stride[pl] + cbx;
                unsigned a1 = 0, a2 = 0;
                for (int q = tid; q < cbw * cbh; q += 256) {
                    const int y = q / cbw, x = q - y * cbw;
                    const unsigned px = cp[(long)y * L0.stride[pl] + x];
                    a1 += px; a2 += px * px;
                }
                a1 = block_sum(a1, S.red); a2 = block_sum(a2, S.red);
                const unsigned var = a2 - (a1 * a1) / (unsigned)(cbw * cbh);
                best_v = var > best_v ? var : best_v;
            }
            mv_[side] = best_v;
        }
        if (mv_[1] > 4 * mv_[0]) want_intra = true;
    }

    if (want_intra) {
        // representability veto: mean of the zero-motion block + clamped residual must reproduce src
        const int mean = (int)zs1 / (bw * bh);
        unsigned bad = 0;
        for (int q = tid; q < bw * bh; q += 256) {
            const int y = q / bw, x = q - y * bw;
            const int px = S.src[y * SP + x];
            const int back = d_sat8(mean + d_sat8(px - mean + 128) - 128);
            bad += (back != px);
        }
        bad = block_sum(bad, S.red);
        if (!bad) {
            int submask = 0xF;
            if (src_tex > 1) {
                const int qw = bw / 2, qh = bh / 2;
                for (int k = 0; k < 4; k++) {
                    const int ox = (k & 1) ? qw : 0, oy = (k & 2) ? qh : 0;
                    unsigned good = 0, evil = 0;
                    for (int q = tid; q < qw * qh; q += 256) {
                        const int y = q / qw, x = q - y * qw;
                        const uint8_t *ra = S.src + (oy + y) * SP + ox + x;
                        const uint8_t *rb = zref + (oy + y) * RP + ox + x;
                        const int pa = ra[0], pb = rb[0];
                        const int la = x ? ra[-1] : pa, lb = x ? rb[-1] : pb;
                        const int ua = y ? ra[-SP] : pa, ub = y ? rb[-RP] : pb;
                        const int dif = abs(pa - pb);
                        good += (unsigned)(abs(pa - la) + abs(pa - ua) + abs(pb - lb) + abs(pb - ub));
                        if (dif == 0) good += 192;
                        else if (dif == 1) good += 128;
                        else if (dif == 2) good += 96;
                        else evil += (unsigned)dif;
                    }
                    good = block_sum(good, S.red);
                    evil = block_sum(evil, S.red);
                    if (good >= (unsigned)((qw + qh) >> 1) * evil) submask &= ~(1 << k);
                }
            }
            if (submask) {
                out.submask = (uint8_t)submask;
                out.mode = 1;
            } else {
                out.submask = 0;
            }
            // note: the reference leaves submask = 0xF & ~votes even when it ends up inter (0)
        }
    }
    if (tid == 0) {
        mf[i + j * A.nxb] = out;
        A.aux_tex[(size_t)pair * A.nblk + i + j * A.nxb] = luma_tex;
        A.aux_var[(size_t)pair * A.nblk + i + j * A.nxb] = src_var;
    }
}

// second pass of level 0: high_detail from the causal neighbours' final flags (hme.c:621-648)
__global__ __launch_bounds__(256) void k_hme_detail(HmeArgs A)
{
    const int pair = blockIdx.y;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= A.nblk) return;
    const int i = b % A.nxb, j = b / A.nxb;
    DMV *mf = A.mvf + ((size_t)pair * (A.levels + 1)) * A.nblk;
    const FrameLayout &L = A.L[0];
    if (i * A.blk_w >= L.w[0] || j * A.blk_h >= L.h[0]) return;
    unsigned thr_tex = 1;
    int thr_var = WIN * WIN;
    if (i > 0) {
        const DMV nb = mf[j * A.nxb + i - 1];
        if (nb.mode == 0 && !nb.lo_tex && !nb.lo_var) { thr_var *= WIN; thr_tex++; }
    }
    if (j > 0) {
        const DMV nb = mf[(j - 1) * A.nxb + i];
        if (nb.mode == 0 && !nb.lo_tex && !nb.lo_var) { thr_var *= WIN; thr_tex++; }
    }
    if (i > 0 && j > 0) {
        const DMV nb = mf[(j - 1) * A.nxb + i - 1];
        if (nb.mode == 0 && !nb.lo_tex && !nb.lo_var) { thr_var *= WIN / 4; thr_tex++; }
    }
    const unsigned tex = A.aux_tex[(size_t)pair * A.nblk + b];
    const int var = A.aux_var[(size_t)pair * A.nblk + b];
    mf[b].high_detail = (tex > thr_tex && var > thr_var) ? 1 : 0;
}

void launch_hme(hipStream_t st, const HmeArgs &A, int npairs, Prof *pf)
{
    for (int level = A.levels; level >= 0; level--) {
        const int step = 1 << level;
        const int nvx = (A.nxb + step - 1) / step, nvy = (A.nyb + step - 1) / step;
        const double px = 2.0 * npairs * (double)A.L[level].w[0] * A.L[level].h[0];     // src + ref luma once
        if (pf) pf->begin(st, level > 0 ? KID_HME_LEVEL : KID_HME_LEVEL0, px);
        if (level > 0) hipLaunchKernelGGL((k_hme_level<false>), dim3(nvx * nvy, npairs), dim3(256), 0, st, A, level);
        else           hipLaunchKernelGGL((k_hme_level<true>), dim3(nvx * nvy, npairs), dim3(256), 0, st, A, level);
        if (pf) pf->end(st);
    }
    if (pf) pf->begin(st, KID_HME_DETAIL, 0.0);
    hipLaunchKernelGGL(k_hme_detail, dim3((A.nblk + 255) / 256, npairs), dim3(256), 0, st, A);
    if (pf) pf->end(st);
}
