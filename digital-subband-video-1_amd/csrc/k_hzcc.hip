// k_hzcc.hip -- adaptive dead-zone quantisation + interleaved exp-Golomb coefficient packing for
// gfx950 (MI355X).  Integer / bit work, HBM-bound on the coefficient plane; no MFMA.
//
// Replaces hzcc_enc (hzcc.c:137-293) + the bit writer it drives (bs.c:105-206) and the scatter half
// of hzcc_dec (hzcc.c:295-435).  The sequential coder
//     for cell in scan order: if v != 0: UEG(run); NEG(previous non-zero); ...; final NEG(last)
// is parallelised as
//   k_hz_quant<false>  one workgroup per 2048 scan cells: quantise, write the DEQUANTISED value back in place
//               (hzcc.c:172-184), compact the non-zeros in scan order with wave ballots, and sum the bit
//               lengths of every symbol whose predecessor lies in the same chunk (operator calls)
//   k_hz_collect / k_hz_collect_list   the encoder's pictures are quantised where their coefficients appear (the detail
//               bands in the forward transform, the LL region in k_fwd_haar_mid<4> / k_tail_q): what is left is the
//               compaction.  Dense pictures (I): a wave per chunk.  Sparse pictures (P): a workgroup per 64 chunks spread over
//               the picture, a lane per chunk flag, the waves share out the flagged chunks.  (k_hz_quant<true>: the LL region
//               quantised by a kernel of its own, DSV1_NO_LLQ)
//   k_hz_scan   one workgroup per plane: carries (position,value) of the last non-zero across
//               chunks (max-scan), adds each chunk's first-symbol length, prefix-sums bit offsets
//   k_hz_emit / k_hz_emit_list   a wave per chunk with entries: rounds of 64 entries, both codes of an entry in one
//               pattern where they are short (the usual round), the round's bits assembled in LDS; complete words leave
//               by plain stores, the partial word is carried into the next round, a chunk's first and last word (shared
//               with its neighbours) by atomicOr after the loop -- no round waits for a store
//   k_hz_parse  the decoder's entropy parse (state-machine scan) + k_hz_scatter_lv
// Scan regions can overlap for some plane sizes (960x540: SURVEY.md Q7); a cell seen by two
// regions is processed twice exactly like the sequential reference: the later region quantises
// the earlier region's dequantised value and owns the final store.
#include "dsvg_dev.hpp"
#include "dsvg_kernels.hpp"
#include "dsvg_host.hpp"

#define MINQ HZ_MINQ
#define q_lo hzq_lo
#define dq_lo hzdq_lo
#define q_hi hzq_hi
#define dq_hi hzdq_hi
#define cell_tq hz_cell_tq
#define quant_any hz_quant_any
#define dequant_any hz_dequant_any

// inclusive prefix sum over the wave in six DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16; row_bcast 15 / 31 across them)
static __device__ __forceinline__ unsigned wave_scan_incl(unsigned x)
{
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
static __device__ __forceinline__ int len_ueg(unsigned v) { return 2 * (31 - __clz((int)(v + 1))) + 1; }
static __device__ __forceinline__ int len_neg(int v) { return len_ueg((unsigned)(v < 0 ? -v : v) - 1u) + 1; }

static __device__ __forceinline__ int find_region(const HzPlane &hp, int p)
{
    int r = 0;
#pragma unroll
    for (int i = 1; i < 10; i++) r += (p >= hp.r[i].base) ? 1 : 0;
    return r;
}

// -------------------------------------------------------------------------------------------------
// one scan cell, general path: region lookup, overlap handling, in-place dequantised store
static __device__ __forceinline__ int hz_cell(const JobDev &jb, const HzPlane &hp, int c, int32_t *plane, int p)
{
    const HzRegion r = hp.r[find_region(hp, p)];
    const int local = p - r.base;
    const int y = local / r.sw, x = local - y * r.sw;
    const int gx = r.x0 + x, gy = r.y0 + y;
    if (p == 0) {
        jb.psum[c].dc = plane[0];                     // DC travels separately (hzcc.c:161,457-460)
        return 0;
    }
    int cv = plane[(size_t)gy * hp.w + gx];
    const int l = r.level;
    // Cells covered by two regions are processed twice by the sequential reference.
    // Both passes only READ the original value here (each emits its own symbol); the final
    // value is stored by hz_fix_overlaps() in the next kernel, so there is no in-place race.
    bool shared_cell = false;
    if (l >= 1 && gx < 2 * hp.s_w[l - 1] && gy < 2 * hp.s_h[l - 1]) {
        const int rx = gx >= hp.s_w[l - 1], ry = gy >= hp.s_h[l - 1];
        if (rx + ry) {                                // the previous level's pass ran first on this cell
            const HzRegion e = hp.r[1 + 3 * (l - 1) + (rx + 2 * ry) - 1];
            const int etq = cell_tq(e, jb.stable, hp.nbh, gx - e.x0, gy - e.y0);
            const int ev = quant_any(e, cv, etq);
            cv = ev ? dequant_any(e, ev, etq) : 0;
            shared_cell = true;
        }
    }
    if (l >= 0 && l <= 1 && (gx >= hp.s_w[l + 1] || gy >= hp.s_h[l + 1])) shared_cell = true;
    const int tq = cell_tq(r, jb.stable, hp.nbh, x, y);
    const int v = quant_any(r, cv, tq);
    if (!shared_cell) plane[(size_t)gy * hp.w + gx] = v ? dequant_any(r, v, tq) : 0;
    return v;
}

// flat chunk index of a job -> plane c and chunk inside the plane (planes with nchunks == 0 are skipped)
static __device__ __forceinline__ bool flat_chunk(const JobDev &jb, int fc, int &c, int &chunk)
{
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const int n = jb.hz[p].nchunks;
        if (fc < n) { c = p; chunk = fc; return true; }
        fc -= n;
    }
    return false;
}

// SYM: the detail regions were already quantised by the forward transform (k_fwd_haar_pix<true>), which left
// their symbols in jb.sym in scan order; only the LL region (scan cells below r[1].base) is still quantised here.
template <bool SYM>
__global__ __launch_bounds__(256) void k_hz_quant(const JobDev *__restrict__ jobs, int ll_chunks)
{
    __shared__ int s_pos[HZ_CHUNK];
    __shared__ int s_val[HZ_CHUNK];
    __shared__ int s_wcnt[4];
    __shared__ unsigned s_bits;
    const int job = blockIdx.y;
    const JobDev &jb = jobs[job];
    int c, chunk;
    if (SYM) {                      // grid.x = 3 x ll_chunks: only the chunks that reach into the LL region
        c = blockIdx.x / ll_chunks; chunk = blockIdx.x - c * ll_chunks;
        if (chunk >= jb.hz[c].nchunks || chunk * HZ_CHUNK >= jb.hz[c].r[1].base) return;
    } else if (!flat_chunk(jb, blockIdx.x, c, chunk)) return;
    const HzPlane &hp = jb.hz[c];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t *plane = jb.coef + jb.hz_coef_off[c];
    const uint8_t *stable = jb.stable;
    const int W = hp.w;
    // any region overlap in this plane?  (then rows/columns touching it take the general path)
    const bool any_ov = (2 * hp.s_w[0] > hp.s_w[1]) || (2 * hp.s_h[0] > hp.s_h[1]) ||
                        (2 * hp.s_w[1] > hp.s_w[2]) || (2 * hp.s_h[1] > hp.s_h[2]);
    if (tid == 0) s_bits = 0;

    // each wave owns 512 consecutive scan cells: 2 rounds of 256, four adjacent cells per lane
    const int wbase = chunk * HZ_CHUNK + wv * 512;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    int wcount = 0;
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        const int p0 = wbase + k * 256 + 4 * lane;
        int v[4] = {0, 0, 0, 0};
        if (SYM && p0 >= hp.r[1].base) {
            if (p0 < hp.nscan) {                       // nz_off and the scan size are multiples of 4 cells (see fill_job)
                const uint2 sv = *reinterpret_cast<const uint2 *>(jb.sym + jb.nz_off[c] + p0);
                v[0] = (int16_t)(sv.x & 0xffff); v[1] = (int)sv.x >> 16;
                v[2] = (int16_t)(sv.y & 0xffff); v[3] = (int)sv.y >> 16;
#pragma unroll
                for (int j = 1; j < 4; j++)
                    if (p0 + j >= hp.nscan) v[j] = 0;
            }
        } else if (p0 < hp.nscan) {
            const HzRegion r = hp.r[find_region(hp, p0)];
            const int local = p0 - r.base;
            const int y = local / r.sw, x = local - y * r.sw;
            const size_t idx = (size_t)(r.y0 + y) * W + r.x0 + x;
            bool fast = (p0 != 0) && (x + 3 < r.sw) && ((idx & 3) == 0);
            if (fast && any_ov) {
                const int l = r.level, gx = r.x0 + x, gy = r.y0 + y;
                if (l >= 1 && gx < 2 * hp.s_w[l - 1] && gy < 2 * hp.s_h[l - 1]) fast = false;
                if (l >= 0 && l <= 1 && (gx + 3 >= hp.s_w[l + 1] || gy >= hp.s_h[l + 1])) fast = false;
            }
            if (fast) {
                // fast path: 4 cells of one region row, 16-byte load and store
                int4 cv = *reinterpret_cast<const int4 *>(plane + idx);
                int in[4] = {cv.x, cv.y, cv.z, cv.w}, out[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int tq = cell_tq(r, stable, hp.nbh, x + j, y);
                    v[j] = quant_any(r, in[j], tq);
                    out[j] = v[j] ? dequant_any(r, v[j], tq) : 0;
                }
                *reinterpret_cast<int4 *>(plane + idx) = make_int4(out[0], out[1], out[2], out[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (p0 + j < hp.nscan)
                        v[j] = (SYM && p0 + j >= hp.r[1].base) ? (int)jb.sym[jb.nz_off[c] + p0 + j]
                                                               : hz_cell(jb, hp, c, plane, p0 + j);
            }
        }
        // compact the non-zeros in scan order: lanes in order, the 4 cells of a lane in order
        const unsigned long long b0 = __ballot(v[0] != 0), b1 = __ballot(v[1] != 0);
        const unsigned long long b2 = __ballot(v[2] != 0), b3 = __ballot(v[3] != 0);
        int rank = wcount + __popcll(b0 & ltmask) + __popcll(b1 & ltmask) + __popcll(b2 & ltmask) + __popcll(b3 & ltmask);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (v[j] != 0) {
                s_pos[wv * 512 + rank] = p0 + j;
                s_val[wv * 512 + rank] = v[j];
                rank++;
            }
        wcount += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
    }
    if (lane == 0) s_wcnt[wv] = wcount;
    __syncthreads();

    int woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int n = s_wcnt[i];
        if (i < wv) woff += n;
        total += n;
    }
    // copy this wave's segment to the plane's chunk slot and add up the in-chunk symbol lengths
    int32_t *gpos = jb.nzpos + jb.nz_off[c] + (size_t)chunk * HZ_CHUNK;
    int32_t *gval = jb.nzval + jb.nz_off[c] + (size_t)chunk * HZ_CHUNK;
    unsigned bits = 0;
    for (int j = lane; j < wcount; j += 64) {
        const int pos = s_pos[wv * 512 + j], val = s_val[wv * 512 + j];
        const int gr = woff + j;
        gpos[gr] = pos;
        gval[gr] = val;
        if (gr > 0) {
            int ppos, pval;
            if (j > 0) {
                ppos = s_pos[wv * 512 + j - 1];
                pval = s_val[wv * 512 + j - 1];
            } else {                                   // predecessor = last entry of the nearest non-empty wave
                int pw = wv - 1;
                while (s_wcnt[pw] == 0) pw--;
                ppos = s_pos[pw * 512 + s_wcnt[pw] - 1];
                pval = s_val[pw * 512 + s_wcnt[pw] - 1];
            }
            bits += (unsigned)(len_ueg((unsigned)(pos - ppos - 1)) + len_neg(pval));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits += __shfl_down(bits, o);
    if (lane == 0 && bits) atomicAdd(&s_bits, bits);
    __syncthreads();
    if (tid == 0) {
        HzChunkSum &cs = jb.chunks[jb.chunk_off[c] + chunk];
        cs.nnz = total;
        cs.bits_inner = s_bits;
        int fw = 0, lw = 3;
        while (fw < 4 && s_wcnt[fw] == 0) fw++;
        while (lw >= 0 && s_wcnt[lw] == 0) lw--;
        cs.first_pos = total ? s_pos[fw * 512] : -1;
        cs.last_pos = total ? s_pos[lw * 512 + s_wcnt[lw] - 1] : -1;
        cs.last_val = total ? s_val[lw * 512 + s_wcnt[lw] - 1] : 0;
        cs.packed = 0;
    }
}

// -------------------------------------------------------------------------------------------------
// P pictures of the encoder: the forward transform already left every detail symbol in jb.sym in scan order
// (k_fwd_haar_pix<true>); what remains is the compaction.  One WAVE per 2048-cell chunk, four chunks per
// workgroup, no LDS and no barrier: the chunk is read as 4 rounds of 16 bytes (8 symbols) per lane, all four
// loads issued up front, and the non-zeros go straight to the chunk's ordered list.  The few chunks that
// reach into the LL region (scan cells below r[1].base) are left to k_hz_quant<true>.
#ifdef EMIT_STATS
// diagnostic build (tools/ab/emit_stats.py): lifetimes of the emit waves, by chunk density
__device__ unsigned long long g_emit_stat[8][64];
__device__ unsigned long long g_coll_stat[8][64];
static int debug_stats(const void *sym, unsigned long long *out);
extern "C" int dsvg_debug_coll_stats(unsigned long long *out) { return debug_stats(HIP_SYMBOL(g_coll_stat), out); }
extern "C" int dsvg_debug_emit_stats(unsigned long long *out) { return debug_stats(HIP_SYMBOL(g_emit_stat), out); }
static int debug_stats(const void *sym, unsigned long long *out)
{
    unsigned long long h[8][64];
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, sym, sizeof h) != hipSuccess) return -1;
    for (int k = 0; k < 8; k++) {
        out[k] = 0;
        for (int i = 0; i < 64; i++) out[k] = (k == 6) ? (h[k][i] > out[k] ? h[k][i] : out[k]) : out[k] + h[k][i];
    }
    memset(h, 0, sizeof h);
    return hipMemcpyToSymbol(sym, h, sizeof h) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EMIT_STATS
#define COLL_T0 const unsigned long long t_in = wall_clock64();
#define COLL_T1(nnz_, ll_) do { if (lane == 0) { const unsigned long long dt = wall_clock64() - t_in; const int sh = (chunk * 7 + c) & 63, k = (ll_) ? 3 : 0; \
    atomicAdd(&g_coll_stat[k][sh], 1ull); atomicAdd(&g_coll_stat[k + 1][sh], (unsigned long long)(nnz_)); atomicAdd(&g_coll_stat[k + 2][sh], dt); atomicMax(&g_coll_stat[6][sh], dt); } } while (0)
#else
#define COLL_T0
#define COLL_T1(nnz_, ll_)
#endif
// compaction state of one chunk and one round of it: 8 consecutive cells per lane (v[j] = symbol of scan cell p0 + j), the
// non-zeros appended to the chunk's ordered list, in-chunk code lengths summed
struct CollectState {
    int run = 0;                    // entries written so far
    int cpos = -1, cval = 0;        // last entry so far (wave-uniform)
    int first_pos = -1;
    unsigned bits = 0;              // per-lane partial of bits_inner
};
static __device__ __forceinline__ void collect_round(CollectState &st, const int (&v)[8], int p0, int lane, int32_t *gpos, int32_t *gval)
{
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) cnt += (v[j] != 0);
    const unsigned long long have = __ballot(cnt != 0);
    if (have == 0ull) return;                                 // wave-uniform: a round of zeros costs nothing more
    const int incl = (int)wave_scan_incl((unsigned)cnt);
    // this lane's last entry, for the lanes after it
    int lpos = -1, lval = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (v[j] != 0) { lpos = p0 + j; lval = v[j]; }
    // predecessor of this lane's first entry: the nearest earlier lane with entries, else the carry
    const unsigned long long before = have & ltmask;
    const int src = before ? 63 - __clzll(before) : 0;
    int ppos = __shfl(lpos, src), pval = __shfl(lval, src);
    if (!before) { ppos = st.cpos; pval = st.cval; }
    int at = st.run + incl - cnt;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (v[j] != 0) {
            gpos[at] = p0 + j;
            gval[at] = v[j];
            if (at > 0) st.bits += (unsigned)(len_ueg((unsigned)(p0 + j - ppos - 1)) + len_neg(pval));
            ppos = p0 + j; pval = v[j];
            at++;
        }
    }
    const int lastl = 63 - __clzll(have), firstl = __ffsll((long long)have) - 1;
    if (st.first_pos < 0) {                                   // first entry of the chunk = first entry of lane firstl
        int fp = -1;
#pragma unroll
        for (int j = 7; j >= 0; j--)
            if (v[j] != 0) fp = p0 + j;
        st.first_pos = __builtin_amdgcn_readlane(fp, firstl);
    }
    st.cpos = __builtin_amdgcn_readlane(lpos, lastl); st.cval = __builtin_amdgcn_readlane(lval, lastl);
    st.run += __builtin_amdgcn_readlane(incl, 63);
}
static __device__ __forceinline__ void collect_finish(const JobDev &jb, int c, int chunk, int lane, CollectState &st, int packed = 0)
{
    unsigned bits = st.bits;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits += __shfl_down(bits, o);
    if (lane == 0) {
        HzChunkSum &cs = jb.chunks[jb.chunk_off[c] + chunk];
        cs.nnz = st.run;
        cs.bits_inner = bits;
        cs.first_pos = st.run ? st.first_pos : -1;
        cs.last_pos = st.run ? st.cpos : -1;
        cs.last_val = st.run ? st.cval : 0;
        cs.packed = packed;
    }
}

// Round 4: the detail chunks' lists in ONE word per entry -- (symbol << 16) | position inside the chunk (11 bits; the symbols
// of the detail regions are int16 by construction: jb.sym) -- and compacted through the wave's LDS window first.  The lanes
// used to store their up to eight entries of a round one by one (eight predicated position / value store pairs per round: 64
// vector-memory instructions per chunk, most of them for a few lanes -- the kernel ran at the rate the CU takes such
// instructions, 0.42 of the HBM peak and half of the VALU issue rate); now a round's entries leave as full 64-lane stores, and
// the in-chunk code lengths are taken from the compacted list (an entry and its predecessor sit next to each other) instead
// of along each lane's serial chain.  rw: the lane's eight int16 symbols as loaded, p0rel: the first one's position in the chunk
#define COLL_STAGE_WORDS 512
static __device__ __forceinline__ void collect_round_pk(CollectState &st, const uint4 &rw, int p0rel, int cbase, int lane, unsigned *stg,
                                                         DSVG_GLOBAL unsigned *gent)
{
    const unsigned w[4] = {rw.x, rw.y, rw.z, rw.w};
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) cnt += ((w[j] & 0xffffu) != 0u) + ((w[j] >> 16) != 0u);
    const unsigned incl = wave_scan_incl((unsigned)cnt);
    const int n = __builtin_amdgcn_readlane((int)incl, 63);
    unsigned at = incl - (unsigned)cnt;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if ((w[j] & 0xffffu) != 0u) stg[at++] = (w[j] << 16) | (unsigned)(p0rel + 2 * j);
        if ((w[j] >> 16) != 0u) stg[at++] = (w[j] & 0xffff0000u) | (unsigned)(p0rel + 2 * j + 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    const unsigned cent = ((unsigned)st.cval << 16) | (unsigned)(st.cpos - cbase);      // last entry of the rounds before (if any)
    for (int i = lane; i < n; i += 64) {
        const unsigned e = stg[i], pe = i ? stg[i - 1] : cent;
        gent[(unsigned)(st.run + i)] = e;
        if (st.run + i > 0) st.bits += (unsigned)(len_ueg((e & 0x7ffu) - (pe & 0x7ffu) - 1u) + len_neg((int)pe >> 16));
    }
    const unsigned e0 = stg[0], el = stg[n - 1];                  // (broadcast reads)
    if (st.first_pos < 0) st.first_pos = cbase + (int)(__builtin_amdgcn_readfirstlane((int)e0) & 0x7ff);
    const int eli = __builtin_amdgcn_readfirstlane((int)el);
    st.cpos = cbase + (eli & 0x7ff); st.cval = eli >> 16;
    st.run += n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

// a chunk that reaches into the LL region of a job whose LL symbols were written by the transform (jb.llq): cells below
// ll_end come from the int32 LL symbol plane, the detail cells of the chunk that straddles ll_end from the int16 plane
// (sparse jobs: taken down again, as in collect_chunk)
static __device__ __forceinline__ void collect_chunk_ll(const JobDev &jb, int c, int chunk, int lane)
{
    const HzPlane &hp = jb.hz[c];
    const int ll_end = hp.r[1].base, nscan = hp.nscan, cbase = chunk * HZ_CHUNK;
    const int32_t *ls = jb.llsym + jb.ll_off[c];
    int16_t *sym = jb.sym + jb.nz_off[c];
    uint8_t *nzf = jb.nzf ? jb.nzf + (jb.nz_off[c] >> 2) : nullptr;
    int32_t *gpos = jb.nzpos + jb.nz_off[c] + (size_t)chunk * HZ_CHUNK;
    int32_t *gval = jb.nzval + jb.nz_off[c] + (size_t)chunk * HZ_CHUNK;
    COLL_T0
    CollectState st;
    int v[4][8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p0 = cbase + k * 512 + 8 * lane;
#pragma unroll
        for (int j = 0; j < 8; j++) v[k][j] = 0;
        if (p0 + 8 <= ll_end) {
            const int4 a = *reinterpret_cast<const int4 *>(ls + p0), b = *reinterpret_cast<const int4 *>(ls + p0 + 4);
            v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w; v[k][4] = b.x; v[k][5] = b.y; v[k][6] = b.z; v[k][7] = b.w;
        } else if (p0 < nscan) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int p = p0 + j;
                if (p < ll_end) v[k][j] = ls[p];
                else if (p < nscan && (!nzf || nzf[p >> 2])) {
                    v[k][j] = sym[p];
                    if (nzf) sym[p] = 0;
                }
            }
        }
    }
    if (nzf && cbase + HZ_CHUNK > ll_end) {
        // the flags of the detail cells read above (a flag byte covers 4 cells: cleared after every lane has read its cells)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p0 = cbase + k * 512 + 8 * lane;
            if (p0 + 8 > ll_end && p0 < nscan) {
                if (p0 + 3 >= ll_end) nzf[p0 >> 2] = 0;
                if (p0 + 7 >= ll_end) nzf[(p0 >> 2) + 1] = 0;
            }
        }
        if (lane == 0) jb.cflag[jb.chunk_off[c] + chunk] = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) collect_round(st, v[k], cbase + k * 512 + 8 * lane, lane, gpos, gval);
    collect_finish(jb, c, chunk, lane, st);
    COLL_T1(st.run, 1);
}

// FLAGGED: the caller has seen the chunk's flag up (k_hz_collect_list): it is not fetched again -- a memory round trip in front of the group flags'
template <bool FLAGGED = false>
static __device__ __forceinline__ void collect_chunk(const JobDev &jb, int c, int chunk, int lane, unsigned *stg)
{
    const HzPlane &hp = jb.hz[c];
    const int ll_end = hp.r[1].base, nscan = hp.nscan;
    const int16_t *sym = jb.sym + jb.nz_off[c];
    int32_t *gpos = jb.nzpos + jb.nz_off[c] + (size_t)chunk * HZ_CHUNK;
    const int cbase = chunk * HZ_CHUNK;
    int16_t *symw = jb.sym + jb.nz_off[c];
    uint8_t *cfl = jb.nzf ? jb.cflag + jb.chunk_off[c] + chunk : nullptr;
    if (cbase < ll_end && jb.llq) { collect_chunk_ll(jb, c, chunk, lane); return; }
    if (cbase < ll_end) {
        // chunks that reach into the LL region were compacted by k_hz_quant<true>; in sparse mode the detail symbols of
        // the chunk that straddles the end of the LL region still have to be taken down (this kernel is their last reader)
        if (cfl && cbase + HZ_CHUNK > ll_end && (FLAGGED || *cfl)) {
            uint8_t *nzf = jb.nzf + (jb.nz_off[c] >> 2);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int p0 = cbase + k * 512 + 8 * lane;
                if (p0 < nscan && *reinterpret_cast<const unsigned short *>(nzf + (p0 >> 2))) {
                    *reinterpret_cast<uint4 *>(symw + p0) = make_uint4(0, 0, 0, 0);
                    *reinterpret_cast<unsigned short *>(nzf + (p0 >> 2)) = 0;
                }
            }
            if (lane == 0) *cfl = 0;
        }
        return;
    }
    if (!FLAGGED && cfl && *cfl == 0) { // sparse mode: nothing was stored into this chunk
        if (lane == 0) {
            HzChunkSum &cs = jb.chunks[jb.chunk_off[c] + chunk];
            cs.nnz = 0; cs.bits_inner = 0; cs.first_pos = -1; cs.last_pos = -1; cs.last_val = 0;
        }
        return;
    }
    COLL_T0
    uint4 raw[4];
    if (jb.nzf) {
        // P pictures: the forward transform flagged every group of four cells that holds a non-zero symbol -- two flag
        // bytes per lane and round instead of sixteen symbol bytes; symbols are fetched only where a flag is up, and
        // symbols, flags and the chunk flag are taken down again for the next picture (the plane stays zero)
        uint8_t *nzf = jb.nzf + (jb.nz_off[c] >> 2);
        unsigned short fl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p0 = cbase + k * 512 + 8 * lane;
            fl[k] = p0 < nscan ? *reinterpret_cast<const unsigned short *>(nzf + (p0 >> 2)) : (unsigned short)0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p0 = cbase + k * 512 + 8 * lane;
            raw[k] = make_uint4(0, 0, 0, 0);
            if (fl[k]) {
                raw[k] = *reinterpret_cast<const uint4 *>(sym + p0);
                *reinterpret_cast<uint4 *>(symw + p0) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<unsigned short *>(nzf + (p0 >> 2)) = 0;
            }
        }
        if (lane == 0) *cfl = 0;
    } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p0 = cbase + k * 512 + 8 * lane;
        raw[k] = make_uint4(0, 0, 0, 0);
        if (p0 < nscan) raw[k] = *reinterpret_cast<const uint4 *>(sym + p0);   // the planes are padded to whole chunks
    }
    }
    CollectState st;
    DSVG_GLOBAL unsigned *gent = dsvg_global(reinterpret_cast<unsigned *>(gpos));
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p0 = cbase + k * 512 + 8 * lane;
        uint4 rw = raw[k];
        if (p0 + 8 > nscan) {           // cells past the end of the scan (the plane's last chunk)
            unsigned *w = reinterpret_cast<unsigned *>(&rw);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (p0 + 2 * j >= nscan) w[j] = 0u;
                else if (p0 + 2 * j + 1 >= nscan) w[j] &= 0xffffu;
            }
        }
        // most rounds of a P picture hold no symbol at all: one OR and a ballot decide that before anything else
        if (__ballot((rw.x | rw.y | rw.z | rw.w) != 0u) == 0ull) continue;
        collect_round_pk(st, rw, k * 512 + 8 * lane, cbase, lane, stg, gent);
    }
    collect_finish(jb, c, chunk, lane, st, 1);
    COLL_T1(st.run, 0);
}

__global__ __launch_bounds__(256) void k_hz_collect(const JobDev *__restrict__ jobs)
{
    const JobDev &jb = jobs[blockIdx.y];
    int c, chunk;
    __shared__ unsigned s_cstage[4][COLL_STAGE_WORDS];
    if (!flat_chunk(jb, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c, chunk)) return;
    collect_chunk(jb, c, chunk, threadIdx.x & 63, s_cstage[threadIdx.x >> 6]);
}

// Sparse pictures: 93 % of a P picture's chunks hold nothing, and a launch of one wave per chunk is bound by the rate at
// which waves can be started (121 k waves that read one flag byte each).  Here a workgroup owns 64 * Q chunks, spread
// over the picture (chunk = workgroup + k * workgroups: flagged chunks cluster where something moves); every wave reads
// the same 64 * Q chunk flags, Q per lane, and takes every fourth flagged chunk -- no LDS, no barrier, and on
// average well under one chunk per wave, so nothing is serialised.  Chunks without a flag get their empty summary from
// wave 0.  (A dense job in such a launch -- a scene change inside a step -- is handled too: every chunk counts as flagged.)
#ifndef HZ_LIST_Q
#define HZ_LIST_Q 1               // chunks per workgroup / 64 of the large launches (measured per 320-GOP step, collect + emit: Q = 1 1.26 + 0.80 ms, 2: 1.24 + 0.80, 4: 1.23 + 0.80, 8: 1.24 + 0.85 -- not worth a second variant)
#endif
#ifndef HZ_LIST_Q_MIN_WGS
#define HZ_LIST_Q_MIN_WGS 4096     // ... which are those that still start this many workgroups
#endif
// Q: the workgroup owns 64 * Q chunks (Q flags per lane).  Fewer than 64 chunks per workgroup cost workgroup starts (per 320-GOP
// step: 16 chunks 1.84 ms, 32: 1.45, 64: 1.28); more than 64 gain nothing (HZ_LIST_Q), what is left is the flagged chunks' chain
// of dependent fetches (chunk flags, group flags, symbols) at two waves' lifetimes per launch
template <int Q>
__global__ __launch_bounds__(256) void k_hz_collect_list(const JobDev *__restrict__ jobs)
{
    __shared__ unsigned s_cstage[4][COLL_STAGE_WORDS];
    const JobDev &jb = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int c[Q], chunk[Q];
    unsigned long long m[Q];
    const int ll0 = jb.hz[0].r[1].base, ll1 = jb.hz[1].r[1].base, ll2 = jb.hz[2].r[1].base;
    const int co0 = jb.chunk_off[0], co1 = jb.chunk_off[1], co2 = jb.chunk_off[2];
    const bool sparse = jb.nzf != nullptr;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        c[q] = chunk[q] = 0;
        const bool have = flat_chunk(jb, (int)blockIdx.x + (64 * q + lane) * (int)gridDim.x, c[q], chunk[q]);
        bool work = false;
        if (have) {
            // (the plane's facts by scalar loads and a select: indexed by the lane's plane they were vector loads from the job table,
            // a memory round trip in front of the chunk flag's own)
            const int ll_end = c[q] == 0 ? ll0 : (c[q] == 1 ? ll1 : ll2), coff = c[q] == 0 ? co0 : (c[q] == 1 ? co1 : co2), cbase = chunk[q] * HZ_CHUNK;
            const bool fl = sparse ? dsvg_global(static_cast<const uint8_t *>(jb.cflag))[(unsigned)(coff + chunk[q])] != 0 : true;
            if (cbase < ll_end) work = jb.llq || (sparse && cbase + HZ_CHUNK > ll_end && fl);   // llq: the LL chunks are compacted here; else only the straddling chunk's cleanup
            else {
                work = fl;
                if (!fl && wv == 0) jb.chunks[coff + chunk[q]].nnz = 0;     // (of an empty chunk's summary only the count is ever read)
            }
        }
        m[q] = __ballot(work);
    }
    // collect_chunk takes the chunk flags down: every wave must have its view of them before any wave starts (a late wave
    // would count fewer flagged chunks, take the wrong ones, and overwrite a finished summary with an empty one)
    __syncthreads();
    int k = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        unsigned long long mm = m[q];
        for (; mm; k++) {
            const int l = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            if ((k & 3) != wv) continue;
            if (sparse) collect_chunk<true>(jb, __builtin_amdgcn_readlane(c[q], l), __builtin_amdgcn_readlane(chunk[q], l), lane, s_cstage[wv]);
            else collect_chunk<false>(jb, __builtin_amdgcn_readlane(c[q], l), __builtin_amdgcn_readlane(chunk[q], l), lane, s_cstage[wv]);      // (scalars: the job table is then read with scalar loads)
        }
    }
}

// -------------------------------------------------------------------------------------------------
// block-wide inclusive scans over 1024 threads (16 waves)
template <typename T, typename Op>
static __device__ __forceinline__ T block_scan_incl(T v, T identity, Op op, T *s_w /*[16]*/)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T n = __shfl_up(v, o);
        if (lane >= o) v = op(v, n);
    }
    __syncthreads();
    if (lane == 63) s_w[wv] = v;
    __syncthreads();
    T pre = identity;
    for (int i = 0; i < wv; i++) pre = op(pre, s_w[i]);
    return op(pre, v);
}
struct OpAddU64 { __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a + b; } };
struct OpAddI   { __device__ int operator()(int a, int b) const { return a + b; } };
struct OpMaxI   { __device__ int operator()(int a, int b) const { return a > b ? a : b; } };

// final value of every cell shared by two scan regions: quantise/dequantise with the earlier
// region's rule, then with the later region's (hzcc.c:172-184 applied twice in scan order)
static __device__ void hz_fix_overlaps(const JobDev &jb, const HzPlane &hp, int c, int nthreads)
{
    int32_t *plane = jb.coef + jb.hz_coef_off[c];
    for (int l = 0; l < 2; l++) {
        const int cw = 2 * hp.s_w[l], ch = 2 * hp.s_h[l];        // extent covered up to level l
        const bool col = cw > hp.s_w[l + 1], row = ch > hp.s_h[l + 1];
        const int ncol = col ? ch : 0, nrow = row ? cw : 0;
        for (int i = threadIdx.x; i < ncol + nrow; i += nthreads) {
            int gx, gy;
            if (i < ncol) { gx = hp.s_w[l + 1]; gy = i; }
            else {
                gx = i - ncol; gy = hp.s_h[l + 1];
                if (col && gx == hp.s_w[l + 1]) continue;          // corner already handled by the column
            }
            const int ex = gx >= hp.s_w[l], ey = gy >= hp.s_h[l];
            if (!(ex + ey)) continue;
            const HzRegion e = hp.r[1 + 3 * l + (ex + 2 * ey) - 1];
            const int lx = gx >= hp.s_w[l + 1], ly = gy >= hp.s_h[l + 1];
            const HzRegion r = hp.r[1 + 3 * (l + 1) + (lx + 2 * ly) - 1];
            int v = plane[(size_t)gy * hp.w + gx];
            const int etq = cell_tq(e, jb.stable, hp.nbh, gx - e.x0, gy - e.y0);
            const int ev = quant_any(e, v, etq);
            v = ev ? dequant_any(e, ev, etq) : 0;
            const int tq = cell_tq(r, jb.stable, hp.nbh, gx - r.x0, gy - r.y0);
            const int rv = quant_any(r, v, tq);
            plane[(size_t)gy * hp.w + gx] = rv ? dequant_any(r, rv, tq) : 0;
        }
    }
}

// 256 threads, not 1024: beside another stream's saturating kernel a 16-wave workgroup waits for a whole CU's worth of
// wave slots to come free at once -- measured 118 us per launch in the two-stream timed region against 19 us alone on the
// chip, on the critical path of every frame step.  Planes with more than 2048 chunks are walked in tiles with carries.
#define SCAN_THREADS 256
#define SCAN_ITEMS 8            // chunks per thread and tile

// Round 4: the scan sits on the critical path of every frame step of a small batch (27 us per 4K 4:4:4 picture) for its latency:
// a thread read its eight chunk summaries one dependent load after the other, twice.  Now every field a pass needs of the
// thread's chunks is requested in ONE batch of independent loads (nnz; then first / last position, last value, inner bits), the
// only dependent fetch left is the carried-in predecessor's two words, and a launch with few workgroups takes NT = 1024 threads.
template <int NT>
__global__ __launch_bounds__(NT) void k_hz_scan(const JobDev *__restrict__ jobs)
{
    constexpr int TILE = NT * SCAN_ITEMS;
    __shared__ unsigned long long s_u64[16];
    __shared__ int s_i[16];
    __shared__ int s_incl_ne[NT];
    __shared__ int s_c_ne, s_c_nnz;                            // carries into the next tile: last non-empty chunk, entries so far,
    __shared__ unsigned long long s_c_bits;                    // bits so far
    const int job = blockIdx.y, c = blockIdx.x;
    const JobDev &jb = jobs[job];
    const HzPlane &hp = jb.hz[c];
    HzChunkSum *cs = jb.chunks + jb.chunk_off[c];
    const int n = hp.nchunks;
    if (n > 0 && !jb.fused) hz_fix_overlaps(jb, hp, c, NT);   // the fused path stores final values itself
    if (threadIdx.x == 0) { s_c_ne = -1; s_c_nnz = 0; s_c_bits = 0ull; }
    __syncthreads();
    for (int t0 = 0; t0 < n; t0 += TILE) {
        const int limit = min(n, t0 + TILE);
        const int per = (limit - t0 + NT - 1) / NT;     // <= SCAN_ITEMS
        const int first = t0 + threadIdx.x * per;
        const int c_ne = s_c_ne, c_nnz = s_c_nnz;
        const unsigned long long c_bits = s_c_bits;

        // pass 1: index of the last non-empty chunk at or before each chunk (max-scan), nnz prefix
        int z[SCAN_ITEMS];
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) z[i] = (i < per && first + i < limit) ? cs[first + i].nnz : 0;
        int lastne = -1, nnzsum = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (z[i] > 0) lastne = first + i;
            nnzsum += z[i];
        }
        // the fields pass 2 needs of this thread's non-empty chunks: requested now, under the two block scans
        int fpos[SCAN_ITEMS], lpos[SCAN_ITEMS], lval[SCAN_ITEMS];
        unsigned inner[SCAN_ITEMS];
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            fpos[i] = lpos[i] = lval[i] = 0; inner[i] = 0;
            if (z[i] > 0) { const HzChunkSum &q = cs[first + i]; fpos[i] = q.first_pos; lpos[i] = q.last_pos; lval[i] = q.last_val; inner[i] = q.bits_inner; }
        }
        const int incl_ne = max(block_scan_incl<int>(lastne, -1, OpMaxI(), s_i), c_ne);
        __syncthreads();
        s_incl_ne[threadIdx.x] = incl_ne;
        __syncthreads();
        // exclusive: last non-empty chunk strictly before this thread's first chunk
        const int carry_ne = threadIdx.x ? s_incl_ne[threadIdx.x - 1] : c_ne;
        int ppos = -1, pval = 0;                               // ... and its last entry (the one dependent fetch of the tile)
        if (carry_ne >= 0) { ppos = cs[carry_ne].last_pos; pval = cs[carry_ne].last_val; }

        const int incl_nnz = block_scan_incl<int>(nnzsum, 0, OpAddI(), s_i);
        int nzbase = c_nnz + incl_nnz - nnzsum;

        // pass 2: per chunk first-symbol length, chunk bit totals
        unsigned long long mybits = 0;
        bool have_prev = carry_ne >= 0;
        unsigned long long cb[SCAN_ITEMS];
        int pp[SCAN_ITEMS], pvv[SCAN_ITEMS];
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            cb[i] = 0; pp[i] = ppos; pvv[i] = pval;
            if (z[i] > 0) {
                unsigned b = (unsigned)len_ueg((unsigned)(fpos[i] - ppos - 1));
                if (have_prev) b += (unsigned)len_neg(pval);
                cb[i] = (unsigned long long)b + inner[i];
                have_prev = true; ppos = lpos[i]; pval = lval[i];
            }
            mybits += cb[i];
        }
        const unsigned long long incl_bits = block_scan_incl<unsigned long long>(mybits, 0ull, OpAddU64(), s_u64);
        unsigned long long off = c_bits + incl_bits - mybits;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const int ch = first + i;
            if (i < per && ch < limit) {
                cs[ch].bit_off = off;
                cs[ch].prev_pos = pp[i];
                cs[ch].prev_val = pvv[i];
                cs[ch].nz_base = nzbase;
                off += cb[i];
                nzbase += z[i];
            }
        }
        __syncthreads();                                       // every thread has read the carries of this tile
        if (threadIdx.x == NT - 1) { s_c_ne = incl_ne; s_c_nnz = c_nnz + incl_nnz; s_c_bits = c_bits + incl_bits; }
        __syncthreads();
    }
    __shared__ unsigned long long s_total;
    if (threadIdx.x == 0) {
        HzPlaneSum &ps = jb.psum[c];
        const int lne = s_c_ne;                                // last non-empty chunk of the plane
        unsigned long long tb = s_c_bits;
        if (lne >= 0) tb += (unsigned long long)len_neg(cs[lne].last_val);   // trailing NEG (hzcc.c:283-285)
        ps.total_bits = tb;
        ps.nruns = (unsigned)s_c_nnz;
        ps.last_chunk = lne;
        ps.overflow = ((tb + 7) >> 3) > jb.bits_cap[c] ? 1 : 0;
        s_total = ps.overflow ? 0ull : tb;
    }
    __syncthreads();
    // the emit kernel ORs into the payload (bs.c:50-63 semantics): clear exactly the words it will touch
    unsigned *out32 = reinterpret_cast<unsigned *>(jb.bits + jb.bits_off[c]);
    const unsigned nwords = (unsigned)((s_total + 31) >> 5) + 4;
    for (unsigned i = threadIdx.x; i < nwords; i += NT) out32[i] = 0;
}

// compact the packed planes of many pictures into one contiguous buffer (one D2H instead of 3 per picture)
__global__ __launch_bounds__(256) void k_gather_bits(const uint8_t *__restrict__ bits, size_t bits_per_job,
                                                     const unsigned long long *__restrict__ tab, uint8_t *__restrict__ dst)
{
    // tab[3*i+0] = source offset inside `bits`, tab[3*i+1] = destination offset, tab[3*i+2] = byte count (mult. of 4 ok)
    const unsigned long long so = tab[3 * blockIdx.y], d0 = tab[3 * blockIdx.y + 1], n = tab[3 * blockIdx.y + 2];
    const unsigned *s32 = reinterpret_cast<const unsigned *>(bits + so);
    unsigned *d32 = reinterpret_cast<unsigned *>(dst + d0);
    const unsigned long long nw = (n + 3) >> 2;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < nw; i += (unsigned long long)gridDim.x * 256ull) d32[i] = s32[i];
    (void)bits_per_job;
}

// -------------------------------------------------------------------------------------------------
static __device__ __forceinline__ unsigned long long spread_bits(unsigned long long x)   // bit i -> bit 2i
{
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
// UEG(v) as an MSB-first integer pattern (bs.c:129-145): k x ('0', bit) then '1'
static __device__ __forceinline__ unsigned long long pat_ueg(unsigned v, int &len)
{
    const unsigned m = v + 1;
    const int k = 31 - __clz((int)m);
    len = 2 * k + 1;
    return (spread_bits((unsigned long long)(m & ((1u << k) - 1u))) << 1) | 1ull;
}
static __device__ __forceinline__ unsigned long long pat_neg(int v, int &len)      // bs.c:191-206
{
    const unsigned mag = (unsigned)(v < 0 ? -v : v);
    int l;
    const unsigned long long u = pat_ueg(mag - 1u, l);
    len = l + 1;
    return (u << 1) | (v < 0 ? 1ull : 0ull);
}

// OR `len` (<= 48) bits of `pat` into the MSB-first byte stream at bit position `pos`
static __device__ __forceinline__ void or_bits(unsigned *out32, unsigned long long pos, unsigned long long pat, int len)
{
    const unsigned long long w = pos >> 5;
    const int o = (int)(pos & 31);
    const int sh = 64 - o - len;             // left shift of pat inside the first 64-bit window
    unsigned long long hi, lo = 0;
    if (sh >= 0) hi = pat << sh;
    else { hi = pat >> (-sh); lo = pat << (64 + sh); }
    const unsigned w0 = (unsigned)(hi >> 32), w1 = (unsigned)hi, w2 = (unsigned)(lo >> 32);
    if (w0) atomicOr(out32 + w, __builtin_bswap32(w0));
    if (w1) atomicOr(out32 + w + 1, __builtin_bswap32(w1));
    if (w2) atomicOr(out32 + w + 2, __builtin_bswap32(w2));
}

// OR `len` (<= 48) bits of `pat` into a word array held MSB-first per 32-bit VALUE (no byte swap) at bit `pos`
static __device__ __forceinline__ void or_bits_lds(unsigned *w32, unsigned pos, unsigned long long pat, int len)
{
    const unsigned w = pos >> 5;
    const int o = (int)(pos & 31);
    const int sh = 64 - o - len;
    unsigned long long hi, lo = 0;
    if (sh >= 0) hi = pat << sh;
    else { hi = pat >> (-sh); lo = pat << (64 + sh); }
    const unsigned w0 = (unsigned)(hi >> 32), w1 = (unsigned)hi, w2 = (unsigned)(lo >> 32);
    if (w0) atomicOr(w32 + w, w0);
    if (w1) atomicOr(w32 + w + 1, w1);
    if (w2) atomicOr(w32 + w + 2, w2);
}

// One WAVE per chunk (most chunks of a P picture hold a few dozen symbols), four chunks per workgroup.
// Each round of 64 symbols is assembled in the wave's LDS window with LDS atomics and flushed with plain
// coalesced stores; only the first and last word of a round can be shared (with the neighbouring round or
// chunk) and go out as global atomicOr -- an I picture would otherwise issue ~2.5 global atomics per symbol.
// the fences order the wave's LDS traffic only: a fence over all address spaces would also wait for the round's global
// stores to be acknowledged -- a round trip per round of 64 symbols on the serial path of a dense chunk
#ifdef EMIT_FULL_FENCE
#define EMIT_FENCE(o) __builtin_amdgcn_fence(o, "wavefront")
#else
#define EMIT_FENCE(o) __builtin_amdgcn_fence(o, "wavefront", "local")
#endif
#define EMIT_STAGE_WORDS 256            // rounds of 64 entries: 64 x (<= 47 + 49 bits) = 6144 bits = 192 words, + straddle; of 256 entries with codes <= 31 bits: 249
// lanes that have no word to store in a round store here instead: every round then issues the same vector-memory
// operations, which lets the compiler wait for the prefetched entries alone (s_waitcnt vmcnt(1)) instead of for everything
__device__ unsigned g_emit_dump[128];

// One chunk by one wave.  A round takes 64 entries (one per lane), builds their codes, assembles the round's bits in the
// wave's LDS window and stores the COMPLETE words it filled with plain stores; the incomplete last word is carried into
// the next round's first word.  Only the chunk's first and last word can be shared with a neighbouring chunk: those two
// go out as global atomicOr after the loop.  Nothing in a round waits for a store: the serial path of a dense chunk
// (32 rounds) is arithmetic + LDS only.
struct EmitState {
    unsigned at0, wbase;             // bit position relative to the chunk's first word; that word's index in the payload
    unsigned carry, firstv;          // the incomplete last word so far; the chunk's first word once it is complete
    bool first_pending;              // the chunk's first word has not been taken out of the stage yet
    int cpos, cval;                  // the entry before the round's first one (before the first round: the chunk's predecessor)
};
// entry j = (pos, val) of this lane; prev_in: the chunk has a predecessor in the plane (cs.prev_pos >= 0)
static __device__ __forceinline__ void emit_round64(EmitState &S, bool prev_in, int j, int nnz, int pos, int val, int lane, unsigned *stg,
                                                    DSVG_GLOBAL unsigned *out32)
{
    int ppos = __builtin_amdgcn_update_dpp(0, pos, 0x138, 0xf, 0xf, true);      // wave_shr:1
    int pval = __builtin_amdgcn_update_dpp(0, val, 0x138, 0xf, 0xf, true);
    if (lane == 0) { ppos = S.cpos; pval = S.cval; }
    S.cpos = __builtin_amdgcn_readlane(pos, 63); S.cval = __builtin_amdgcn_readlane(val, 63);
    const bool valid = j < nnz, hasn = valid && (j > 0 || prev_in);
    const unsigned m = valid ? (unsigned)(pos - ppos - 1) + 1u : 1u;              // UEG codes run + 1
    const unsigned mag = hasn ? (unsigned)(pval < 0 ? -pval : pval) : 1u;         // NEG codes UEG(|v| - 1), then the sign
    const unsigned at0 = S.at0, wbase = S.wbase, carry = S.carry;
    const bool first_pending = S.first_pending;
    const unsigned o0 = at0 & 31, w0 = at0 >> 5;
    unsigned tot;
    if (__ballot(m >= 256u || mag >= 256u) == 0ull) {
        // the common round: runs and values below 256, i.e. at most 15 + 16 bits per lane -- both codes in one 32-bit
        // pattern (both bit spreads in one go), a round of at most 63 words, two LDS atomics per lane
        const int k1 = 31 - __clz((int)m), k2 = 31 - __clz((int)mag);
        unsigned x = ((m & ((1u << k1) - 1u)) << 16) | (mag & ((1u << k2) - 1u));
        x = (x | (x << 4)) & 0x0F0F0F0Fu;
        x = (x | (x << 2)) & 0x33333333u;
        x = (x | (x << 1)) & 0x55555555u;
        unsigned pat = ((x >> 16) << 1) | 1u;
        unsigned len = valid ? 2u * k1 + 1u : 0u;
        if (hasn) {
            pat = (pat << (2 * k2 + 2)) | ((((x & 0xffffu) << 1) | 1u) << 1) | (pval < 0 ? 1u : 0u);
            len += 2u * k2 + 2u;
        }
        const unsigned incl = wave_scan_incl(len);
        tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        stg[lane] = lane ? 0u : carry;
        if (lane < 2) stg[64 + lane] = 0;
        EMIT_FENCE(__ATOMIC_RELEASE);
        __builtin_amdgcn_wave_barrier();
        if (len) {
            const unsigned rel = o0 + (incl - len);
            const unsigned long long hi = (unsigned long long)pat << (64 - (rel & 31) - len);
            const unsigned h0 = (unsigned)(hi >> 32), h1 = (unsigned)hi;
            if (h0) atomicOr(stg + (rel >> 5), h0);
            if (h1) atomicOr(stg + (rel >> 5) + 1, h1);
        }
        EMIT_FENCE(__ATOMIC_ACQ_REL);
        __builtin_amdgcn_wave_barrier();
        const int nfull = (int)((o0 + tot) >> 5);                     // complete words of the round, <= 63
        const unsigned v = stg[lane];
        S.carry = (unsigned)__builtin_amdgcn_readlane((int)v, nfull & 63);
        if (nfull == 64) S.carry = 0;                                 // (cannot happen: 31 + 64 * 31 bits)
        const bool mine = lane < nfull && !(first_pending && lane == 0);
        DSVG_GLOBAL unsigned *d = mine ? out32 + (wbase + w0 + (unsigned)lane) : (DSVG_GLOBAL unsigned *)g_emit_dump + lane;
        *d = __builtin_bswap32(v);
        ((DSVG_GLOBAL unsigned *)g_emit_dump)[64 + lane] = 0;         // (every path of the round: two stores per lane)
        if (first_pending && nfull > 0) { S.firstv = (unsigned)__builtin_amdgcn_readfirstlane((int)v); S.first_pending = false; }
    } else if (__ballot(m >= 32768u || mag >= 32768u) == 0ull) {
        // runs and values below 2^15 (the LL region of a picture): codes of at most 29 + 30 bits in one 64-bit pattern,
        // a round of at most 120 words, two words per lane
        const int k1 = 31 - __clz((int)m), k2 = 31 - __clz((int)mag);
        unsigned x1 = m & ((1u << k1) - 1u), x2 = mag & ((1u << k2) - 1u);
        x1 = (x1 | (x1 << 8)) & 0x00FF00FFu; x2 = (x2 | (x2 << 8)) & 0x00FF00FFu;
        x1 = (x1 | (x1 << 4)) & 0x0F0F0F0Fu; x2 = (x2 | (x2 << 4)) & 0x0F0F0F0Fu;
        x1 = (x1 | (x1 << 2)) & 0x33333333u; x2 = (x2 | (x2 << 2)) & 0x33333333u;
        x1 = (x1 | (x1 << 1)) & 0x55555555u; x2 = (x2 | (x2 << 1)) & 0x55555555u;
        unsigned long long pat = ((unsigned long long)x1 << 1) | 1ull;
        unsigned len = valid ? 2u * k1 + 1u : 0u;
        if (hasn) {
            pat = (pat << (2 * k2 + 2)) | ((((unsigned long long)x2 << 1) | 1ull) << 1) | (pval < 0 ? 1ull : 0ull);
            len += 2u * k2 + 2u;
        }
        const unsigned incl = wave_scan_incl(len);
        tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        stg[lane] = lane ? 0u : carry;
        stg[64 + lane] = 0;
        EMIT_FENCE(__ATOMIC_RELEASE);
        __builtin_amdgcn_wave_barrier();
        if (len) or_bits_lds(stg, o0 + (incl - len), pat, (int)len);
        EMIT_FENCE(__ATOMIC_ACQ_REL);
        __builtin_amdgcn_wave_barrier();
        const int nfull = (int)((o0 + tot) >> 5);                     // <= 118
        const unsigned v0 = stg[lane], v1 = stg[64 + lane];
        S.carry = stg[nfull];
        const bool mine0 = lane < nfull && !(first_pending && lane == 0), mine1 = 64 + lane < nfull;
        DSVG_GLOBAL unsigned *d0 = mine0 ? out32 + (wbase + w0 + (unsigned)lane) : (DSVG_GLOBAL unsigned *)g_emit_dump + lane;
        DSVG_GLOBAL unsigned *d1 = mine1 ? out32 + (wbase + w0 + 64u + (unsigned)lane) : (DSVG_GLOBAL unsigned *)g_emit_dump + 64 + lane;
        *d0 = __builtin_bswap32(v0);
        *d1 = __builtin_bswap32(v1);
        if (first_pending && nfull > 0) { S.firstv = (unsigned)__builtin_amdgcn_readfirstlane((int)v0); S.first_pending = false; }
    } else {
        int l1 = 0, l2 = 0;
        unsigned long long p1 = 0, p2 = 0;
        if (valid) {
            p1 = pat_ueg(m - 1u, l1);
            if (hasn) p2 = pat_neg(pval, l2);
        }
        const unsigned len = (unsigned)(l1 + l2);
        const unsigned incl = wave_scan_incl(len);
        tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        const int nw = (int)((o0 + tot + 31) >> 5);                   // words touched, <= 193
        for (int i = lane; i < nw + 2; i += 64) stg[i] = i ? 0u : carry;
        EMIT_FENCE(__ATOMIC_RELEASE);
        __builtin_amdgcn_wave_barrier();
        const unsigned rel = o0 + (incl - len);
        if (l1) or_bits_lds(stg, rel, p1, l1);
        if (l2) or_bits_lds(stg, rel + l1, p2, l2);
        EMIT_FENCE(__ATOMIC_ACQ_REL);
        __builtin_amdgcn_wave_barrier();
        const int nfull = (int)((o0 + tot) >> 5);
        for (int i = lane; i < nfull; i += 64)
            if (!(first_pending && i == 0)) out32[wbase + w0 + (unsigned)i] = __builtin_bswap32(stg[i]);
        S.carry = stg[nfull];
        if (first_pending && nfull > 0) { S.firstv = stg[0]; S.first_pending = false; }
        ((DSVG_GLOBAL unsigned *)g_emit_dump)[lane] = 0;              // (two stores at the end of every path: see above)
        ((DSVG_GLOBAL unsigned *)g_emit_dump)[64 + lane] = 0;
    }
    EMIT_FENCE(__ATOMIC_ACQ_REL);
    __builtin_amdgcn_wave_barrier();
    S.at0 += tot;
}

// Round 4, packed chunks (collect_round_pk): a round takes 256 entries, four CONSECUTIVE ones per lane (one 16-byte load).  The
// per-round work that does not depend on the number of entries -- the wave scan of the code lengths, clearing and reading the
// stage, fences, the carried word, the loads and stores themselves -- is paid once per 256 entries instead of once per 64, a
// lane's four codes (<= 31 bits each in the common case) are joined in registers to two strings of <= 62 bits before they go
// to the stage (<= 6 LDS atomics instead of 8), and a round's complete words leave as <= 4 coalesced stores that are issued at
// the start of the NEXT round, behind that round's entries: the loop never waits for a store it has just issued.
// A round with a run or a value of 256 or more (rare outside the LL chunks, which are not packed) is done as four rounds of 64.
// a record at a wave-uniform address, read through the constant address space (scalar loads): only for data no kernel writes while this one runs
template <typename T>
static __device__ __forceinline__ T emit_sload(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "dwords");
    typedef const __attribute__((address_space(4))) unsigned *CU;
    const CU q = (CU)p;
    T v;
    unsigned *d = reinterpret_cast<unsigned *>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) d[i] = q[i];
    return v;
}
struct EmitPend { unsigned w[4]; unsigned base; int n; bool skip0; };
static __device__ __forceinline__ void emit_flush_pend(const EmitPend &P, int lane, DSVG_GLOBAL unsigned *out32)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (64 * k >= P.n) break;                                     // (wave-uniform)
        const int i = lane + 64 * k;
        if (i < P.n && !(P.skip0 && i == 0)) out32[P.base + (unsigned)i] = __builtin_bswap32(P.w[k]);
    }
}

template <bool PK>
static __device__ __forceinline__ void emit_chunk_t(const JobDev &jb, int c, int chunk, int lane, unsigned *stg)
{
#ifdef EMIT_STATS
    const unsigned long long t_in = wall_clock64();
#endif
    // the plane's and the chunk's summaries (written by k_hz_scan, the launch before): both requested at once, by scalar loads -- c and
    // chunk are wave-uniform -- instead of two vector round trips one after the other in front of the chunk's first entries
    const HzPlaneSum ps = emit_sload(jb.psum + c);
    const HzChunkSum cs = emit_sload(jb.chunks + (jb.chunk_off[c] + chunk));
    if (ps.overflow) return;
    DSVG_GLOBAL unsigned *out32 = dsvg_global(reinterpret_cast<unsigned *>(jb.bits + jb.bits_off[c]));
    const DSVG_GLOBAL int32_t *gpos = dsvg_global(jb.nzpos + jb.nz_off[c] + (size_t)chunk * HZ_CHUNK);
    const DSVG_GLOBAL int32_t *gval = dsvg_global(jb.nzval + jb.nz_off[c] + (size_t)chunk * HZ_CHUNK);
    const int nnz = cs.nnz;
    if (nnz > 0) {
        EmitState S;
        S.at0 = (unsigned)(cs.bit_off & 31);
        S.wbase = (unsigned)(cs.bit_off >> 5);
        S.carry = 0; S.firstv = 0; S.first_pending = true;
        S.cpos = cs.prev_pos; S.cval = cs.prev_val;
        const bool prev_in = cs.prev_pos >= 0;
        if (PK) {
            const int cbase = chunk * HZ_CHUNK;
            const DSVG_GLOBAL unsigned *gent = reinterpret_cast<const DSVG_GLOBAL unsigned *>(gpos);
            EmitPend P;
            P.w[0] = P.w[1] = P.w[2] = P.w[3] = 0; P.base = 0; P.n = 0; P.skip0 = false;
            // rounds of 256 while more than 128 entries remain, rounds of 64 for the rest (a sparse picture's chunk holds a few
            // dozen entries: one round of 64, one 4-byte load); the entries of a round are requested one round ahead
            uint4 nev = make_uint4(0u, 0u, 0u, 0u);
            unsigned ne1 = 0u;
            auto request = [&](int b) {
                if (nnz - b > 128) nev = dsvg_ld4(gent + (b + 4 * lane));       // (the chunk's slot holds HZ_CHUNK words: no load leaves it)
                else if (b < nnz) ne1 = dsvg_at(gent, (unsigned)min(b + lane, nnz - 1));
            };
            request(0);
            for (int base = 0; base < nnz;) {
                const uint4 ev = nev;
                const unsigned e1 = ne1;
                emit_flush_pend(P, lane, out32);
                P.n = 0;
                if (nnz - base <= 128) {
                    request(base + 64);
                    emit_round64(S, prev_in, base + lane, nnz, cbase + (int)(e1 & 0x7ffu), (int)e1 >> 16, lane, stg, out32);
                    base += 64;
                    continue;
                }
                request(base + 256);
                const int base0 = base;
                base += 256;
                const unsigned e[4] = {ev.x, ev.y, ev.z, ev.w};
                const unsigned pe3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)e[3], 0x138, 0xf, 0xf, true);      // wave_shr:1
                int pos[4], val[4];
#pragma unroll
                for (int t = 0; t < 4; t++) { pos[t] = cbase + (int)(e[t] & 0x7ffu); val[t] = (int)e[t] >> 16; }
                int ppos0 = cbase + (int)(pe3 & 0x7ffu), pval0 = (int)pe3 >> 16;
                if (lane == 0) { ppos0 = S.cpos; pval0 = S.cval; }
                unsigned m[4], mag[4];
                bool valid[4], hasn[4], neg[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int j = base0 + 4 * lane + t, pp = t ? pos[t - 1] : ppos0, pv = t ? val[t - 1] : pval0;
                    valid[t] = j < nnz; hasn[t] = valid[t] && (j > 0 || prev_in);
                    m[t] = valid[t] ? (unsigned)(pos[t] - pp) : 1u;
                    mag[t] = hasn[t] ? (unsigned)(pv < 0 ? -pv : pv) : 1u;
                    neg[t] = pv < 0;
                }
                if (__ballot((m[0] | m[1] | m[2] | m[3] | mag[0] | mag[1] | mag[2] | mag[3]) >= 256u) != 0ull) {
                    // (rare) as four rounds of 64: every lane fetches the entries again in that arrangement
#ifdef AB_EMIT_BREAK_FALLBACK              // (test of the tests: tests/test_gpu_stream.py::test_dense_chunks_... must fail with this)
                    S.carry |= 1u;
#endif
                    for (int r = 0; r < 4 && base0 + 64 * r < nnz; r++) {
                        const int j = base0 + 64 * r + lane;
                        const unsigned en = dsvg_at(gent, (unsigned)min(j, nnz - 1));
                        emit_round64(S, prev_in, j, nnz, cbase + (int)(en & 0x7ffu), (int)en >> 16, lane, stg, out32);
                    }
                    continue;
                }
                S.cpos = __builtin_amdgcn_readlane(pos[3], 63); S.cval = __builtin_amdgcn_readlane(val[3], 63);
                unsigned pat[4], len[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int k1 = 31 - __clz((int)m[t]), k2 = 31 - __clz((int)mag[t]);
                    unsigned x = ((m[t] & ((1u << k1) - 1u)) << 16) | (mag[t] & ((1u << k2) - 1u));
                    x = (x | (x << 4)) & 0x0F0F0F0Fu;
                    x = (x | (x << 2)) & 0x33333333u;
                    x = (x | (x << 1)) & 0x55555555u;
                    pat[t] = ((x >> 16) << 1) | 1u;
                    len[t] = valid[t] ? 2u * k1 + 1u : 0u;
                    if (hasn[t]) {
                        pat[t] = (pat[t] << (2 * k2 + 2)) | ((((x & 0xffffu) << 1) | 1u) << 1) | (neg[t] ? 1u : 0u);
                        len[t] += 2u * k2 + 2u;
                    }
                    if (!valid[t]) pat[t] = 0u;
                }
                const unsigned long long sa = ((unsigned long long)pat[0] << len[1]) | pat[1], sb = ((unsigned long long)pat[2] << len[3]) | pat[3];
                const unsigned la = len[0] + len[1], lb = len[2] + len[3];
                const unsigned incl = wave_scan_incl(la + lb);
                const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                const unsigned o0 = S.at0 & 31, w0 = S.at0 >> 5;
                reinterpret_cast<uint4 *>(stg)[lane] = make_uint4(lane ? 0u : S.carry, 0u, 0u, 0u);      // 256 words: <= 31 + 64 * 124 bits
                EMIT_FENCE(__ATOMIC_RELEASE);
                __builtin_amdgcn_wave_barrier();
                const unsigned rel = o0 + (incl - (la + lb));
                if (la) or_bits_lds(stg, rel, sa, (int)la);
                if (lb) or_bits_lds(stg, rel + la, sb, (int)lb);
                EMIT_FENCE(__ATOMIC_ACQ_REL);
                __builtin_amdgcn_wave_barrier();
                const int nfull = (int)((o0 + tot) >> 5);             // complete words of the round, <= 249
#pragma unroll
                for (int k = 0; k < 4; k++) P.w[k] = stg[lane + 64 * k];
                S.carry = (unsigned)__builtin_amdgcn_readfirstlane((int)stg[nfull]);
                P.base = S.wbase + w0; P.n = nfull; P.skip0 = S.first_pending;
                if (S.first_pending && nfull > 0) { S.firstv = (unsigned)__builtin_amdgcn_readfirstlane((int)P.w[0]); S.first_pending = false; }
                EMIT_FENCE(__ATOMIC_ACQ_REL);
                __builtin_amdgcn_wave_barrier();
                S.at0 += tot;
            }
            emit_flush_pend(P, lane, out32);
        } else {
            // a lane reads its own entry only -- the predecessor comes from the lane below (the last lane's of the round before
            // is carried) -- and the next round's entries are requested before this round is assembled
            int npos = dsvg_at(gpos, (unsigned)min(lane, nnz - 1)), nval = dsvg_at(gval, (unsigned)min(lane, nnz - 1));
            // (the same two stores behind the first loads as behind every round's: the loop head then waits for exactly "all
            // but the two youngest operations")
            ((DSVG_GLOBAL unsigned *)g_emit_dump)[lane] = 0;
            ((DSVG_GLOBAL unsigned *)g_emit_dump)[64 + lane] = 0;
            for (int base = 0; base < nnz; base += 64) {
                const int j = base + lane;
                const int pos = npos, val = nval;
                npos = dsvg_at(gpos, (unsigned)min(j + 64, nnz - 1)); nval = dsvg_at(gval, (unsigned)min(j + 64, nnz - 1));
                emit_round64(S, prev_in, j, nnz, pos, val, lane, stg, out32);
            }
        }
        if (lane == 0) {
            // the first and the last word of the chunk, possibly shared with the neighbouring chunks (or with each other)
            unsigned *og = reinterpret_cast<unsigned *>(jb.bits + jb.bits_off[c]);
            if (S.first_pending) { if (S.carry) atomicOr(og + S.wbase, __builtin_bswap32(S.carry)); }
            else {
                if (S.firstv) atomicOr(og + S.wbase, __builtin_bswap32(S.firstv));
                if (S.carry) atomicOr(og + (S.wbase + (S.at0 >> 5)), __builtin_bswap32(S.carry));
            }
        }
    }
    if (lane == 0 && chunk == ps.last_chunk) {       // trailing value of the plane
        int l;
        const unsigned long long p = pat_neg(cs.last_val, l);
        or_bits(reinterpret_cast<unsigned *>(jb.bits + jb.bits_off[c]), ps.total_bits - (unsigned long long)l, p, l);
    }
#ifdef EMIT_STATS
    if (lane == 0 && cs.nnz > 0) {
        const unsigned long long dt = wall_clock64() - t_in;
        const int rounds = (cs.nnz + 63) >> 6, sh = (chunk * 7 + c) & 63, k = rounds >= 8 ? 3 : 0;
        atomicAdd(&g_emit_stat[k][sh], 1ull);
        atomicAdd(&g_emit_stat[k + 1][sh], (unsigned long long)rounds);
        atomicAdd(&g_emit_stat[k + 2][sh], dt);
        atomicMax(&g_emit_stat[6][sh], dt);
        atomicAdd(&g_emit_stat[7][sh], (unsigned long long)cs.nnz);
    }
#endif
}

#ifndef EMIT_WPE
#define EMIT_WPE 6
#endif
#if EMIT_WPE
#define EMIT_WPE_ATTR __attribute__((amdgpu_waves_per_eu(EMIT_WPE, EMIT_WPE)))
#else
#define EMIT_WPE_ATTR
#endif
static __device__ __forceinline__ void emit_chunk(const JobDev &jb, int c, int chunk, int lane, unsigned *stg)
{
    if (emit_sload(&jb.chunks[jb.chunk_off[c] + chunk].packed)) emit_chunk_t<true>(jb, c, chunk, lane, stg);      // (wave-uniform)
    else emit_chunk_t<false>(jb, c, chunk, lane, stg);
}

__global__ __launch_bounds__(256) EMIT_WPE_ATTR void k_hz_emit(const JobDev *__restrict__ jobs)
{
    __shared__ __attribute__((aligned(16))) unsigned s_stage[4][EMIT_STAGE_WORDS];
    const JobDev &jb = jobs[blockIdx.y];
    int c, chunk;
    if (!flat_chunk(jb, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c, chunk)) return;
    emit_chunk(jb, c, chunk, threadIdx.x & 63, s_stage[threadIdx.x >> 6]);
}

// the same for sparse pictures, organised like k_hz_collect_list: a lane per chunk reads the entry count of its summary,
// the waves share out the chunks that have entries
template <int Q>
__global__ __launch_bounds__(256) EMIT_WPE_ATTR void k_hz_emit_list(const JobDev *__restrict__ jobs)
{
    __shared__ __attribute__((aligned(16))) unsigned s_stage[4][EMIT_STAGE_WORDS];
    const JobDev &jb = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int co0 = jb.chunk_off[0], co1 = jb.chunk_off[1], co2 = jb.chunk_off[2];
    int k = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        int c = 0, chunk = 0;
        const bool have = flat_chunk(jb, (int)blockIdx.x + (64 * q + lane) * (int)gridDim.x, c, chunk);
        const int coff = c == 0 ? co0 : (c == 1 ? co1 : co2);           // (scalar loads + select, see k_hz_collect_list)
        const bool work = have && jb.chunks[coff + chunk].nnz > 0;
        unsigned long long m = __ballot(work);
        for (; m; k++) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            if ((k & 3) != wv) continue;
            emit_chunk(jb, __builtin_amdgcn_readlane(c, l), __builtin_amdgcn_readlane(chunk, l), lane, s_stage[wv]);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// decoder: entropy PARSE on the device (hzcc_dec hzcc.c:295-435, bit reader bs.c:148-219).
// After the plane header (SEG(DC), 32-bit run count -- read by the host) the payload is a chain of interleaved
// exp-Golomb codes   U(run_1) | U(run_2) N(v_1) | U(run_3) N(v_2) | ... | U(run_n) N(v_{n-1}) | N(v_n)
// with U = k x ('0', bit) then '1' and N = U(|v|-1) + sign bit.  Where a code starts depends on every bit
// before it, so the bits are walked by a 5-state machine {U flag, U data, N flag, N data, sign} whose per-chunk
// transition maps compose associatively: one workgroup per plane, every thread owns 128 bits per pass,
//   1. per thread: the state map of its 16 bytes from a byte table in LDS (5 states in parallel);
//   2. workgroup scan of the maps -> the state each thread really enters with; one more table walk gives the
//      code-end bit mask of the chunk, its count and its last end; prefix sum / max scan of those;
//   3. each thread decodes the codes that END in its chunk (their start is the previous code end) straight into
//      the run / value arrays; the pass carries state, code count and last end to the next pass;
// then a prefix sum of (run + 1) turns runs into scan positions.  After U(run_1), which thread 0 reads serially,
// the chain alternates U,N strictly except for the last code, an N where a U would be due: its U part ends where
// the machine says, the sign bit is the next bit.
#define PARSE_THREADS 1024
#define PARSE_BITS 128
static __device__ __forceinline__ unsigned long long bits_at(const uint8_t *p, unsigned long long bitpos)
{   // 64 bits of the MSB-first stream starting at bitpos: two aligned 8-byte loads (p is 8-byte aligned and the payload
    // buffer has 64 bytes of slack)
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p) + (bitpos >> 6);
    const unsigned sh = (unsigned)(bitpos & 63);
    const unsigned long long a = __builtin_bswap64(q[0]), b = __builtin_bswap64(q[1]);
    return sh ? (a << sh) | (b >> (64 - sh)) : a;
}
// 128 bits from bitpos: three loads
static __device__ __forceinline__ void bits128_at(const uint8_t *p, unsigned long long bitpos, unsigned long long &w0, unsigned long long &w1)
{
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p) + (bitpos >> 6);
    const unsigned sh = (unsigned)(bitpos & 63);
    const unsigned long long a = __builtin_bswap64(q[0]), b = __builtin_bswap64(q[1]), c = __builtin_bswap64(q[2]);
    w0 = sh ? (a << sh) | (b >> (64 - sh)) : a;
    w1 = sh ? (b << sh) | (c >> (64 - sh)) : b;
}
static __device__ __forceinline__ unsigned compress_bits(unsigned long long x)      // bit 2i -> bit i (inverse of spread_bits)
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (unsigned)x;
}
// value of the U code occupying the top `len` (= 2k+1, <= 63) bits of w
static __device__ __forceinline__ unsigned ueg_value(unsigned long long w, int len)
{
    const int k = (len - 1) >> 1;
    if (k <= 0) return 0u;
    const unsigned long long body = w >> (64 - (len - 1));          // k pairs ('0', bit), the data bit is the low bit of each pair
    return ((1u << k) | compress_bits(body)) - 1u;
}
// the same for a code of at most 31 bits in the top bits of a 32-bit word
static __device__ __forceinline__ unsigned ueg_value32(unsigned w, int len)
{
    const int k = (len - 1) >> 1;
    if (k <= 0) return 0u;
    unsigned x = (w >> (32 - (len - 1))) & 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return ((1u << k) | x) - 1u;
}
// one step of the machine: state x bit -> state; *end = a code ends with this bit
static __device__ __forceinline__ int parse_step(int st, int b, bool &end)
{
    end = false;
    switch (st) {
    case 0: if (b) { end = true; return 2; } return 1;               // U flag
    case 1: return 0;                                                // U data
    case 2: return b ? 4 : 3;                                        // N flag
    case 3: return 2;                                                // N data
    default: end = true; return 0;                                   // sign bit
    }
}

// 128 payload bits from B with everything at or after endbits forced to zero
static __device__ __forceinline__ void parse_chunk_bits(const uint8_t *pay, long long B, long long endbits, unsigned long long &w0, unsigned long long &w1)
{
    w0 = 0; w1 = 0;
    if (B < endbits) bits128_at(pay, (unsigned long long)B, w0, w1);
    if (B + 128 > endbits) {
        const long long keep = endbits - B;                           // < 128
        if (keep <= 0) { w0 = 0; w1 = 0; }
        else if (keep < 64) { w0 &= ~0ull << (64 - keep); w1 = 0; }
        else if (keep < 128) { w1 = keep == 64 ? 0ull : (w1 & (~0ull << (128 - keep))); }
    }
}

// A. the serial part, one workgroup per (picture, plane): state maps, their scan, code-end masks; per 128-bit chunk it
// leaves {end masks, index of the first code ending in it, end of the code before} for k_hz_codes.
__global__ __launch_bounds__(PARSE_THREADS) void k_hz_parse(JobDev *__restrict__ jobs, int c0)
{
    const int c = c0 + (int)blockIdx.y;
    __shared__ uint16_t s_tab[5][256];              // [state][byte] -> exit state | (code ends << 3) | (end mask << 8)
    __shared__ unsigned s_wmap[16];                 // per-wave inclusive state maps (5 x 3 bits)
    __shared__ int s_wcnt[16], s_wlast[16];
    __shared__ int s_state, s_ncode;
    __shared__ long long s_lastend;
    JobDev &jb = jobs[blockIdx.x];
    const HzPlane &hp = jb.hz[c];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint8_t *pay = jb.bits + jb.bits_off[c];
    int32_t *R = jb.nzpos + jb.nz_off[c];           // [0] = DC position, [m] = run_m, then scan position of entry m
    int32_t *V = jb.nzval + jb.nz_off[c];
    HzParseChunk *meta = jb.dec_meta[c];
    const long long endbits = 8ll * jb.dec_len[c];
    const int cap = hp.nchunks * HZ_CHUNK - 1;
    const int n = min(jb.dec_runs[c], cap);         // entries announced by the header
    if (tid == 0) { R[0] = 0; V[0] = jb.dec_dc[c]; jb.dec_first_bad[c] = 0x7fffffff; jb.dec_npass[c] = 0; jb.dec_ncode[c] = 0; }
    if (n <= 0) return;

    for (int i = tid; i < 5 * 256; i += PARSE_THREADS) {
        int st = i >> 8, cnt = 0, mask = 0;
        const int byte = i & 255;
#pragma unroll
        for (int b = 7; b >= 0; b--) { bool e; st = parse_step(st, (byte >> b) & 1, e); cnt += e; if (e) mask |= 1 << b; }
        s_tab[i >> 8][byte] = (uint16_t)(st | (cnt << 3) | (mask << 8));
    }
    const long long start0 = jb.dec_bitpos[c];
    if (tid == 0) {                                 // U(run_1), serially
        const unsigned long long w = bits_at(pay, (unsigned long long)start0);
        const int k = w ? __clzll((long long)(w & 0xAAAAAAAAAAAAAAAAull)) >> 1 : 31;   // first '1' at an even offset = the stop flag
        const int len = 2 * k + 1;
        R[1] = (int32_t)ueg_value(w, len);
        s_state = 0; s_ncode = 0;
        s_lastend = start0 + len;
    }
    __syncthreads();
    const int ncodes = 2 * n - 1;
    const long long S0 = s_lastend;                 // first bit of the alternating chain
    int npass = 0;

    for (long long pbase = S0; pbase < endbits && s_ncode < ncodes; pbase += (long long)PARSE_THREADS * PARSE_BITS, npass++) {
        const long long B = pbase + (long long)tid * PARSE_BITS;
        unsigned long long w0, w1;
        parse_chunk_bits(pay, B, endbits, w0, w1);
        // 1. state map of the chunk: exit state for each of the 5 entry states
        unsigned map = 0;
#pragma unroll
        for (int s0 = 0; s0 < 5; s0++) {
            int st = s0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const unsigned byte = (unsigned)((k < 8 ? w0 >> (56 - 8 * k) : w1 >> (56 - 8 * (k - 8))) & 0xff);
                st = s_tab[st][byte] & 7;
            }
            map |= (unsigned)st << (3 * s0);
        }
        // inclusive scan of the maps (compose: first a, then b)
        unsigned inc = map;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned prev = (unsigned)__shfl_up((int)inc, o);
            if (lane >= o) {
                unsigned cmp = 0;
#pragma unroll
                for (int s0 = 0; s0 < 5; s0++) cmp |= ((inc >> (3 * ((prev >> (3 * s0)) & 7))) & 7) << (3 * s0);
                inc = cmp;
            }
        }
        if (lane == 63) s_wmap[wv] = inc;
        __syncthreads();
        int st_in = s_state;                                              // state at the start of the pass
        for (int w = 0; w < wv; w++) st_in = (int)((s_wmap[w] >> (3 * st_in)) & 7);
        {
            const unsigned excl = (unsigned)__shfl_up((int)inc, 1);
            if (lane > 0) st_in = (int)((excl >> (3 * st_in)) & 7);
        }
        // 2. the chunk again from its real entry state: code-end mask (bit 127-i of {m0,m1} = a code ends with chunk bit i;
        // the table gives the ends inside a byte: bit 7-j of the mask byte = a code ends with the byte's j-th bit)
        unsigned long long m0 = 0, m1 = 0;
        int st = st_in;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned e = s_tab[st][(unsigned)(w0 >> (56 - 8 * k)) & 0xff];
            st = (int)(e & 7); m0 |= (unsigned long long)(e >> 8) << (56 - 8 * k);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned e = s_tab[st][(unsigned)(w1 >> (56 - 8 * k)) & 0xff];
            st = (int)(e & 7); m1 |= (unsigned long long)(e >> 8) << (56 - 8 * k);
        }
        const int cnt = __popcll(m0) + __popcll(m1);
        const int lastoff = m1 ? 127 - (__ffsll((long long)m1) - 1) : (m0 ? 63 - (__ffsll((long long)m0) - 1) : -1);   // chunk bit of the last end
        // exclusive prefix sum of counts, exclusive max-scan of last ends
        int csum = cnt, lmax = lastoff >= 0 ? tid * PARSE_BITS + lastoff : -1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(csum, o), b = __shfl_up(lmax, o);
            if (lane >= o) { csum += a; lmax = max(lmax, b); }
        }
        if (lane == 63) { s_wcnt[wv] = csum; s_wlast[wv] = lmax; }
        __syncthreads();
        int cbase = s_ncode, lbefore = -1;
        for (int w = 0; w < wv; w++) { cbase += s_wcnt[w]; lbefore = max(lbefore, s_wlast[w]); }
        {
            const int a = __shfl_up(csum, 1), b = __shfl_up(lmax, 1);
            if (lane > 0) { cbase += a; lbefore = max(lbefore, b); }
        }
        const long long prev_end = lbefore >= 0 ? pbase + lbefore + 1 : s_lastend;    // bit after the previous code
        if (B < endbits) {
            HzParseChunk mc;
            mc.m0 = m0; mc.m1 = m1; mc.cbase = cbase; mc.prev_end = (int)prev_end;
            meta[(size_t)npass * PARSE_THREADS + tid] = mc;
        }
        __syncthreads();
        if (tid == PARSE_THREADS - 1) {                                   // carry to the next pass
            s_state = st;
            s_ncode = cbase + cnt;
            const int l = max(lbefore, lastoff >= 0 ? tid * PARSE_BITS + lastoff : -1);
            if (l >= 0) s_lastend = pbase + l + 1;
        }
        __syncthreads();
    }
    if (tid == 0) { jb.dec_npass[c] = npass; jb.dec_ncode[c] = s_ncode; jb.dec_s0[c] = S0; }
}

// B. every chunk on its own: the codes that END in it go straight into the run / value arrays.  A code is at most
// 63 bits, so it starts no earlier than the second word of the chunk before: three words of payload per thread.
__global__ __launch_bounds__(256) void k_hz_codes(JobDev *__restrict__ jobs, int c0)
{
    const int c = c0 + (int)blockIdx.z;
    JobDev &jb = jobs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jb.dec_npass[c] * PARSE_THREADS) return;
    const int dlen = jb.dec_len[c];
    const long long endbits = 8ll * dlen;
    const long long B = jb.dec_s0[c] + (long long)i * PARSE_BITS;
    if (B >= endbits) return;
    const HzPlane &hp = jb.hz[c];
    const int n = min(jb.dec_runs[c], hp.nchunks * HZ_CHUNK - 1);
    const int ncodes = 2 * n - 1;
    const uint8_t *pay = jb.bits + jb.bits_off[c];
    int32_t *R = jb.nzpos + jb.nz_off[c];
    int32_t *V = jb.nzval + jb.nz_off[c];
    const HzParseChunk mc = jb.dec_meta[c][i];
    int j = mc.cbase;
    if (j >= ncodes || !(mc.m0 | mc.m1)) return;
    unsigned long long w0, w1;
    parse_chunk_bits(pay, B, endbits, w0, w1);
    const unsigned long long wp = B >= 64 ? bits_at(pay, (unsigned long long)(B - 64)) : 0ull;
    long long prev_end = mc.prev_end;
    int first_bad = 0x7fffffff;
    for (int half = 0; half < 2 && j < ncodes; half++) {
        unsigned long long m = half ? mc.m1 : mc.m0;
        while (m && j < ncodes) {
            const int hb = 63 - __clzll((long long)m);                   // highest set bit = earliest end
            m &= ~(1ull << hb);
            const long long endp = B + half * 64 + (63 - hb) + 1;         // bit after the code
            const int len = (int)min(endp - prev_end, 63ll);
            const int rel = (int)(prev_end - (B - 64));
            unsigned long long w;
            if (rel >= 0) {
                const int idx = rel >> 6, sh = rel & 63;
                const unsigned long long wa = idx == 0 ? wp : (idx == 1 ? w0 : w1);
                const unsigned long long wb = idx == 0 ? w0 : (idx == 1 ? w1 : 0ull);
                w = sh ? (wa << sh) | (wb >> (64 - sh)) : wa;
            } else {
                w = bits_at(pay, (unsigned long long)prev_end);           // over-long code of a damaged stream
            }
            // one body for both code kinds: U(run) is all magnitude; N(value) is magnitude + sign bit, except the final N
            // of the plane, which arrives where a U was due (its U part ends here, the sign is the next bit)
            const bool last = j == ncodes - 1, isN = (j & 1) != 0;
            const int ulen = isN ? len - 1 : len;                         // bits of the U part
            const unsigned mag = ulen <= 31 ? ueg_value32((unsigned)(w >> 32), ulen) : ueg_value(w, ulen);
            int sign = (int)((w >> (64 - len)) & 1);
            if (last) sign = (int)((bits_at(pay, (unsigned long long)endp) >> 63) & 1);
            const int mi = last ? n : (j + 1) >> 1;
            const bool val = isN || last;
            int32_t *dst = val ? V + mi : R + (j >> 1) + 2;
            const int v1 = (int)mag + 1;
            *dst = val ? (sign ? -v1 : v1) : (int)mag;
            if (val && ((endp + (last ? 1 : 0)) >> 3) >= dlen) first_bad = min(first_bad, mi);   // hzcc.c:337-339
            prev_end = endp;
            j++;
        }
    }
    if (first_bad != 0x7fffffff) atomicMin(&jb.dec_first_bad[c], first_bad);
}

// C. runs -> scan positions: q_1 = run_1, q_m = q_{m-1} + 1 + run_m, one workgroup per (picture, plane).  Eight
// consecutive entries per thread and pass (two 16-byte loads; entry 0, the DC, adds nothing), wave scan of the thread
// totals, waves chained through LDS.
__global__ __launch_bounds__(PARSE_THREADS) void k_hz_positions(JobDev *__restrict__ jobs, int c0)
{
    const int c = c0 + (int)blockIdx.y;
    __shared__ unsigned long long s_q, s_w64[16];
    __shared__ int s_cnt[16];
    JobDev &jb = jobs[blockIdx.x];
    const HzPlane &hp = jb.hz[c];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t *R = jb.nzpos + jb.nz_off[c];
    const int nscan = hp.nscan;
    const int n = min(jb.dec_runs[c], hp.nchunks * HZ_CHUNK - 1);
    if (n <= 0) { if (tid == 0) jb.dec_cnt[c] = 1; return; }
    const int ncodes = 2 * n - 1, ncode = jb.dec_ncode[c], first_bad = jb.dec_first_bad[c];
    // entries actually usable: announced, fully inside the data, and (below) inside the scan
    const int nent = ncode >= ncodes ? min(n, first_bad - 1) : min(ncode / 2, first_bad - 1);   // truncated data: whole pairs only
    if (tid == 0) s_q = 0ull;
    __syncthreads();
    int count = 0;
    for (int base = 0; base <= nent; base += PARSE_THREADS * 8) {
        const int mb = base + tid * 8;
        int4 ra = make_int4(0, 0, 0, 0), rb = make_int4(0, 0, 0, 0);
        if (mb <= nent) ra = *reinterpret_cast<const int4 *>(R + mb);
        if (mb + 4 <= nent) rb = *reinterpret_cast<const int4 *>(R + mb + 4);
        const int rv[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
        unsigned long long pre[8], tot = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = mb + i;
            tot += (m >= 1 && m <= nent) ? (unsigned long long)(unsigned)rv[i] + (m > 1 ? 1ull : 0ull) : 0ull;
            pre[i] = tot;
        }
        unsigned long long inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long a = __shfl_up(inc, o);
            if (lane >= o) inc += a;
        }
        if (lane == 63) s_w64[wv] = inc;
        __syncthreads();
        unsigned long long q0 = s_q + (inc - tot);                        // positions before this thread's entries
        for (int w = 0; w < wv; w++) q0 += s_w64[w];
        int out[8], nok = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = mb + i;
            const unsigned long long q = q0 + pre[i];
            const bool ok = m >= 1 && m <= nent && q < (unsigned long long)nscan;
            out[i] = ok ? (int)q : rv[i];
            nok += ok;
        }
        if (mb + 7 <= nent) {
            *reinterpret_cast<int4 *>(R + mb) = make_int4(mb == 0 ? 0 : out[0], out[1], out[2], out[3]);
            *reinterpret_cast<int4 *>(R + mb + 4) = make_int4(out[4], out[5], out[6], out[7]);
        } else {                                                          // tail: never write past entry nent
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (mb + i >= 1 && mb + i <= nent) R[mb + i] = out[i];
        }
        // positions are increasing, so the valid entries form a prefix: counting them is enough
        int c64 = nok;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c64 += __shfl_xor(c64, o);
        if (lane == 0) s_cnt[wv] = c64;
        __syncthreads();
        for (int w = 0; w < 16; w++) count += s_cnt[w];
        if (tid == PARSE_THREADS - 1) s_q = q0 + tot;
        __syncthreads();
    }
    if (tid == 0) jb.dec_cnt[c] = 1 + count;      // positions are increasing: the valid entries are a prefix
}

// scatter of one level group of the parsed entries (the groups go in order: a cell two scan regions share takes the
// later region's value): phase -1 / 0 = LL + level 0, 1 = level 1, 2 = level 2
__global__ __launch_bounds__(256) void k_hz_scatter_lv(const JobDev *__restrict__ jobs, int c0, int phase)
{
    const int c = c0 + (int)blockIdx.z;
    const JobDev &jb = jobs[blockIdx.y];
    const HzPlane &hp = jb.hz[c];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jb.dec_cnt[c]) return;
    const int p = jb.nzpos[jb.nz_off[c] + i];
    const int v = jb.nzval[jb.nz_off[c] + i];
    if (p < 0 || p >= hp.nscan) return;
    const int ph = p >= hp.r[7].base ? 2 : (p >= hp.r[4].base ? 1 : 0);
    // phase -1 (the first launch): every entry of a sparse plane -- its symbols land in slots of their own, nothing to order --
    // and level group 0 of the others; launches 1, 2 then only serve the planes that keep int32 coefficients
    if (phase < 0 ? (!jb.dec_sym[c] && ph != 0) : (jb.dec_sym[c] || ph != phase)) return;
    if (p == 0) {                                                     // unquantised DC (hzcc.c:495)
        (jb.coef + jb.hz_coef_off[c])[0] = v;
        if (jb.dec_sym[c] && (v < -jb.dec_lim[0] || v > jb.dec_lim[0])) atomicOr(jb.dec_flag, 1);
        return;
    }
    const HzRegion r = hp.r[find_region(hp, p)];
    const int local = p - r.base;
    const int y = local / r.sw, x = local - y * r.sw;
    const int tq = cell_tq(r, jb.stable, hp.nbh, x, y);
    const int dq = dequant_any(r, v, tq);
    if (jb.dec_sym[c]) {
        // The sparse path ends in the encoder's fused inverse, whose packed int16 level 1 is exact for what an ENCODER of 8-bit
        // video can produce (range chain: tests/test_symbol_range.py).  A parsable stream may hold anything, and the reference
        // decodes it in 32 bits (hzcc.c:295-435): a value beyond the encoder's range at its level -- a symbol beyond int16
        // included -- flags the picture, and its call is decoded again from int32 coefficients (dsvg_decode_pictures).
        int lvl = 3 - r.level;                                       // scan levels 0, 1, 2 = transform levels 3, 2, 1
        if (r.level < 0) {                                           // LL region: the band of level k starts at column w_k or row h_k
            lvl = 4;
            while (lvl < 15 && x < DSVG_RSU(hp.w, lvl) && y < DSVG_RSU(hp.h, lvl)) lvl++;
        }
        if (dq < -jb.dec_lim[lvl] || dq > jb.dec_lim[lvl] || v < -32767 || v > 32767) { atomicOr(jb.dec_flag, 1); if (r.level >= 0) return; }
    }
    if (jb.dec_sym[c] && r.level >= 0) {
        // sparse decode: the SYMBOL goes to its scan slot (the fused inverse dequantises it: levels 1-3 never exist as
        // int32 coefficients), and the 8x8-pixel patch it belongs to is flagged for the inverse's symbol fetch
        (jb.sym + jb.nz_off[c])[p] = (int16_t)v;
        jb.pflag[jb.pf_off[c] + (y >> r.level) * hp.r[0].sw + (x >> r.level)] = 1;
        return;
    }
    (jb.coef + jb.hz_coef_off[c])[(size_t)(r.y0 + y) * hp.w + r.x0 + x] = dq;
}

// decoder, sparse symbol path on a plane whose scan regions share cells (960x540, 250x130): the inverse reads a
// shared cell through the LATER region's slot.  The reference decoder (hzcc.c:295-435) writes non-zero symbols only, so a
// cell whose later symbol is absent keeps the EARLIER region's dequantised value -- which no symbol of the later region's
// quantiser expresses.  Rare (the later, finer quantiser must map a value the earlier one kept to zero): such a picture is
// flagged and decoded again on the int32 coefficient path.  One thread per shared cell, geometry as hz_fix_overlaps.
__global__ __launch_bounds__(256) void k_hz_dec_resolve(const JobDev *__restrict__ jobs, int c0)
{
    const int c = c0 + (int)blockIdx.y;
    const JobDev &jb = jobs[blockIdx.x];
    if (!jb.dec_sym[c]) return;
    const HzPlane &hp = jb.hz[c];
    const int16_t *sym = jb.sym + jb.nz_off[c];
    for (int l = 0; l < 2; l++) {
        const int cw = 2 * hp.s_w[l], ch = 2 * hp.s_h[l];
        const bool col = cw > hp.s_w[l + 1], row = ch > hp.s_h[l + 1];
        const int ncol = col ? ch : 0, nrow = row ? cw : 0;
        for (int i = threadIdx.x; i < ncol + nrow; i += 256) {
            int gx, gy;
            if (i < ncol) { gx = hp.s_w[l + 1]; gy = i; }
            else {
                gx = i - ncol; gy = hp.s_h[l + 1];
                if (col && gx == hp.s_w[l + 1]) continue;
            }
            const int ex = gx >= hp.s_w[l], ey = gy >= hp.s_h[l];
            if (!(ex + ey)) continue;
            const HzRegion &e = hp.r[1 + 3 * l + (ex + 2 * ey) - 1];
            const int lx = gx >= hp.s_w[l + 1], ly = gy >= hp.s_h[l + 1];
            const HzRegion &r = hp.r[1 + 3 * (l + 1) + (lx + 2 * ly) - 1];
            const int es = sym[e.base + (gy - e.y0) * e.sw + (gx - e.x0)];
            const int ls = sym[r.base + (gy - r.y0) * r.sw + (gx - r.x0)];
            if (es != 0 && ls == 0) atomicOr(jb.dec_flag, 2);
        }
    }
}

// decoder, after the reconstruction: the symbols scattered by k_hz_scatter_lv are taken down again (the planes stay zero)
__global__ __launch_bounds__(256) void k_hz_unscatter(const JobDev *__restrict__ jobs, int c0)
{
    const int c = c0 + (int)blockIdx.z;
    const JobDev &jb = jobs[blockIdx.y];
    if (!jb.dec_sym[c]) return;
    const HzPlane &hp = jb.hz[c];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jb.dec_cnt[c]) return;
    const int p = jb.nzpos[jb.nz_off[c] + i];
    if (p >= hp.r[1].base && p < hp.nscan) (jb.sym + jb.nz_off[c])[p] = 0;
}
// decoder, before the scatter: what the scatter does not overwrite must be zero (hzcc_dec writes non-zeros only, hzcc.c:295-435).
// Sparse planes: the LL region of the coefficient plane (levels >= 4) and the patch flags; other planes: the whole plane.
__global__ __launch_bounds__(256) void k_dec_clear(const JobDev *__restrict__ jobs)
{
    const int c = (int)blockIdx.z;
    const JobDev &jb = jobs[blockIdx.y];
    const HzPlane &hp = jb.hz[c];
    int32_t *plane = jb.coef + jb.hz_coef_off[c];
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (jb.dec_sym[c]) {
        const int w3 = hp.r[0].sw, h3 = hp.r[0].sh, n = w3 * h3;
        for (int i = tid; i < n; i += nth) {
            const int y = i / w3, x = i - y * w3;
            plane[(size_t)y * hp.w + x] = 0;
            jb.pflag[jb.pf_off[c] + i] = 0;
        }
    } else {
        const size_t n4 = ((((uintptr_t)plane) & 15) == 0) ? ((size_t)hp.w * hp.h) >> 2 : 0;    // 16-byte stores where the plane allows
        int4 *p4 = reinterpret_cast<int4 *>(plane);
        for (size_t i = tid; i < n4; i += nth) p4[i] = make_int4(0, 0, 0, 0);
        for (size_t i = (n4 << 2) + tid; i < (size_t)hp.w * hp.h; i += nth) plane[i] = 0;
    }
}

// -------------------------------------------------------------------------------------------------
#define PB(kid, bytes) do { if (pf) pf->begin(st, kid, bytes); } while (0)
#define PE() do { if (pf) pf->end(st); } while (0)

void launch_dec_clear(hipStream_t st, const JobDev *jobs, int njobs)
{
    hipLaunchKernelGGL(k_dec_clear, dim3(64, njobs, 3), dim3(256), 0, st, jobs);
}
void launch_hz_dec_resolve(hipStream_t st, const JobDev *jobs, int njobs, int c0, int nplanes)
{
    hipLaunchKernelGGL(k_hz_dec_resolve, dim3(njobs, nplanes), dim3(256), 0, st, jobs, c0);
}
void launch_hz_unscatter(hipStream_t st, const JobDev *jobs, int njobs, int max_entries)
{
    if (max_entries <= 0) return;
    hipLaunchKernelGGL(k_hz_unscatter, dim3((max_entries + 255) / 256, njobs, 3), dim3(256), 0, st, jobs, 0);
}

// The entropy stage in two halves, so a pipeline can put them on different streams.
// launch_hz_quant: everything the RECONSTRUCTION depends on -- jobs [0, nplain) take the full quantiser
// (k_hz_quant<false>); jobs [nplain, njobs) were quantised by the forward transform (JobDev.fused) and only
// their LL region is quantised here.  launch_hz_pack: symbol compaction of the fused jobs, the plane-wide
// scan (which also fixes up the shared cells of non-fused jobs) and the bit emission.
// samples = coefficients per job, job_chunks = scan chunks of one job over its three planes, ll_chunks = chunks
// that reach into the largest plane's LL region.
void launch_hz_quant(hipStream_t st, const JobDev *jobs, int njobs, int job_chunks, Prof *pf, double samples, int nplain, int ll_chunks)
{
    if (nplain < 0 || nplain > njobs) nplain = njobs;
    if (nplain > 0) {
        PB(KID_HZ_QUANT, samples * nplain * 8.0);          // 4 B/sample in, 4 B/sample dequantised back
        hipLaunchKernelGGL((k_hz_quant<false>), dim3(job_chunks, nplain), dim3(256), 0, st, jobs, 0);
        PE();
    }
    if (njobs > nplain) {
        PB(KID_HZ_QUANT_LL, 0.0);
        hipLaunchKernelGGL((k_hz_quant<true>), dim3(3 * ll_chunks, njobs - nplain), dim3(256), 0, st, jobs + nplain, ll_chunks);
        PE();
    }
}
void launch_hz_pack(hipStream_t st, const JobDev *jobs, int njobs, int job_chunks, Prof *pf, double samples, int nplain, int ndense)
{
    if (nplain < 0 || nplain > njobs) nplain = njobs;
    // fused jobs [nplain, nplain + ndense) are dense (I pictures: every chunk has entries: a wave per chunk); the rest are
    // sparse (P pictures: most chunks empty: k_hz_*_list)
    if (ndense < 0 || ndense > njobs - nplain) ndense = njobs - nplain;
    const int nsparse = njobs - nplain - ndense;
    const bool bigq = (long)((job_chunks + 64 * HZ_LIST_Q - 1) / (64 * HZ_LIST_Q)) * nsparse >= HZ_LIST_Q_MIN_WGS;
    const int list_wgs = bigq ? (job_chunks + 64 * HZ_LIST_Q - 1) / (64 * HZ_LIST_Q) : (job_chunks + 63) / 64;
    if (ndense > 0) {
        PB(KID_HZ_COLLECT, samples * ndense * 2.0);             // 2 B/sample of symbols in
        hipLaunchKernelGGL(k_hz_collect, dim3((job_chunks + 3) / 4, ndense), dim3(256), 0, st, jobs + nplain);
        PE();
    }
    if (nsparse > 0) {
        PB(KID_HZ_COLLECT_LIST, 0.0);
        if (bigq) hipLaunchKernelGGL((k_hz_collect_list<HZ_LIST_Q>), dim3(list_wgs, nsparse), dim3(256), 0, st, jobs + nplain + ndense);
        else hipLaunchKernelGGL((k_hz_collect_list<1>), dim3(list_wgs, nsparse), dim3(256), 0, st, jobs + nplain + ndense);
        PE();
    }
    PB(KID_HZ_SCAN, 0.0);
    // few workgroups (small batches: a link of a latency-bound chain): 1024 threads per plane; many: 256 (see above)
    if (3 * njobs <= 96) hipLaunchKernelGGL((k_hz_scan<1024>), dim3(3, njobs), dim3(1024), 0, st, jobs);
    else hipLaunchKernelGGL((k_hz_scan<SCAN_THREADS>), dim3(3, njobs), dim3(SCAN_THREADS), 0, st, jobs);
    PE();
    if (nplain + ndense > 0) {
        PB(KID_HZ_EMIT, 0.0);
        hipLaunchKernelGGL(k_hz_emit, dim3((job_chunks + 3) / 4, nplain + ndense), dim3(256), 0, st, jobs);
        PE();
    }
    if (nsparse > 0) {
        PB(KID_HZ_EMIT_LIST, 0.0);
        if (bigq) hipLaunchKernelGGL((k_hz_emit_list<HZ_LIST_Q>), dim3(list_wgs, nsparse), dim3(256), 0, st, jobs + nplain + ndense);
        else hipLaunchKernelGGL((k_hz_emit_list<1>), dim3(list_wgs, nsparse), dim3(256), 0, st, jobs + nplain + ndense);
        PE();
    }
}
void launch_hz_encode(hipStream_t st, const JobDev *jobs, int njobs, int job_chunks, Prof *pf, double samples, int nplain,
                      int ll_chunks)
{
    launch_hz_quant(st, jobs, njobs, job_chunks, pf, samples, nplain, ll_chunks);
    launch_hz_pack(st, jobs, njobs, job_chunks, pf, samples, nplain);
}

// decoder: parse the uploaded plane payloads of `njobs` jobs (plane c) and scatter the entries; max_entries = largest
// announced run count + 1 over the jobs (sizes the scatter grids)
// planes [c, c + nplanes) of every job: the serial scan (a workgroup per picture and plane), the code decode spread over
// the chip (max_chunks = most 128-bit chunks any of the payloads has), the run -> position pass, then the three ordered
// scatter phases; max_entries = the largest entry count announced by any of those planes
void launch_hz_parse_scatter(hipStream_t st, JobDev *jobs, int njobs, int c, int nplanes, int max_entries, int max_chunks, Prof *pf, bool all_sparse)
{
    PB(KID_HZ_PARSE, 0.0);
    hipLaunchKernelGGL(k_hz_parse, dim3(njobs, nplanes), dim3(PARSE_THREADS), 0, st, jobs, c);
    PE();
    if (max_chunks > 0) {
        PB(KID_HZ_CODES, 0.0);
        hipLaunchKernelGGL(k_hz_codes, dim3((max_chunks + 255) / 256, njobs, nplanes), dim3(256), 0, st, jobs, c);
        PE();
    }
    PB(KID_HZ_POSITIONS, 0.0);
    hipLaunchKernelGGL(k_hz_positions, dim3(njobs, nplanes), dim3(PARSE_THREADS), 0, st, jobs, c);
    PE();
    if (max_entries <= 0) return;
    // all_sparse: every plane of every job of the call takes the symbol path (a call of P pictures): one launch instead of three
    for (int ph = -1; ph < (all_sparse ? 0 : 3); ph += (ph < 0 ? 2 : 1)) {
        PB(KID_HZ_SCATTER, 0.0);
        hipLaunchKernelGGL(k_hz_scatter_lv, dim3((max_entries + 255) / 256, njobs, nplanes), dim3(256), 0, st, jobs, c, ph);
        PE();
    }
}

void launch_gather_bits(hipStream_t st, const uint8_t *bits, const unsigned long long *tab, int nitems, uint8_t *dst)
{
    if (nitems <= 0) return;
    hipLaunchKernelGGL(k_gather_bits, dim3(8, nitems), dim3(256), 0, st, bits, (size_t)0, tab, dst);
}

int hz_scan_items_max() { return 1 << 22; }     // the scan walks a plane in tiles: no practical limit on its chunks
