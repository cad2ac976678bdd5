// k_hme.hip -- hierarchical motion estimation + level-0 mode decision for gfx950 (MI355X).
//
// Replaces dsv_hme / refine_level (hme.c:378-741).  One launch per pyramid level (coarse -> fine), ONE WAVE
// (64-thread workgroup) per visited block and frame pair in an XCD-aware order: 16 column groups (4 px, one dword)
// x 4 row groups, every lane owning a run of adjacent rows.  No cross-wave exchange, 2.4 KB LDS, VALU-issue bound.
//   * the source block lives in registers (one dword per lane-row); every SAD is v_alignbyte_b32 (re-align
//     the reference dwords) + v_sad_u8 (4 pixels per instruction);
//   * inherited candidates and the rows of the 9-point +-1 search are read straight from global memory in
//     back-to-back batches (a memory round trip per batch, not per row); each reference row of the +-1 search
//     serves the three source rows around it;
//   * reductions are DPP wave sums; parents are fetched wave-uniformly and de-duplicated in registers;
//   * every decision that depends on candidate ORDER (first minimum wins, hme.c:503-506,531-534,
//     572-575; the last candidate is the fallback, hme.c:482-509) is evaluated identically by all lanes
//     from the reduced sums, in the reference's order.
// Level 0 adds: the 32x32 half-pel lattice of the centred 16x16 reference patch (hpel hme.c:350-376),
// the 8-point half-pel search on the 14x14 window (hme.c:551-591), the block statistics with 32-bit
// unsigned wrap-around (hme.c:181-300; v_dot4_u32_u8 for the sums of squares), the intra tests
// (hme.c:652-682), the representability veto (hme.c:147-179) and the 4-quadrant vote (hme.c:89-134,
// 689-716).  high_detail needs the left/top/top-left neighbours' final flags (hme.c:621-648) and is
// resolved by the second tiny kernel k_hme_detail.
#include <type_traits>
#include "dsvg_dev.hpp"
#include "dsvg_kernels.hpp"
#include "dsvg_host.hpp"

// which row-load sites of the full-block body take global_load's SGPR-base form (see lane_ro): 1 source block, 2 inherited
// candidates, 4 the +-1 search, 8 zero-motion rows, 16 chroma blocks.  Measured per 160-GOP step: none 4.35 ms, 2+4: 4.30,
// 2+4+8: 4.29, all: 4.35 (the source and chroma loads sit at the head of a dependency chain: scalar row steps in front of
// each of them cost what the vector adds saved)
#ifndef HME_SADDR_SITES
#define HME_SADDR_SITES 14
#endif
#define HME_LRO_BARRIER(site, x) do { if ((HME_SADDR_SITES) & (site)) asm volatile("" : "+v"(x)); } while (0)
// diagnostic builds (DSVG_CLOCK_PROBE): shader-clock stamps at the stage boundaries of a level-0 block, summed per stage
#ifdef DSVG_CLOCK_PROBE
#define HME_MARK(i) do { if (LEVEL0) clk_m_[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#elif defined(AB_HME_ASMMARK)              // region markers in the assembly listing (static instruction counts per stage: tools/ab/hme_asm_regions.py)
#define HME_MARK(i) asm volatile("; HMEMARK " #i)
#else
#define HME_MARK(i) do { } while (0)
#endif
#define NT 64              // threads per block of the picture = ONE wave: 16 column groups x 4 row groups, no cross-wave exchange
// Waves per workgroup: each wave takes a block of its own, the workgroup HME_WPG horizontally adjacent ones.  The waves never
// talk to each other -- they share the CU's L1: a 64-pixel block row is half a 128-byte line and a chroma block row a quarter,
// so neighbours in separate workgroups (separate CUs) each pull the whole line out of L2.
#ifndef HME_WPG
#define HME_WPG 4
#endif
#ifndef HME_PG
// frame pairs that walk the block grid together (see k_hme_level).  1: the waves of a workgroup take horizontally adjacent blocks of ONE
// pair (they share half lines of the source and reference rows); HME_WPG: consecutive pairs at one block position (they share frames).
// Pairs together were ahead while every block fetched the chroma blocks and the zero-motion rows of both frames (4.21 against 4.28 ms);
// since those come from k_hme_csum's table and the zero vector's rows, neighbours are: 6.43-6.51 against 6.59-6.73 ms per 320-GOP step,
// the upper levels 1.05-1.07 against 1.14-1.16
#define HME_PG 1
#endif
#define HME_TID ((int)(threadIdx.x & 63u))
// LDS hand-over inside ONE wave (its DS instructions execute in order): nothing for the hardware to wait for, the compiler
// must not move accesses across
static __device__ __forceinline__ void hme_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define NRG (NT / 16)
#define NK (64 / NRG)       // rows per thread
#define NW (NT / 64)        // waves per workgroup
#define WIN 14
#define LAT 32
#define SP 64              // pitch of the source block in LDS (bytes)
#define RP 72              // pitch of reference windows in LDS (bytes, 18 dwords)
#define RROWS 67
// the +-1 search of a full block stages its window in LDS with a few wide loads (lanes = 16-byte pieces of whole rows) instead of
// fourteen 12-byte loads per lane straight from memory: the kernel pays per vector-memory instruction
#ifndef HME_NINE_LDS
#define HME_NINE_LDS 1
#endif
#define NINE_P 20          // dwords per staged row: 66 window bytes + up to 3 of misalignment, in five 16-byte pieces
#define NINE_ROWS 66       // 4 * 16 + 2

static __device__ __forceinline__ int tap4(int m, int a, int b, int p) { return 9 * (a + b) - (m + p); }
#ifdef AB_HME_COUNT          // diagnostic build: how many full level-0 blocks take the union window, and what the others fail on
__device__ unsigned long long g_hme_cnt[8];
extern "C" void dsvg_hme_counts(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hme_cnt), sizeof(g_hme_cnt)); }
#define HME_COUNT(i) do { if (tid == 0) atomicAdd(&g_hme_cnt[i], 1ull); } while (0)
#else
#define HME_COUNT(i) do { } while (0)
#endif

// stage rows [oy,oy+nh) x cols [ox,ox+nw) of a plane into LDS (pitch P) as ALIGNED dwords; returns the
// byte shift `mis` such that dst[r*P + mis + k] == plane(ox+k, oy+r).  TPR threads share a row (the window
// must span <= TPR dwords), NT/TPR rows are in flight per pass.
template <int TPR, int MAXROWS>
static __device__ __forceinline__ int load_win(uint8_t *dst, int P, const uint8_t *plane, int stride,
                                               int ox, int oy, int nw, int nh)
{
    constexpr int RPP = NT / TPR;                       // rows per pass
    constexpr int NIT = (MAXROWS + RPP - 1) / RPP;
    const uint8_t *g0 = plane + (long)oy * stride + ox;
    const int mis = (int)(((uintptr_t)g0) & 3);
    const int ndw = (mis + nw + 3) >> 2;
    const int d = HME_TID & (TPR - 1), rb = HME_TID / TPR;
    if (d < ndw) {
        // loads are issued in batches of up to 9 before their stores: a memory round trip per batch, not per row
        constexpr int BATCH = 9;
#pragma unroll
        for (int b0 = 0; b0 < NIT; b0 += BATCH) {
            unsigned v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (b0 + u < NIT) {
                    const int r = min(rb + (b0 + u) * RPP, nh - 1);
                    v[u] = *reinterpret_cast<const unsigned *>(g0 - mis + (long)r * stride + 4 * d);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < BATCH; u++) {
                if (b0 + u < NIT) {
                    const int r = rb + (b0 + u) * RPP;
                    if (r < nh) *reinterpret_cast<unsigned *>(dst + r * P + 4 * d) = v[u];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return mis;
}

// unaligned dword from global memory: bytes p[0..3]
static __device__ __forceinline__ unsigned ldg_u32_unaligned(const uint8_t *p)
{
    const unsigned sh = (unsigned)(((uintptr_t)p) & 3);
    const unsigned *a = reinterpret_cast<const unsigned *>(p - sh);
    const unsigned lo = a[0], hi = a[1];
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// Round 4 -- ONE reference window for a level-0 block's inherited candidates, its +-1 search and its half-pel patch.  A wave's life was
// a chain of dependent memory round trips: source + parents -> candidate rows -> +-1 window -> half-pel patch (-> full-pel window again
// when no half-pel candidate won); by ablation the +-1 search alone cost 44 % of the kernel for 17 % of its instructions, the
// candidates 23 %.  The inherited candidates of a block are the parents' vectors: they lie close together.  When the non-zero ones fit
// a box of UW_SPX x UW_SPY pixels, the union of their block footprints plus one pixel all round is staged in LDS by six 16-byte loads
// per lane (instead of 12 loads per candidate + 5 for the +-1 window + 3 for the patch), the zero vector's rows come with the source
// rows in the very first round trip (they are dword aligned and double as the zero-motion block of the statistics), and every later
// stage reads LDS: two round trips per block instead of four or five.  Blocks whose candidates spread further keep the old path.
#ifndef HME_UNION
#define HME_UNION 1
#endif
#ifndef HME_UNION_UPPER
#define HME_UNION_UPPER 0
#endif
// Row pitch: 20 dwords = 80 bytes = 66 + spread (<= 8) + misalignment (<= 3), five 16-byte pieces.  The four row groups of a wave read rows
// 12 apart: 12 * 20 dwords = 16 banks of 32 apart -- conflict-free per half wave, like the +-1 search's own window of round 3; a pitch of
// 24 (spreads up to 24 pixels) put all four groups on the same banks (12 * 24 = 0 mod 32) and cost more than the wider boxes brought
#ifndef UW_P
#define UW_P 20
#endif
#ifndef UW_SPX
#define UW_SPX (4 * UW_P - 72)
#endif
#define UW_PPR (UW_P / 4)                                    // 16-byte pieces per staged row
#define UW_PDIV(p) (((p) * ((262144 + UW_PPR - 1) / UW_PPR)) >> 18)     // p / UW_PPR for p < 1024
// (vertical spread, measured per 320-GOP step: 14 rows 5.57-5.64 ms, 8: 5.41-5.45, 5: 5.44-5.46, 3: 5.46-5.53, 2: 5.44-5.52, 0: 5.84-5.89)
#ifndef UW_SPY
#define UW_SPY 8
#endif
#define UW_ROWS(NKB) (4 * (NKB) + 2 + UW_SPY)
template <int NKB>
struct HmeSharedT {
    // half-pel stage: the 19x20 reference patch (or the full-pel 14x14 window), the unrounded horizontal taps of its rows
    // (int16, pitch 16) and the three half-pel sample planes of the search (bytes, pitch 16): horizontal, vertical, diagonal
    union {
        // the union window (see above), dead once the half-pel patch has been taken out of it
        __attribute__((aligned(16))) unsigned win[(NKB > 0 && HME_UNION) ? UW_ROWS(NKB) * UW_P : 4];
        // the +-1 search's reference window (full blocks: 4 * NKB + 2 rows of 20 dwords), dead before the half-pel stage begins
        __attribute__((aligned(16))) unsigned nine[(NKB > 0 ? 4 * NKB + 2 : NINE_ROWS) * NINE_P];
        struct {
            __attribute__((aligned(16))) uint8_t patch[20 * 24];
            __attribute__((aligned(16))) short h16[20 * 16];
            __attribute__((aligned(16))) uint8_t hb[14 * 16 + 16];
            __attribute__((aligned(16))) uint8_t vb[16 * 16];
            __attribute__((aligned(16))) uint8_t db[16 * 16 + 16];
        } hp;
    } u;
    __attribute__((aligned(16))) uint8_t swin[WIN * 24];
    __attribute__((aligned(16))) uint8_t rwin[WIN * 16];
    unsigned part[2][NW][14];
};

// sum N values over the workgroup; every thread receives all totals (wave-uniform).  `part` is double
// buffered by the caller-maintained phase bit, so one barrier per reduction suffices.
template <int N>
static __device__ __forceinline__ void block_sum_n(unsigned (&v)[N], unsigned (*part)[NW][14], int &phase)
{
    // Four values per pass (gfx950 v_permlane32_swap / v_permlane16_swap, checked in tools/ubench/permlane_swap.hip):
    // swap32 + add folds the two 32-lane halves of a PAIR of values into one register (lanes 0-31: first value,
    // 32-63: second); swap16 + add folds two such registers into one whose four 16-lane rows hold the values
    // a, c, b, d; four DPP steps finish the rows; one v_readlane per value.  14 VALU per 4 values instead of 32.
#pragma unroll
    for (int i = 0; i + 1 < N; i += 4) {
        unsigned a = v[i], b = v[i + 1], c = i + 2 < N ? v[i + 2] : 0u, d = i + 3 < N ? v[i + 3] : 0u;
        const auto p = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        const auto q = __builtin_amdgcn_permlane32_swap(c, d, false, false);
        const auto r = __builtin_amdgcn_permlane16_swap(p[0] + p[1], q[0] + q[1], false, false);
        unsigned t = r[0] + r[1];
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xf, 0xf, true);
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, true);
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x141, 0xf, 0xf, true);
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x140, 0xf, 0xf, true);
        v[i] = (unsigned)__builtin_amdgcn_readlane((int)t, 0);
        v[i + 1] = (unsigned)__builtin_amdgcn_readlane((int)t, 32);
        if (i + 2 < N) v[i + 2] = (unsigned)__builtin_amdgcn_readlane((int)t, 16);
        if (i + 3 < N) v[i + 3] = (unsigned)__builtin_amdgcn_readlane((int)t, 48);
    }
    if (N % 4 == 1) {       // a single left-over value: the plain DPP chain
        unsigned t = v[N - 1];
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xf, 0xf, true);
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, true);
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x141, 0xf, 0xf, true);
        t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x140, 0xf, 0xf, true);
        v[N - 1] = (unsigned)__builtin_amdgcn_readlane((int)t, 0) + (unsigned)__builtin_amdgcn_readlane((int)t, 16) +
                   (unsigned)__builtin_amdgcn_readlane((int)t, 32) + (unsigned)__builtin_amdgcn_readlane((int)t, 48);
    }
    if (NW == 1) return;
    unsigned (*pb)[14] = part[phase];
    phase ^= 1;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) pb[threadIdx.x >> 6][i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; i++) {
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) t += pb[w][i];
        v[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    }
}

// n / d for 32-bit unsigned n and d < 2^13 with rd = 1.0 / d: the double product is within 2^-19 of the true
// quotient and a non-integer quotient is >= 2^-13 away from the next integer, so the 2^-16 bias makes
// the truncation exact.
static __device__ __forceinline__ unsigned udiv_rd(unsigned n, double rd)
{
    return (unsigned)__builtin_fma((double)n, rd, 0x1p-16);
}

static __device__ __forceinline__ int frame_invalid(int fw, int fh, int x, int y, int w, int h)   // invalid_block
{
    const int b = DSVG_BORDER;
    return x < -b || y < -b || x + w > fw + b || y + h > fh + b;
}

static __device__ const int FP_X[9] = {0, 1, -1, 0, 0, -1, 1, -1, 1};
static __device__ const int FP_Y[9] = {0, 0, 0, 1, -1, -1, -1, 1, 1};
static __device__ const int HP_X[8] = {1, -1, 0, 0, -1, 1, -1, 1};
static __device__ const int HP_Y[8] = {0, 0, 1, -1, -1, -1, 1, 1};

// gradient / moment partial sums of a 14x14 byte window in LDS (pitch P), one pixel per thread (tid < 196)
// gradient / moment partial sums of a 14x14 byte window in LDS (pitch P bytes, first pixel at byte `mis` of the
// row): one dword (4 px) per lane -- lane = 4 * row + dword -- with v_sad_u8 / v_dot4 instead of per-pixel loops
static __device__ __forceinline__ void win_partial(const uint8_t *p, int P, int mis, unsigned &gh, unsigned &gv, unsigned &s1, unsigned &s2)
{
    gh = gv = s1 = s2 = 0;
    const int r = HME_TID >> 2, d = HME_TID & 3;
    if (HME_TID < 4 * WIN) {
        const unsigned sh = (unsigned)(mis & 3);
        const unsigned *w = reinterpret_cast<const unsigned *>(p + r * P) + (mis >> 2) + d;
        const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
        const unsigned m = d == 3 ? 0xffffu : 0xffffffffu;          // pixels 12,13 only in the last dword
        const unsigned pm = d == 3 ? 0xffu : 0xffffffffu;           // pairs (x, x+1) exist while x + 1 < 14
        const unsigned cur = __builtin_amdgcn_alignbyte(w1, w0, sh);
        const unsigned nxt = __builtin_amdgcn_alignbyte(w2, w1, sh);
        const unsigned right = __builtin_amdgcn_alignbyte(nxt, cur, 1u);
        gh = __builtin_amdgcn_sad_u8(cur & pm, right & pm, 0u);
        if (r > 0) {
            const unsigned *u = reinterpret_cast<const unsigned *>(p + (r - 1) * P) + (mis >> 2) + d;
            const unsigned up = __builtin_amdgcn_alignbyte(u[1], u[0], sh);
            gv = __builtin_amdgcn_sad_u8(cur & m, up & m, 0u);
        }
        s1 = __builtin_amdgcn_sad_u8(cur & m, 0u, 0u);
        s2 = __builtin_amdgcn_udot4(cur & m, cur & m, 0u, false);
    }
}

// The work of one block.  NKB > 0: the block is FULL (bw == 64: all sixteen column groups, bh == 4 * NKB: every lane owns
// exactly NKB rows) -- row counts, masks and "does this row exist" are compile-time facts, so the search loops carry no
// per-row scalar state (the generic form keeps ~40 row offsets and row-exists masks in SGPRs, which the compiler spills
// to VGPR lanes and reads back with two v_readlane per row).  NKB == 0: any block (edges, small frames).
template <bool LEVEL0, int NKB, typename SHARED>
static __device__ __forceinline__ void hme_block(const HmeArgs &A, int level, int pair, int i, int j, SHARED &S)
{
    constexpr bool FAST = NKB > 0;
    // the union-window path (HmeSharedT): full blocks of level 0 (on the upper levels -- HME_UNION_UPPER -- it is parity-clean and changes
    // nothing: 1.18 against 1.12-1.19 ms per 320-GOP step)
    constexpr bool UNI = FAST && (LEVEL0 || HME_UNION_UPPER != 0) && HME_UNION != 0;
    constexpr bool ZE = UNI && LEVEL0;                         // ... and the zero vector's rows in the first round trip (level 0: they are the statistics' rows too)
#ifdef DSVG_CLOCK_PROBE
    unsigned clk_m_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    HME_MARK(0);
    constexpr int NKR = FAST ? NKB : NK;                // rows a lane may own
    const int tid = HME_TID;
    const int cg = tid & 15, rg = tid >> 4;             // column group (4 px) / row group
    const int step = 1 << level;
    const FrameLayout &L = A.L[level];
    const int fw = L.w[0], fh = L.h[0];
    int stride = L.stride[0];
    const int BW = A.blk_w, BH = A.blk_h;
    const int bx = (i * BW) >> level, by = (j * BH) >> level;
    // (the slot tables were written by the host before the launch: read through the constant address space they come by scalar
    // loads -- as generic pointers the compiler took the vector path for them, a memory round trip in front of the block's first rows)
    typedef const __attribute__((address_space(4))) int *HmeCInt;
    const int cur = ((HmeCInt)A.cur_slots)[pair], rf = ((HmeCInt)A.ref_slots)[pair];
    const uint8_t *sp = A.slab[level] + (size_t)cur * L.pitch + L.off[0];
    const uint8_t *rp = A.slab[level] + (size_t)rf * L.pitch + L.off[0];
    const int bw = FAST ? 64 : min(max(fw - bx, 0), BW), bh = FAST ? 4 * NKB : min(max(fh - by, 0), BH);
    int sstride = stride;                               // the SOURCE frame's row stride (its rows are read inside the picture only)
    if constexpr (LEVEL0) {
        // Round 5: frames whose luma stays in the caller's clip (HmeArgs.slot_y; of their bordered copies only a ring exists).  The block's
        // SOURCE rows come from the clip whenever the frame has it there (row stride = the picture's width).  Its REFERENCE rows do when
        // the reference frame has it too and the block lies `deep_r` pixels or more from every edge: it can then touch no pixel outside the
        // picture -- a level-0 vector is at most 2^(levels+1) - 1 full pixels long (a level adds +-1 of its own pixels to twice its parent's
        // vector, hme.c:452-541), the nine-point search, the half-pel lattice and the windows' slack add a few.  Every other block reads the
        // reference's bordered copy, whose ring reaches a block and twice deep_r in from each edge.  One stride per frame for the whole block,
        // so every site below is as it was.  Wave-uniform: a wave is a block.
        if (A.slot_y) {
            typedef const __attribute__((address_space(4))) unsigned long long *HmeCU64;
            const unsigned long long cy = ((HmeCU64)A.slot_y)[cur], ry = ((HmeCU64)A.slot_y)[rf];
            const int R = A.deep_r;
            // (the 14x14 statistics window around the block's centre leaves the picture beside a narrow edge block: those blocks keep the
            // bordered copy, whose ring holds them)
            const int wx0 = bx + (bw >> 1) - WIN / 2, wy0 = by + (bh >> 1) - WIN / 2;
            if (cy && wx0 >= 0 && wy0 >= 0 && wx0 + WIN <= fw && wy0 + WIN <= fh) { sp = reinterpret_cast<const uint8_t *>(cy); sstride = fw; }
            if (ry && bx >= R && by >= R && bx + bw + R <= fw && by + bh + R <= fh) { rp = reinterpret_cast<const uint8_t *>(ry); stride = fw; }
        }
    }
    DMV *mf = A.mvf + ((size_t)pair * (A.levels + 1) + level) * A.nblk;
    const DMV *parent = level < A.levels ? A.mvf + ((size_t)pair * (A.levels + 1) + level + 1) * A.nblk : nullptr;

    // this thread's pixels: columns 4cg..4cg+3 of the nkb ADJACENT rows rg*nkb .. rg*nkb+nkb-1 (adjacent rows let the
    // +-1 search reuse every reference row for three source rows)
    const int nkb = FAST ? NKB : (bh + NRG - 1) / NRG;
    const int r0 = rg * nkb;
#define ROWOK(k) (FAST || ((k) < nkb && r0 + (k) < bh))
    const bool uni = FAST || (bh % NRG) == 0;           // every lane owns exactly nkb rows
    const int xcol = 4 * cg;
    // full blocks: a row load is (wave-uniform row base in SGPRs) + (this lane's constant offset: its first row and column) --
    // global_load's SGPR-base form, the row step is two scalar adds instead of a 64-bit vector add per row and lane (the
    // launcher takes this body only when the stride is a multiple of 4: the misalignment of a row is then wave-uniform too).
    // The compiler picks that form only when it sees the offset's zero extension in the block of the load: each site
    // takes its copy through an empty asm, or the hoisted 64-bit pair ends in a v_lshl_add_u64 per row again.
    const unsigned lane_ro = (unsigned)(r0 * stride + xcol);
    const unsigned cmask = FAST ? 0xffffffffu : (xcol >= bw ? 0u : (xcol + 4 <= bw ? 0xffffffffu : ((1u << (8 * (bw - xcol))) - 1u)));
    // level 0: the parents' vectors (one per lane, lanes 0..4) are requested BEFORE the block's rows: memory answers in order, so the
    // candidates can be set up and the union window requested while the source rows are still on their way (behind the rows, waiting
    // for the vectors meant waiting for all 25 loads of the block's first round trip)
    int par_early = 0;
    if constexpr (LEVEL0) {
        if (parent) {
            const unsigned pmask = ~(unsigned)((step << 1) - 1);
            const int pi = (int)((unsigned)i & pmask), pj = (int)((unsigned)j & pmask);
            const int m = tid;
            const int ox = m == 1 ? -2 : (m == 2 ? 2 : 0), oy = m == 3 ? -2 : (m == 4 ? 2 : 0);
            const int x = pi + ox * step, y = pj + oy * step;
            if (m < 5 && x >= 0 && x < A.nxb && y >= 0 && y < A.nyb)
                par_early = *reinterpret_cast<const DSVG_GLOBAL int *>(dsvg_global(parent) + (x + y * A.nxb));      // DMV starts with int16 x, y: one dword = x | y << 16
        }
    }
    unsigned srcw[NKR];
    if constexpr (FAST) {
        auto sq = dsvg_global(sp + (long)by * sstride + bx);
        unsigned lro = (unsigned)(r0 * sstride + xcol);          // (lane_ro with the source frame's stride)
        HME_LRO_BARRIER(1, lro);
#pragma unroll
        for (int k = 0; k < NKR; k++) { srcw[k] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(sq + lro); sq += sstride; }
    } else {
#pragma unroll
        for (int k = 0; k < NKR; k++) {
            const int r = r0 + k;
            srcw[k] = 0;
            if (cmask && ROWOK(k)) srcw[k] = *reinterpret_cast<const unsigned *>(sp + (size_t)(by + r) * sstride + bx + xcol);
            srcw[k] &= cmask;
        }
    }
    // UNI: the zero vector's reference rows = the co-located block (bx + 4cg is dword aligned): requested with the source rows, in the
    // block's FIRST round trip -- they are the zero candidate's rows, the zero-motion block of the statistics (hme.c:181-300) and of
    // the veto / quadrant votes, and they depend on no decision
    unsigned zw[NKR];
    if constexpr (ZE) {
        auto zq = dsvg_global(rp + (long)by * stride + bx);
        unsigned lro = lane_ro;
        HME_LRO_BARRIER(8, lro);
#pragma unroll
        for (int k = 0; k < NKR; k++) { zw[k] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(zq + lro); zq += stride; }
    }
    // parents (hme.c:452-480): every lane reads the same five vectors (wave-uniform addresses -> SGPRs) and
    // de-duplicates them in registers in the reference's order -- no LDS hand-off, no barrier
    int cand[6] = {0, 0, 0, 0, 0, 0};
    int n = 1;                                          // cand[0] = the zero vector
    if (parent) {
        const unsigned pmask = ~(unsigned)((step << 1) - 1);
        const int pi = (int)((unsigned)i & pmask), pj = (int)((unsigned)j & pmask);
        // Lanes 0..4 fetch one parent each (a neighbour outside the grid counts as the zero vector) and de-duplicate among themselves:
        // a vector is new iff it is non-zero and differs from the (up to four) parents before it -- the same set, in the same order,
        // as comparing against the accepted candidates one by one.  The scalar form of this (five s_load, a chain of scalar compares
        // and selects per parent) was ~230 of the kernel's ~970 scalar instructions per block, and the kernel pays for scalar
        // instructions what it pays for vector ones (profiles/r04_hme_issue_probe.txt)
#ifdef HME_SCALAR_PARENTS
        int par[5];
        bool pok[5];
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const int ox = m == 1 ? -2 : (m == 2 ? 2 : 0), oy = m == 3 ? -2 : (m == 4 ? 2 : 0);
            const int x = pi + ox * step, y = pj + oy * step;
            pok[m] = x >= 0 && x < A.nxb && y >= 0 && y < A.nyb;
            const int idx = min(max(x, 0), A.nxb - 1) + min(max(y, 0), A.nyb - 1) * A.nxb;      // DMV starts with int16 x, y: one dword = x | y << 16
            par[m] = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(parent + idx));
        }
#pragma unroll
        for (int m = 0; m < 5; m++) par[m] = pok[m] ? par[m] : 0;
#pragma unroll
        for (int m = 0; m < 5; m++) {
            const int all = par[m];
            bool dup = (all == 0);
#pragma unroll
            for (int k = 1; k < 6; k++) dup |= (cand[k] == all);   // unused slots hold 0 and all != 0 here
            if (!dup) {
#pragma unroll
                for (int k = 1; k < 6; k++) cand[k] = (k == n) ? all : cand[k];
                n++;
            }
        }
#else
        int par_l = par_early;
        if constexpr (!LEVEL0) {
            // the upper levels' blocks are small and wait for their parents: those keep the scalar fetches (the scalar cache answers
            // faster than the vector path: 1.14 against 1.43 ms per 320-GOP step), only the de-duplication moves to the lanes
            int par[5];
#pragma unroll
            for (int m = 0; m < 5; m++) {
                const int ox = m == 1 ? -2 : (m == 2 ? 2 : 0), oy = m == 3 ? -2 : (m == 4 ? 2 : 0);
                const int x = pi + ox * step, y = pj + oy * step;
                const bool ok = x >= 0 && x < A.nxb && y >= 0 && y < A.nyb;
                const int idx = min(max(x, 0), A.nxb - 1) + min(max(y, 0), A.nyb - 1) * A.nxb;
                par[m] = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(parent + idx));
                par[m] = ok ? par[m] : 0;
            }
            par_l = tid == 0 ? par[0] : (tid == 1 ? par[1] : (tid == 2 ? par[2] : (tid == 3 ? par[3] : (tid == 4 ? par[4] : 0))));
        }
        const int s1 = __builtin_amdgcn_update_dpp(0, par_l, 0x111, 0xf, 0xf, true), s2 = __builtin_amdgcn_update_dpp(0, par_l, 0x112, 0xf, 0xf, true);      // row_shr:1..4,
        const int s3 = __builtin_amdgcn_update_dpp(0, par_l, 0x113, 0xf, 0xf, true), s4 = __builtin_amdgcn_update_dpp(0, par_l, 0x114, 0xf, 0xf, true);      // zero shifted in
        unsigned um = (unsigned)__ballot(par_l != 0 && par_l != s1 && par_l != s2 && par_l != s3 && par_l != s4) & 0x1fu;
        n = 1 + __popc(um);
#pragma unroll
        for (int k = 1; k < 6; k++) {
            const int l = __ffs((int)um) - 1;
            cand[k] = um ? __builtin_amdgcn_readlane(par_l, l & 63) : 0;
            um &= um - 1u;
        }
#endif
    }
    int phase = 0;
#ifdef AB_HME_DUMMY_SALU        // sensitivity probes (timing only): N extra scalar / vector instructions per block
    { int ds_ = level;
#pragma unroll
      for (int u = 0; u < AB_HME_DUMMY_SALU; u++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ds_));
      if (ds_ == 0x7fffffff) return; }
#endif
#ifdef AB_HME_DUMMY_VALU
    { unsigned dv_ = srcw[0];
#pragma unroll
      for (int u = 0; u < AB_HME_DUMMY_VALU; u++) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(dv_) : "v"(srcw[1]));
      if (dv_ == 0x12345u) return; }
#endif
    HME_MARK(1);
    // best inherited candidate by SAD, all candidates in one pass, reference pixels straight from HBM/L2
    int pick = n - 1;
    // level 0, full blocks: sum and sum of squares of the zero-motion reference block (hme.c:181-300), taken from the zero
    // vector's rows while they are here for its SAD -- the statistics stage then has no rows of its own to fetch
    unsigned zc1 = 0, zc2 = 0;
    bool have_z = false;
    // UNI: the union window of the non-zero candidates (pixels relative to (bx, by): columns xmin - 1 .. xmax + bw, rows ymin - 1 ..
    // ymax + bh), staged at S.u.win with `wmis` bytes of misalignment in front of column xmin - 1
    bool have_win = false;
    int uxmin = 0, uxmax = 0, uymin = 0, uymax = 0;
    unsigned wmis = 0;
#ifdef AB_HME_NO_CAND
    if (0) {
#else
    if (n > 1) {
#endif
        unsigned acc[6];
        unsigned validmask = 0;
        const bool src_ok = !frame_invalid(fw, fh, bx, by, bw, bh);
        if constexpr (UNI) {
            // which candidates count, and the box of the non-zero ones
            int xmn = 0x7fffffff, xmx = -0x7fffffff, ymn = 0x7fffffff, ymx = -0x7fffffff;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                acc[k] = 0;
                if (k < n) {
                    const int all = cand[k];
                    const int cdx = ((int)(int16_t)(all & 0xffff)) >> level, cdy = ((int)(int16_t)((unsigned)all >> 16)) >> level;
                    if (src_ok && !frame_invalid(fw, fh, bx + cdx, by + cdy, bw, bh)) {
                        validmask |= 1u << k;
                        if (k > 0) { xmn = min(xmn, cdx); xmx = max(xmx, cdx); ymn = min(ymn, cdy); ymx = max(ymx, cdy); }
                    }
                }
            }
            have_win = (validmask >> 1) != 0u && xmx - xmn <= UW_SPX && ymx - ymn <= UW_SPY;
            HME_COUNT(0);                                      // blocks with candidates beside the zero vector
            if (have_win) HME_COUNT(1);
            else if ((validmask >> 1) == 0u) HME_COUNT(2);     // none of them valid
            else HME_COUNT(3);                                 // spread too wide
            if (have_win) {
                uxmin = xmn; uxmax = xmx; uymin = ymn; uymax = ymx;
                const uint8_t *g0 = rp + (long)(by + ymn - 1) * stride + (bx + xmn - 1);
                wmis = (unsigned)(((uintptr_t)g0) & 3);
                auto q = dsvg_global(g0 - wmis);
                const int nrows = 4 * NKB + 2 + (ymx - ymn);                 // <= UW_ROWS(NKB)
                constexpr int NPW = (UW_ROWS(NKB) * UW_PPR + 63) / 64;       // 16-byte pieces per lane
                dsvg_u32x4a4 pw[NPW];
#pragma unroll
                for (int u = 0; u < NPW; u++) {
                    const int p_ = tid + 64 * u, prow = UW_PDIV(p_), pc = p_ - UW_PPR * prow;
                    if (prow < nrows) pw[u] = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(q + (unsigned)(prow * stride + 16 * pc));
                }
                // (under the window's round trip: the zero candidate from the rows that came with the source block -- level 0 -- or from
                // loads of its own in the same batch: the co-located rows are dword aligned)
                if (validmask & 1u) {
                    if constexpr (!ZE) {
                        auto zq = dsvg_global(rp + (long)by * stride + bx);
                        unsigned lro = lane_ro;
                        HME_LRO_BARRIER(8, lro);
#pragma unroll
                        for (int u = 0; u < NKB; u++) { zw[u] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(zq + lro); zq += stride; }
                    }
#pragma unroll
                    for (int u = 0; u < NKB; u++) acc[0] = __builtin_amdgcn_sad_u8(srcw[u], zw[u], acc[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < NPW; u++) {
                    const int p_ = tid + 64 * u, prow = UW_PDIV(p_), pc = p_ - UW_PPR * prow;
                    if (prow < nrows) *reinterpret_cast<dsvg_u32x4a4 *>(S.u.win + prow * UW_P + 4 * pc) = pw[u];
                }
                hme_sync();
#pragma unroll
                for (int k = 1; k < 6; k++) {
                    if ((validmask >> k) & 1u) {
                        const int all = cand[k];
                        const int cdx = ((int)(int16_t)(all & 0xffff)) >> level, cdy = ((int)(int16_t)((unsigned)all >> 16)) >> level;
                        const int ox = cdx - xmn + 1 + (int)wmis;            // byte of the staged row the candidate's column 0 sits at
                        const unsigned sh = (unsigned)(ox & 3);
                        const unsigned *wl = S.u.win + (cdy - ymn + 1 + r0) * UW_P + (ox >> 2) + cg;
                        // (all of the candidate's rows requested before the first is used: one LDS round trip per candidate -- left to
                        // itself the compiler waited for every row's two dwords in turn, twelve round trips in a chain)
                        unsigned lo_[NKB], hi_[NKB];
#pragma unroll
                        for (int u = 0; u < NKB; u++) { lo_[u] = wl[u * UW_P]; hi_[u] = wl[u * UW_P + 1]; }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < NKB; u++) {
                            const unsigned rw = __builtin_amdgcn_alignbyte(hi_[u], lo_[u], sh);
                            acc[k] = __builtin_amdgcn_sad_u8(srcw[u], rw, acc[k]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        const bool via_win = UNI && have_win;
        if (!via_win) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            acc[k] = 0;
            if (k < n) {
                const int all = cand[k];
                const int cdx = ((int)(int16_t)(all & 0xffff)) >> level, cdy = ((int)(int16_t)((unsigned)all >> 16)) >> level;
                if (src_ok && !frame_invalid(fw, fh, bx + cdx, by + cdy, bw, bh)) {
                    validmask |= 1u << k;
                    if constexpr (FAST) {
                        if (ZE && k == 0) {                        // (the zero vector's rows are in registers)
#pragma unroll
                            for (int u = 0; u < NKB; u++) acc[0] = __builtin_amdgcn_sad_u8(srcw[u], zw[u], acc[0]);
                            continue;
                        }
                        // every row exists in every lane: NKB 8-byte loads back to back off a running pointer, then the SADs
                        const uint8_t *ub = rp + (long)(by + cdy) * stride + (bx + cdx);
                        const unsigned sh = (unsigned)(((uintptr_t)ub) & 3);
                        auto q = dsvg_global(ub - sh);
                        unsigned lro = lane_ro;
                        HME_LRO_BARRIER(2, lro);              // (the zero extension stays next to the loads: see lane_ro)
                        uint2 w[NKB];
#pragma unroll
                        for (int u = 0; u < NKB; u++) { w[u] = dsvg_ld2(q + lro); q += stride; }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < NKB; u++) {
                            const unsigned rw = __builtin_amdgcn_alignbyte(w[u].y, w[u].x, sh);
                            acc[k] = __builtin_amdgcn_sad_u8(srcw[u], rw, acc[k]);
                            if (LEVEL0 && !ZE && k == 0) {         // the zero vector's rows are the zero-motion block of the statistics
                                zc1 = __builtin_amdgcn_sad_u8(rw, 0u, zc1);
                                zc2 = __builtin_amdgcn_udot4(rw, rw, zc2, false);
                            }
                        }
                        if (LEVEL0 && !ZE && k == 0) have_z = true;
                        __builtin_amdgcn_sched_barrier(0);
                    } else
                    if (cmask && r0 < bh) {
                        // the candidate's rows are fetched back to back, then scored
                        const uint8_t *p0 = rp + (long)(by + cdy + r0) * stride + bx + cdx + xcol;
                        const unsigned sh = (unsigned)(((uintptr_t)p0) & 3);
                        const unsigned *pa = reinterpret_cast<const unsigned *>(p0 - sh);
                        const int sdw = stride >> 2;
                        constexpr int CB = 8;                      // rows per batch (registers)
                        // uni: bh is a multiple of the row groups, so "row k exists" is the wave-uniform k < nkb -- rows past
                        // nkb are skipped by scalar branches and no per-lane select is needed
                        auto score = [&](auto UNI_) {
#pragma unroll
                            for (int b0 = 0; b0 < NK; b0 += CB) {
                                if (decltype(UNI_)::value && b0 >= nkb) break;
                                unsigned lo[CB], hi[CB];
#pragma unroll
                                for (int u = 0; u < CB; u++) {
                                    const long o = (long)min(b0 + u, nkb - 1) * sdw;
                                    lo[u] = pa[o]; hi[u] = pa[o + 1];
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int u = 0; u < CB; u++) {
                                    if (decltype(UNI_)::value && b0 + u >= nkb) break;
                                    const unsigned rw = __builtin_amdgcn_alignbyte(hi[u], lo[u], sh) & cmask;
                                    const unsigned a6 = __builtin_amdgcn_sad_u8(srcw[b0 + u], rw, acc[k]);
                                    if (decltype(UNI_)::value) acc[k] = a6;
                                    else acc[k] = ROWOK(b0 + u) ? a6 : acc[k];
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        };
                        if (uni) score(std::true_type{}); else score(std::false_type{});
                    }
                }
            }
        }
        }
        block_sum_n<6>(acc, S.part, phase);
        // first minimum over the valid candidates (hme.c:503-506): the minimum of (score << 3 | index) -- a SAD is below 2^20
        unsigned bestkey = 0x7fffffffu;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const unsigned key = (k < n && ((validmask >> k) & 1u)) ? ((acc[k] << 3) | (unsigned)k) : 0x7fffffffu;
            bestkey = min(bestkey, key);
        }
        if (bestkey != 0x7fffffffu) pick = (int)(bestkey & 7u);
    }
    if constexpr (ZE) {
        // the zero-motion block's sum and sum of squares (the statistics stage, hme.c:181-300) from the same rows
#pragma unroll
        for (int u = 0; u < NKB; u++) { zc1 = __builtin_amdgcn_sad_u8(zw[u], 0u, zc1); zc2 = __builtin_amdgcn_udot4(zw[u], zw[u], zc2, false); }
        have_z = true;
    }
    HME_MARK(2);
    int dx, dy;
    {
        int all = cand[0];
#pragma unroll
        for (int k = 1; k < 6; k++) all = (pick == k) ? cand[k] : all;
        dx = ((int)(int16_t)(all & 0xffff)) >> level;
        dy = ((int)(int16_t)((unsigned)all >> 16)) >> level;
        dx = d_clamp(dx, -bw - bx, fw - bx);
        dy = d_clamp(dy, -bh - by, fh - by);
    }
    // 9-point +-1 search around (dx,dy).  The thread walks reference rows r0-1 .. r0+nkb of the (bw+2)x(bh+2)
    // window at (bx+dx-1, by+dy-1) straight from global memory (three aligned dwords per row -> the three
    // horizontal offsets by v_alignbyte); every reference row serves the source rows above, at and below it.
    int best, bestk;
    {
        constexpr int FX[9] = {0, 1, -1, 0, 0, -1, 1, -1, 1}, FY[9] = {0, 0, 0, 1, -1, -1, -1, 1, 1};   // = FP_X, FP_Y
        unsigned acc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) acc[k] = 0;
#ifdef AB_HME_NO_NINE
        if constexpr (false) {
#else
        if constexpr (FAST) {
#endif
            constexpr int NR = NKB + 2, HB = (NR + 2) / 3;     // reference rows, rows per batch (three batches: registers)
            unsigned v[3][3];                                  // rolling: v[t % 3][ox] = reference row t, offset ox
            // the 9 SADs of this lane's rows from a staged window: row t of the lane's NR rows at nl[t * pitch .. + 2], the window's
            // column 0 at byte m9 of the first dword
            auto nine_lds = [&](const unsigned *nl, int pitch, unsigned m9) {
#pragma unroll
                for (int t = 0; t < NR; t++) {
                    const unsigned dx_ = nl[t * pitch], dy_ = nl[t * pitch + 1], dz_ = nl[t * pitch + 2];
                    const unsigned lo = __builtin_amdgcn_alignbyte(dy_, dx_, m9);        // window bytes 0..3 of the row
                    const unsigned hi = __builtin_amdgcn_alignbyte(dz_, dy_, m9);        //              4..7
                    v[t % 3][0] = lo;
                    v[t % 3][1] = __builtin_amdgcn_alignbyte(hi, lo, 1u);
                    v[t % 3][2] = __builtin_amdgcn_alignbyte(hi, lo, 2u);
                    const int k = t - 2;                           // source row whose three reference rows are now complete
                    if (k >= 0) {
#pragma unroll
                        for (int c9 = 0; c9 < 9; c9++) acc[c9] = __builtin_amdgcn_sad_u8(srcw[k], v[(k + 1 + FY[c9]) % 3][1 + FX[c9]], acc[c9]);
                    }
                }
            };
            // UNI: the picked vector's (bw + 2) x (bh + 2) window lies inside the union window whenever the vector lies in the
            // candidates' box (always, unless the zero vector won from outside it): no memory access at all
            if (UNI && have_win && !(dx >= uxmin && dx <= uxmax && dy >= uymin && dy <= uymax)) have_win = false;
            const bool nine_in_win = UNI && have_win;
            if (UNI) { HME_COUNT(4); if (nine_in_win) HME_COUNT(5); if (n == 1) HME_COUNT(6); }
            if (nine_in_win) {
                const int ob = dx - uxmin + (int)wmis;             // byte of the staged row that window column 0 (pixel dx - 1) sits at
                nine_lds(S.u.win + (dy - uymin + r0) * UW_P + (ob >> 2) + cg, UW_P, (unsigned)(ob & 3));
            } else {
            const uint8_t *g0 = rp + (long)(by + dy - 1) * stride + (bx + dx - 1);
            const unsigned mis = (unsigned)(((uintptr_t)g0) & 3);
            auto q = dsvg_global(g0 - mis);
            unsigned lro = lane_ro;
            HME_LRO_BARRIER(4, lro);
#if HME_NINE_LDS
            {
                // stage rows 0 .. 4 NKB + 1 of the window: lane = (row of a pass of twelve, 16-byte piece 0..4)
                constexpr int WR = 4 * NKB + 2, NP = (WR + 11) / 12;
                const int prow = (tid * 205) >> 10, pc = tid - 5 * prow;           // tid / 5, tid % 5 (tid < 64)
                dsvg_u32x4a4 pw[NP];
                {
                    auto qs = q + (unsigned)(prow * stride + 16 * pc);
#pragma unroll
                    for (int u = 0; u < NP; u++) {
                        if (tid < 60 && 12 * u + prow < WR) pw[u] = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(qs);
                        qs += 12 * stride;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < NP; u++)
                    if (tid < 60 && 12 * u + prow < WR) *reinterpret_cast<dsvg_u32x4a4 *>(S.u.nine + (12 * u + prow) * NINE_P + 4 * pc) = pw[u];
                hme_sync();
                nine_lds(S.u.nine + r0 * NINE_P + cg, NINE_P, mis);
                hme_sync();                                        // (the half-pel stage reuses the space)
            }
            if (false)
#endif
#pragma unroll
            for (int b0 = 0; b0 < NR; b0 += HB) {
                struct __attribute__((aligned(4))) U3 { unsigned x, y, z; } d[HB];
#pragma unroll
                for (int u = 0; u < HB; u++)
#if defined(AB_HME_NINE_X2)
                    if (b0 + u < NR) { const uint2 t2 = dsvg_ld2(q + lro); d[u] = U3{t2.x, t2.y, t2.y}; q += stride; }
#elif defined(AB_HME_NINE_X1)
                    if (b0 + u < NR) { const unsigned t1 = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(q + lro); d[u] = U3{t1, t1, t1}; q += stride; }
#else
                    if (b0 + u < NR) { const dsvg_u32x3a4 t3 = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x3a4 *>(q + lro); d[u] = U3{t3.x, t3.y, t3.z}; q += stride; }
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < HB; u++) {
                    const int t = b0 + u;
                    if (t >= NR) break;
                    const unsigned lo = __builtin_amdgcn_alignbyte(d[u].y, d[u].x, mis);       // window bytes 0..3 of the row
                    const unsigned hi = __builtin_amdgcn_alignbyte(d[u].z, d[u].y, mis);       //              4..7
                    v[t % 3][0] = lo;
                    v[t % 3][1] = __builtin_amdgcn_alignbyte(hi, lo, 1u);
                    v[t % 3][2] = __builtin_amdgcn_alignbyte(hi, lo, 2u);
                    const int k = t - 2;                           // source row whose three reference rows are now complete
                    if (k >= 0) {
#pragma unroll
                        for (int c9 = 0; c9 < 9; c9++) acc[c9] = __builtin_amdgcn_sad_u8(srcw[k], v[(k + 1 + FY[c9]) % 3][1 + FX[c9]], acc[c9]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }       // (not from the union window)
        } else
#ifdef AB_HME_NO_NINE
        if (0) {
#else
        if (cmask && r0 < bh) {
#endif
            const uint8_t *g0 = rp + (long)(by + dy - 1 + r0) * stride + (bx + dx - 1 + xcol);
            const unsigned mis = (unsigned)(((uintptr_t)g0) & 3);
            const unsigned *ga = reinterpret_cast<const unsigned *>(g0 - mis);
            const int sdw = stride >> 2;
            // all row loads are issued back to back (no branch in between: one memory round trip, not NK+2);
            // rows past the thread's last reference row re-read that row
            // three batches of six row loads, each issued back to back (one memory round trip per batch instead of
            // one per row) -- a single batch would cost occupancy in registers
            auto nine = [&](auto UNI) {
                unsigned v[3][3];                                  // rolling: v[t % 3][ox] = reference row t, offset ox
#pragma unroll
                for (int half = 0; half < 3; half++) {
                    constexpr int HB = (NK + 2) / 3;
                    if (decltype(UNI)::value && half * HB >= nkb + 2) break;
                    unsigned d[HB][3];
#pragma unroll
                    for (int u = 0; u < HB; u++) {
                        const long o = (long)min(half * HB + u, nkb + 1) * sdw;
                        d[u][0] = ga[o]; d[u][1] = ga[o + 1]; d[u][2] = ga[o + 2];
                    }
                    __builtin_amdgcn_sched_barrier(0);             // keep the scheduler from sinking loads between the SADs
#pragma unroll
                    for (int u = 0; u < HB; u++) {
                        const int t = half * HB + u;
                        if (decltype(UNI)::value && t >= nkb + 2) break;
                        const unsigned lo = __builtin_amdgcn_alignbyte(d[u][1], d[u][0], mis);   // window bytes 0..3 of the row
                        const unsigned hi = __builtin_amdgcn_alignbyte(d[u][2], d[u][1], mis);   //              4..7
                        v[t % 3][0] = lo & cmask;
                        v[t % 3][1] = __builtin_amdgcn_alignbyte(hi, lo, 1u) & cmask;
                        v[t % 3][2] = __builtin_amdgcn_alignbyte(hi, lo, 2u) & cmask;
                        const int k = t - 2;                       // source row whose three reference rows are now complete
                        if (k >= 0) {
                            const unsigned sw = srcw[k];
                            const bool ok = ROWOK(k);
#pragma unroll
                            for (int c9 = 0; c9 < 9; c9++) {
                                const unsigned a9 = __builtin_amdgcn_sad_u8(sw, v[(k + 1 + FY[c9]) % 3][1 + FX[c9]], acc[c9]);
                                if (decltype(UNI)::value) acc[c9] = a9;
                                else acc[c9] = ok ? a9 : acc[c9];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (uni) nine(std::true_type{}); else nine(std::false_type{});
        }
        block_sum_n<9>(acc, S.part, phase);
        best = 0x7fffffff; bestk = 0;
#pragma unroll
        for (int k = 0; k < 9; k++)
            if (best > (int)acc[k]) { best = (int)acc[k]; bestk = k; }
    }
    HME_MARK(3);
    dx += FP_X[bestk];
    dy += FP_Y[bestk];
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
    // Full blocks: the rows of the zero-motion reference block depend on no decision -- requested now, consumed by the
    // statistics further down, so their memory round trip runs under the half-pel stage.  (The four chroma blocks of the
    // variance test were requested here too until round 3: their 16 registers, held across the half-pel stage, were what
    // put the kernel at 75 VGPRs = 6 waves per SIMD; fetched where they are used it fits 8.)
    unsigned zpre[NKR];
    if constexpr (FAST && !UNI) {
        auto zq = dsvg_global(rp + (long)by * stride + bx);
        unsigned lro = lane_ro;
        HME_LRO_BARRIER(8, lro);
#pragma unroll
        for (int kk = 0; kk < NKR; kk++) {
            zpre[kk] = 0u;
            if (!have_z) { zpre[kk] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(zq + lro); zq += stride; }      // (wave-uniform)
        }
    }
    const unsigned yarea = (unsigned)(bw * bh), yareasq = yarea * yarea;
    const double ryarea = 1.0 / (double)yarea;
    const int wx = bx + ((bw >> 1) - WIN / 2), wy = by + ((bh >> 1) - WIN / 2);
#ifdef AB_HME_NO_HP
    const bool do_hp = false;
#else
    const bool do_hp = best > BW * BH;
#endif
    // stage: source 14x14 window, and either the 19x20 patch for the lattice or the full-pel 14x14 window
    int smis, pmis;
    if constexpr (UNI) {
        // the source window's 14 rows x 4 dwords are in the registers of the lanes that own them (columns 24..39 = column groups 6..9,
        // rows 2 NKB - 7 ..): no load
        constexpr int WYO = 2 * NKB - WIN / 2;
        if (cg >= 6 && cg <= 9) {
#pragma unroll
            for (int k = 0; k < NKB; k++) {
                const int r = r0 + k - WYO;
                if (r >= 0 && r < WIN) reinterpret_cast<unsigned *>(S.swin)[r * 6 + (cg - 6)] = srcw[k];
            }
        }
        smis = 1;                                              // pixel wx = bx + 25 sits at byte 1 of the staged dwords
    } else smis = load_win<8, WIN>(S.swin, 24, sp, sstride, wx, wy, WIN, WIN);
    if (UNI && have_win) {
        // ... and the reference patch (19x20 around the vector for the lattice, or the full-pel 14x14 window) lies inside the union
        // window: taken out of it through registers (the patch's storage overlaps the window's)
        constexpr int WYO = 2 * NKB - WIN / 2;
        const int e = do_hp ? 2 : 0, nrp = do_hp ? 20 : WIN;
        const int prow = WYO + mvy - e - (uymin - 1), pcb = ((bw >> 1) - WIN / 2) + mvx - e - (uxmin - 1) + (int)wmis;
        unsigned pv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int jj = tid + 64 * u, R = (jj * 43691) >> 18, d = jj - 6 * R;
            pv[u] = jj < 6 * nrp ? S.u.win[(prow + R) * UW_P + (pcb >> 2) + d] : 0u;
        }
        hme_sync();
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int jj = tid + 64 * u;
            if (jj < 6 * nrp) reinterpret_cast<unsigned *>(S.u.hp.patch)[jj] = pv[u];
        }
        pmis = pcb & 3;
    } else
    if (do_hp) pmis = load_win<8, 20>(S.u.hp.patch, 24, rp, stride, wx + mvx - 2, wy + mvy - 2, 19, 20);
    else       pmis = load_win<8, 20>(S.u.hp.patch, 24, rp, stride, wx + mvx, wy + mvy, WIN, WIN);
    hme_sync();
    HME_MARK(4);
    bool have_hp = false;
    // lane = (row y, dword d) of a 14x14 window: 4 pixels per lane, the last dword of a row holds two
    const int wy_ = tid >> 2, wd_ = tid & 3;
    // 14 rows x 4 dwords of a byte plane in LDS (pitch P, first pixel at byte `mis` of the row) -> rwin (pitch 16)
    auto copy_win = [&](const uint8_t *pl, int P, int mis) {
        if (tid < 4 * WIN) {
            const int b = mis + 4 * wd_;
            const unsigned *w = reinterpret_cast<const unsigned *>(pl + wy_ * P) + (b >> 2);
            reinterpret_cast<unsigned *>(S.rwin + wy_ * 16)[wd_] = __builtin_amdgcn_alignbyte(w[1], w[0], (unsigned)(b & 3));
        }
    };
    if (do_hp) {
        static_assert(NT == 64, "the half-pel stage maps its items onto one wave");
        // The reference builds a 32x32 lattice of the 16x16 patch cells (full-pel F, horizontal H, vertical V, diagonal D
        // sample of every cell, hpel hme.c:350-376) and reads eight shifted 14x14 windows out of it (hme.c:551-591).  Only
        // H of cells (0..14, 1..14), V of (1..14, 0..14) and D of (0..14, 0..14) are ever read, and the 4-tap filters are
        // byte dot products: with patch row R, column c as P[R][c], cell (li, lj) has
        //   h(li, R) = 9 (P[R][li+1] + P[R][li+2]) - (P[R][li] + P[R][li+3])          (two v_dot4_u32_u8 on a window dword)
        //   H = sat8((h(li, lj+1) + 8) >> 4),  D = sat8((9 (h(li,lj+1) + h(li,lj+2)) - (h(li,lj) + h(li,lj+3)) + 128) >> 8)
        //   V = sat8((9 (P[lj+1][li+1] + P[lj+2][li+1]) - (P[lj][li+1] + P[lj+3][li+1]) + 8) >> 4)   (int16 pairs)
        // stage A: lane = (patch row R, five columns): h -> h16[R][li], H of rows 2..15 -> hb[R-2][li]
        if (tid < 60) {
            const int R = tid / 3, g = tid - 3 * R;
            const int b0 = pmis + 5 * g;
            const unsigned *w = reinterpret_cast<const unsigned *>(S.u.hp.patch + R * 24) + (b0 >> 2);
            const unsigned sh = (unsigned)(b0 & 3);
            const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
            const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
            const unsigned win[5] = {lo, __builtin_amdgcn_alignbyte(hi, lo, 1u), __builtin_amdgcn_alignbyte(hi, lo, 2u), __builtin_amdgcn_alignbyte(hi, lo, 3u), hi};
            short *hrow = S.u.hp.h16 + R * 16 + 5 * g;
            const bool hrow_ok = R >= 2 && R < 16;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int h = (int)__builtin_amdgcn_udot4(win[q], 0x00090900u, 0u, false) - (int)__builtin_amdgcn_udot4(win[q], 0x01000001u, 0u, false);
                hrow[q] = (short)h;
                if (hrow_ok) S.u.hp.hb[(R - 2) * 16 + 5 * g + q] = (uint8_t)d_sat8((h + 8) >> 4);
            }
        }
        // stage V: lane = (row lj = 0..15, four columns li = 4d+1 .. 4d+4): vertical taps on int16 pairs -> vb[lj][li-1]
        {
            typedef short v2s __attribute__((ext_vector_type(2)));
            const int b0 = pmis + 4 * wd_ + 2;
            const unsigned sh = (unsigned)(b0 & 3);
            v2s e[4], o[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned *w = reinterpret_cast<const unsigned *>(S.u.hp.patch + (wy_ + q) * 24) + (b0 >> 2);
                const unsigned x = __builtin_amdgcn_alignbyte(w[1], w[0], sh);
                e[q] = __builtin_bit_cast(v2s, x & 0x00ff00ffu);
                o[q] = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, x, 0x0c030c01u));
            }
            const v2s nine = {9, 9}, eight = {8, 8};
            const v2s ve = ((e[1] + e[2]) * nine + eight - (e[0] + e[3])) >> 4;
            const v2s vo = ((o[1] + o[2]) * nine + eight - (o[0] + o[3])) >> 4;
            unsigned eb, ob;
            asm("v_sat_pk_u8_i16 %0, %1" : "=v"(eb) : "v"(__builtin_bit_cast(unsigned, ve)));
            asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ob) : "v"(__builtin_bit_cast(unsigned, vo)));
            reinterpret_cast<unsigned *>(S.u.hp.vb + wy_ * 16)[wd_] = __builtin_amdgcn_perm(ob, eb, 0x05010400u);      // bytes e0 o0 e1 o1
        }
        hme_sync();
        // stage D: lane = (column li = 0..15, four rows lj = 4g .. 4g+3): seven h values serve four vertical taps
        {
            const int c = tid & 15, g4 = tid >> 4;
            const short *hc = S.u.hp.h16 + (4 * g4) * 16 + c;
            int hv[7];
#pragma unroll
            for (int q = 0; q < 7; q++) hv[q] = hc[q * 16];
#pragma unroll
            for (int q = 0; q < 4; q++)
                S.u.hp.db[(4 * g4 + q) * 16 + c] = (uint8_t)d_sat8((9 * (hv[q + 1] + hv[q + 2]) - (hv[q] + hv[q + 3]) + 128) >> 8);
        }
        hme_sync();
        // the eight candidates (HP_X, HP_Y order): H(x+1,y+1) H(x,y+1) V(x+1,y+1) V(x+1,y) D(x,y) D(x+1,y) D(x,y+1) D(x+1,y+1)
        unsigned acc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = 0;
        if (tid < 4 * WIN) {
            const int b = smis + 4 * wd_;
            const unsigned *sw = reinterpret_cast<const unsigned *>(S.swin + wy_ * 24) + (b >> 2);
            const unsigned m = wd_ == 3 ? 0xffffu : 0xffffffffu;
            const unsigned sv = __builtin_amdgcn_alignbyte(sw[1], sw[0], (unsigned)(b & 3)) & m;
            const unsigned *ph = reinterpret_cast<const unsigned *>(S.u.hp.hb + wy_ * 16) + wd_;
            const unsigned *pv = reinterpret_cast<const unsigned *>(S.u.hp.vb + wy_ * 16) + wd_;
            const unsigned *pd = reinterpret_cast<const unsigned *>(S.u.hp.db + wy_ * 16) + wd_;
            const unsigned h0 = ph[0], h1 = ph[1], v0 = pv[0], v1 = pv[4], d0 = pd[0], d1 = pd[1], d2 = pd[4], d3 = pd[5];
            acc[0] = __builtin_amdgcn_sad_u8(sv, __builtin_amdgcn_alignbyte(h1, h0, 1u) & m, 0u);
            acc[1] = __builtin_amdgcn_sad_u8(sv, h0 & m, 0u);
            acc[2] = __builtin_amdgcn_sad_u8(sv, v1 & m, 0u);
            acc[3] = __builtin_amdgcn_sad_u8(sv, v0 & m, 0u);
            acc[4] = __builtin_amdgcn_sad_u8(sv, d0 & m, 0u);
            acc[5] = __builtin_amdgcn_sad_u8(sv, __builtin_amdgcn_alignbyte(d1, d0, 1u) & m, 0u);
            acc[6] = __builtin_amdgcn_sad_u8(sv, d2 & m, 0u);
            acc[7] = __builtin_amdgcn_sad_u8(sv, __builtin_amdgcn_alignbyte(d3, d2, 1u) & m, 0u);
        }
        block_sum_n<8>(acc, S.part, phase);
        int best_hp = (int)udiv_rd((unsigned)(best * (WIN * WIN)), ryarea);
        int hm = -1;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (best_hp > (int)acc[k]) { best_hp = (int)acc[k]; hm = k; }
        mvx = (int)(int16_t)(mvx << 1);
        mvy = (int)(int16_t)(mvy << 1);
        if (hm >= 0) {
            best = (int)((unsigned)best_hp * yarea / (WIN * WIN));
            mvx = (int)(int16_t)(mvx + HP_X[hm]);
            mvy = (int)(int16_t)(mvy + HP_Y[hm]);
            // the winner's window: its plane, first row (0 / 1) and first column (0 / 1)
            const uint8_t *pl = hm < 2 ? S.u.hp.hb : (hm < 4 ? S.u.hp.vb : S.u.hp.db);
            const int ro = (hm == 2 || hm >= 6) ? 1 : 0, co = (hm == 0 || hm == 5 || hm == 7) ? 1 : 0;
            copy_win(pl + ro * 16, 16, co);
            have_hp = true;
        }
    } else {
        mvx = (int)(int16_t)(mvx << 1);
        mvy = (int)(int16_t)(mvy << 1);
    }
    if (!have_hp) {
        // half-pel search found nothing better: the full-pel window at the (doubled) vector is rows / columns 2 .. 15 of the 19x20
        // patch that is still staged (round 4: it was fetched again until now -- one more round trip for every such block)
        if (do_hp) copy_win(S.u.hp.patch + 2 * 24, 24, pmis + 2);
        else copy_win(S.u.hp.patch, 24, pmis);
    }
    // the zero-motion reference block (variance test, veto, quadrant votes) is read straight from global memory:
    // bx + 4cg is dword aligned, every thread takes the rows it owns
    hme_sync();                                    // rwin complete
    HME_MARK(5);
    unsigned zrow[NKR];
    if constexpr (UNI) {
#pragma unroll
        for (int kk = 0; kk < NKR; kk++) zrow[kk] = zw[kk];
    } else if constexpr (FAST) {
#pragma unroll
        for (int kk = 0; kk < NKR; kk++) zrow[kk] = zpre[kk];
    } else {
        const unsigned *zp = reinterpret_cast<const unsigned *>(rp + (long)(by + r0) * stride + bx + xcol);
        const int sdw = stride >> 2;
#pragma unroll
        for (int kk = 0; kk < NKR; kk++) zrow[kk] = (cmask && r0 < bh) ? zp[(long)min(kk, nkb - 1) * sdw] : 0u;
    }
    // ---- statistics: one fused pass + one 8-value and one 10-value reduction
    unsigned st[8];                 // src block gh,gv,s1,s2 ; zref s1,s2 ; spare
#pragma unroll
    for (int k = 0; k < 8; k++) st[k] = 0;
    {
        // the row above this thread's first row belongs to the previous row group: lane - 16, its last row
        unsigned lastw = srcw[0];
#pragma unroll
        for (int k = 1; k < NKR; k++) lastw = (k == nkb - 1) ? srcw[k] : lastw;
        unsigned upw = (unsigned)__shfl_up((int)lastw, 16);
#pragma unroll
        for (int kk = 0; kk < NKR; kk++) {
            const int r = r0 + kk;
            const unsigned curw = srcw[kk];
            // horizontal neighbours: bytes x+1..x+4 of the same row; the 4th comes from the next column group = lane + 1
            // of the same 16-lane row (DPP row_shl:1, zero past the row's end -- masked by pm there anyway)
            const unsigned nxt = (unsigned)__builtin_amdgcn_update_dpp(0, (int)curw, 0x101, 0xf, 0xf, true);
#ifdef AB_HME_NO_STATS
            if (0) {
#else
            if (cmask && ROWOK(kk)) {
#endif
                const unsigned right = __builtin_amdgcn_alignbyte(nxt, curw, 1u);
                // pairs (x,x+1) count only while x+1 < bw
                const int npair = min(4, bw - 1 - xcol);
                const unsigned pm = npair <= 0 ? 0u : (npair >= 4 ? 0xffffffffu : ((1u << (8 * npair)) - 1u));
                st[0] = __builtin_amdgcn_sad_u8(curw & pm, right & pm, st[0]);
                if (r > 0) st[1] = __builtin_amdgcn_sad_u8(curw, upw, st[1]);
                st[2] = __builtin_amdgcn_sad_u8(curw, 0u, st[2]);
                st[3] = __builtin_amdgcn_udot4(curw, curw, st[3], false);
                if (!have_z) {
                    const unsigned zw = zrow[kk] & cmask;
                    st[4] = __builtin_amdgcn_sad_u8(zw, 0u, st[4]);
                    st[5] = __builtin_amdgcn_udot4(zw, zw, st[5], false);
                }
            }
            upw = curw;
        }
        if (have_z) { st[4] = zc1; st[5] = zc2; }
    }
    // block statistics and the two 14x14 window statistics share one reduction
    unsigned ws[14];
    win_partial(S.swin, 24, smis, ws[0], ws[1], ws[2], ws[3]);
    win_partial(S.rwin, 16, 0, ws[4], ws[5], ws[6], ws[7]);
#pragma unroll
    for (int k = 0; k < 6; k++) ws[8 + k] = st[k];
    block_sum_n<14>(ws, S.part, phase);
    HME_MARK(6);
#pragma unroll
    for (int k = 0; k < 6; k++) st[k] = ws[8 + k];
    const unsigned luma_tex = udiv_rd((st[0] + st[1]) / 2, ryarea);
    const unsigned luma_var = st[3] - udiv_rd(st[2] * st[2], ryarea);
    const unsigned zs1 = st[4];
    const unsigned zvar = st[5] - udiv_rd(st[4] * st[4], ryarea);
    const int src_tex = (int)(((ws[0] + ws[1]) / 2) / (WIN * WIN));
    const int src_avg = (int)(ws[2] / (WIN * WIN));
    const int src_var = (int)(ws[3] - (ws[2] * ws[2]) / (WIN * WIN));
    const int ref_tex = (int)(((ws[4] + ws[5]) / 2) / (WIN * WIN));
    const int ref_avg = (int)(ws[6] / (WIN * WIN));
    const int ref_var = (int)(ws[7] - (ws[6] * ws[6]) / (WIN * WIN));
    out.x = (int16_t)mvx; out.y = (int16_t)mvy;
    out.lo_tex = (luma_tex <= 2);
    out.lo_var = (luma_var < yareasq);

    bool want_intra = false;
    if (src_tex < 2 && zvar > luma_var * 2) want_intra = true;
    else if (ref_var > src_var * 2) want_intra = true;
    else if (src_tex == 0 && ref_tex != 0) want_intra = true;
    else if (abs(src_avg - ref_avg) > 8) want_intra = true;
    else if (luma_tex <= 10 && (unsigned)best > yareasq / 16) want_intra = true;
#ifdef AB_HME_NO_CHROMA
    else if (0) {
#else
    else {
#endif
        // chroma variance test (c_maxvar hme.c:269-300): the four chroma blocks straight from HBM, one reduction
        const FrameLayout &L0 = A.L[0];
        const int cbx = i * (BW >> L0.hs), cby = j * (BH >> L0.vs);
        const int cbw = bw >> L0.hs, cbh = bh >> L0.vs;
        unsigned cs[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cs[k] = 0;
        const int ndw = (cbw + 3) >> 2;
        // (chroma planes by the per-slot tables: the bordered frame, or the caller's packed clip for frames loaded in place)
        const uint8_t *cpl[4] = {reinterpret_cast<const uint8_t *>(A.slot_cu[cur]), reinterpret_cast<const uint8_t *>(A.slot_cv[cur]),
                                 reinterpret_cast<const uint8_t *>(A.slot_cu[rf]), reinterpret_cast<const uint8_t *>(A.slot_cv[rf])};
        const int cst[4] = {A.slot_cs[cur], A.slot_cs[cur], A.slot_cs[rf], A.slot_cs[rf]};
        const bool aligned = (((unsigned)cst[0] | (unsigned)cst[2] | (unsigned)cbx | (unsigned)(uintptr_t)cpl[0] | (unsigned)(uintptr_t)cpl[1] |
                               (unsigned)(uintptr_t)cpl[2] | (unsigned)(uintptr_t)cpl[3]) & 3u) == 0;
        bool done_c = false, cs_from_table = false;
        if constexpr (FAST) {
            if (A.csum && ((((unsigned)cst[0] | (unsigned)cst[2] | (unsigned)(uintptr_t)cpl[0] | (unsigned)(uintptr_t)cpl[1] | (unsigned)(uintptr_t)cpl[2] |
                             (unsigned)(uintptr_t)cpl[3]) & 3u) == 0) && (cbw & 15) == 0 && A.nxb <= 64) {
                // the sums of this block's chroma blocks in both frames, from k_hme_csum's table (the same conditions as there)
                const unsigned *ts = A.csum + ((size_t)cur * A.nblk + i + j * A.nxb) * 4, *tr = A.csum + ((size_t)rf * A.nblk + i + j * A.nxb) * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) { cs[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)ts[k]); cs[4 + k] = (unsigned)__builtin_amdgcn_readfirstlane((int)tr[k]); }
                done_c = cs_from_table = true;
            }
        }
        if constexpr (FAST) if (!done_c) {
            unsigned cpre[4][4];
            bool cpre_ok = false;
            const int rpp = ndw ? NT / ndw : NT + 1;
            const uint8_t *su = reinterpret_cast<const uint8_t *>(A.slot_cu[cur]), *sv_ = reinterpret_cast<const uint8_t *>(A.slot_cv[cur]);
            const uint8_t *ru = reinterpret_cast<const uint8_t *>(A.slot_cu[rf]), *rv_ = reinterpret_cast<const uint8_t *>(A.slot_cv[rf]);
            const int ss = A.slot_cs[cur], rs = A.slot_cs[rf];          // (source and reference may differ: in place / bordered)
            cpre_ok = (((unsigned)ss | (unsigned)rs | (unsigned)cbx | (unsigned)(uintptr_t)su | (unsigned)(uintptr_t)sv_ | (unsigned)(uintptr_t)ru | (unsigned)(uintptr_t)rv_ | (unsigned)cbw) & 3u) == 0 &&
                      ndw >= 4 && (ndw & (ndw - 1)) == 0 && ndw <= 16 && cbh % rpp == 0 && cbh / rpp <= 4;
            if (cpre_ok) {
                // lane = (row, dword) of a pass of 64 / ndw rows; at most four passes
                const int sh = 31 - __clz(ndw), npass = cbh / rpp;
                // (wave-uniform bases + one 32-bit lane offset, as the luma rows)
                const long obs = (long)cby * ss + cbx, obr = (long)cby * rs + cbx, advs = (long)rpp * ss, advr = (long)rpp * rs;
                const unsigned clos = (unsigned)((tid >> sh) * ss + 4 * (tid & (ndw - 1))), clor = (unsigned)((tid >> sh) * rs + 4 * (tid & (ndw - 1)));
                auto q0 = dsvg_global(su + obs), q1 = dsvg_global(sv_ + obs), q2 = dsvg_global(ru + obr), q3 = dsvg_global(rv_ + obr);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    cpre[u][0] = cpre[u][1] = cpre[u][2] = cpre[u][3] = 0u;
                    if (u < npass) {                                       // wave-uniform
                        unsigned c2 = clos, c3 = clor;
                        HME_LRO_BARRIER(16, c2);                               // (per block of code: see lane_ro)
                        HME_LRO_BARRIER(16, c3);
                        cpre[u][0] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(q0 + c2);
                        cpre[u][1] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(q1 + c2);
                        cpre[u][2] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(q2 + c3);
                        cpre[u][3] = *reinterpret_cast<const DSVG_GLOBAL unsigned *>(q3 + c3);
                    }
                    q0 += advs; q1 += advs; q2 += advr; q3 += advr;
                }
            }
            if (cpre_ok) {
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int o = (k >> 1) * 4 + (k & 1) * 2;
                        cs[o] = __builtin_amdgcn_sad_u8(cpre[u][k], 0u, cs[o]);
                        cs[o + 1] = __builtin_amdgcn_udot4(cpre[u][k], cpre[u][k], cs[o + 1], false);
                    }
                done_c = true;
            }
        }
        if (done_c) {
        } else if (aligned && ndw * cbh <= 4 * NT) {
            // the usual case (block origins are multiples of 4 in the chroma planes): one aligned dword per item and
            // plane, all of a lane's loads in flight together -- one memory round trip for the whole test
            unsigned cw4[4][4], cm[4];
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int q = tid + it * NT;
                const bool ok = q < ndw * cbh;
                const int y = ok ? q / ndw : 0, xd = ok ? 4 * (q - y * ndw) : 0;
                const int nb = min(4, cbw - xd);
                cm[it] = !ok ? 0u : (nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u));
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    cw4[it][k] = *reinterpret_cast<const unsigned *>(cpl[k] + (long)(cby + y) * cst[k] + cbx + xd);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 4; it++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned w = cw4[it][k] & cm[it];
                    const int o = (k >> 1) * 4 + (k & 1) * 2;
                    cs[o] = __builtin_amdgcn_sad_u8(w, 0u, cs[o]);
                    cs[o + 1] = __builtin_amdgcn_udot4(w, w, cs[o + 1], false);
                }
        } else
        for (int q = tid; q < ndw * cbh; q += NT) {
            const int y = q / ndw, xd = 4 * (q - y * ndw);
            const int nb = min(4, cbw - xd);
            const unsigned m = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
#pragma unroll
            for (int side = 0; side < 2; side++) {
#pragma unroll
                for (int pl = 1; pl <= 2; pl++) {
                    const unsigned w = ldg_u32_unaligned(cpl[side * 2 + pl - 1] + (long)(cby + y) * cst[side * 2] + cbx + xd) & m;
                    const int o = side * 4 + (pl - 1) * 2;
                    cs[o] = __builtin_amdgcn_sad_u8(w, 0u, cs[o]);
                    cs[o + 1] = __builtin_amdgcn_udot4(w, w, cs[o + 1], false);
                }
            }
        }
        if (!cs_from_table) block_sum_n<8>(cs, S.part, phase);
        const unsigned carea = (unsigned)(cbw * cbh);
        // full block: the chroma area is the luma area over a power of two -- the reciprocal is an exact scaling of ryarea
        const double rcarea = FAST ? ryarea * (double)(1 << (L0.hs + L0.vs)) : 1.0 / (double)carea;
        const unsigned vsu = cs[1] - udiv_rd(cs[0] * cs[0], rcarea), vsv = cs[3] - udiv_rd(cs[2] * cs[2], rcarea);
        const unsigned vru = cs[5] - udiv_rd(cs[4] * cs[4], rcarea), vrv = cs[7] - udiv_rd(cs[6] * cs[6], rcarea);
        const unsigned cvs = vsu > vsv ? vsu : vsv, cvr = vru > vrv ? vru : vrv;
        if (cvr > 4 * cvs) want_intra = true;
    }

    HME_MARK(7);
    if (want_intra) {
        // representability veto + the four quadrant votes in one pass, one 9-value reduction
        const int mean = (int)udiv_rd(zs1, ryarea);
        const int qw = bw / 2, qh = bh / 2;
        unsigned qv[9];                 // [0] bad count, [1+2q] good, [2+2q] evil
#pragma unroll
        for (int k = 0; k < 9; k++) qv[k] = 0;
        for (int kk = 0; kk < NK; kk++) {
            const int r = rg + NRG * kk;
            if (r >= bh) continue;
#pragma unroll
            for (int b4 = 0; b4 < 4; b4++) {
                const int x = xcol + b4;
                if (x >= bw) continue;
                const uint8_t *sx = sp + (size_t)(by + r) * sstride + bx + x;    // rare path: source pixels straight from global
                const uint8_t *zx = rp + (size_t)(by + r) * stride + bx + x;     // and the co-located reference pixels
                const int pa = sx[0], pb = zx[0];
                const int back = d_sat8(mean + d_sat8(pa - mean + 128) - 128);
                qv[0] += (back != pa);
                if (x < 2 * qw && r < 2 * qh) {
                    const int qx = x >= qw, qy = r >= qh;
                    const int lx = x - qx * qw, ly = r - qy * qh;        // position inside the quadrant
                    const int la = lx ? sx[-1] : pa, lb = lx ? zx[-1] : pb;
                    const int ua = ly ? sx[-sstride] : pa, ub = ly ? zx[-stride] : pb;
                    const int dif = abs(pa - pb);
                    unsigned good = (unsigned)(abs(pa - la) + abs(pa - ua) + abs(pb - lb) + abs(pb - ub));
                    unsigned evil = 0;
                    if (dif == 0) good += 192;
                    else if (dif == 1) good += 128;
                    else if (dif == 2) good += 96;
                    else evil = (unsigned)dif;
                    const int q = qx + 2 * qy;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        qv[1 + 2 * qq] += (q == qq) ? good : 0u;
                        qv[2 + 2 * qq] += (q == qq) ? evil : 0u;
                    }
                }
            }
        }
        block_sum_n<9>(qv, S.part, phase);
        if (!qv[0]) {
            int submask = 0xF;
            if (src_tex > 1) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (qv[1 + 2 * k] >= (unsigned)((qw + qh) >> 1) * qv[2 + 2 * k]) submask &= ~(1 << k);
            }
            if (submask) {
                out.submask = (uint8_t)submask;
                out.mode = 1;
            }
        }
    }
    if (tid == 0) {
        mf[i + j * A.nxb] = out;
        A.aux_tex[(size_t)pair * A.nblk + i + j * A.nxb] = luma_tex;
        A.aux_var[(size_t)pair * A.nblk + i + j * A.nxb] = src_var;
    }
#ifdef DSVG_CLOCK_PROBE
    HME_MARK(8);
    if (FAST && tid == 0 && (blockIdx.x & 127) == 5) {
#pragma unroll
        for (int q = 0; q < 8; q++) { atomicAdd(&dsvg_clk_acc[8 + q][0], (unsigned long long)(clk_m_[q + 1] - clk_m_[q])); atomicAdd(&dsvg_clk_acc[8 + q][1], 1ull); }
    }
#endif
}

// one wave per visited block and frame pair; NKBF = rows per lane of a FULL block of this geometry (0: none is special)
// PART 1: the full blocks only (64 wide, 4 * NKBF rows: the specialised body), grid = fullx x fully blocks per pair;
// PART 2: the rest of the block grid (right column / bottom row of partial blocks), the generic body -- as kernels of their
// own: in one kernel the register allocation is the generic body's (106 SGPRs + 246 spilled to VGPR lanes, 79 VGPRs; the
// specialised body alone: 64 SGPRs, no spills).  PART 0: every block, generic body (geometries without full blocks).
// PART 3: every block, either body (the small upper levels of the pyramid: a second launch costs more than it saves).
// n / d for n < 2^32 with inv = min(floor(2^32 / d), 2^32 - 1) from the launcher: the estimate is the quotient or one below it
// (the item -> pair / block row / block column divisions of every wave were two ~25-instruction division sequences)
static __device__ __forceinline__ unsigned hme_udiv(unsigned n, unsigned d, unsigned inv, unsigned &r)
{
    unsigned q = __umulhi(n, inv);
    r = n - q * d;
    if (r >= d) { q++; r -= d; }
    return q;
}
#ifdef HME_WPE
#define HME_WPE_ATTR __attribute__((amdgpu_waves_per_eu(HME_WPE, HME_WPE)))
#else
#define HME_WPE_ATTR
#endif
template <bool LEVEL0, int NKBF, int PART>
__global__ __launch_bounds__(NT * HME_WPG) HME_WPE_ATTR void k_hme_level(HmeArgs A, int level, int npairs, int fullx, int fully, unsigned inv_per, unsigned inv_row)
{
    typedef HmeSharedT<((LEVEL0 || HME_UNION_UPPER != 0) && PART != 2 && PART != 0) ? NKBF : 0> HmeShared;      // (the window is sized for the launch's full blocks)
    __shared__ HmeShared SS[HME_WPG];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    HmeShared &S = SS[wave];
    const int step = 1 << level;
    const int nvx = (A.nxb + step - 1) / step, nvy = (A.nyb + step - 1) / step;
    const int per = PART == 1 ? fullx * fully : (PART == 2 ? nvx * nvy - fullx * fully : nvx * nvy);      // blocks of a pair in this launch
    const int nwg = (per * npairs + HME_WPG - 1) / HME_WPG;
    const int wg = d_xcd_remap(blockIdx.x, nwg);
    const int item = wg * HME_WPG + wave;
    if (wg >= nwg || item >= per * npairs) return;
    // item -> (pair, block).  HME_PG consecutive frame pairs take a block position one after the other (with HME_PG = HME_WPG: the
    // waves of a workgroup): the current frame of pair p is the reference of pair p + 1, so the block's source rows and the
    // next pair's reference rows around the same position are the same lines -- fetched from HBM once per group instead of
    // once per pair
    int pair, vb, vi, vj;
    if (HME_PG > 1) {
        const int gsz = HME_PG * per, q = item / gsz, r = item - q * gsz;
        const int gl = min(HME_PG, npairs - q * HME_PG);        // pairs in this group (the last one may be short)
        vb = r / gl;
        pair = q * HME_PG + (r - vb * gl);
    } else {
        unsigned r_;
        pair = (int)hme_udiv((unsigned)item, (unsigned)per, inv_per, r_);
        vb = (int)r_;
    }
    if (PART == 1) { unsigned r_; vj = (int)hme_udiv((unsigned)vb, (unsigned)fullx, inv_row, r_); vi = (int)r_; }
    else if (PART == 2) {
        const int nr = (nvx - fullx) * nvy;                 // right strip (all rows), then the bottom strip under the full blocks
        if (vb < nr) { vj = vb / (nvx - fullx); vi = fullx + vb - vj * (nvx - fullx); }
        else { vb -= nr; vj = vb / fullx; vi = vb - vj * fullx; vj += fully; }
    } else { unsigned r_; vj = (int)hme_udiv((unsigned)vb, (unsigned)nvx, inv_row, r_); vi = (int)r_; }
    const int i = vi * step, j = vj * step;
    const int fw = A.L[level].w[0], fh = A.L[level].h[0];
    const int bx = (i * A.blk_w) >> level, by = (j * A.blk_h) >> level;
    if (bx >= fw || by >= fh) {                              // a zero inter vector (hme.c:441-444): written here, so that the vector
        if (HME_TID == 0) {                                 // field needs no clearing before the launch (116 MB per 320-GOP step)
            DMV z;
            z.x = z.y = 0; z.mode = z.submask = z.lo_var = z.lo_tex = z.high_detail = 0; z.pad[0] = z.pad[1] = z.pad[2] = 0;
            A.mvf[((size_t)pair * (A.levels + 1) + level) * A.nblk + i + j * A.nxb] = z;
        }
        return;
    }
    if constexpr (PART == 3) {                              // every block in one launch, each with the body that fits it
        if (fw - bx >= 64 && fh - by >= 4 * NKBF) { hme_block<LEVEL0, NKBF>(A, level, pair, i, j, S); return; }
    }
    DSVG_CLK_BEGIN();
    if constexpr (PART == 1) hme_block<LEVEL0, NKBF>(A, level, pair, i, j, S);
    else hme_block<LEVEL0, 0>(A, level, pair, i, j, S);
    DSVG_CLK_END(LEVEL0 ? (PART == 1 ? 0 : 1) : 2);
}

// Sums of the chroma blocks (c_maxvar's inputs, hme.c:269-300), once per source frame.  The variance test reads a frame's blocks twice -- as
// the current frame of one pair and as the reference frame of the next -- and a wave reads them as 32-byte pieces of 128-byte lines
// (12 loads, 96 L2 requests per block: 0.9 ms of the level-0 kernel by ablation).  Here a workgroup streams one row of chroma blocks of one
// frame, whole lines, both planes, and leaves four numbers per block; the motion search fetches eight scalars.
// Grid: (block row, 2 * pair + role); role 0 = the pair's current frame, role 1 = its reference frame unless the pair before has it as its
// current frame.  Only full luma blocks (i < fullx, j < fully) and planes / strides / block widths that are multiples of 4.
__global__ __launch_bounds__(256) void k_hme_csum(HmeArgs A, int fullx, int fully)
{
    __shared__ unsigned acc[64 * 4];
    const int j = blockIdx.x, pr = blockIdx.y, pair = pr >> 1, role = pr & 1;
    const int slot = role ? A.ref_slots[pair] : A.cur_slots[pair];
    if (role && pair > 0 && A.cur_slots[pair - 1] == slot) return;
    const FrameLayout &L0 = A.L[0];
    const int cbw = A.blk_w >> L0.hs, cbh = A.blk_h >> L0.vs;
    const int cstr = A.slot_cs[slot];
    const unsigned long long pu = A.slot_cu[slot], pv = A.slot_cv[slot];
    if ((((unsigned)cstr | (unsigned)pu | (unsigned)pv) & 3u) || (cbw & 15) || fullx > 64) return;      // (the search then fetches the blocks itself)
    for (int t = threadIdx.x; t < fullx * 4; t += 256) acc[t] = 0;
    __syncthreads();
    // item = (row y, block i): the block's row piece, cbw bytes in 16-byte loads -- neighbouring lanes read neighbouring pieces of the row
    const size_t row0 = (size_t)j * cbh * cstr;
    const auto bu = dsvg_global(reinterpret_cast<const uint8_t *>(pu) + row0), bv = dsvg_global(reinterpret_cast<const uint8_t *>(pv) + row0);
    for (int c = threadIdx.x; c < fullx * cbh; c += 256) {
        const int y = c / fullx, i = c - y * fullx;
        const unsigned o = (unsigned)(y * cstr + i * cbw);
        unsigned s[4] = {0, 0, 0, 0};
        for (int k = 0; k < cbw; k += 16) {
            const dsvg_u32x4a4 wu = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(bu + o + k), wv = *reinterpret_cast<const DSVG_GLOBAL dsvg_u32x4a4 *>(bv + o + k);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                s[0] = __builtin_amdgcn_sad_u8(wu[q], 0u, s[0]); s[1] = __builtin_amdgcn_udot4(wu[q], wu[q], s[1], false);
                s[2] = __builtin_amdgcn_sad_u8(wv[q], 0u, s[2]); s[3] = __builtin_amdgcn_udot4(wv[q], wv[q], s[3], false);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) atomicAdd(&acc[i * 4 + q], s[q]);
    }
    __syncthreads();
    unsigned *out = A.csum + ((size_t)slot * A.nblk + (size_t)j * A.nxb) * 4;
    for (int t = threadIdx.x; t < fullx * 4; t += 256) out[t] = acc[t];
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
    if (A.csum) {
        // (full level-0 blocks, as the level loop below defines them)
        const bool fullb = A.blk_w == 64 && (A.blk_h == 64 || A.blk_h == 48 || A.blk_h == 32) && (A.L[0].stride[0] & 3) == 0;
        const int fullx = fullb ? std::min(A.nxb, A.L[0].w[0] / 64) : 0, fully = fullb ? std::min(A.nyb, A.L[0].h[0] / A.blk_h) : 0;
        if (fullx > 0 && fully > 0) {
            if (pf) pf->begin(st, KID_HME_CSUM, (double)npairs * ((double)A.L[0].w[1] * A.L[0].h[1] + (double)A.L[0].w[2] * A.L[0].h[2]));      // every frame's chroma once
            hipLaunchKernelGGL(k_hme_csum, dim3(fully, 2 * npairs), dim3(256), 0, st, A, fullx, fully);
            if (pf) pf->end(st);
        }
    }
    for (int level = A.levels; level >= 0; level--) {
        const int step = 1 << level;
        const int nvx = (A.nxb + step - 1) / step, nvy = (A.nyb + step - 1) / step;
        double px = 2.0 * npairs * (double)A.L[level].w[0] * A.L[level].h[0];           // src + ref luma once
        if (level == 0 && !A.csum)  // level 0 also reads both frames' chroma planes (c_maxvar hme.c:269-300,669-681) unless k_hme_csum has read them
            px += 2.0 * npairs * ((double)A.L[0].w[1] * A.L[0].h[1] + (double)A.L[0].w[2] * A.L[0].h[2]);
        if (pf) pf->begin(st, level > 0 ? KID_HME_LEVEL : KID_HME_LEVEL0, px);
        const dim3 blk(NT * HME_WPG);
        // rows per lane of a full block (64 wide, blk_h = 4 * rows): those blocks take the specialised body, in a launch of
        // their own; the partial blocks at the right / bottom edge of the level's frame the generic one
        const int nkbf = (A.blk_w == 64 && (A.blk_h == 64 || A.blk_h == 48 || A.blk_h == 32) && (A.L[level].stride[0] & 3) == 0) ? A.blk_h / 4 : 0;
        const int fw = A.L[level].w[0], fh = A.L[level].h[0];
        const int fullx = nkbf ? std::min(nvx, fw / 64) : 0, fully = nkbf ? std::min(nvy, fh / A.blk_h) : 0;
        const int nfull = fullx * fully, nrest = nvx * nvy - nfull;
        auto uinv = [](int d) { return (unsigned)std::min<unsigned long long>(0x100000000ull / (unsigned long long)std::max(d, 1), 0xffffffffull); };
        // (inv_row: the row length the launch's block index is split by -- PART 1: fullx; PART 0 / 3: nvx; PART 2 divides by its two strip widths itself)
#define HME_LAUNCH(L0, N, P, cnt) hipLaunchKernelGGL((k_hme_level<L0, N, P>), dim3(xcd_grid(((cnt) * npairs + HME_WPG - 1) / HME_WPG)), blk, 0, st, A, level, npairs, fullx, fully, \
                                                     uinv(cnt), uinv((P) == 1 ? fullx : nvx))
#define HME_FULL(L0) do { switch (nkbf) { case 16: HME_LAUNCH(L0, 16, 1, nfull); break; case 12: HME_LAUNCH(L0, 12, 1, nfull); break; \
                                          default: HME_LAUNCH(L0, 8, 1, nfull); } } while (0)
        (void)nrest;
        if (nfull > 0 && level > 0) {
            switch (nkbf) { case 16: HME_LAUNCH(false, 16, 3, nvx * nvy); break; case 12: HME_LAUNCH(false, 12, 3, nvx * nvy); break;
                            default: HME_LAUNCH(false, 8, 3, nvx * nvy); }
        } else if (nfull > 0) {
            HME_FULL(true);
            if (nrest > 0) HME_LAUNCH(true, 0, 2, nrest);
        } else if (level > 0) HME_LAUNCH(false, 0, 0, nvx * nvy);
        else HME_LAUNCH(true, 0, 0, nvx * nvy);
        if (pf) pf->end(st);
    }
    if (pf) pf->begin(st, KID_HME_DETAIL, 0.0);
    hipLaunchKernelGGL(k_hme_detail, dim3((A.nblk + 255) / 256, npairs), dim3(256), 0, st, A);
    if (pf) pf->end(st);
}

#ifdef DSVG_CLOCK_PROBE
DSVG_CLK_DUMP_FN(dsvg_clk_dump_hme, "k_hme_level<true> full", "k_hme_level<true> edge", "k_hme_level<false>", "", "", "", "", "",
                 "hme0 stage: source+parents", "hme0 stage: candidates", "hme0 stage: nine-point", "hme0 stage: windows in", "hme0 stage: half-pel",
                 "hme0 stage: statistics", "hme0 stage: intra tests+chroma", "hme0 stage: vote+store")
#endif
