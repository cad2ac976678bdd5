// dsvg_kernels.hpp -- launcher prototypes and by-value kernel argument blocks.
#pragma once
#include "dsvg_dev.hpp"

#define DSVG_BORDER 64

struct Prof;
struct SbtGeo3 { SbtGeo g[3]; };

struct McGeo {
    int blk_w, blk_h, nbh, nbv, hs, vs;
    int w[3], h[3], stride[3];
    size_t off[3];
    int cw_extra[3];         // coefficient plane one column wider than the pixel plane
};

struct HmeArgs {
    FrameLayout L[6];        // level 0 = full frames, level i = pyramid level i
    const uint8_t *slab[6];
    const int *cur_slots, *ref_slots;
    // chroma planes of every source slot (level 0's variance test, hme.c:269-300,667-681): pixel (0,0) of U / V and the row stride --
    // the bordered frame's, or the caller's packed clip for frames loaded in place (dsvg_load_frames_map_ex)
    const unsigned long long *slot_cu, *slot_cv;
    const int *slot_cs;
    DMV *mvf;                // [pair][level][nblk]
    unsigned *aux_tex;       // [pair][nblk] block texture (for the high_detail pass)
    int *aux_var;            // [pair][nblk] centre-window variance
    // (round 5) per source slot: the frame's luma plane in the caller's packed clip (row stride = the picture's width), or 0: the level-0 search
    // reads a block's source and reference rows there when everything it can touch lies inside the picture (hme_block, `deep`) -- of the
    // bordered copies only a ring exists for such frames (k_unpack, slot-table bit 29)
    const unsigned long long *slot_y;
    int deep_r;
    // [slot][nblk][4]: sum / sum of squares of the U and of the V block of every FULL block of a source slot (k_hme_csum), or nullptr:
    // level 0's chroma variance test then fetches the blocks itself
    unsigned *csum;
    int levels, nxb, nyb, nblk, blk_w, blk_h;
};

// k_sbt.hip
int  sbt_tail_supported(const SbtGeo &g);
void sbt_set_func_attributes();
void launch_fwd_sbt(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, int isP, int from_src, Prof *pf = nullptr, int with_tail = 1, int fused = 0,
                    const struct McGeo *mc = nullptr, const DMV *mvs0 = nullptr,    // mc: motion compensation fused into the P forward transform
                    int general_whole = 1);   // mc: 0 = the caller knows that no block of these pictures is intra (the general kernel then only runs on the strips of the grid the geometry asks for)
bool mc_fusable(const struct McGeo &MG);
void launch_inv_sbt(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, int isP, Prof *pf = nullptr, int with_tail = 1,
                    int insym = 0, int patch_kernel = 0,    // patch_kernel: sparse P pictures (flags valid, prediction given): unfiltered planes take k_inv_patch_c
                    int fuse_border = 0);                   // the kernels also write the reconstruction's border (JobDev.ext) where inv_sbt_fuses_border says they can
bool inv_sbt_fuses_border(const SbtGeo3 &G, int insym_c, int patch_kernel_c);   // the chroma patch kernel writes the borders of all three planes
void launch_inv54_all(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, Prof *pf = nullptr);   // levels 5..4 of all planes: then launch_inv_sbt(.., with_tail | 2)
void launch_sbt_tail(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, int inverse, Prof *pf = nullptr);
void launch_fwd_mid4(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, bool llq, Prof *pf = nullptr);   // levels 4..5: launch_fwd_sbt does it itself unless fused >= 3
void launch_tail_q(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, int c0, int npl, Prof *pf = nullptr);   // fused >= 2 pictures: forward tail + LL quantiser + inverse tail
// k_hzcc.hip
void launch_hz_encode(hipStream_t st, const JobDev *jobs, int njobs, int job_chunks, Prof *pf = nullptr, double samples = 0,
                      int nplain = -1, int ll_chunks = 1);
void launch_hz_quant(hipStream_t st, const JobDev *jobs, int njobs, int job_chunks, Prof *pf, double samples, int nplain, int ll_chunks);
void launch_hz_pack(hipStream_t st, const JobDev *jobs, int njobs, int job_chunks, Prof *pf, double samples, int nplain, int ndense = -1);
void launch_hz_parse_scatter(hipStream_t st, JobDev *jobs, int njobs, int c, int nplanes, int max_entries, int max_chunks, Prof *pf = nullptr, bool all_sparse = false);
void launch_dec_clear(hipStream_t st, const JobDev *jobs, int njobs);                       // decoder: zero what the scatter leaves untouched
void launch_mc_patch(hipStream_t st, const JobDev *jobs, int njobs, const SbtGeo3 &G, const McGeo &MG, const DMV *mvs0, int ex0, int ey0, Prof *pf);
void launch_hz_dec_resolve(hipStream_t st, const JobDev *jobs, int njobs, int c0, int nplanes);
void launch_hz_unscatter(hipStream_t st, const JobDev *jobs, int njobs, int max_entries);   // decoder: take the scattered symbols down again
int  hz_scan_items_max();
void launch_gather_bits(hipStream_t st, const uint8_t *bits, const unsigned long long *tab, int nitems, uint8_t *dst);
// k_rc.hip: device-resident ABR -- mode 0: quantiser of jobs [d0, d0 + n) from their streams' state; mode 1: after their k_hz_scan,
// packet size -> statistics -> quantiser tables of each stream's next job (RcJobDev.next)
void launch_rc(hipStream_t st, JobDev *jobs, const RcJobDev *rcj, dsvg_rc_state *state, int d0, int n, int mode);
// k_bmc.hip
// mvs0: the jobs' vector arrays when they are contiguous (job j at mvs0 + j * nblocks), else null (JobDev.mvs is used)
void launch_mc(hipStream_t st, const JobDev *jobs, int njobs, const McGeo &G, int do_sub, Prof *pf = nullptr, const DMV *mvs0 = nullptr,
               const int *list = nullptr, int nlist = 0);   // list: only these blocks (job * nblocks + block)
// k_frame.hip
int  unpack_fuses_level1(const FrameLayout &L);
void launch_unpack(hipStream_t st, const uint8_t *yuv, size_t yuv_pitch, uint8_t *slab, const FrameLayout &L, int first, int n, Prof *pf = nullptr, const int *slot_tab = nullptr,
                   uint8_t *slab1 = nullptr, const FrameLayout *L1 = nullptr, bool sides = false, bool sides1 = false,
                   uint8_t *slab2 = nullptr, const FrameLayout *L2 = nullptr, bool sides2 = false, int n_chroma = -1,
                   int ring_x16 = 0, int ring_y4 = 0, int n_ring = 0);    // slot-table entries with bit 29: only the luma plane's outer ring_x16 x 16 columns / ring_y4 x 4 rows are copied
int  unpack_fuses_level2(const FrameLayout &L, const FrameLayout &L1, const FrameLayout &L2);
bool unpack_writes_sides(const uint8_t *yuv, size_t yuv_pitch, const uint8_t *slab, const FrameLayout &L);
void launch_pack(hipStream_t st, uint8_t *yuv, const uint8_t *frame, const FrameLayout &L);
void launch_pack_n(hipStream_t st, uint8_t *yuv, size_t out_pitch, const uint8_t *slab, const FrameLayout &L, const int *slot_tab, int n, Prof *pf = nullptr);
void launch_extend(hipStream_t st, uint8_t *slab, const FrameLayout &L, int first, int n, int nplanes, const int *slot_tab, Prof *pf = nullptr, const JobDev *jobs = nullptr, bool tb_only = false);
void launch_ds2x(hipStream_t st, const uint8_t *sslab, const FrameLayout &SL, uint8_t *dslab, const FrameLayout &DL, int first, int n, Prof *pf = nullptr, const int *slot_tab = nullptr, bool sides = false);
bool level_sides_ok(const uint8_t *slab, const FrameLayout &L);
void launch_luma_sum(hipStream_t st, const uint8_t *slab, const FrameLayout &L, int first, int n, unsigned *sums, Prof *pf = nullptr, const int *slot_tab = nullptr);
void launch_frame_add(hipStream_t st, uint8_t *dst, const FrameLayout &DL, const uint8_t *src, const FrameLayout &SL);
// k_hme.hip
void launch_hme(hipStream_t st, const HmeArgs &A, int npairs, Prof *pf = nullptr);
