// k_frame.hip -- frame plumbing kernels for gfx950: planar -> bordered layout, border replication,
// 2x luma box downsample for the motion pyramid, mean luma.  Pure HBM streaming.
//
// Replaces dsv_frame_copy/dsv_clone_frame (frame.c:166-221), dsv_extend_frame(_luma)
// (frame.c:263-327), dsv_ds2x_frame_luma (frame.c:240-261), dsv_frame_avg_luma (frame.c:223-238).
#include "dsvg_dev.hpp"
#include "dsvg_kernels.hpp"
#include "dsvg_host.hpp"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// tightly packed planar frames -> interiors of bordered frames (grid.y = plane, grid.z = frame)
__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ yuv, size_t yuv_pitch,
                                                uint8_t *__restrict__ slab, FrameLayout L, int first_slot,
                                                const int *__restrict__ slot_tab, uint8_t *__restrict__ slab1, FrameLayout L1, int sides_ring,
                                                uint8_t *__restrict__ slab2, FrameLayout L2)
{
    const int sides = sides_ring & 7;                   // (bits 8..23: the ring's extents, below)
    // sides: the 64-byte left / right borders of every row are written here too (launch_unpack checked that every plane
    // takes a 16-byte path): they complete the 128-byte lines the row's first and last pixels lie in, where a separate
    // border kernel writes half lines; the rows above and below the picture are left to k_extend16
    const int c = blockIdx.y, f = blockIdx.z;
    // slot table entries: bit 30 = this frame's chroma stays where the caller has it (dsvg_load_frames_map_ex, "in place"):
    // only its luma plane is unpacked, bordered and fed to the pyramid
    const int raw = slot_tab ? slot_tab[f] : first_slot + f;
    const int slot = raw & 0x1fffffff;
    if ((raw & 0x40000000) && c > 0) return;
    // bit 29 (round 5): the frame's LUMA stays in the caller's clip too -- of the bordered copy only a ring is written, the picture's outer
    // (sides_ring >> 8 & 0xff) x 16 columns and (sides_ring >> 16 & 0xff) x 4 rows (+ the border itself): what the level-0 motion search reads of a
    // frame through the bordered layout (blocks near an edge, whose candidates may leave the picture); every other block reads the clip
    // (hme_block, `deep`), and so do the forward transforms (JobDev.srcp).  The pyramid levels are produced as ever.
    const bool ring = (raw & 0x20000000) != 0;
    const int ring_x = (sides_ring >> 8) & 0xff, ring_y = (sides_ring >> 16) & 0xff;
    const int w = L.w[c], h = L.h[c];
    size_t poff = 0;
    for (int k = 0; k < c; k++) poff += (size_t)L.w[k] * L.h[k];
    const uint8_t *src = yuv + (size_t)f * yuv_pitch + poff;
    uint8_t *dst = slab + (size_t)slot * L.pitch + L.off[c];
    const bool vec = ((w & 15) == 0) && ((((uintptr_t)src) & 15) == 0);
    if (vec && c == 0 && slab1 && slab2 && (h & 3) == 0) {
        // luma with the first TWO pyramid levels fused in: four rows x 16 pixels per thread -> two rows x 8 of level 1 and
        // one row x 4 of level 2 (each level the rounded 2x2 mean of the one below, frame.c:240-261); sides & 4: level 2's
        // side borders too
        uint8_t *dst1 = slab1 + (size_t)slot * L1.pitch + L1.off[0];
        uint8_t *dst2 = slab2 + (size_t)slot * L2.pitch + L2.off[0];
        const int nv = w >> 4, hq = h >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nv * hq; i += gridDim.x * 256) {
            const int y4 = i / nv, x = i - y4 * nv;
            u32x4 q[4];
#pragma unroll
            for (int r = 0; r < 4; r++) q[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + (size_t)(4 * y4 + r) * w) + x);
            if (!ring || ring_x == 0 || x < ring_x || x >= nv - ring_x || y4 < ring_y || y4 >= hq - ring_y) {
#pragma unroll
                for (int r = 0; r < 4; r++) __builtin_nontemporal_store(q[r], reinterpret_cast<u32x4 *>(dst + (size_t)(4 * y4 + r) * L.stride[0]) + x);
            }
            const bool edge = x == 0 || x == nv - 1;
            if (sides && edge) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned v = (x == 0 ? (q[r].x & 0xff) : (q[r].w >> 24)) * 0x01010101u;
                    uint4 *b = reinterpret_cast<uint4 *>(dst + (size_t)(4 * y4 + r) * L.stride[0] + (x == 0 ? -DSVG_BORDER : w));
#pragma unroll
                    for (int k = 0; k < DSVG_BORDER / 16; k++) b[k] = make_uint4(v, v, v, v);
                }
            }
            if (sides && nv == 1) {                              // (a 16-pixel-wide plane: the one column is first and last)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const unsigned v = (q[r].w >> 24) * 0x01010101u;
                    uint4 *b = reinterpret_cast<uint4 *>(dst + (size_t)(4 * y4 + r) * L.stride[0] + w);
#pragma unroll
                    for (int k = 0; k < DSVG_BORDER / 16; k++) b[k] = make_uint4(v, v, v, v);
                }
            }
            unsigned o1[2][2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const unsigned a[4] = {q[2 * p].x, q[2 * p].y, q[2 * p].z, q[2 * p].w}, b[4] = {q[2 * p + 1].x, q[2 * p + 1].y, q[2 * p + 1].z, q[2 * p + 1].w};
                o1[p][0] = o1[p][1] = 0u;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const unsigned wa = a[k >> 1], wb = b[k >> 1];
                    const int sh = 16 * (k & 1);
                    const unsigned sum = ((wa >> sh) & 0xff) + ((wa >> (sh + 8)) & 0xff) + ((wb >> sh) & 0xff) + ((wb >> (sh + 8)) & 0xff);
                    o1[p][k >> 2] |= ((sum + 2) >> 2) << (8 * (k & 3));
                }
                *reinterpret_cast<uint2 *>(dst1 + (size_t)(2 * y4 + p) * L1.stride[0] + 8 * x) = make_uint2(o1[p][0], o1[p][1]);
                if (sides > 1 && (sides & 2) && edge) {
                    const unsigned v = (x == 0 ? (o1[p][0] & 0xff) : (o1[p][1] >> 24)) * 0x01010101u;
                    uint4 *bb = reinterpret_cast<uint4 *>(dst1 + (size_t)(2 * y4 + p) * L1.stride[0] + (x == 0 ? -DSVG_BORDER : L1.w[0]));
#pragma unroll
                    for (int k = 0; k < DSVG_BORDER / 16; k++) bb[k] = make_uint4(v, v, v, v);
                }
            }
            unsigned o2 = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned wa = o1[0][k >> 1], wb = o1[1][k >> 1];
                const int sh = 16 * (k & 1);
                const unsigned sum = ((wa >> sh) & 0xff) + ((wa >> (sh + 8)) & 0xff) + ((wb >> sh) & 0xff) + ((wb >> (sh + 8)) & 0xff);
                o2 |= ((sum + 2) >> 2) << (8 * k);
            }
            *reinterpret_cast<unsigned *>(dst2 + (size_t)y4 * L2.stride[0] + 4 * x) = o2;
            if ((sides & 4) && edge) {
                const unsigned v = (x == 0 ? (o2 & 0xff) : (o2 >> 24)) * 0x01010101u;
                uint4 *bb = reinterpret_cast<uint4 *>(dst2 + (size_t)y4 * L2.stride[0] + (x == 0 ? -DSVG_BORDER : L2.w[0]));
#pragma unroll
                for (int k = 0; k < DSVG_BORDER / 16; k++) bb[k] = make_uint4(v, v, v, v);
            }
        }
        return;
    }
    if (vec && c == 0 && slab1 && (h & 1) == 0) {
        // luma with the first pyramid level fused in (dsv_ds2x_frame_luma frame.c:240-261: (p1+p2+p3+p4+2)>>2): two rows x 16
        // pixels per thread, so the bordered frame is not read again for the 2x2 means (even dims: no border pixel enters)
        uint8_t *dst1 = slab1 + (size_t)slot * L1.pitch + L1.off[0];
        const int nv = w >> 4, hr = h >> 1;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nv * hr; i += gridDim.x * 256) {
            const int y2 = i / nv, x = i - y2 * nv;
            // non-temporal on both sides: 6 GB per step stream through here once -- kept out of the caches they do not evict what
            // the coding streams' kernels re-read (23.9 -> 23.0 ms per step, the gain is in the other kernels)
            const u32x4 q0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + (size_t)(2 * y2) * w) + x);
            const u32x4 q1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + (size_t)(2 * y2 + 1) * w) + x);
            const uint4 r0 = make_uint4(q0.x, q0.y, q0.z, q0.w), r1 = make_uint4(q1.x, q1.y, q1.z, q1.w);
            __builtin_nontemporal_store(q0, reinterpret_cast<u32x4 *>(dst + (size_t)(2 * y2) * L.stride[0]) + x);
            __builtin_nontemporal_store(q1, reinterpret_cast<u32x4 *>(dst + (size_t)(2 * y2 + 1) * L.stride[0]) + x);
            if (sides && (x == 0 || x == nv - 1)) {
                const unsigned v0 = (x == 0 ? (r0.x & 0xff) : (r0.w >> 24)) * 0x01010101u, v1 = (x == 0 ? (r1.x & 0xff) : (r1.w >> 24)) * 0x01010101u;
                uint4 *b0 = reinterpret_cast<uint4 *>(dst + (size_t)(2 * y2) * L.stride[0] + (x == 0 ? -DSVG_BORDER : w));
                uint4 *b1 = reinterpret_cast<uint4 *>(dst + (size_t)(2 * y2 + 1) * L.stride[0] + (x == 0 ? -DSVG_BORDER : w));
#pragma unroll
                for (int k = 0; k < DSVG_BORDER / 16; k++) { b0[k] = make_uint4(v0, v0, v0, v0); b1[k] = make_uint4(v1, v1, v1, v1); }
            }
            if (sides && nv == 1) {                              // (a 16-pixel-wide plane: the one column is first and last)
                const unsigned v0 = (r0.w >> 24) * 0x01010101u, v1 = (r1.w >> 24) * 0x01010101u;
                uint4 *b0 = reinterpret_cast<uint4 *>(dst + (size_t)(2 * y2) * L.stride[0] + w);
                uint4 *b1 = reinterpret_cast<uint4 *>(dst + (size_t)(2 * y2 + 1) * L.stride[0] + w);
#pragma unroll
                for (int k = 0; k < DSVG_BORDER / 16; k++) { b0[k] = make_uint4(v0, v0, v0, v0); b1[k] = make_uint4(v1, v1, v1, v1); }
            }
            const unsigned a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
            unsigned o[2] = {0u, 0u};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned wa = a[k >> 1], wb = b[k >> 1];
                const int sh = 16 * (k & 1);
                const unsigned sum = ((wa >> sh) & 0xff) + ((wa >> (sh + 8)) & 0xff) + ((wb >> sh) & 0xff) + ((wb >> (sh + 8)) & 0xff);
                o[k >> 2] |= ((sum + 2) >> 2) << (8 * (k & 3));
            }
            *reinterpret_cast<uint2 *>(dst1 + (size_t)y2 * L1.stride[0] + 8 * x) = make_uint2(o[0], o[1]);
            if (sides > 1 && (x == 0 || x == nv - 1)) {          // (sides & 2: the first pyramid level's rows get theirs too)
                const unsigned v = (x == 0 ? (o[0] & 0xff) : (o[1] >> 24)) * 0x01010101u;
                uint4 *b = reinterpret_cast<uint4 *>(dst1 + (size_t)y2 * L1.stride[0] + (x == 0 ? -DSVG_BORDER : L1.w[0]));
#pragma unroll
                for (int k = 0; k < DSVG_BORDER / 16; k++) b[k] = make_uint4(v, v, v, v);
            }
        }
        return;
    }
    if (vec) {
        const int nv = w >> 4;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nv * h; i += gridDim.x * 256) {
            const int y = i / nv, x = i - y * nv;
            const u32x4 qv = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + (size_t)y * w) + x);
            const uint4 r = make_uint4(qv.x, qv.y, qv.z, qv.w);
            __builtin_nontemporal_store(qv, reinterpret_cast<u32x4 *>(dst + (size_t)y * L.stride[c]) + x);
            if (sides && x == 0) {
                const unsigned v = (r.x & 0xff) * 0x01010101u;
                uint4 *b = reinterpret_cast<uint4 *>(dst + (size_t)y * L.stride[c] - DSVG_BORDER);
#pragma unroll
                for (int k = 0; k < DSVG_BORDER / 16; k++) b[k] = make_uint4(v, v, v, v);
            }
            if (sides && x == nv - 1) {
                const unsigned v = (r.w >> 24) * 0x01010101u;
                uint4 *b = reinterpret_cast<uint4 *>(dst + (size_t)y * L.stride[c] + w);
#pragma unroll
                for (int k = 0; k < DSVG_BORDER / 16; k++) b[k] = make_uint4(v, v, v, v);
            }
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
            const int y = i / w, x = i - y * w;
            dst[(size_t)y * L.stride[c] + x] = src[(size_t)y * w + x];
        }
    }
}

// bordered interiors -> tightly packed planar (download path)
__global__ __launch_bounds__(256) void k_pack(uint8_t *__restrict__ yuv, const uint8_t *__restrict__ frame, FrameLayout L)
{
    const int c = blockIdx.y;
    const int w = L.w[c], h = L.h[c];
    size_t poff = 0;
    for (int k = 0; k < c; k++) poff += (size_t)L.w[k] * L.h[k];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        yuv[poff + (size_t)y * w + x] = frame[L.off[c] + (size_t)y * L.stride[c] + x];
    }
}

// batched form for the decoder: reconstruction slot slot_tab[z] -> packed planar frame z of the output (16 bytes per
// lane when every plane row is 16-byte aligned on both sides)
template <bool V16>
__global__ __launch_bounds__(256) void k_pack_n(uint8_t *__restrict__ yuv, size_t out_pitch, const uint8_t *__restrict__ slab, FrameLayout L,
                                                const int *__restrict__ slot_tab)
{
    const int c = blockIdx.y;
    const int f = slot_tab[blockIdx.z];
    const int w = L.w[c], h = L.h[c], s = L.stride[c];
    size_t poff = 0;
    for (int k = 0; k < c; k++) poff += (size_t)L.w[k] * L.h[k];
    const uint8_t *src = slab + (size_t)f * L.pitch + L.off[c];
    uint8_t *dst = yuv + (size_t)blockIdx.z * out_pitch + poff;
    if (V16) {
        const int wq = w >> 4, n = wq * h;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const int y = i / wq, x = (i - y * wq) << 4;
            *reinterpret_cast<uint4 *>(dst + (size_t)y * w + x) = *reinterpret_cast<const uint4 *>(src + (size_t)y * s + x);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
            const int y = i / w, x = i - y * w;
            dst[(size_t)y * w + x] = src[(size_t)y * s + x];
        }
    }
}

// border replication of planes [0, nplanes) of frames: every border byte is the nearest interior pixel
// (identical to the row memsets + row copies of frame.c:278-292).  One dword per thread-iteration:
// the 64-byte left/right borders of all h+128 rows, then the top/bottom 64 rows over the interior.
__global__ __launch_bounds__(256) void k_extend(uint8_t *__restrict__ slab, FrameLayout L, int first, int nplanes,
                                                const int *__restrict__ slot_tab)
{
    const int c = blockIdx.y;
    if (c >= nplanes) return;
    const int raw = slot_tab ? slot_tab[blockIdx.z] : first + (int)blockIdx.z;
    if (raw < 0) return;
    const int f = raw & 0x1fffffff;
    if ((raw & 0x40000000) && c > 0) return;            // (source frames whose chroma stays in the caller's clip: see k_unpack)
    const int w = L.w[c], h = L.h[c], s = L.stride[c];
    uint8_t *p = slab + (size_t)f * L.pitch + L.off[c];
    const int B = DSVG_BORDER;
    const bool al = (w & 3) == 0;                     // right border / interior rows dword-aligned
    // part 1: side borders: (h + 2B) rows x 32 dwords (16 left, 16 right)
    const int n1 = (h + 2 * B) * 32;
    // part 2: top + bottom B rows over the interior columns, ceil(w/4) dwords per row
    const int wq = (w + 3) >> 2;
    const int n2 = 2 * B * wq;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n1 + n2; i += gridDim.x * 256) {
        if (i < n1) {
            const int r = i >> 5, k = i & 31;
            const int y = r - B;
            const int sy = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            if (k < 16) {
                const unsigned v = p[(long)sy * s] * 0x01010101u;
                *reinterpret_cast<unsigned *>(p + (long)y * s - B + 4 * k) = v;
            } else {
                const unsigned v = p[(long)sy * s + w - 1] * 0x01010101u;
                uint8_t *d = p + (long)y * s + w + 4 * (k - 16);
                if (al) *reinterpret_cast<unsigned *>(d) = v;
                else { d[0] = (uint8_t)v; d[1] = (uint8_t)v; d[2] = (uint8_t)v; d[3] = (uint8_t)v; }
            }
        } else {
            const int j = i - n1;
            const int r = j / wq, x = 4 * (j - r * wq);
            const int y = r < B ? r - B : h + (r - B);
            const uint8_t *src = p + (long)(r < B ? 0 : h - 1) * s + x;
            uint8_t *d = p + (long)y * s + x;
            if (x + 4 <= w) *reinterpret_cast<unsigned *>(d) = *reinterpret_cast<const unsigned *>(src);
            else for (int q = 0; x + q < w; q++) d[q] = src[q];
        }
    }
}

// same for planes whose width is a multiple of 16 (every store a 16-byte one).  A thread has one load round trip and then
// several stores behind it (one store per thread left the kernel waiting on 94 k wave launches of one round trip each):
// side borders: 4 rows of one 16-byte column of the 128 border bytes of a row; top / bottom: 8 rows of a 16-byte column
// (non-temporal stores here, as in k_unpack, were measured slower: 0.77 -> 0.87 ms, nothing gained elsewhere)
#define EXT_ST(ptr, val) (*reinterpret_cast<uint4 *>(ptr) = (val))
#define EXT16_SIDE_ROWS 4
#define EXT16_TB_ROWS 8
// `jobs` (encoder): frame z is the reconstruction of jobs[z], whose ext[] says how far the pictures that predict from it
// reach into the border (dsvg_code_batch works that out from their motion vectors) -- only that much is written: columns
// in units of 16, rows in units of EXT16_TB_ROWS.  null: the whole border.
__global__ __launch_bounds__(256) void k_extend16(uint8_t *__restrict__ slab, FrameLayout L, int first, int nplanes,
                                                  const int *__restrict__ slot_tab, const JobDev *__restrict__ jobs, int tb_only)
{
    // tb_only: the side borders of the picture rows exist already (k_unpack): only the rows above and below, over the whole
    // width of the allocation (corners included), as copies of the bordered first / last row
    const int c = blockIdx.y;
    if (c >= nplanes) return;
    const int raw = slot_tab ? slot_tab[blockIdx.z] : first + (int)blockIdx.z;
    if (raw < 0) return;
    const int f = raw & 0x1fffffff;
    if ((raw & 0x40000000) && c > 0) return;            // (source frames whose chroma stays in the caller's clip: see k_unpack)
    const int w = L.w[c], h = L.h[c], s = L.stride[c];
    uint8_t *p = slab + (size_t)f * L.pitch + L.off[c];
    const int B = DSVG_BORDER;
    int el = B, er = B, et = B, eb = B;
    if (jobs) {
        const short *e = jobs[blockIdx.z].ext + (c ? 4 : 0);
        el = min(B, (e[0] + 15) & ~15); er = min(B, (e[1] + 15) & ~15);
        et = min(B, (e[2] + EXT16_TB_ROWS - 1) & ~(EXT16_TB_ROWS - 1)); eb = min(B, (e[3] + EXT16_TB_ROWS - 1) & ~(EXT16_TB_ROWS - 1));
    }
    const int nc = tb_only ? 0 : (el + er) >> 4, ncl = el >> 4;      // 16-byte columns of the side borders of a row
    const int rg = (h + et + eb + EXT16_SIDE_ROWS - 1) / EXT16_SIDE_ROWS;
    const int n1 = rg * nc;
    const int wq = (w + (tb_only ? 2 * B : 0)) >> 4, xo = tb_only ? -B : 0;
    const int n2 = ((et + eb) / EXT16_TB_ROWS) * wq;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n1 + n2; i += gridDim.x * 256) {
        if (i < n1) {
            const int g = i / nc, k = i - g * nc, r0 = g * EXT16_SIDE_ROWS - et;      // first of this item's rows
            const int sx = k < ncl ? 0 : w - 1, dx = k < ncl ? -el + 16 * k : w + 16 * (k - ncl);
            unsigned v[EXT16_SIDE_ROWS];
#pragma unroll
            for (int j = 0; j < EXT16_SIDE_ROWS; j++) {
                const int y = r0 + j;
                const int sy = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
                v[j] = p[(long)sy * s + sx] * 0x01010101u;
            }
#pragma unroll
            for (int j = 0; j < EXT16_SIDE_ROWS; j++) {
                const int y = r0 + j;
                if (y < h + eb) EXT_ST(p + (long)y * s + dx, make_uint4(v[j], v[j], v[j], v[j]));
            }
        } else {
            const int j = i - n1;
            const int g = j / wq, x = xo + 16 * (j - g * wq);
            const int r0 = g * EXT16_TB_ROWS;                        // 0 .. et+eb-1: the rows above first, then the rows below
            const bool top = r0 < et;
            const uint4 v = *reinterpret_cast<const uint4 *>(p + (long)(top ? 0 : h - 1) * s + x);
            uint8_t *d = p + (long)(top ? r0 - et : h + (r0 - et)) * s + x;
#pragma unroll
            for (int q = 0; q < EXT16_TB_ROWS; q++) EXT_ST(d + (long)q * s, v);
        }
    }
}

// 2x2 box downsample of the luma plane: (p1+p2+p3+p4+2)>>2 (frame.c:240-261)
__global__ __launch_bounds__(256) void k_ds2x(const uint8_t *__restrict__ sslab, FrameLayout SL,
                                              uint8_t *__restrict__ dslab, FrameLayout DL, int first,
                                              const int *__restrict__ slot_tab, int sides)
{
    // sides: the level's width is a multiple of 16: the threads of a row's first and last pixels also write its 64-byte side
    // borders (whole 128-byte lines; k_extend16 then adds the rows above and below only)
    const int f = (slot_tab ? slot_tab[blockIdx.z] : first + (int)blockIdx.z) & 0x1fffffff;
    const uint8_t *sp = sslab + (size_t)f * SL.pitch + SL.off[0];
    uint8_t *dp = dslab + (size_t)f * DL.pitch + DL.off[0];
    const int dw = DL.w[0], dh = DL.h[0];
    const int nq = (dw + 3) >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nq * dh; i += gridDim.x * 256) {
        const int y = i / nq, x4 = 4 * (i - y * nq);
        const uint8_t *a = sp + (size_t)(2 * y) * SL.stride[0] + 2 * x4;
        const uint2 r0 = *reinterpret_cast<const uint2 *>(a);
        const uint2 r1 = *reinterpret_cast<const uint2 *>(a + SL.stride[0]);
        unsigned out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned w0 = k < 2 ? r0.x : r0.y, w1 = k < 2 ? r1.x : r1.y;
            const int shf = 16 * (k & 1);
            const int v = (int)((w0 >> shf) & 0xff) + (int)((w0 >> (shf + 8)) & 0xff) +
                          (int)((w1 >> shf) & 0xff) + (int)((w1 >> (shf + 8)) & 0xff);
            out |= (unsigned)((v + 2) >> 2) << (8 * k);
        }
        uint8_t *d = dp + (size_t)y * DL.stride[0] + x4;
        if (x4 + 4 <= dw) *reinterpret_cast<unsigned *>(d) = out;
        else
            for (int k = 0; k < 4; k++)
                if (x4 + k < dw) d[k] = (uint8_t)(out >> (8 * k));
        if (sides && (x4 == 0 || x4 + 4 == dw)) {
            const unsigned v = (x4 == 0 ? (out & 0xff) : (out >> 24)) * 0x01010101u;
            uint4 *b = reinterpret_cast<uint4 *>(dp + (size_t)y * DL.stride[0] + (x4 == 0 ? -DSVG_BORDER : dw));
#pragma unroll
            for (int k = 0; k < DSVG_BORDER / 16; k++) b[k] = make_uint4(v, v, v, v);
            if (dw == 4) {                                           // (one thread holds both ends)
                const unsigned v2 = (out >> 24) * 0x01010101u;
                uint4 *b2 = reinterpret_cast<uint4 *>(dp + (size_t)y * DL.stride[0] + dw);
#pragma unroll
                for (int k = 0; k < DSVG_BORDER / 16; k++) b2[k] = make_uint4(v2, v2, v2, v2);
            }
        }
    }
}

// sum of the luma plane of frames [first, first+n) -> sums[first+f] (host divides, frame.c:237)
__global__ __launch_bounds__(256) void k_luma_sum(const uint8_t *__restrict__ slab, FrameLayout L, int first,
                                                  unsigned *__restrict__ sums, const int *__restrict__ slot_tab)
{
    const int f = (slot_tab ? slot_tab[blockIdx.z] : first + (int)blockIdx.z) & 0x1fffffff;
    const uint8_t *p = slab + (size_t)f * L.pitch + L.off[0];
    const int w = L.w[0], h = L.h[0];
    unsigned acc = 0;
    if (((w | L.stride[0]) & 3) == 0 && (((uintptr_t)p) & 3) == 0) {
        // four samples per load and per v_sad_u8 (against zero: the sum of the four bytes)
        const int wq = w >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < wq * h; i += gridDim.x * 256) {
            const int y = i / wq, x = i - y * wq;
            acc = __builtin_amdgcn_sad_u8(*reinterpret_cast<const unsigned *>(p + (size_t)y * L.stride[0] + 4 * x), 0u, acc);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
            const int y = i / w, x = i - y * w;
            acc += p[(size_t)y * L.stride[0] + x];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&sums[f], acc);
}

// dst = clamp(dst + src - 128) over the plane interiors (dsv_frame_add / addf bmc.c:29-41,304-316)
__global__ __launch_bounds__(256) void k_frame_add(uint8_t *__restrict__ dst, FrameLayout DL,
                                                   const uint8_t *__restrict__ src, FrameLayout SL)
{
    const int c = blockIdx.y;
    const int w = DL.w[c], h = DL.h[c];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        uint8_t *d = dst + DL.off[c] + (size_t)y * DL.stride[c] + x;
        const int v = (int)*d + (int)src[SL.off[c] + (size_t)y * SL.stride[c] + x] - 128;
        *d = (uint8_t)d_sat8(v);
    }
}

static inline int nblk(long items, int cap) { long b = (items + 255) / 256; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

// slab1 / L1: when given (and the luma plane qualifies, see unpack_fuses_level1) the first pyramid level is produced
// by the same kernel
int unpack_fuses_level1(const FrameLayout &L) { return (L.w[0] & 15) == 0 && (L.h[0] & 1) == 0; }
// ... and the second one too: four source rows per thread, both levels exact halvings
int unpack_fuses_level2(const FrameLayout &L, const FrameLayout &L1, const FrameLayout &L2)
{
    return unpack_fuses_level1(L) && (L.h[0] & 3) == 0 && L1.w[0] * 2 == L.w[0] && L1.h[0] * 2 == L.h[0] && L2.w[0] * 4 == L.w[0] && L2.h[0] * 4 == L.h[0] &&
           (L2.stride[0] & 3) == 0 && (L2.off[0] & 3) == 0 && (L2.pitch & 3) == 0;
}
// can k_unpack write the side borders (and launch_extend be told tb_only)?  every plane on a 16-byte path of k_unpack and
// of k_extend16
bool unpack_writes_sides(const uint8_t *yuv, size_t yuv_pitch, const uint8_t *slab, const FrameLayout &L)
{
    bool ok = ((uintptr_t)yuv % 16) == 0 && (yuv_pitch % 16) == 0 && ((uintptr_t)slab % 16) == 0 && (L.pitch % 16) == 0 &&
              ((size_t)L.w[0] * L.h[0] % 16) == 0 && ((size_t)L.w[1] * L.h[1] % 16) == 0;
    for (int c = 0; c < 3; c++) ok = ok && (L.w[c] % 16) == 0 && (L.stride[c] % 16) == 0 && (L.off[c] % 16) == 0;
    return ok;
}
void launch_unpack(hipStream_t st, const uint8_t *yuv, size_t yuv_pitch, uint8_t *slab, const FrameLayout &L, int first, int n, Prof *pf, const int *slot_tab,
                   uint8_t *slab1, const FrameLayout *L1, bool sides, bool sides1, uint8_t *slab2, const FrameLayout *L2, bool sides2, int n_chroma, int ring_x16, int ring_y4, int n_ring)
{
    FrameLayout dummy = L;
    // algorithmic bytes: every copied plane read once and written once (n_chroma: the frames whose chroma is copied too -- the
    // others keep it in the caller's clip), + the pyramid levels written
    if (n_chroma < 0) n_chroma = n;
    // (n_ring, round 5: the frames whose luma stays in the clip as well -- read once for the pyramid, written only as the ring)
    double ring_saved = 0.0;
    if (n_ring > 0 && ring_x16 > 0) {
        const double iw = std::max(0, L.w[0] - 2 * 16 * ring_x16), ih = std::max(0, L.h[0] - 2 * 4 * ring_y4);
        ring_saved = (double)n_ring * iw * ih;                              // the interior that is not written
    }
    if (pf) pf->begin(st, KID_UNPACK, 2.0 * ((double)n * L.w[0] * L.h[0] + 2.0 * n_chroma * L.w[1] * L.h[1]) - ring_saved + (slab1 ? 0.25 * n * L.w[0] * L.h[0] : 0.0) + (slab2 ? 0.0625 * n * L.w[0] * L.h[0] : 0.0));
    // The grid is sized for the work there is (the kernel's loops stride by the grid, so any size is correct): the fused luma body takes
    // 4 rows x 16 pixels per thread, and a batch whose chroma stays in the caller's clip has no chroma planes to copy -- sized for 16
    // pixels per thread and three planes, five workgroups in six of the bench's launch found nothing to do (2.4 million per step).
    const bool fused = slab1 && slab2 && (L.w[0] & 15) == 0 && (L.h[0] & 3) == 0;
    const long items = fused ? (long)(L.w[0] >> 4) * (L.h[0] >> 2) : (long)L.w[0] * L.h[0] / 16;
    const long citems = n_chroma > 0 ? (long)L.w[1] * L.h[1] / 16 : 0;
    hipLaunchKernelGGL(k_unpack, dim3(nblk(std::max(items, citems), 512), n_chroma > 0 ? 3 : 1, n), dim3(256), 0, st, yuv, yuv_pitch, slab, L, first, slot_tab,
                       slab1, L1 ? *L1 : dummy, (sides ? 1 : 0) | (sides && sides1 ? 2 : 0) | (sides && sides2 ? 4 : 0) | ((ring_x16 & 0xff) << 8) | ((ring_y4 & 0xff) << 16), slab2, L2 ? *L2 : dummy);
    if (pf) pf->end(st);
}
void launch_pack(hipStream_t st, uint8_t *yuv, const uint8_t *frame, const FrameLayout &L)
{
    hipLaunchKernelGGL(k_pack, dim3(nblk((long)L.w[0] * L.h[0], 1024), 3, 1), dim3(256), 0, st, yuv, frame, L);
}
void launch_pack_n(hipStream_t st, uint8_t *yuv, size_t out_pitch, const uint8_t *slab, const FrameLayout &L, const int *slot_tab, int n, Prof *pf)
{
    bool v16 = (L.pitch % 16) == 0 && (out_pitch % 16) == 0 && ((uintptr_t)yuv % 16) == 0 && ((uintptr_t)slab % 16) == 0;
    for (int c = 0; c < 3; c++) v16 = v16 && (L.w[c] % 16) == 0 && (L.stride[c] % 16) == 0 && (L.off[c] % 16) == 0 && ((size_t)L.w[c] * L.h[c] % 16) == 0;
    const double fb = (double)L.w[0] * L.h[0] + 2.0 * L.w[1] * L.h[1];
    if (pf) pf->begin(st, v16 ? KID_PACK16 : KID_PACK, 2.0 * fb * n);
    if (v16) hipLaunchKernelGGL(k_pack_n<true>, dim3(nblk((long)L.w[0] * L.h[0] / 16, 256), 3, n), dim3(256), 0, st, yuv, out_pitch, slab, L, slot_tab);
    else     hipLaunchKernelGGL(k_pack_n<false>, dim3(nblk((long)L.w[0] * L.h[0], 1024), 3, n), dim3(256), 0, st, yuv, out_pitch, slab, L, slot_tab);
    if (pf) pf->end(st);
}
void launch_extend(hipStream_t st, uint8_t *slab, const FrameLayout &L, int first, int n, int nplanes, const int *slot_tab, Prof *pf, const JobDev *jobs, bool tb_only)
{
    const long items = (long)(L.h[0] + 128) * 32 + 32L * L.w[0];        // dwords of the luma border
    bool v16 = (L.pitch % 16) == 0;
    for (int c = 0; c < nplanes; c++) v16 = v16 && (L.w[c] % 16) == 0 && (L.stride[c] % 16) == 0 && (L.off[c] % 16) == 0;
    if (pf) pf->begin(st, v16 ? KID_EXTEND16 : KID_EXTEND, 8.0 * items * n);
    const long it16 = ((L.h[0] + 128 + EXT16_SIDE_ROWS - 1) / EXT16_SIDE_ROWS) * 8L + (128 / EXT16_TB_ROWS) * (long)(L.w[0] >> 4);
    if (v16) hipLaunchKernelGGL(k_extend16, dim3(nblk(it16, 256), nplanes, n), dim3(256), 0, st, slab, L, first, nplanes, slot_tab, jobs, tb_only ? 1 : 0);
    else     hipLaunchKernelGGL(k_extend, dim3(nblk(items, 256), nplanes, n), dim3(256), 0, st, slab, L, first, nplanes, slot_tab);
    if (pf) pf->end(st);
}
// can the kernel that writes a level's luma rows (k_ds2x, or k_unpack for the first level) write their side borders too?
bool level_sides_ok(const uint8_t *slab, const FrameLayout &L)
{
    return (L.w[0] % 16) == 0 && (L.stride[0] % 16) == 0 && (L.off[0] % 16) == 0 && (L.pitch % 16) == 0 && ((uintptr_t)slab % 16) == 0;
}
void launch_ds2x(hipStream_t st, const uint8_t *sslab, const FrameLayout &SL, uint8_t *dslab, const FrameLayout &DL, int first, int n, Prof *pf, const int *slot_tab, bool sides)
{
    if (pf) pf->begin(st, KID_DS2X, 5.0 * n * (double)DL.w[0] * DL.h[0]);
    hipLaunchKernelGGL(k_ds2x, dim3(nblk((long)DL.w[0] * DL.h[0] / 4, 512), 1, n), dim3(256), 0, st, sslab, SL, dslab, DL, first, slot_tab, sides ? 1 : 0);
    if (pf) pf->end(st);
}
void launch_luma_sum(hipStream_t st, const uint8_t *slab, const FrameLayout &L, int first, int n, unsigned *sums, Prof *pf, const int *slot_tab)
{
    if (pf) pf->begin(st, KID_LUMA_SUM, 1.0 * n * (double)L.w[0] * L.h[0]);
    hipLaunchKernelGGL(k_luma_sum, dim3(nblk((long)L.w[0] * L.h[0] / 32, 64), 1, n), dim3(256), 0, st, slab, L, first, sums, slot_tab);
    if (pf) pf->end(st);
}
void launch_frame_add(hipStream_t st, uint8_t *dst, const FrameLayout &DL, const uint8_t *src, const FrameLayout &SL)
{
    hipLaunchKernelGGL(k_frame_add, dim3(nblk((long)DL.w[0] * DL.h[0], 1024), 3, 1), dim3(256), 0, st, dst, DL, src, SL);
}

