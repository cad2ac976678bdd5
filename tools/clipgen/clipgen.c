/*
 * clipgen.c -- integer-only synthetic YUV clip generator: the input of bench.py, the tests and the golden
 * fixtures.  Not part of the product library and not part of the oracle (it holds no arithmetic of the coded path);
 * built into tools/clipgen/libclipgen.so by __graft_entry__.build().
 *
 * Stateless: frame t of clip (w,h,subsamp,seed,style) is a pure function of its arguments, so
 * the golden fixtures, the CPU baseline and the GPU bench all see identical bytes.
 *   luma   = blend of a 5x5-box-blurred hash-noise texture panning (1.5, 1) px/frame (half-pel
 *            horizontal motion on odd frames), two triangle-wave ramps, and +-1 per-pixel dither
 *   chroma = drifting triangle waves + a little texture
 *   style 1 adds a fast flat square and a flat band whose level changes every frame (these
 *   provoke intra blocks / forced-intra frames on the reference encoder; style 0 does not).
 *   style 2: static background (no pan), a fast TEXTURED bright square (blocks it half covers
 *   become intra with partial sub-block masks) and a global +14 luma step from frame 5 on
 *   (scene-change detection).
 *   style 4: style 1 with a flat square of a third of the frame height and a flat band over its bottom quarter (a third
 *   of all blocks intra, no picture forced intra: the content that leaves the encoder's lean kernels).
 *   style 5: style 3 with a four times faster pan (the blocks' stability accumulators leave zero).
 *   style 6 (round 4): a STATIC noise texture cut into 16x16 cells (the block size of frames up to 352 wide);
 *   a third of the cells change between consecutive frames in a chosen subset of their four 8x8 quadrants (brightness +-60),
 *   the subset being a fixed function of the cell: every partial intra submask of hme.c:689-716 occurs, including "all four
 *   quadrants prefer the zero vector" (the whole cell changes by +-10 over a texture strong enough for intra_metric).
 *   style 7 (round 5): style 0 with STRONG per-pixel noise that changes every frame (+-24 on luma, +-12 on chroma) -- at a high -qp nearly
 *   every 8x8 patch of every P picture carries level-1 symbols: the dense-residual floor of the sparse inverse / list entropy kernels.
 *   style 3: style 0 with SCENE CUTS: every 7 frames the texture is another one and the brightness steps by 12 (the mean
 *   luma of the smallest pyramid level moves by more than the default scene_change_delta of 4: dsv_encoder.c:538-554).
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>

#define RSHIFT_UP(x, s) (((x) + (1 << (s)) - 1) >> (s))   /* DSV_ROUND_SHIFT dsv.h:62 */
#define HSHIFT(fmt) (((fmt) >> 2) & 3)                /* dsv.h:83 */
#define VSHIFT(fmt) ((fmt) & 3)                       /* dsv.h:84 */

static inline uint32_t mix(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}
static inline uint32_t hash3(uint32_t x, uint32_t y, uint32_t s)
{
    return mix(x * 0x9E3779B1u ^ mix(y * 0x85EBCA77u ^ mix(s)));
}
static inline int tri(int v, int period)          /* 0 .. period/2 */
{
    int m = v % period;
    if (m < 0) m += period;
    int half = period / 2;
    return m < half ? m : period - m;
}
static inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

size_t clipgen_frame_bytes(int w, int h, int subsamp)
{
    size_t cw = (size_t)RSHIFT_UP(w, HSHIFT(subsamp)), ch = (size_t)RSHIFT_UP(h, VSHIFT(subsamp));
    return (size_t)w * h + 2 * cw * ch;
}

#define MARGIN 192

void clipgen_frame(uint8_t *out, int w, int h, int subsamp, uint32_t seed, int t, int style)
{
    if (style == 6) {
        const int cw6 = RSHIFT_UP(w, HSHIFT(subsamp)), ch6 = RSHIFT_UP(h, VSHIFT(subsamp));
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int bi = x >> 4, bj = y >> 4;
                int v = 80 + (int)(hash3((uint32_t)x, (uint32_t)y, seed) % 96u);              /* static, strong texture */
                if ((bi + bj) % 3 == 0) {                                                     /* an active cell */
                    const uint32_t hc = hash3((uint32_t)bi, (uint32_t)bj, seed ^ 0x6A5Cu);
                    const int m = (int)(hc & 15u), on = (t + (int)((hc >> 4) & 1u)) & 1;      /* half of the cells are up on even frames, half on odd ones: the frame mean stays (no scene cut) */
                    const int q = ((x >> 3) & 1) | (((y >> 3) & 1) << 1);                     /* quadrant: bit 0 = right, bit 1 = lower */
                    if (m == 0) v += on ? 10 : 0;
                    else if ((m >> q) & 1) v += on ? 60 : 0;
                }
                out[(size_t)y * w + x] = sat8(v);
            }
        memset(out + (size_t)w * h, 128, 2 * (size_t)cw6 * ch6);
        return;
    }
    int lift = 0;                                   /* style 3: brightness step of the scene */
    int big = 0;                                    /* style 4: style 1 with flat objects that cover a third of the frame */
    if (style == 4) { big = 1; style = 1; }
    int fast = 1;                                   /* style 5: style 3 with a pan of (6, 4) pixels per frame -- the stability */
    if (style == 5) { fast = 4; style = 3; }
    int noisy = 0;                                  /* style 7: style 0 with strong noise that is new in every frame */
    if (style == 7) { noisy = 1; style = 0; }        /* accumulators (|mv| >> 2 per block and picture, dsv_encoder.c:367-372) fill up */
    if (style == 3) {
        const int scene = t / 7;
        seed ^= mix(0x5CE9Eu + (uint32_t)scene);
        lift = 12 * (scene % 3);
        style = 0;
    }
    const int TW = w + 2 * MARGIN, TH = h + 2 * MARGIN;
    /* the texture depends on (w, h, seed) only: kept from one frame of a clip to the next (one entry; callers are single
     * threaded -- the tests and bench.py generate their clips up front) */
    static uint8_t *tex_cache = NULL;
    static int cw_ = 0, ch_ = 0;
    static uint32_t cseed_ = 0;
    uint8_t *tex;
    if (tex_cache && cw_ == w && ch_ == h && cseed_ == seed) tex = tex_cache;
    else {
        uint8_t *noise = (uint8_t *)malloc((size_t)TW * TH);
        uint16_t *colsum = (uint16_t *)malloc((size_t)TW * sizeof(uint16_t));
        free(tex_cache);
        tex = tex_cache = (uint8_t *)malloc((size_t)TW * TH);
        cw_ = w; ch_ = h; cseed_ = seed;
        for (int y = 0; y < TH; y++)
            for (int x = 0; x < TW; x++)
                noise[(size_t)y * TW + x] = (uint8_t)(hash3((uint32_t)x, (uint32_t)y, seed) >> 24);
        /* 5x5 box blur, clamped addressing, rounded */
        for (int y = 0; y < TH; y++) {
            for (int x = 0; x < TW; x++) {
                int s = 0;
                for (int dy = -2; dy <= 2; dy++) {
                    int yy = y + dy; yy = yy < 0 ? 0 : (yy >= TH ? TH - 1 : yy);
                    s += noise[(size_t)yy * TW + x];
                }
                colsum[x] = (uint16_t)s;
            }
            for (int x = 0; x < TW; x++) {
                int s = 0;
                for (int dx = -2; dx <= 2; dx++) {
                    int xx = x + dx; xx = xx < 0 ? 0 : (xx >= TW ? TW - 1 : xx);
                    s += colsum[xx];
                }
                /* stretch contrast a little so the texture survives quantisation */
                int v = (s + 12) / 25;
                tex[(size_t)y * TW + x] = sat8(128 + (v - 128) * 3);
            }
        }
        free(noise); free(colsum);
    }

    const int hx = style == 2 ? 0 : 3 * t * fast, ypan = style == 2 ? 0 : t * fast;   /* half-pel x shift, integer y shift */
    const int xo = hx >> 1, xfrac = hx & 1;
    uint8_t *Y = out;
    for (int y = 0; y < h; y++) {
        const uint8_t *tr = tex + (size_t)((y + ypan) % TH + 0) * TW;
        for (int x = 0; x < w; x++) {
            int xi = (x + xo) % (TW - 1);
            int tv = xfrac ? (tr[xi] + tr[xi + 1] + 1) >> 1 : tr[xi];
            int ramp = 64 + (tri(x, 74) * 128 / 37 + tri(y, 46) * 128 / 23) / 2;
            int d = (int)(hash3((uint32_t)x, (uint32_t)y, seed ^ (0xD17Du + (uint32_t)t * 977u)) & 3) - 1;
            if (noisy) d = (int)((hash3((uint32_t)x, (uint32_t)y, seed ^ (0xD17Du + (uint32_t)t * 977u)) >> 8) % 49u) - 24;
            Y[(size_t)y * w + x] = sat8((3 * tv + 2 * ramp) / 5 + d + lift);
        }
    }
    if (style == 1) {
        int sq = big ? h / 3 : (w < 192 ? w / 4 : 96);
        if (sq > h - 8) sq = h - 8;                   /* very flat frames: keep the square inside */
        if (sq > w - 8) sq = w - 8;                   /* ... and narrow ones (style 4's square is a third of the height) */
        int sx = (37 * t) % (w - sq), sy = (23 * t) % (h - sq);
        int lvl = 40 + 15 * t; if (lvl > 250) lvl = 250;
        for (int y = 0; y < sq; y++) memset(Y + (size_t)(sy + y) * w + sx, lvl, (size_t)sq);
        int band = big ? h / 4 : h / 8, bl = 100 + 12 * (t % 3);
        for (int y = h - band; y < h; y++) memset(Y + (size_t)y * w, bl, (size_t)w);
    }

    if (style == 2) {
        int sq = w < 192 ? w / 4 : 88;
        if (sq > h - 8) sq = h - 8;
        int sx = 9 + (37 * t) % (w - sq - 9), sy = 5 + (23 * t) % (h - sq - 5);
        for (int y = 0; y < sq; y++)
            for (int x = 0; x < sq; x++) {
                int tv = tex[(size_t)(y + 7) * TW + x + 11];
                Y[(size_t)(sy + y) * w + sx + x] = sat8(70 + tv / 2 + ((x ^ y) & 8 ? 40 : 0));
            }
        if (t >= 5)
            for (size_t i = 0; i < (size_t)w * h; i++) Y[i] = sat8(Y[i] + 14);
    }

    const int cw = RSHIFT_UP(w, HSHIFT(subsamp)), ch = RSHIFT_UP(h, VSHIFT(subsamp));
    uint8_t *U = out + (size_t)w * h, *V = U + (size_t)cw * ch;
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) {
            int tv = tex[(size_t)((y + MARGIN / 2) % TH) * TW + (x + MARGIN / 2) % TW];
            const uint32_t hn = noisy ? hash3((uint32_t)x, (uint32_t)y, seed ^ (0xC0DEu + (uint32_t)t * 613u)) : 0u;
            const int nu = noisy ? (int)((hn >> 4) % 25u) - 12 : 0, nv = noisy ? (int)((hn >> 12) % 25u) - 12 : 0;
            U[(size_t)y * cw + x] = sat8(96 + tri(x + 3 * t, 64) * 2 + (tv >> 5) - 4 + nu);
            V[(size_t)y * cw + x] = sat8(104 + tri(y - 2 * t, 48) * 2 + tri(x + y, 90) / 2 + (tv >> 6) + nv);
        }
    if (style == 1) {
        int band = big ? ch / 4 : ch / 8;
        for (int y = ch - band; y < ch; y++) {
            memset(U + (size_t)y * cw, 120 + 6 * (t % 3), (size_t)cw);
            memset(V + (size_t)y * cw, 136 - 5 * (t % 3), (size_t)cw);
        }
    }
}
