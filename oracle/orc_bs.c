/*
 * orc_bs.c -- ORACLE (test infrastructure): MSB-first bit I/O and the three interleaved
 * exp-Golomb codes + zero-bit run-length coder of DSV1.
 * Follows bs.c of the reference: put/get bits bs.c:76-125, UEG bs.c:129-157,
 * SEG bs.c:160-188, NEG bs.c:191-219, ZBRLE bs.c:222-267, align bs.c:28-35,
 * concat bs.c:37-46.  The writer only ever ORs into a pre-zeroed buffer (bs.c:50-63),
 * which is what makes length back-patching work.
 */
#include <string.h>
#include "orc.h"

void orc_bs_init(orc_bs *bs, uint8_t *buf)
{
    bs->start = buf;
    bs->pos = 0;
}

void orc_bs_align(orc_bs *bs)
{
    bs->pos = (bs->pos + 7u) & ~7u;
}

static inline void set_bit(orc_bs *bs, unsigned one)
{
    if (one)
        bs->start[bs->pos >> 3] |= (uint8_t)(0x80u >> (bs->pos & 7));
    bs->pos++;
}

static inline unsigned read_bit(orc_bs *bs)
{
    unsigned b = (bs->start[bs->pos >> 3] >> (7 - (bs->pos & 7))) & 1u;
    bs->pos++;
    return b;
}

void orc_bs_put_bits(orc_bs *bs, unsigned n, unsigned v)
{
    while (n--)
        set_bit(bs, (v >> n) & 1u);
}

unsigned orc_bs_get_bits(orc_bs *bs, unsigned n)
{
    unsigned out = 0;
    while (n--)
        out = (out << 1) | read_bit(bs);
    return out;
}

/* UEG(v): with m = v+1 and k = floor(log2 m): k pairs ('0', next-lower bit of m), then '1' */
void orc_bs_put_ueg(orc_bs *bs, unsigned v)
{
    unsigned m = v + 1;
    int k = 31 - __builtin_clz(m);
    while (k-- > 0) {
        bs->pos++;                      /* the '0' of the pair: buffer is already zero */
        set_bit(bs, (m >> k) & 1u);
    }
    set_bit(bs, 1);
}

unsigned orc_bs_get_ueg(orc_bs *bs)
{
    unsigned m = 1;
    while (!read_bit(bs))
        m = (m << 1) | read_bit(bs);
    return m - 1;
}

void orc_bs_put_seg(orc_bs *bs, int v)
{
    unsigned mag = v < 0 ? (unsigned)-v : (unsigned)v;
    orc_bs_put_ueg(bs, mag);
    if (mag)
        set_bit(bs, v < 0);
}

int orc_bs_get_seg(orc_bs *bs)
{
    int mag = (int)orc_bs_get_ueg(bs);
    if (mag && read_bit(bs))
        return -mag;
    return mag;
}

/* NEG: value is known to be non-zero, so |v|-1 is coded, then the sign */
void orc_bs_put_neg(orc_bs *bs, int v)
{
    unsigned mag = v < 0 ? (unsigned)-v : (unsigned)v;
    orc_bs_put_ueg(bs, mag - 1);
    if (mag)
        set_bit(bs, v < 0);
}

int orc_bs_get_neg(orc_bs *bs)
{
    int mag = (int)orc_bs_get_ueg(bs) + 1;
    if (mag && read_bit(bs))
        return -mag;
    return mag;
}

void orc_bs_append(orc_bs *bs, const uint8_t *data, int len)
{
    memcpy(bs->start + (bs->pos >> 3), data, (size_t)len);
    bs->pos += (unsigned)len * 8u;
}

/* ZBRLE: every '1' is coded as UEG(number of zeros since the previous '1') */
void orc_rle_init(orc_zbrle *r, uint8_t *buf)
{
    orc_bs_init(&r->bs, buf);
    r->nz = 0;
}

void orc_rle_put(orc_zbrle *r, int bit)
{
    if (!bit) {
        r->nz++;
        return;
    }
    orc_bs_put_ueg(&r->bs, (unsigned)r->nz);
    r->nz = 0;
}

int orc_rle_get(orc_zbrle *r)
{
    if (r->nz == 0)
        r->nz = (int)orc_bs_get_ueg(&r->bs);
    else
        r->nz--;
    return r->nz == 0;
}

int orc_rle_finish_write(orc_zbrle *r)
{
    orc_bs_put_ueg(&r->bs, (unsigned)r->nz);
    r->nz = 0;
    orc_bs_align(&r->bs);
    return (int)orc_bs_bytepos(&r->bs);
}
