/* hpcdc_oracle.c -- TEST INFRASTRUCTURE (see oracle.h). CPU restatement of longtail's Buzhash-48
 * content-defined chunker, reference lib/hpcdcchunker/longtail_hpcdcchunker.c.
 *
 * Two independent formulations that the tests prove equal to each other and to the reference:
 *   (1) lto_hpcdc_chunk_stream  : the reference's own shape -- a 4*max byte buffer refilled through a
 *       feeder, a 48-byte circular window and a rolling hash restarted at every chunk.
 *   (2) lto_hpcdc_candidates + lto_hpcdc_select : the shape the HIP kernels use -- a position-pure
 *       window hash evaluated everywhere, then a sparse sequential selection walk.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

const uint32_t lto_buzhash_table[256] = {
#include "../longtail_amd/csrc/buzhash_table.inc"
};

static inline uint32_t rotl32(uint32_t x, unsigned r)
{
    r &= 31u;
    return r ? ((x << r) | (x >> (32u - r))) : x;
}

/* hpcdcchunker.c:126-129 -- double arithmetic, truncated */
uint32_t lto_hpcdc_discriminator(uint32_t avg)
{
    double a = (double)avg;
    return (uint32_t)(a / (-1.42888852e-7 * a + 1.33237515));
}

/* ------------------------------------------------------------------------------------------------
 * (1) streaming restatement
 * ---------------------------------------------------------------------------------------------- */
struct stream_src
{
    const uint8_t* data;
    uint64_t size;
    uint64_t pos;
};

/* StorageChunkFeederFunc (src/longtail.c:1923-1960): deliver min(requested, remaining) bytes */
static uint32_t feed(struct stream_src* s, uint32_t requested, uint8_t* dst)
{
    uint64_t n = s->size - s->pos;
    if (n > requested)
        n = requested;
    if (n)
        memcpy(dst, s->data + s->pos, (size_t)n);
    s->pos += n;
    return (uint32_t)n;
}

uint64_t lto_hpcdc_chunk_stream(const uint8_t* data, uint64_t size, uint32_t min, uint32_t avg, uint32_t max,
                                uint32_t* out_lens, uint64_t cap)
{
    /* chunker state, hpcdcchunker.c:112-124, 148-177 */
    uint64_t ring_cap = (uint64_t)max * 4u;
    if (ring_cap >= 0xffffffffu)
        ring_cap = 0xffffffffu;
    uint8_t* ring = (uint8_t*)malloc((size_t)ring_cap);
    uint32_t have = 0; /* buf.len */
    uint32_t off = 0;  /* c->off   */
    const uint32_t d = lto_hpcdc_discriminator(avg);
    struct stream_src src = {data, size, 0};
    uint64_t count = 0;
    uint8_t win[LTO_WINDOW];

    if (!ring)
        return 0;
    for (;;)
    {
        /* refill when fewer than max bytes are buffered (:241-249, FeedChunker :199-209) */
        if (have - off < max)
        {
            if (off)
            {
                memmove(ring, ring + off, have - off);
                have -= off;
                off = 0;
            }
            have += feed(&src, (uint32_t)(ring_cap - have), ring + have);
        }
        if (off == have)
            break; /* :250-255 -> ESPIPE at the API level */

        uint32_t left = have - off;
        uint32_t len;
        if (left <= min)
        {
            len = left; /* :257-264 */
        }
        else
        {
            const uint8_t* p = ring + off;
            uint32_t h = 0;
            /* seed the window with bytes [min-48, min) of this chunk (:268-279) */
            for (uint32_t i = 0; i < LTO_WINDOW; ++i)
            {
                uint8_t b = p[min - LTO_WINDOW + i];
                h ^= rotl32(lto_buzhash_table[b], (LTO_WINDOW - i - 1u) & 31u);
                win[i] = b;
            }
            uint32_t pos = min;
            uint32_t idx = 0;
            uint32_t end = left > max ? max : left;
            while (pos < end) /* :289-306 */
            {
                uint8_t in = p[pos++];
                uint8_t out = win[idx];
                win[idx++] = in;
                h = rotl32(h, 1) ^ rotl32(lto_buzhash_table[out], LTO_WINDOW & 31u) ^ lto_buzhash_table[in];
                if ((h % d) == d - 1u)
                    break;
                if (idx == LTO_WINDOW)
                    idx = 0;
            }
            len = pos;
        }
        if (count < cap)
            out_lens[count] = len;
        ++count;
        off += len;
    }
    free(ring);
    return count;
}

uint64_t lto_hpcdc_next_from_buffer(const uint8_t* buf, uint64_t size, uint32_t min, uint32_t avg, uint32_t max)
{
    if (size <= min)
        return size; /* :479-484 */
    const uint32_t d = lto_hpcdc_discriminator(avg);
    uint8_t win[LTO_WINDOW];
    uint32_t h = 0;
    /* QUIRK (:488-494): window seeded from the FIRST 48 bytes of the buffer, not from [min-48,min) */
    for (uint32_t i = 0; i < LTO_WINDOW; ++i)
    {
        uint8_t b = buf[i];
        h ^= rotl32(lto_buzhash_table[b], (LTO_WINDOW - i - 1u) & 31u);
        win[i] = b;
    }
    uint32_t pos = min;
    uint32_t idx = 0;
    uint32_t end = (uint32_t)(size > max ? max : size);
    while (pos < end)
    {
        uint8_t in = buf[pos++];
        uint8_t out = win[idx];
        win[idx++] = in;
        h = rotl32(h, 1) ^ rotl32(lto_buzhash_table[out], LTO_WINDOW & 31u) ^ lto_buzhash_table[in];
        if ((h % d) == d - 1u)
            break;
        if (idx == LTO_WINDOW)
            idx = 0;
    }
    return pos;
}

/* ------------------------------------------------------------------------------------------------
 * (2) position-pure formulation
 *   H(p) = XOR_{j=0..47} rotl32(T[data[p-1-j]], j mod 32)
 * which is what the rolling recurrence above evaluates once >= 48 bytes have been rolled in, and what
 * the seeding loop evaluates directly at p = chunk_start + min.
 * ---------------------------------------------------------------------------------------------- */
uint32_t lto_buzhash_at(const uint8_t* data, uint64_t p)
{
    uint32_t h = 0;
    for (uint32_t j = 0; j < LTO_WINDOW; ++j)
        h ^= rotl32(lto_buzhash_table[data[p - 1 - j]], j & 31u);
    return h;
}

void lto_hpcdc_candidates(const uint8_t* data, uint64_t size, uint32_t d, uint8_t* bitmap)
{
    if (size < LTO_WINDOW)
        return;
    uint32_t h = lto_buzhash_at(data, LTO_WINDOW);
    for (uint64_t p = LTO_WINDOW;; ++p)
    {
        if ((h % d) == d - 1u)
            bitmap[p >> 3] |= (uint8_t)(1u << (p & 7u));
        if (p == size)
            break;
        h = rotl32(h, 1) ^ rotl32(lto_buzhash_table[data[p - LTO_WINDOW]], LTO_WINDOW & 31u) ^
            lto_buzhash_table[data[p]];
    }
}

uint64_t lto_hpcdc_select(const uint8_t* bitmap, uint64_t size, uint32_t min, uint32_t max, uint32_t* out_lens,
                          uint64_t cap)
{
    uint64_t s = 0, count = 0;
    while (s < size)
    {
        uint64_t left = size - s;
        uint64_t len;
        if (left <= min)
            len = left;
        else
        {
            uint64_t end = left > max ? max : left;
            len = end;
            for (uint64_t L = (uint64_t)min + 1; L <= end; ++L)
            {
                uint64_t p = s + L;
                if (bitmap[p >> 3] & (1u << (p & 7u)))
                {
                    len = L;
                    break;
                }
            }
        }
        if (count < cap)
            out_lens[count] = (uint32_t)len;
        ++count;
        s += len;
    }
    return count;
}

uint64_t lto_hpcdc_chunk_pure(const uint8_t* data, uint64_t size, uint32_t min, uint32_t avg, uint32_t max,
                              uint32_t* out_lens, uint64_t cap)
{
    uint8_t* bm = (uint8_t*)calloc((size_t)(size / 8 + 2), 1);
    if (!bm)
        return 0;
    lto_hpcdc_candidates(data, size, lto_hpcdc_discriminator(avg), bm);
    uint64_t n = lto_hpcdc_select(bm, size, min, max, out_lens, cap);
    free(bm);
    return n;
}
