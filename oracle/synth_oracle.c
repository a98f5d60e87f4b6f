/* synth_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *  - lto_synth_fill: CPU side of the counter-based synthetic asset generator (include/longtail_synth.h).
 *  - lto_ingest: single-thread "port" CPU baseline for bench.py: the reference's job structure for
 *    the hot path (src/longtail.c:2396-2458 part split, :1985-1987 min/avg/max, :6801-6860 greedy
 *    block packing, compressblockstore.c:67-141 per-block compress) on top of the restated kernels.
 */
#include "oracle.h"
#include "../include/longtail_synth.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

uint64_t lto_synth_asset_seed(uint64_t tree_seed, uint64_t index) { return lt_synth_asset_seed(tree_seed, index); }

void lto_synth_fill(uint8_t* dst, uint64_t nbytes, uint64_t seed, uint64_t byte_offset, int kind)
{
    uint64_t i = 0;
    while (i < nbytes)
    {
        uint64_t pos = byte_offset + i;
        uint64_t w = pos >> 3;
        unsigned sh = (unsigned)(pos & 7u);
        uint64_t v = lt_synth_word(seed, w, kind);
        unsigned take = 8u - sh;
        if (take > nbytes - i)
            take = (unsigned)(nbytes - i);
        for (unsigned k = 0; k < take; ++k)
            dst[i + k] = (uint8_t)(v >> (8u * (sh + k)));
        i += take;
    }
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int lto_ingest(const uint8_t* data, uint64_t size, uint64_t part_size, uint32_t target_chunk_size,
               uint32_t block_size, int do_compress, struct lto_ingest_result* out)
{
    /* src/longtail.c:1985-1987 with chunker minimum 48 */
    uint32_t mn = target_chunk_size / 8 < LTO_WINDOW ? LTO_WINDOW : target_chunk_size / 8;
    uint32_t av = target_chunk_size / 2 < LTO_WINDOW ? LTO_WINDOW : target_chunk_size / 2;
    uint32_t mx = target_chunk_size * 2 < LTO_WINDOW ? LTO_WINDOW : target_chunk_size * 2;
    memset(out, 0, sizeof *out);
    if (part_size == 0)
        part_size = (uint64_t)target_chunk_size * 1024u; /* :2396 */

    uint64_t cap = size / (mn ? mn : 1) + size / part_size + 16;
    uint32_t* lens = (uint32_t*)malloc((size_t)cap * sizeof(uint32_t));
    uint64_t* hashes = (uint64_t*)malloc((size_t)cap * sizeof(uint64_t));
    if (!lens || !hashes)
    {
        free(lens);
        free(hashes);
        return 12;
    }
    uint64_t n = 0;
    double t0 = now_s();
    for (uint64_t off = 0; off < size; off += part_size)
    {
        uint64_t psz = size - off < part_size ? size - off : part_size;
        if (psz <= LTO_WINDOW)
        { /* :2051-2108: tiny part = one chunk, chunker not involved */
            lens[n++] = (uint32_t)psz;
            continue;
        }
        n += lto_hpcdc_chunk_stream(data + off, psz, mn, av, mx, lens + n, cap - n);
    }
    double t1 = now_s();
    {
        uint64_t o = 0;
        for (uint64_t i = 0; i < n; ++i)
        {
            hashes[i] = lto_blake3_u64(data + o, lens[i]);
            out->hash_xor ^= hashes[i];
            out->hash_sum += hashes[i];
            o += lens[i];
        }
    }
    double t2 = now_s();
    if (do_compress)
    {
        /* greedy packing, :6801-6860 (all chunks unique, one tag, <= 1024 chunks, 10 % overshoot) */
        uint64_t limit = (uint64_t)block_size + block_size / 10;
        uint8_t* dst = (uint8_t*)malloc(lto_lz4_bound((size_t)limit + 16));
        uint64_t i = 0, o = 0;
        if (!dst)
        {
            free(lens);
            free(hashes);
            return 12;
        }
        while (i < n)
        {
            uint64_t bsz = lens[i], cnt = 1;
            while (i + cnt < n && cnt < 1024 && bsz + lens[i + cnt] <= limit)
            {
                bsz += lens[i + cnt];
                ++cnt;
            }
            int c = lto_lz4_compress(data + o, (int)bsz, dst, (int)lto_lz4_bound((size_t)bsz));
            if (c <= 0)
            {
                free(dst);
                free(lens);
                free(hashes);
                return 5;
            }
            out->compressed_bytes += (uint64_t)c + 8; /* + [u32 raw][u32 comp] header, compressblockstore.c:135-137 */
            o += bsz;
            i += cnt;
        }
        free(dst);
    }
    double t3 = now_s();
    out->chunk_count = n;
    out->seconds_chunk = t1 - t0;
    out->seconds_hash = t2 - t1;
    out->seconds_compress = t3 - t2;
    free(lens);
    free(hashes);
    return 0;
}

void lto_xorshift_fill(uint8_t* dst, uint64_t nbytes, uint64_t seed)
{
    uint64_t s = seed;
    for (uint64_t o = 0; o < nbytes; o += 8)
    {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        uint64_t n = nbytes - o < 8 ? nbytes - o : 8;
        for (uint64_t k = 0; k < n; ++k)
            dst[o + k] = (uint8_t)(s >> (8 * k));
    }
}
