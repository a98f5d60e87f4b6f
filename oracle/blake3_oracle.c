/* blake3_oracle.c -- TEST INFRASTRUCTURE (see oracle.h). Portable restatement of unkeyed BLAKE3
 * (hash mode) as vendored by the reference: lib/blake3/ext/blake3_portable.c:8-98 (compression
 * function), blake3_impl.h:85-97 (IV, message schedule), blake3.c:118-166 (leaf = 1024-byte "chunk"
 * state machine, left_len rule), :216-249/:576-618 (parents, ROOT finalisation).
 * Written recursively (hash the left power-of-two subtree, then the rest) instead of the reference's
 * incremental CV stack; both define the same tree.
 */
#include "oracle.h"

#include <string.h>

enum
{
    F_CHUNK_START = 1, /* blake3_impl.h:14-22 */
    F_CHUNK_END = 2,
    F_PARENT = 4,
    F_ROOT = 8
};

static const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                  0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

/* new_m[i] = m[B3_PERM[i]] between rounds (row r+1 of MSG_SCHEDULE is row r composed with this) */
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};

static inline uint32_t rotr32(uint32_t x, unsigned r) { return (x >> r) | (x << (32u - r)); }

#define B3_G(a, b, c, d, mx, my)          \
    do                                    \
    {                                     \
        s[a] = s[a] + s[b] + (mx);        \
        s[d] = rotr32(s[d] ^ s[a], 16);   \
        s[c] = s[c] + s[d];               \
        s[b] = rotr32(s[b] ^ s[c], 12);   \
        s[a] = s[a] + s[b] + (my);        \
        s[d] = rotr32(s[d] ^ s[a], 8);    \
        s[c] = s[c] + s[d];               \
        s[b] = rotr32(s[b] ^ s[c], 7);    \
    } while (0)

/* out16 = full 16-word output of the compression function */
static void b3_compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len,
                        uint32_t flags, uint32_t out16[16])
{
    uint32_t s[16], m[16], t[16];
    memcpy(s, cv, 32);
    memcpy(s + 8, B3_IV, 16);
    s[12] = (uint32_t)counter;
    s[13] = (uint32_t)(counter >> 32);
    s[14] = block_len;
    s[15] = flags;
    memcpy(m, block, 64);
    for (int r = 0; r < 7; ++r)
    {
        B3_G(0, 4, 8, 12, m[0], m[1]);
        B3_G(1, 5, 9, 13, m[2], m[3]);
        B3_G(2, 6, 10, 14, m[4], m[5]);
        B3_G(3, 7, 11, 15, m[6], m[7]);
        B3_G(0, 5, 10, 15, m[8], m[9]);
        B3_G(1, 6, 11, 12, m[10], m[11]);
        B3_G(2, 7, 8, 13, m[12], m[13]);
        B3_G(3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; ++i)
            t[i] = m[B3_PERM[i]];
        memcpy(m, t, 64);
    }
    for (int i = 0; i < 8; ++i)
    {
        out16[i] = s[i] ^ s[i + 8];
        out16[i + 8] = s[i + 8] ^ cv[i];
    }
}

static void load_block(const uint8_t* p, size_t n, uint32_t w[16])
{
    uint8_t tmp[64];
    memset(tmp, 0, 64);
    if (n)
        memcpy(tmp, p, n);
    for (int i = 0; i < 16; ++i)
        w[i] = (uint32_t)tmp[4 * i] | ((uint32_t)tmp[4 * i + 1] << 8) | ((uint32_t)tmp[4 * i + 2] << 16) |
               ((uint32_t)tmp[4 * i + 3] << 24);
}

/* One leaf (<= 1024 bytes, possibly 0 only for the empty message): chain its 64-byte blocks.
 * extra_flags = F_ROOT when the leaf is the whole message. Writes the 16-word output of the LAST
 * block compression (words 0..7 = chaining value / root bytes 0..31). */
static void b3_leaf(const uint8_t* p, size_t len, uint64_t leaf_index, uint32_t extra_flags, uint32_t out16[16])
{
    uint32_t cv[8], w[16];
    memcpy(cv, B3_IV, 32);
    size_t nblocks = len ? (len + 63) / 64 : 1;
    for (size_t b = 0; b < nblocks; ++b)
    {
        size_t bl = (b + 1 == nblocks) ? len - 64 * b : 64;
        uint32_t fl = 0;
        if (b == 0)
            fl |= F_CHUNK_START;
        if (b + 1 == nblocks)
            fl |= F_CHUNK_END | extra_flags;
        load_block(p + 64 * b, bl, w);
        b3_compress(cv, w, leaf_index, (uint32_t)bl, fl, out16);
        memcpy(cv, out16, 32);
    }
}

/* blake3.c:161-166: largest power-of-two number of whole leaves that leaves at least 1 byte on the right */
static size_t left_len(size_t len)
{
    size_t full = (len - 1) / 1024;
    size_t p = 1;
    while (p * 2 <= full)
        p *= 2;
    return p * 1024;
}

/* chaining value of the (non-root) subtree covering p[0..len), whose first leaf has index leaf0 */
static void b3_subtree_cv(const uint8_t* p, size_t len, uint64_t leaf0, uint32_t cv[8])
{
    uint32_t out16[16];
    if (len <= 1024)
    {
        b3_leaf(p, len, leaf0, 0, out16);
    }
    else
    {
        uint32_t pair[16];
        size_t ll = left_len(len);
        b3_subtree_cv(p, ll, leaf0, pair);
        b3_subtree_cv(p + ll, len - ll, leaf0 + ll / 1024, pair + 8);
        b3_compress(B3_IV, pair, 0, 64, F_PARENT, out16);
    }
    memcpy(cv, out16, 32);
}

void lto_blake3(const void* data, size_t len, uint8_t out32[32])
{
    const uint8_t* p = (const uint8_t*)data;
    uint32_t out16[16];
    if (len <= 1024)
    {
        b3_leaf(p, len, 0, F_ROOT, out16);
    }
    else
    {
        uint32_t pair[16];
        size_t ll = left_len(len);
        b3_subtree_cv(p, ll, 0, pair);
        b3_subtree_cv(p + ll, len - ll, ll / 1024, pair + 8);
        b3_compress(B3_IV, pair, 0, 64, F_PARENT | F_ROOT, out16);
    }
    for (int i = 0; i < 8; ++i)
    {
        out32[4 * i] = (uint8_t)out16[i];
        out32[4 * i + 1] = (uint8_t)(out16[i] >> 8);
        out32[4 * i + 2] = (uint8_t)(out16[i] >> 16);
        out32[4 * i + 3] = (uint8_t)(out16[i] >> 24);
    }
}

uint64_t lto_blake3_u64(const void* data, size_t len)
{
    uint8_t o[32];
    uint64_t v = 0;
    lto_blake3(data, len, o);
    for (int i = 7; i >= 0; --i)
        v = (v << 8) | o[i];
    return v;
}

void lto_blake3_u64_many(const uint8_t* data, const uint64_t* offsets, const uint32_t* lens, uint64_t count,
                         uint64_t* hashes)
{
    for (uint64_t i = 0; i < count; ++i)
        hashes[i] = lto_blake3_u64(data + offsets[i], lens[i]);
}
