/* lz4_oracle.c -- TEST INFRASTRUCTURE (see oracle.h). Restatement of the LZ4 *block* codec as used by
 * the reference's LZ4 CompressionAPI (lib/lz4/longtail_lz4.c:52-102): LZ4_compress_fast(acc=1) and
 * LZ4_decompress_safe from the vendored LZ4 1.10.0 (lib/lz4/ext/lz4.c).
 *
 * Encoder: greedy single-probe LZ77, lz4.c:930-1338 specialised to what longtail reaches: no
 * dictionary, 64-bit little-endian host, acceleration 1; table of 8192 u16 for inputs < 64 KiB + 11
 * (hash of 4 bytes -> 13 bits, lz4.c:777-783) else 4096 u32 (hash of 5 bytes -> 12 bits, :785-795).
 * Bit-exact with the reference (tests/test_oracle_lz4.py) -- the GPU encoder is NOT required to be
 * (payloads must round-trip through the reference decoder).
 * Decoder: the strict full-block rules of LZ4_decompress_generic's safe loop, lz4.c:2215-2435.  The reference's
 * "shortcut" / fast-loop paths (lz4.c:2083-2208, 2230-2261) skip some end-of-input checks on MALFORMED streams;
 * this restatement applies the strict rules everywhere, so it accepts a subset of what the reference accepts and
 * decodes identically whenever it accepts (tests/test_oracle_vs_ref.py) -- the right polarity for a checker.
 */
#include "oracle.h"

#include <string.h>

enum
{
    MINMATCH = 4,
    MFLIMIT = 12,     /* lz4.c:242-246 */
    LASTLITERALS = 5,
    MIN_LENGTH = 13,  /* LZ4_minLength = MFLIMIT+1 */
    MAX_DISTANCE = 65535,
    SKIP_TRIGGER = 6, /* lz4.c:711 */
    LIMIT_64K = 65536 + MFLIMIT - 1 /* lz4.c:710 */
};
#define LZ4_MAX_INPUT 0x7E000000

size_t lto_lz4_bound(size_t n) { return n > LZ4_MAX_INPUT ? 0 : n + n / 255 + 16; }

static inline uint32_t rd32(const uint8_t* p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline uint64_t rd64(const uint8_t* p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

static inline uint32_t hash_small(const uint8_t* p) { return (rd32(p) * 2654435761u) >> (32 - 13); }
static inline uint32_t hash_large(const uint8_t* p) { return (uint32_t)(((rd64(p) << 24) * 889523592379ull) >> (64 - 12)); }

/* number of equal bytes at a[] / b[] while a < lim (LZ4_count, lz4.c:660-703) */
static inline uint32_t common_len(const uint8_t* a, const uint8_t* b, const uint8_t* lim)
{
    const uint8_t* s = a;
    while (a < lim && *a == *b)
    {
        ++a;
        ++b;
    }
    return (uint32_t)(a - s);
}

static uint8_t* put_len(uint8_t* op, uint32_t v) /* v already reduced by the 15 stored in the token */
{
    for (; v >= 255; v -= 255)
        *op++ = 255;
    *op++ = (uint8_t)v;
    return op;
}

int lto_lz4_compress(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    if (n < 0 || (uint32_t)n > LZ4_MAX_INPUT)
        return 0;
    const int limited = cap < (int)lto_lz4_bound((size_t)n); /* lz4.c:1388 */
    if (n == 0)
    { /* lz4.c:1361-1371 */
        if (limited && cap <= 0)
            return 0;
        dst[0] = 0;
        return 1;
    }
    const int small = n < LIMIT_64K;
    uint32_t tab32[4096];
    uint16_t tab16[8192];
    if (small)
        memset(tab16, 0, sizeof tab16);
    else
        memset(tab32, 0, sizeof tab32);
#define HASH(pos) (small ? hash_small(src + (pos)) : hash_large(src + (pos)))
#define TGET(h) (small ? (uint32_t)tab16[h] : tab32[h])
#define TPUT(h, v)                  \
    do                              \
    {                               \
        if (small)                  \
            tab16[h] = (uint16_t)(v); \
        else                        \
            tab32[h] = (v);         \
    } while (0)

    uint8_t* op = dst;
    uint8_t* const olimit = dst + cap;
    uint32_t ip = 0, anchor = 0;
    const uint32_t iend = (uint32_t)n;
    const uint32_t mflimit1 = iend - MFLIMIT + 1; /* only meaningful when n >= 13 */
    const uint32_t matchlimit = iend - LASTLITERALS;

    if (n >= MIN_LENGTH)
    {
        TPUT(HASH(0), 0);
        ip = 1;
        uint32_t fh = HASH(ip);
        for (;;)
        {
            uint32_t match;
            uint8_t* token;
            /* -- search (lz4.c:1042-1101) -- */
            {
                uint32_t fwd = ip, step = 1, tries = 1u << SKIP_TRIGGER;
                for (;;)
                {
                    uint32_t h = fh, cur = fwd;
                    match = TGET(h);
                    ip = fwd;
                    fwd += step;
                    step = tries++ >> SKIP_TRIGGER;
                    if (fwd > mflimit1)
                        goto last_literals;
                    fh = HASH(fwd);
                    TPUT(h, cur);
                    if (!small && match + MAX_DISTANCE < cur)
                        continue;
                    if (rd32(src + match) == rd32(src + ip))
                        break;
                }
            }
            /* -- extend backwards (lz4.c:1104-1109) -- */
            while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1])
            {
                --ip;
                --match;
            }
            /* -- literals (lz4.c:1111-1136) -- */
            {
                uint32_t lit = ip - anchor;
                token = op++;
                if (limited && op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > olimit)
                    return 0;
                if (lit >= 15)
                {
                    *token = 15 << 4;
                    op = put_len(op, lit - 15);
                }
                else
                    *token = (uint8_t)(lit << 4);
                memcpy(op, src + anchor, lit);
                op += lit;
            }
            for (;;)
            { /* _next_match (lz4.c:1138-1226) */
                uint32_t off = ip - match;
                *op++ = (uint8_t)off;
                *op++ = (uint8_t)(off >> 8);
                uint32_t mcode = common_len(src + ip + MINMATCH, src + match + MINMATCH, src + matchlimit);
                ip += mcode + MINMATCH;
                if (limited && op + (1 + LASTLITERALS) + (mcode + 240) / 255 > olimit)
                    return 0;
                if (mcode >= 15)
                {
                    *token += 15;
                    op = put_len(op, mcode - 15);
                }
                else
                    *token += (uint8_t)mcode;
                anchor = ip;
                if (ip >= mflimit1)
                    goto last_literals;
                /* lz4.c:1235-1242 */
                TPUT(HASH(ip - 2), ip - 2);
                /* immediate re-test at ip (lz4.c:1253-1294) */
                {
                    uint32_t h = HASH(ip), cur = ip;
                    match = TGET(h);
                    TPUT(h, cur);
                    if ((small || match + MAX_DISTANCE >= cur) && rd32(src + match) == rd32(src + ip))
                    {
                        token = op++;
                        *token = 0;
                        continue;
                    }
                }
                break;
            }
            fh = HASH(++ip);
        }
    }
last_literals:
{
    uint32_t run = iend - anchor;
    if (limited && op + run + 1 + ((run + 255 - 15) / 255) > olimit)
        return 0;
    if (run >= 15)
    {
        *op++ = 15 << 4;
        op = put_len(op, run - 15);
    }
    else
        *op++ = (uint8_t)(run << 4);
    memcpy(op, src + anchor, run);
    op += run;
}
    return (int)(op - dst);
#undef HASH
#undef TGET
#undef TPUT
}

int lto_lz4_decompress(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    if (cap == 0)
        return (n == 1 && src[0] == 0) ? 0 : -1; /* lz4.c:2064-2068 */
    if (n <= 0)
        return -1;
    int64_t ip = 0, op = 0;
    for (;;)
    {
        if (ip >= n)
            return -1;
        uint32_t token = src[ip++];
        int64_t len = token >> 4;
        if (len == 15)
        { /* read_variable_length(&ip, iend-RUN_MASK, 1), lz4.c:1979-2013 */
            uint32_t b;
            if (ip >= (int64_t)n - 15)
                return -1;
            do
            {
                b = src[ip++];
                len += b;
                if (ip > (int64_t)n - 15)
                    return -1;
            } while (b == 255);
        }
        /* lz4.c:2279-2329: close to either end => must be the final, literal-only sequence */
        if (op + len > (int64_t)cap - MFLIMIT || ip + len > (int64_t)n - (2 + 1 + LASTLITERALS))
        {
            if (ip + len != n || op + len > cap)
                return -1;
            memmove(dst + op, src + ip, (size_t)len);
            op += len;
            return (int)op;
        }
        memcpy(dst + op, src + ip, (size_t)len);
        ip += len;
        op += len;
        uint32_t off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8);
        ip += 2;
        if (off == 0 || off > op)
            return -1; /* lz4.c:2356 */
        int64_t ml = token & 15;
        if (ml == 15)
        {
            uint32_t b;
            do
            {
                if (ip >= (int64_t)n - LASTLITERALS + 1)
                    return -1; /* lz4.c:2346 */
                b = src[ip++];
                ml += b;
            } while (b == 255);
        }
        ml += MINMATCH;
        if (op + ml > (int64_t)cap - LASTLITERALS)
            return -1; /* lz4.c:2423 */
        for (int64_t i = 0; i < ml; ++i)
            dst[op + i] = dst[op + i - off];
        op += ml;
    }
}
