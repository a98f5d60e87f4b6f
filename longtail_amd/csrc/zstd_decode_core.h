/* zstd_decode_core.h -- decoder for zstd frames (RFC 8878), the other half of ZStdCompressionAPI.
 *
 * What the reference does on this path: `ZStdCompressionAPI_Decompress` (lib/zstd/longtail_zstd.c:144-177) hands the
 * stored block's payload to `ZSTD_decompressDCtx`, which accepts one or more concatenated frames
 * (ext/decompress/zstd_decompress.c:1068-1085).  This is our own decoder, written against the format as that decoder
 * implements it (citations are to /root/reference/lib/zstd/ext):
 *   frame header, skippable frames, block headers       decompress/zstd_decompress.c:438-560, 560-640, 1105-1195
 *   literals section, Huffman tree, 1 / 4 streams        decompress/zstd_decompress_block.c:135-345, huf_decompress.c
 *   FSE table description and decoding tables            common/entropy_common.c:42-187, zstd_decompress_block.c:484-603
 *   sequences: modes, decode order, repeat offsets       decompress/zstd_decompress_block.c:700-760, 1240-1345
 * Not supported (reported as malformed): dictionaries.  A content checksum, if present, is skipped, not verified.
 *
 * Same two execution models as zstd_block_core.h (whose format helpers it reuses): lane 0 parses -- entropy decoding
 * is a serial bit-stream walk -- and all lanes copy (literal runs, matches, raw / RLE blocks).  One wavefront decodes
 * one payload on the GPU; the host build (ZB_LANES 1) is the model tests run against reference-ENCODED frames.
 * Every read is bounds-checked against the payload and every write against the destination capacity: malformed input
 * gives ZD_ERROR, never an out-of-range access.
 */
#ifndef ZSTD_DECODE_CORE_H
#define ZSTD_DECODE_CORE_H

#include "zstd_block_core.h"

#define ZD_ERROR 0xFFFFFFFFu
#define ZD_PREPARED 0xFFFFFFFEu /* see ZDV_PREP: literals are in lits[], the three tables in sh->fse[]; v[ZDV_LL] = number of sequences,
                                 * v[ZDV_ML] = number of literals, v[ZDV_OFF] / v[ZDV_LEN] = offset (in src) / size of the bit-stream */
/* where the payload was found malformed (source line, diagnostics only) */
#define ZD_SET_ERR(sh) ((sh)->v[ZDV_SRC] = __LINE__, (sh)->v[ZDV_ERR] = 1)
#define ZD_FAIL_AT(sh) ((sh)->v[ZDV_SRC] = (sh)->v[ZDV_ERR] ? (sh)->v[ZDV_SRC] : __LINE__, ZD_ERROR)
#define ZD_LIT_MAX ZB_BLOCK_MAX

typedef struct ZdFse /* one FSE decoding table */
{
    uint16_t base[512];  /* new state = base + bits */
    uint8_t sym[512];
    uint8_t nb[512];
    uint32_t log;        /* 0 with valid == 2: RLE (single symbol sym[0]) */
    uint32_t valid;      /* 0 none, 1 table, 2 RLE */
} ZdFse;

enum
{
    ZDV_ERR,
    ZDV_LL,
    ZDV_ML,
    ZDV_OFF,
    ZDV_OUT,    /* bytes produced so far (frame relative) */
    ZDV_LITPOS, /* literals consumed in this block */
    ZDV_NLIT,
    ZDV_MODE,   /* copy mode of the current step */
    ZDV_SRC,    /* source line of the check that failed */
    ZDV_LEN,
    ZDV_BYTE,
    ZDV_DONE,
    ZDV_PREP,   /* != 0 (kernel build, piece mode): stop in front of the sequence loop and hand the block's state over (ZD_PREPARED) */
    ZDV_COUNT
};

typedef struct ZdShared
{
    uint16_t huf[1u << (ZB_HUF_MAXBITS + 1u)]; /* nbBits << 8 | symbol; tables of up to 12 bits */
    uint32_t huf_log, huf_valid;
    ZdFse fse[3]; /* ZT_LL, ZT_OF, ZT_ML */
    ZdFse wtab;   /* FSE table of the Huffman weights */
    int16_t norm[256];
    uint8_t weights[256];
    /* small tables the builders index dynamically: in shared memory, because a local array indexed by data is private memory on
     * the GPU (one memory round trip, or a 64-way select chain, per access -- the FSE builder alone cost 1.7 ms per block) */
    uint16_t next[64];
    uint16_t cum[64 + 2]; /* zd_build_fse_par: first table cell (in spread order) of every symbol */
    uint32_t tb_maxsym[3], tb_log[3], tb_build[3]; /* the sequence tables whose description was read and that are still to be built */
    uint32_t rank_start[ZB_HUF_MAXBITS + 3u];
    uint32_t rank_cnt[ZB_HUF_MAXBITS + 3u];
    uint32_t rep[3];
    uint32_t v[ZDV_COUNT];
} ZdShared;

/* ---- little-endian reads with bounds ---- */
ZB_FN uint32_t zd_le(const uint8_t* p, uint32_t n)
{
    uint32_t v = 0;
    for (uint32_t i = 0; i < n; ++i)
        v |= (uint32_t)p[i] << (8u * i);
    return v;
}

/* ---- forward bit reader (NCount) ---- */
typedef struct ZdFwd
{
    const uint8_t* p;
    uint32_t size, bit;
} ZdFwd;
ZB_FN uint32_t zd_fwd_peek(const ZdFwd* r, uint32_t n) /* n <= 16; bits past the end read as 0 */
{
    uint32_t v = 0;
    const uint32_t b0 = r->bit >> 3;
    for (uint32_t i = 0; i < 4u; ++i)
        if (b0 + i < r->size)
            v |= (uint32_t)r->p[b0 + i] << (8u * i);
    return (v >> (r->bit & 7u)) & ((1u << n) - 1u);
}

/* NCount (entropy_common.c:42-187).  Returns bytes consumed or ZD_ERROR; norm[0..*maxsym], *maxsym updated. */
ZB_FN uint32_t zd_read_ncount(const uint8_t* p, uint32_t size, int16_t* norm, uint32_t* maxsym, uint32_t maxlog, uint32_t* tlog)
{
    ZdFwd r;
    r.p = p;
    r.size = size;
    r.bit = 0;
    if (size == 0u)
        return ZD_ERROR;
    const uint32_t tl = zd_fwd_peek(&r, 4) + 5u;
    r.bit = 4;
    if (tl > maxlog)
        return ZD_ERROR;
    int remaining = (int)(1u << tl) + 1, threshold = (int)(1u << tl);
    uint32_t nbits = tl + 1u, sym = 0;
    int prev0 = 0;
    for (uint32_t s = 0; s <= *maxsym; ++s)
        norm[s] = 0;
    while (remaining > 1 && sym <= *maxsym)
    {
        if (prev0)
        {
            for (;;)
            {
                const uint32_t rp = zd_fwd_peek(&r, 2);
                r.bit += 2;
                sym += rp;
                if (rp != 3u)
                    break;
                if (sym > *maxsym + 1u)
                    return ZD_ERROR;
            }
            if (sym > *maxsym)
                return ZD_ERROR;
        }
        {
            const int maxv = (2 * threshold - 1) - remaining;
            int count;
            const uint32_t lo = zd_fwd_peek(&r, nbits - 1u);
            if ((int)lo < maxv)
            {
                count = (int)lo;
                r.bit += nbits - 1u;
            }
            else
            {
                count = (int)zd_fwd_peek(&r, nbits);
                if (count >= threshold)
                    count -= maxv;
                r.bit += nbits;
            }
            --count; /* -1: "less than one" */
            remaining -= count < 0 ? -count : count;
            norm[sym++] = (int16_t)count;
            prev0 = count == 0;
            if (remaining < 1)
                return ZD_ERROR;
            while (remaining < threshold && threshold > 1)
            {
                --nbits;
                threshold >>= 1;
            }
        }
        if (r.bit > size * 8u)
            return ZD_ERROR;
    }
    if (remaining != 1 || sym == 0u)
        return ZD_ERROR;
    *maxsym = sym - 1u;
    *tlog = tl;
    return (r.bit + 7u) >> 3;
}

/* decoding table from a normalised distribution (zstd_decompress_block.c:484-603) */
ZB_FN int zd_build_fse(ZdFse* t, const int16_t* norm, uint32_t maxsym, uint32_t tl, uint16_t* next /* 64 entries of scratch */)
{
    const uint32_t size = 1u << tl, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t high = size - 1u, pos = 0;
    if (maxsym > 63u || tl > 9u)
        return 1;
    for (uint32_t s = 0; s <= maxsym; ++s)
    {
        if (norm[s] == -1)
        {
            t->sym[high--] = (uint8_t)s;
            next[s] = 1;
        }
        else
            next[s] = (uint16_t)norm[s];
    }
    for (uint32_t s = 0; s <= maxsym; ++s)
        for (int i = 0; i < norm[s]; ++i)
        {
            t->sym[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high)
                pos = (pos + step) & mask;
        }
    if (pos != 0u)
        return 1; /* the counts did not fill the table exactly */
    for (uint32_t u = 0; u < size; ++u)
    {
        const uint32_t ns = next[t->sym[u]]++;
        const uint32_t nb = tl - zb_highbit(ns);
        t->nb[u] = (uint8_t)nb;
        t->base[u] = (uint16_t)((ns << nb) - size);
    }
    t->log = tl;
    t->valid = 1;
    return 0;
}

#ifndef ZD_MARK
#define ZD_MARK(i) ((void)0) /* profiling hook of the kernel build */
#endif
#ifndef ZD_ABLATE
#define ZD_ABLATE 0u /* timing experiments of the kernel build */
#endif

/* The same table built by ALL lanes together (the three sequence tables of a block; the serial builder above spends ~250 cycles
 * per cell on a chain of dependent shared-memory updates: 0.7 M cycles per block).  Called by every lane, outside ZB_SERIAL; the
 * caller synchronises before and after.  Every step is a ZB_PAR_FOR:
 *   1. per symbol: the low-probability symbols (-1) take the top cells, in symbol order from the top down;
 *      cum[s] = number of cells of the symbols before s (counts > 0 only);
 *   2. per step j of the spread walk (cell p = j * step mod size, cells above the threshold are skipped): its rank among the
 *      cells that count is j minus the skipped cells met so far, each of which is met at step q * step^-1 mod size; the symbol
 *      whose run [cum[s], cum[s + 1]) holds that rank gets the cell.
 *      Steps 1 and 2 also set the cell's bit in the symbol's own occupancy mask (one bit per cell, 32 cells per word);
 *   3. per symbol: running count of its cells before every 32-cell word;
 *   4. per cell: its number among the symbol's cells IN CELL ORDER = that count + the mask bits below it; state bits and base
 *      follow from the number as in the serial builder.
 * masks: 64 * 16 words, pre: 64 * 16 halfwords of scratch.  Returns nonzero for what the serial builder rejects. */
#define ZD_FSE_PAR_MASK_WORDS (64u * 16u)
ZB_FN int zd_build_fse_par(ZdFse* t, const int16_t* norm, uint32_t maxsym, uint32_t tl, uint16_t* cum, uint32_t* masks, uint16_t* pre,
                           uint32_t zl)
{
    const uint32_t size = 1u << tl, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    const uint32_t W = (size + 31u) >> 5; /* mask words per symbol */
    uint32_t nlow = 0, total = 0, stepinv;
    if (maxsym > 63u || tl > 9u || tl < 4u) /* (step is odd from 16 cells on; the format's tables have 32 or more) */
        return 1;
    for (uint32_t s = 0; s <= maxsym; ++s) /* (every lane: a few dozen reads of the same words) */
    {
        nlow += norm[s] == -1 ? 1u : 0u;
        total += norm[s] > 0 ? (uint32_t)norm[s] : 0u;
    }
    if (total + nlow != size)
        return 1; /* the counts do not fill the table exactly */
    {
        /* step is odd: its inverse modulo the power of two by Newton's iteration (3 bits -> 6 -> 12) */
        uint32_t x = step; /* correct to 3 bits for every odd number */
        x *= 2u - step * x;
        x *= 2u - step * x;
        stepinv = x & mask;
    }
    const uint32_t high = size - 1u - nlow;
    ZB_PAR_FOR(i, (maxsym + 1u) * W) masks[i] = 0;
    ZB_SYNC_LDS();
    ZB_PAR_FOR(s, maxsym + 1u)
    {
        uint32_t before = 0, low_before = 0;
        for (uint32_t q = 0; q < s; ++q)
        {
            before += norm[q] > 0 ? (uint32_t)norm[q] : 0u;
            low_before += norm[q] == -1 ? 1u : 0u;
        }
        cum[s] = (uint16_t)before;
        if (norm[s] == -1)
        {
            const uint32_t p = size - 1u - low_before;
            t->sym[p] = (uint8_t)s;
            zb_atomic_or(&masks[s * W + (p >> 5)], 1u << (p & 31u));
        }
        if (s == maxsym)
            cum[s + 1u] = (uint16_t)(before + (norm[s] > 0 ? (uint32_t)norm[s] : 0u));
    }
    ZB_SYNC_LDS();
    ZB_PAR_FOR(j, size)
    {
        const uint32_t p = (j * step) & mask;
        if (p <= high)
        {
            uint32_t skipped = 0, lo = 0, hi = maxsym + 1u;
            for (uint32_t q = high + 1u; q < size; ++q)
                skipped += ((q * stepinv) & mask) < j ? 1u : 0u;
            {
                const uint32_t g = j - skipped; /* rank of this cell among the cells that count */
                while (hi - lo > 1u) /* last symbol whose run starts at or before g (runs of absent symbols are empty: never them) */
                {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (cum[mid] <= g)
                        lo = mid;
                    else
                        hi = mid;
                }
            }
            t->sym[p] = (uint8_t)lo;
            zb_atomic_or(&masks[lo * W + (p >> 5)], 1u << (p & 31u));
        }
    }
    ZB_SYNC_LDS();
    ZB_PAR_FOR(s, maxsym + 1u)
    {
        uint32_t running = norm[s] == -1 ? 1u : norm[s] > 0 ? (uint32_t)norm[s] : 0u; /* the first cell's number */
        for (uint32_t w = 0; w < W; ++w)
        {
            pre[s * W + w] = (uint16_t)running;
            running += (uint32_t)__builtin_popcount(masks[s * W + w]);
        }
    }
    ZB_SYNC_LDS();
    ZB_PAR_FOR(u, size)
    {
        const uint32_t k = (uint32_t)t->sym[u] * W + (u >> 5);
        const uint32_t ns = pre[k] + (uint32_t)__builtin_popcount(masks[k] & ((1u << (u & 31u)) - 1u));
        const uint32_t nb = tl - zb_highbit(ns);
        t->nb[u] = (uint8_t)nb;
        t->base[u] = (uint16_t)((ns << nb) - size);
    }
    ZB_SERIAL(zl)
    {
        t->log = tl;
        t->valid = 1;
    }
    return 0;
}

/* ---- backward bit reader: the stream ends with a 1 bit followed by zero padding ---- */
typedef struct ZdBack
{
    const uint8_t* p;
    uint32_t size;
    uint32_t pos; /* bits not yet consumed: the next read takes the n bits just below bit `pos` */
    int over;     /* a read went below bit 0 */
    uint64_t cont; /* cached bits [cpos, cpos + 64) of the stream; cpos is a multiple of 8 */
    uint32_t cpos, cvalid;
} ZdBack;

ZB_FN int zd_back_open(ZdBack* r, const uint8_t* p, uint32_t size)
{
    r->p = p;
    r->size = size;
    r->over = 0;
    r->cvalid = 0;
    r->cpos = 0;
    r->cont = 0;
    if (size == 0u || p[size - 1u] == 0u)
        return 1;
    r->pos = (size - 1u) * 8u + zb_highbit(p[size - 1u]);
    return 0;
}
ZB_FN uint32_t zd_back_peek_at(ZdBack* r, uint32_t bitpos, uint32_t n) /* n <= 24, bits [bitpos, bitpos+n) */
{
    /* a 64-bit window is kept in registers and slid down the stream: about one 8-byte load per 40 bits consumed */
    if (!r->cvalid || bitpos < r->cpos || bitpos + n > r->cpos + 64u)
    {
        const uint32_t top = bitpos + n; /* <= 8 * size */
        uint32_t b0 = top > 64u ? (top - 64u + 7u) >> 3 : 0u;
        if (b0 + 8u > r->size)
            b0 = r->size >= 8u ? r->size - 8u : 0u;
        if (b0 + 8u <= r->size)
            __builtin_memcpy(&r->cont, r->p + b0, 8);
        else
        {
            r->cont = 0;
            for (uint32_t i = 0; i < r->size; ++i)
                r->cont |= (uint64_t)r->p[i] << (8u * i);
        }
        r->cpos = b0 * 8u;
        r->cvalid = 1;
    }
    return (uint32_t)(r->cont >> (bitpos - r->cpos)) & ((1u << n) - 1u);
}
ZB_FN uint32_t zd_back_read(ZdBack* r, uint32_t n) /* n <= 24; reading past the start yields zeros and sets over */
{
    if (n == 0u)
        return 0;
    if (n > r->pos)
    {
        /* the part of the field that exists forms its HIGH bits */
        const uint32_t have = r->pos;
        const uint32_t v = have ? zd_back_peek_at(r, 0, have) << (n - have) : 0u;
        r->pos = 0;
        r->over = 1;
        return v;
    }
    r->pos -= n;
    return zd_back_peek_at(r, r->pos, n);
}
ZB_FN uint32_t zd_back_read32(ZdBack* r, uint32_t n) /* n <= 32 */
{
    if (n <= 24u)
        return zd_back_read(r, n);
    {
        const uint32_t hi = zd_back_read(r, n - 16u);
        return (hi << 16) | zd_back_read(r, 16u);
    }
}

/* ---- Huffman ---- */
/* weights[0..n) (the last one already completed) -> single-symbol decoding table (RFC 8878 §4.2.1) */
ZB_FN int zd_build_huf(ZdShared* sh, uint32_t n, uint32_t tlog)
{
    uint32_t* rank_start = sh->rank_start;
    uint32_t* cnt = sh->rank_cnt;
    if (tlog == 0u || tlog > ZB_HUF_MAXBITS + 1u)
        return 1;
    for (uint32_t w = 0; w <= tlog + 1u; ++w)
        cnt[w] = 0;
    for (uint32_t s = 0; s < n; ++s)
    {
        if (sh->weights[s] > tlog)
            return 1;
        ++cnt[sh->weights[s]];
    }
    {
        uint32_t pos = 0;
        for (uint32_t w = 1; w <= tlog; ++w)
        {
            rank_start[w] = pos;
            pos += cnt[w] << (w - 1u);
        }
        if (pos != (1u << tlog))
            return 1;
    }
    for (uint32_t s = 0; s < n; ++s)
    {
        const uint32_t w = sh->weights[s];
        if (w)
        {
            const uint32_t len = 1u << (w - 1u), e = ((tlog + 1u - w) << 8) | s;
            const uint32_t first = rank_start[w]; /* read once: the stores below may alias it as far as the compiler knows */
            for (uint32_t u = 0; u < len; ++u)
                sh->huf[first + u] = (uint16_t)e;
            rank_start[w] = first + len;
        }
    }
    sh->huf_log = tlog;
    sh->huf_valid = 1;
    return 0;
}

/* Huffman tree description (HUF_readStats, entropy_common.c:236-327).  Returns bytes consumed or ZD_ERROR. */
ZB_FN uint32_t zd_read_huf_tree(ZdShared* sh, const uint8_t* p, uint32_t size)
{
    uint32_t n = 0, used;
    if (size == 0u)
        return ZD_ERROR;
    const uint32_t hb = p[0];
    if (hb >= 128u)
    {
        n = hb - 127u;
        used = 1u + ((n + 1u) >> 1);
        if (used > size)
            return ZD_ERROR;
        for (uint32_t i = 0; i < n; ++i)
        {
            const uint32_t b = p[1u + (i >> 1)];
            sh->weights[i] = (uint8_t)((i & 1u) ? (b & 15u) : (b >> 4));
        }
    }
    else
    {
        /* FSE-compressed weights: table log <= 6, two interleaved states (fse_decompress.c:174-238) */
        uint32_t maxsym = 12, tl = 0;
        ZdFse* t = &sh->wtab;
        ZdBack br;
        used = 1u + hb;
        if (hb == 0u || used > size)
            return ZD_ERROR;
        {
            const uint32_t hs = zd_read_ncount(p + 1, hb, sh->norm, &maxsym, 6u, &tl);
            if (hs == ZD_ERROR || hs >= hb)
                return ZD_ERROR;
            if (zd_build_fse(t, sh->norm, maxsym, tl, sh->next))
                return ZD_ERROR;
            if (zd_back_open(&br, p + 1u + hs, hb - hs))
                return ZD_ERROR;
        }
        {
            uint32_t st[2];
            st[0] = zd_back_read(&br, tl);
            st[1] = zd_back_read(&br, tl);
            if (br.over)
                return ZD_ERROR;
            for (uint32_t k = 0;; k ^= 1u)
            {
                if (n >= 255u)
                    return ZD_ERROR;
                sh->weights[n++] = t->sym[st[k]];
                st[k] = t->base[st[k]] + zd_back_read(&br, t->nb[st[k]]);
                if (br.over)
                {
                    /* the update wanted more bits than exist: the other state holds the last weight */
                    if (n >= 255u)
                        return ZD_ERROR;
                    sh->weights[n++] = t->sym[st[k ^ 1u]];
                    break;
                }
            }
        }
    }
    /* complete the last weight: the total must become a power of two */
    {
        uint32_t total = 0;
        for (uint32_t i = 0; i < n; ++i)
        {
            if (sh->weights[i] > ZB_HUF_MAXBITS + 1u)
                return ZD_ERROR;
            total += sh->weights[i] ? 1u << (sh->weights[i] - 1u) : 0u;
        }
        if (total == 0u)
            return ZD_ERROR;
        {
            const uint32_t tlog = zb_highbit(total) + 1u;
            const uint32_t rest = (1u << tlog) - total;
            if (tlog > ZB_HUF_MAXBITS + 1u || (rest & (rest - 1u)) != 0u)
                return ZD_ERROR;
            sh->weights[n++] = (uint8_t)(zb_highbit(rest) + 1u);
            if (zd_build_huf(sh, n, tlog))
                return ZD_ERROR;
        }
    }
    return used;
}

/* one Huffman stream of `count` symbols into out[] (serial); zd_huf_stream_from: the first i0 symbols are there already and pos0
 * bits of the stream are left (a caller that decodes the bulk its own way: k_zstd_sub_entropy, two streams per lane) */
ZB_FN int zd_huf_stream_from(const ZdShared* sh, const uint8_t* p, uint32_t size, uint8_t* out, uint32_t count, uint32_t pos0, uint32_t i0)
{
    ZdBack br;
    const uint32_t tl = sh->huf_log;
    uint32_t i = i0;
    if (zd_back_open(&br, p, size) || pos0 > br.pos)
        return 1;
    br.pos = pos0;
    /* Four symbols per refill while at least 64 bits are left (4 x huf_log <= 44 of the >= 57 bits of one 8-byte load, so no
     * symbol of a group can run out of bits and nothing has to be checked inside it); the careful loop below finishes the
     * stream and reports every malformation exactly as before. */
    while (count - i >= 4u && br.pos >= 64u)
    {
        const uint32_t b0 = ((br.pos + 7u) >> 3) - 8u; /* the 8 bytes that end at the byte holding bit pos-1 */
        uint64_t t;
        uint32_t used = 0, packed = 0;
        __builtin_memcpy(&t, p + b0, 8);
        t <<= (8u - (br.pos & 7u)) & 7u; /* bit pos-1 on top */
        for (uint32_t k = 0; k < 4u; ++k)
        {
            const uint32_t e = sh->huf[(uint32_t)(t >> (64u - tl))];
            const uint32_t nb = e >> 8;
            t <<= nb;
            used += nb;
            packed |= (e & 255u) << (8u * k);
        }
        __builtin_memcpy(out + i, &packed, 4);
        br.pos -= used;
        i += 4u;
    }
    for (; i < count; ++i)
    {
        /* peek tl bits (zero-filled below the start), consume only the code's length */
        uint32_t idx;
        if (br.pos >= tl)
            idx = zd_back_peek_at(&br, br.pos - tl, tl);
        else
            idx = br.pos ? zd_back_peek_at(&br, 0, br.pos) << (tl - br.pos) : 0u;
        {
            const uint32_t e = sh->huf[idx];
            const uint32_t nb = e >> 8;
            if (nb > br.pos)
                return 1;
            br.pos -= nb;
            out[i] = (uint8_t)e;
        }
    }
    return br.pos != 0u; /* the stream must be consumed exactly */
}
ZB_FN int zd_huf_stream(const ZdShared* sh, const uint8_t* p, uint32_t size, uint8_t* out, uint32_t count)
{
    if (size == 0u || p[size - 1u] == 0u)
        return 1;
    return zd_huf_stream_from(sh, p, size, out, count, (size - 1u) * 8u + zb_highbit(p[size - 1u]), 0);
}

/* ---- sequences ---- */
ZB_FN int zd_set_table(ZdShared* sh, int t, uint32_t mode, const uint8_t* p, uint32_t size, uint32_t* used)
{
    ZdFse* f = &sh->fse[t];
    *used = 0;
    int16_t* const norm = sh->norm + 64 * t; /* one 64-entry slice per table: all three are built together after the descriptions */
    sh->tb_build[t] = 0;
    if (mode == 0u)
    {
        const uint32_t nsym = zb_table_nsym(t);
        for (uint32_t s = 0; s < nsym; ++s)
            norm[s] = (int16_t)zb_default_norm(t, s);
        sh->tb_maxsym[t] = nsym - 1u;
        sh->tb_log[t] = zb_table_default_log(t);
        sh->tb_build[t] = 1;
        f->valid = 0; /* until zd_build_fse_par has run */
        return 0;
    }
    if (mode == 1u)
    {
        if (size < 1u)
            return 1;
        if (p[0] > (t == ZT_LL ? 35u : t == ZT_ML ? 52u : 31u))
            return 1;
        f->sym[0] = p[0];
        f->nb[0] = 0;
        f->base[0] = 0;
        f->log = 0;
        f->valid = 2;
        *used = 1;
        return 0;
    }
    if (mode == 2u)
    {
        uint32_t maxsym = t == ZT_LL ? 35u : t == ZT_ML ? 52u : 31u, tl = 0;
        const uint32_t hs = zd_read_ncount(p, size, norm, &maxsym, zb_table_max_log(t), &tl);
        if (hs == ZD_ERROR)
            return 1;
        *used = hs;
        if (maxsym > 63u || tl > 9u)
            return 1;
        sh->tb_maxsym[t] = maxsym;
        sh->tb_log[t] = tl;
        sh->tb_build[t] = 1;
        f->valid = 0;
        return 0;
    }
    return f->valid == 0u; /* Repeat_Mode needs a previous table */
}

/* ------------------------------------------------------------------------------------------------------------
 * One payload (one or more frames) -> dst.  Returns the number of bytes produced or ZD_ERROR.
 * lits: scratch of ZD_LIT_MAX + 32 bytes (global memory on the GPU).
 * ---------------------------------------------------------------------------------------------------------- */
/* piece_out0 == ZD_WHOLE: the payload as described above.
 * Otherwise PIECE mode, for frames whose blocks are known to be independent of each other (this library's own encoder, which says
 * so in a trailing skippable frame -- k_zstd.hip): src is ONE block (3-byte header + content) that regenerates the bytes from
 * dst[piece_out0] on; returns the number of bytes it produced.  The mode is STRICT: a match may not reach below piece_out0, repeat
 * offsets, treeless literals and repeated FSE tables are errors (they would need the state the previous block left) -- so a frame
 * that carries the marker without keeping its promise is rejected, never decoded differently from the serial decoder. */
#define ZD_WHOLE 0xFFFFFFFFu
ZB_FN uint32_t zd_decode_payload_ex(const uint8_t* src, uint32_t src_size, uint8_t* dst, uint32_t dst_cap, uint8_t* lits, ZdShared* sh,
                                    uint32_t zl, uint32_t piece_out0)
{
    const int piece = piece_out0 != ZD_WHOLE;
    uint32_t ip = 0;     /* every lane tracks the parse position through sh->v broadcasts */
    uint32_t out_total = piece ? piece_out0 : 0u;
    ZB_SERIAL(zl) { sh->v[ZDV_ERR] = 0; }
    ZB_SYNC();
    for (;;) /* frames */
    {
        uint32_t frame_start_out = out_total;
        uint32_t hdr_ok = 0, checksum = 0, skip = 0;
        if (ip == src_size)
            break;
        ZB_SERIAL(zl)
        {
            sh->v[ZDV_DONE] = 0;
            sh->v[ZDV_LEN] = 0;
            if (piece)
            {
                /* no frame header: the state a frame starts with */
                sh->v[ZDV_NLIT] = 0xFFFFFFFFu;
                sh->v[ZDV_MODE] = ZB_BLOCK_MAX;
                sh->v[ZDV_DONE] = 1;
                sh->v[ZDV_BYTE] = 0;
                sh->rep[0] = 1;
                sh->rep[1] = 4;
                sh->rep[2] = 8;
                sh->huf_valid = 0;
                sh->fse[0].valid = sh->fse[1].valid = sh->fse[2].valid = 0;
            }
            else if (src_size - ip < 4u)
                ZD_SET_ERR(sh);
            else
            {
                const uint32_t magic = zd_le(src + ip, 4);
                if ((magic & 0xFFFFFFF0u) == 0x184D2A50u)
                {
                    /* skippable frame: magic, u32 size, that many bytes */
                    if (src_size - ip < 8u || zd_le(src + ip + 4u, 4) > src_size - ip - 8u)
                        ZD_SET_ERR(sh);
                    else
                    {
                        sh->v[ZDV_DONE] = 2;
                        sh->v[ZDV_LEN] = 8u + zd_le(src + ip + 4u, 4);
                    }
                }
                else if (magic != 0xFD2FB528u || src_size - ip < 6u)
                    ZD_SET_ERR(sh);
                else
                {
                    const uint32_t fhd = src[ip + 4u];
                    const uint32_t dict = fhd & 3u, single = (fhd >> 5) & 1u, fcs = fhd >> 6;
                    const uint32_t dsz = dict == 3u ? 4u : dict;
                    const uint32_t fsz = fcs == 0u ? single : fcs == 1u ? 2u : fcs == 2u ? 4u : 8u;
                    const uint32_t h = 5u + (single ? 0u : 1u) + dsz + fsz;
                    if ((fhd & 8u) || src_size - ip < h)
                        ZD_SET_ERR(sh);
                    else if (dsz && zd_le(src + ip + 5u + (single ? 0u : 1u), dsz) != 0u)
                        ZD_SET_ERR(sh); /* dictionaries are not supported */
                    else
                    {
                        /* content size (0xFFFFFFFF = not stated) and the largest block the frame may hold */
                        const uint8_t* fp = src + ip + 5u + (single ? 0u : 1u) + dsz;
                        uint32_t content = 0xFFFFFFFFu, wsize = 0xFFFFFFFFu;
                        if (fsz == 8u && zd_le(fp + 4u, 4) != 0u)
                            ZD_SET_ERR(sh); /* more than 4 GiB cannot be a stored block */
                        else if (fsz)
                            content = zd_le(fp, fsz > 4u ? 4u : fsz) + (fsz == 2u ? 256u : 0u);
                        if (single)
                            wsize = content;
                        else
                        {
                            const uint32_t wd = src[ip + 5u], wlog = 10u + (wd >> 3);
                            if (wlog > 31u)
                                ZD_SET_ERR(sh); /* ZSTD_WINDOWLOG_MAX */
                            else if (wlog < 31u)
                                wsize = (1u << wlog) + ((1u << wlog) >> 3) * (wd & 7u);
                        }
                        sh->v[ZDV_NLIT] = content;
                        sh->v[ZDV_MODE] = wsize < ZB_BLOCK_MAX ? wsize : ZB_BLOCK_MAX;
                        sh->v[ZDV_DONE] = 1;
                        sh->v[ZDV_LEN] = h;
                        sh->v[ZDV_BYTE] = (fhd >> 2) & 1u;
                        sh->rep[0] = 1;
                        sh->rep[1] = 4;
                        sh->rep[2] = 8;
                        sh->huf_valid = 0;
                        sh->fse[0].valid = sh->fse[1].valid = sh->fse[2].valid = 0;
                    }
                }
            }
        }
        ZB_SYNC();
        if (sh->v[ZDV_ERR])
            return ZD_FAIL_AT(sh);
        hdr_ok = sh->v[ZDV_DONE];
        skip = sh->v[ZDV_LEN];
        checksum = sh->v[ZDV_BYTE];
        const uint32_t content_size = sh->v[ZDV_NLIT], block_max = sh->v[ZDV_MODE];
        ip += skip;
        if (hdr_ok == 2u)
            continue;

        for (;;) /* blocks */
        {
            uint32_t last, type, bsize;
            ZB_SYNC();
            if (src_size - ip < 3u)
                return ZD_FAIL_AT(sh);
            {
                const uint32_t bh = zd_le(src + ip, 3);
                last = bh & 1u;
                type = (bh >> 1) & 3u;
                bsize = bh >> 3;
            }
            ip += 3u;
            if (type == 3u || bsize > ZB_BLOCK_MAX || (type != 2u && bsize > block_max))
                return ZD_FAIL_AT(sh);
            const uint32_t block_out0 = out_total;
            if (type == 0u)
            {
                if (bsize > src_size - ip || bsize > dst_cap - out_total)
                    return ZD_FAIL_AT(sh);
                ZB_PAR_FOR(i, bsize) dst[out_total + i] = src[ip + i];
                ip += bsize;
                out_total += bsize;
            }
            else if (type == 1u)
            {
                if (src_size - ip < 1u || bsize > dst_cap - out_total)
                    return ZD_FAIL_AT(sh);
                {
                    const uint8_t b = src[ip];
                    ZB_PAR_FOR(i, bsize) dst[out_total + i] = b;
                }
                ip += 1u;
                out_total += bsize;
            }
            else
            {
                /* ---------------- Compressed_Block ---------------- */
                ZD_MARK(11);
                const uint8_t* blk = src + ip;
                uint32_t nlit = 0, lit_mode = 0, lit_src = 0, lit_csize = 0, lit_hdr = 0, seq_pos = 0;
                if (bsize > src_size - ip || bsize < 2u)
                    return ZD_FAIL_AT(sh);
                /* literals section header (every lane parses the few bytes itself) */
                {
                    const uint32_t b0 = blk[0];
                    const uint32_t sf = (b0 >> 2) & 3u;
                    lit_mode = b0 & 3u;
                    if (lit_mode < 2u)
                    {
                        lit_hdr = (sf & 1u) == 0u ? 1u : sf == 1u ? 2u : 3u;
                        if (lit_hdr > bsize)
                            return ZD_FAIL_AT(sh);
                        nlit = lit_hdr == 1u ? b0 >> 3 : lit_hdr == 2u ? zd_le(blk, 2) >> 4 : zd_le(blk, 3) >> 4;
                        lit_csize = lit_mode == 0u ? nlit : 1u;
                    }
                    else
                    {
                        lit_hdr = sf < 2u ? 3u : sf == 2u ? 4u : 5u;
                        if (lit_hdr > bsize)
                            return ZD_FAIL_AT(sh);
                        if (lit_hdr == 3u)
                        {
                            const uint32_t h = zd_le(blk, 3);
                            nlit = (h >> 4) & 0x3FFu;
                            lit_csize = h >> 14;
                        }
                        else if (lit_hdr == 4u)
                        {
                            const uint32_t h = zd_le(blk, 4);
                            nlit = (h >> 4) & 0x3FFFu;
                            lit_csize = h >> 18;
                        }
                        else
                        {
                            const uint32_t h = zd_le(blk, 4);
                            nlit = (h >> 4) & 0x3FFFFu;
                            lit_csize = (h >> 22) | ((uint32_t)blk[4] << 10);
                        }
                        lit_src = sf; /* stream count: 0 -> one stream */
                    }
                    if (nlit > ZD_LIT_MAX || lit_csize > bsize - lit_hdr)
                        return ZD_FAIL_AT(sh);
                }
                /* literals */
                if (lit_mode == 0u)
                {
                    ZB_PAR_FOR(i, nlit) lits[i] = blk[lit_hdr + i];
                }
                else if (lit_mode == 1u)
                {
                    const uint8_t b = blk[lit_hdr];
                    ZB_PAR_FOR(i, nlit) lits[i] = b;
                }
                else
                {
                    const uint8_t* lp = blk + lit_hdr;
                    uint32_t tree = 0;
                    ZB_SERIAL(zl)
                    {
                        sh->v[ZDV_LEN] = 0;
                        if (lit_mode == 2u)
                        {
                            tree = zd_read_huf_tree(sh, lp, lit_csize);
                            if (tree == ZD_ERROR)
                                ZD_SET_ERR(sh);
                            else
                                sh->v[ZDV_LEN] = tree;
                        }
                        else if (!sh->huf_valid)
                            ZD_SET_ERR(sh); /* Treeless without a previous tree */
                    }
                    ZB_SYNC();
                    if (sh->v[ZDV_ERR])
                        return ZD_FAIL_AT(sh);
                    tree = sh->v[ZDV_LEN];
                    if (lit_src == 0u)
                    {
                        ZB_SERIAL(zl)
                        {
                            if (zd_huf_stream(sh, lp + tree, lit_csize - tree, lits, nlit))
                                ZD_SET_ERR(sh);
                        }
                    }
                    else
                    {
                        /* four streams: jump table of three u16 sizes, then the streams; lanes 0..3 decode one each */
                        if (lit_csize - tree < 10u)
                            return ZD_FAIL_AT(sh);
                        {
                            const uint32_t s1 = zd_le(lp + tree, 2), s2 = zd_le(lp + tree + 2u, 2), s3 = zd_le(lp + tree + 4u, 2);
                            const uint32_t body = lit_csize - tree - 6u;
                            const uint32_t seg = (nlit + 3u) >> 2;
                            if (s1 + s2 + s3 >= body || 3u * seg > nlit)
                                return ZD_FAIL_AT(sh);
                            if (!(ZD_ABLATE & 2u))
                            ZB_PAR_FOR(k, 4u)
                            {
                                const uint32_t so = k == 0u ? 0u : k == 1u ? s1 : k == 2u ? s1 + s2 : s1 + s2 + s3;
                                const uint32_t ss = k == 0u ? s1 : k == 1u ? s2 : k == 2u ? s3 : body - s1 - s2 - s3;
                                const uint32_t cnt = k == 3u ? nlit - 3u * seg : seg;
                                if (zd_huf_stream(sh, lp + tree + 6u + so, ss, lits + k * seg, cnt))
                                    { zb_atomic_or(&sh->v[ZDV_ERR], 1u); sh->v[ZDV_SRC] = __LINE__; }
                            }
                        }
                    }
                }
                ZB_SYNC();
                if (sh->v[ZDV_ERR])
                    return ZD_FAIL_AT(sh);
                ZD_MARK(12);
                /* sequences section */
                seq_pos = lit_hdr + lit_csize;
                {
                    uint32_t nbseq = 0;
                    if (seq_pos >= bsize)
                        return ZD_FAIL_AT(sh);
                    {
                        const uint32_t b0 = blk[seq_pos];
                        if (b0 < 128u)
                        {
                            nbseq = b0;
                            seq_pos += 1u;
                        }
                        else if (b0 < 255u)
                        {
                            if (seq_pos + 2u > bsize)
                                return ZD_FAIL_AT(sh);
                            nbseq = ((b0 - 128u) << 8) + blk[seq_pos + 1u];
                            seq_pos += 2u;
                        }
                        else
                        {
                            if (seq_pos + 3u > bsize)
                                return ZD_FAIL_AT(sh);
                            nbseq = zd_le(blk + seq_pos + 1u, 2) + 0x7F00u;
                            seq_pos += 3u;
                        }
                    }
                    if (nbseq == 0u)
                    {
                        if (seq_pos != bsize || nlit > dst_cap - out_total)
                            return ZD_FAIL_AT(sh);
                        ZB_PAR_FOR(i, nlit) dst[out_total + i] = lits[i];
                        out_total += nlit;
                    }
                    else
                    {
                        /* tables (lane 0), then the sequence loop: lane 0 decodes one sequence, all lanes copy it */
                        ZD_MARK(16);
                        ZB_SERIAL(zl)
                        {
                            uint32_t p = seq_pos;
                            sh->v[ZDV_LEN] = 0;
                            if (p >= bsize)
                                ZD_SET_ERR(sh);
                            else
                            {
                                const uint32_t modes = blk[p++];
                                if (modes & 3u)
                                    ZD_SET_ERR(sh);
                                for (int t = 0; t < 3 && !sh->v[ZDV_ERR]; ++t) /* LL, OF, ML */
                                {
                                    uint32_t used = 0;
                                    if (zd_set_table(sh, t, (modes >> (6 - 2 * t)) & 3u, blk + p, bsize - p, &used))
                                        ZD_SET_ERR(sh);
                                    p += used;
                                }
                                sh->v[ZDV_LEN] = p;
                            }
                        }
                        ZB_SYNC();
                        if (sh->v[ZDV_ERR])
                            return ZD_FAIL_AT(sh);
                        ZD_MARK(17); /* 17: table descriptions (lane 0), 18: tables (all lanes) */
                        if (piece) /* the tables described above; a piece is done with its Huffman table here: scratch for all lanes */
                        {
                            for (int t = 0; t < 3; ++t)
                                if (sh->tb_build[t])
                                {
                                    if (zd_build_fse_par(&sh->fse[t], sh->norm + 64 * t, sh->tb_maxsym[t], sh->tb_log[t], sh->cum,
                                                         (uint32_t*)sh->huf, sh->huf + 2u * ZD_FSE_PAR_MASK_WORDS, zl))
                                        return ZD_FAIL_AT(sh);
                                    ZB_SYNC_LDS();
                                }
                            sh->huf_valid = 0;
                        }
                        else /* a later block may still want the Huffman table (treeless literals) */
                        {
                            ZB_SERIAL(zl)
                            {
                                for (int t = 0; t < 3; ++t)
                                    if (sh->tb_build[t] &&
                                        zd_build_fse(&sh->fse[t], sh->norm + 64 * t, sh->tb_maxsym[t], sh->tb_log[t], sh->next))
                                        ZD_SET_ERR(sh);
                            }
                            ZB_SYNC();
                            if (sh->v[ZDV_ERR])
                                return ZD_FAIL_AT(sh);
                        }
                        ZB_SYNC();
                        ZD_MARK(18);
                        seq_pos = sh->v[ZDV_LEN];
                        if (seq_pos >= bsize)
                            return ZD_FAIL_AT(sh);
                        if (piece && sh->v[ZDV_PREP])
                        {
                            ZB_SERIAL(zl)
                            {
                                sh->v[ZDV_LL] = nbseq;
                                sh->v[ZDV_ML] = nlit;
                                sh->v[ZDV_OFF] = ip + seq_pos;
                                sh->v[ZDV_LEN] = bsize - seq_pos;
                            }
                            ZB_SYNC();
                            return ZD_PREPARED;
                        }
                        {
                            ZdBack br;
                            uint32_t st[3] = {0, 0, 0};
                            uint32_t litpos = 0;
                            br.p = 0;
                            br.size = 0;
                            br.pos = 0;
                            br.over = 0;
                            br.cont = 0;
                            br.cpos = br.cvalid = 0;
                            ZB_SERIAL(zl)
                            {
                                if (zd_back_open(&br, blk + seq_pos, bsize - seq_pos))
                                    ZD_SET_ERR(sh);
                                else
                                {
                                    st[ZT_LL] = zd_back_read(&br, sh->fse[ZT_LL].log);
                                    st[ZT_OF] = zd_back_read(&br, sh->fse[ZT_OF].log);
                                    st[ZT_ML] = zd_back_read(&br, sh->fse[ZT_ML].log);
                                    if (br.over)
                                        ZD_SET_ERR(sh);
                                }
                            }
                            /* Output positions below `synced` are known to have landed (the last full ZB_SYNC); a match only
                             * waits for the stores before it when its source reaches past that -- near matches do, the
                             * others (and every literal run) are issued back to back. */
                            uint32_t synced = out_total;
                            ZD_MARK(13); /* 12: literals section, 13: sequence tables + setup */
                            for (uint32_t n = 0; n < nbseq; ++n)
                            {
                                uint32_t ll, ml, off;
                                ZB_SERIAL(zl)
                                {
                                    if (!sh->v[ZDV_ERR])
                                    {
                                        const ZdFse* fl = &sh->fse[ZT_LL];
                                        const ZdFse* fo = &sh->fse[ZT_OF];
                                        const ZdFse* fm = &sh->fse[ZT_ML];
                                        const uint32_t lc = fl->sym[st[ZT_LL]], oc = fo->sym[st[ZT_OF]], mc = fm->sym[st[ZT_ML]];
                                        uint32_t ov, o;
                                        if (oc > 31u || lc > 35u || mc > 52u)
                                            ZD_SET_ERR(sh);
                                        /* All bit fields of a sequence -- offset, match length and literal length extra bits,
                                         * then the three state updates -- come out of ONE 8-byte load when they add up to at
                                         * most 56 bits and 64 bits are left (nothing can run out, so nothing is checked); the
                                         * field-by-field reader handles everything else, and every malformation. */
                                        const uint32_t ob = oc & 31u, mb = zb_ml_bits(mc & 63u), lb = zb_ll_bits(lc & 63u);
                                        const uint32_t more = n + 1u < nbseq ? 1u : 0u;
                                        const uint32_t nbl = more ? fl->nb[st[ZT_LL]] : 0u, nbm = more ? fm->nb[st[ZT_ML]] : 0u,
                                                       nbo = more ? fo->nb[st[ZT_OF]] : 0u;
                                        const uint32_t total = ob + mb + lb + nbl + nbm + nbo;
                                        uint64_t t = 0;
                                        const int fastbits = total <= 56u && br.pos >= 64u && !sh->v[ZDV_ERR];
                                        if (fastbits)
                                        {
                                            const uint32_t b0 = ((br.pos + 7u) >> 3) - 8u;
                                            __builtin_memcpy(&t, br.p + b0, 8);
                                            t <<= (8u - (br.pos & 7u)) & 7u; /* bit pos-1 on top */
                                            br.pos -= total;
                                        }
#define ZD_TAKE(nb) ((uint32_t)((t >> 1) >> (63u - (nb)))) /* the top nb bits (0 for nb == 0) */
                                        if (fastbits)
                                        {
                                            ov = (1u << ob) + ZD_TAKE(ob);
                                            t <<= ob;
                                            ml = zb_ml_base(mc) + 3u + ZD_TAKE(mb);
                                            t <<= mb;
                                            ll = zb_ll_base(lc) + ZD_TAKE(lb);
                                            t <<= lb;
                                        }
                                        else
                                        {
                                        ov = (1u << (oc & 31u)) + zd_back_read32(&br, oc & 31u);
                                        ml = zb_ml_base(mc) + 3u + zd_back_read(&br, zb_ml_bits(mc));
                                        ll = zb_ll_base(lc) + zd_back_read(&br, zb_ll_bits(lc));
                                        }
                                        if (ov > 3u)
                                        {
                                            o = ov - 3u;
                                            sh->rep[2] = sh->rep[1];
                                            sh->rep[1] = sh->rep[0];
                                            sh->rep[0] = o;
                                        }
                                        else
                                        {
                                            const uint32_t idx = ov + (ll == 0u ? 1u : 0u); /* 1..4 */
                                            if (idx == 1u)
                                                o = sh->rep[0];
                                            else
                                            {
                                                o = idx == 4u ? sh->rep[0] - 1u : sh->rep[idx - 1u];
                                                if (idx >= 3u)
                                                    sh->rep[2] = sh->rep[1];
                                                sh->rep[1] = sh->rep[0];
                                                sh->rep[0] = o;
                                            }
                                        }
                                        if (fastbits)
                                        {
                                            if (more)
                                            {
                                                st[ZT_LL] = fl->base[st[ZT_LL]] + ZD_TAKE(nbl);
                                                t <<= nbl;
                                                st[ZT_ML] = fm->base[st[ZT_ML]] + ZD_TAKE(nbm);
                                                t <<= nbm;
                                                st[ZT_OF] = fo->base[st[ZT_OF]] + ZD_TAKE(nbo);
                                            }
                                        }
                                        else if (n + 1u < nbseq)
                                        {
                                            st[ZT_LL] = fl->base[st[ZT_LL]] + zd_back_read(&br, fl->nb[st[ZT_LL]]);
                                            st[ZT_ML] = fm->base[st[ZT_ML]] + zd_back_read(&br, fm->nb[st[ZT_ML]]);
                                            st[ZT_OF] = fo->base[st[ZT_OF]] + zd_back_read(&br, fo->nb[st[ZT_OF]]);
                                        }
#undef ZD_TAKE
                                        if (br.over || o == 0u || ll > nlit - litpos || ll > dst_cap - out_total ||
                                            ml > dst_cap - out_total - ll || o > out_total - frame_start_out + ll || (piece && ov <= 3u))
                                            ZD_SET_ERR(sh);
                                        if (n + 1u == nbseq && br.pos != 0u)
                                            ZD_SET_ERR(sh); /* the bit-stream must be consumed exactly */
                                        sh->v[ZDV_LL] = ll;
                                        sh->v[ZDV_ML] = ml;
                                        sh->v[ZDV_OFF] = o;
                                    }
                                }
                                ZB_SYNC_LDS();
                                ZD_MARK(14); /* sequence decode (lane 0) */
                                if (sh->v[ZDV_ERR])
                                    return ZD_FAIL_AT(sh);
                                ll = sh->v[ZDV_LL];
                                ml = sh->v[ZDV_ML];
                                off = sh->v[ZDV_OFF];
                                if (!(ZD_ABLATE & 1u))
                                {
                                    uint8_t* o = dst + out_total;
                                    const uint8_t* l = lits + litpos;
                                    ZB_PAR_FOR(i, ll) o[i] = l[i];
                                    {
                                        const uint32_t mpos = out_total + ll;
                                        const uint32_t src_hi = off >= ml ? mpos - off + ml : mpos; /* one past the last byte read */
                                        if (src_hi > synced)
                                        {
                                            ZB_SYNC(); /* the match reads bytes (maybe the literals just written) that are in flight */
                                            synced = mpos;
                                        }
                                    }
                                    {
                                        uint8_t* m = o + ll;
                                        const uint8_t* ref = m - off;
                                        if (off >= ml)
                                        {
                                            ZB_PAR_FOR(i, ml) m[i] = ref[i];
                                        }
                                        else
                                        {
                                            ZB_PAR_FOR(i, ml) m[i] = ref[i % off]; /* overlapping match == periodic pattern */
                                        }
                                    }
                                }
                                litpos += ll;
                                out_total += ll + ml;
                                ZB_SYNC_LDS(); /* lane 0 rewrites the sequence slots next */
                                ZD_MARK(15); /* copies */
                            }
                            ZB_SYNC();
                            /* literals after the last sequence */
                            if (nlit - litpos > dst_cap - out_total)
                                return ZD_FAIL_AT(sh);
                            ZB_PAR_FOR(i, nlit - litpos) dst[out_total + i] = lits[litpos + i];
                            out_total += nlit - litpos;
                        }
                    }
                }
                ip += bsize;
            }
            if (out_total - block_out0 > block_max)
                return ZD_FAIL_AT(sh); /* Block_Maximum_Size = min(window, 128 KiB) */
            if (last || piece)
                break;
        }
        if (piece)
        {
            if (ip != src_size)
                return ZD_FAIL_AT(sh);
            break;
        }
        if (content_size != 0xFFFFFFFFu && out_total - frame_start_out != content_size)
            return ZD_FAIL_AT(sh);
        if (checksum)
        {
            if (src_size - ip < 4u)
                return ZD_FAIL_AT(sh);
            ip += 4u; /* XXH64 content checksum: skipped, not verified */
        }
    }
    ZB_SYNC();
    if (sh->v[ZDV_ERR])
        return ZD_FAIL_AT(sh);
    return piece ? out_total - piece_out0 : out_total;
}

ZB_FN uint32_t zd_decode_payload(const uint8_t* src, uint32_t src_size, uint8_t* dst, uint32_t dst_cap, uint8_t* lits, ZdShared* sh,
                                 uint32_t zl)
{
    return zd_decode_payload_ex(src, src_size, dst, dst_cap, lits, sh, zl, ZD_WHOLE);
}

#endif /* ZSTD_DECODE_CORE_H */
