/* zstd_block_core.h -- encoder for ONE zstd Compressed_Block (RFC 8878 §3.1.1.2-3.1.1.4) from LZ sequences.
 *
 * What the reference does on this path: `ZStdCompressionAPI_Compress` hands each stored block to
 * `ZSTD_compressCCtx` (lib/zstd/longtail_zstd.c:105-142); the bytes that come back only have to be a zstd frame the
 * reference's `ZSTD_decompressDCtx` (longtail_zstd.c:144-177) turns back into the block.  This file is NOT a
 * restatement of zstd's compressor: it is our own entropy stage, written against the FORMAT as the reference's
 * decoder implements it (file:line citations are to /root/reference/lib/zstd/ext):
 *   literals section, Huffman tree description, 4-stream layout   decompress/zstd_decompress_block.c:135-345,
 *                                                                 common/entropy_common.c:236-327 (HUF_readStats)
 *   FSE table description (NCount)                                common/entropy_common.c:42-187
 *   FSE decoding table construction (symbol spread, nbBits/base)  decompress/zstd_decompress_block.c:484-603
 *   sequences section header, modes, decode order of the fields   decompress/zstd_decompress_block.c:700-760, 1240-1345
 *   predefined distributions, LL/ML code bit counts               common/zstd_internal.h:123-168
 *   two-state FSE stream of the Huffman weights                   common/fse_decompress.c:174-238
 *
 * ONE source, two execution models.  The includer defines
 *   ZB_LANES          number of cooperating lanes (64 on the GPU: one wavefront per zstd block; 1 on the host)
 *   ZB_FN             function qualifiers
 *   ZB_SYNC()         make earlier writes of all lanes visible to all lanes
 *   zb_atomic_add(p,v) / zb_atomic_or(p,v)   32-bit atomics on shared / global words
 *   zb_scan_excl(v, &total)                  exclusive prefix sum of v over the lanes (lane 0 first) and its total;
 *                                            called by all lanes together (with one lane: 0 and v)
 *   with ZB_LANES > 1 also: zb_ballot(pred) (64-bit mask of the lanes with pred), zb_shfl(v, lane) (v of another lane; all
 *                                            lanes call), zb_reduce_max(v) -- the table builders below have an all-lanes form
 *                                            (zb_normalize_par, zb_build_enc_table_par) that produces the SAME tables as the
 *                                            serial form the one-lane build runs: the tables are functions of the counts alone
 * Every phase is either `ZB_SERIAL(zl)` (lane 0), or a `ZB_PAR_FOR` whose iterations only interact through those
 * commutative atomics, so the bytes produced do not depend on ZB_LANES: the one-lane host build (zstd_model.c in the test
 * infrastructure) is a bit-exact model of the kernel (k_zstd.hip), and runs here without a GPU against the
 * reference decoder.
 */
#ifndef ZSTD_BLOCK_CORE_H
#define ZSTD_BLOCK_CORE_H

#include <stdint.h>

#ifndef ZB_SYNC_LDS /* orders the lanes' LDS (shared-struct) accesses only; the host model needs nothing */
#define ZB_SYNC_LDS() ZB_SYNC()
#endif
#ifndef ZB_DBG
#define ZB_DBG 0u /* host model only: 1 raw literals, 2 never predefined, 4 never FSE-compressed tables, 8 no sampled noise test */
#endif
#ifndef ZB_MARK
#define ZB_MARK(i) ((void)0) /* profiling hook of the kernel build (phase boundaries) */
#endif
#define ZB_PAR_FOR(i, n) for (uint32_t i = zl; i < (uint32_t)(n); i += ZB_LANES)
/* n items handled K per lane and trip, lanes next to each other in every one of the K accesses: trip t covers the items
 * [t K LANES, (t + 1) K LANES), lane zl takes t K LANES + q LANES + zl for q < K (each to be checked against n); every lane makes
 * every trip */
#define ZB_PAR_FOR_K(t, n, K) for (uint32_t t = 0; t * (K) * ZB_LANES < (uint32_t)(n); ++t)
#ifndef ZB_UNROLL
#define ZB_UNROLL /* the kernel build: _Pragma("unroll") -- arrays indexed by such a loop's counter must stay in registers */
#endif
#define ZB_SERIAL(zl) if ((zl) == 0)

#define ZB_BLOCK_MAX (128u * 1024u) /* Block_Maximum_Size, zstd.h:142-143 */
#define ZB_UNIT 4096u               /* bytes of input per match-finder unit */
#define ZB_UNIT_SEQ_MAX 1024u       /* a sequence covers >= 4 bytes */
#define ZB_MAX_UNITS (ZB_BLOCK_MAX / ZB_UNIT)
#define ZB_SEQ_MAX (ZB_MAX_UNITS * ZB_UNIT_SEQ_MAX)
#define ZB_CHUNKS 64u               /* bit-stream work items: 4 literal streams x 16, or 64 runs of sequences */
#define ZB_OUT_BYTES (ZB_BLOCK_MAX + 2048u)
#define ZB_HUF_MAXBITS 11u          /* LitHufLog, zstd_internal.h:105 */

/* match-finder record of one sequence inside a unit: literals before the match, match length, offset */
#define ZB_REC(lit, mlen, off) ((uint64_t)(lit) | ((uint64_t)(mlen) << 16) | ((uint64_t)(off) << 32))
/* merged sequence of the block: lit 20 bits | mlen 16 bits | off 17 bits */
#define ZB_SEQ_LIT(s) ((uint32_t)((s) & 0xFFFFFu))
#define ZB_SEQ_ML(s) ((uint32_t)(((s) >> 20) & 0xFFFFu))
#define ZB_SEQ_OFF(s) ((uint32_t)((s) >> 36))
/* sub-block layout: the offset field holds either the offset (Offset_Value = offset + 3) or, with bit 27 set, a repeat code 1..3 */
#define ZB_OFF_REP 0x8000000u
#define ZB_SEQ_OFV(s) ((ZB_SEQ_OFF(s) & ZB_OFF_REP) ? (ZB_SEQ_OFF(s) & 3u) : ZB_SEQ_OFF(s) + 3u)

typedef struct ZbUnitMeta
{
    uint32_t nseq; /* sequences found in the unit */
    uint32_t nlit; /* literal bytes of the unit (all runs, tail included), stored contiguously */
    uint32_t tail; /* literals after the unit's last sequence (== nlit when nseq == 0) */
    uint32_t uniform; /* 0x100 | b when every byte of the unit equals b, else 0 */
} ZbUnitMeta;

typedef struct ZbInput
{
    const ZbUnitMeta* meta;  /* [nunits] */
    const uint8_t* unit_lits; /* unit u at + u * ZB_UNIT */
    const uint64_t* unit_recs; /* unit u at + u * ZB_UNIT_SEQ_MAX */
    uint32_t nunits;
    uint32_t raw_size;
    const uint8_t* src; /* the block's raw bytes (any alignment) or NULL.  When given, a unit WITHOUT a sequence has no
                         * literal buffer: its literals are its own bytes, src + u * ZB_UNIT (the match finder does not
                         * copy what nobody may ever need) */
    uint32_t flags;     /* ZB_F_* */
} ZbInput;
#define ZB_F_REPCODES 1u /* sub-block layout: repeat-offset codes whose history entry was SET INSIDE THE BLOCK (zb_encode_piece_sub, phase 1) */

typedef struct ZbScratch /* global memory owned by the lanes of one block encoder */
{
    uint64_t* seqs;  /* [ZB_SEQ_MAX]        merged sequences */
    uint16_t* sbits; /* [4 * ZB_SEQ_MAX]    FSE state-transition bits per sequence: nbBits << 10 | bits -- three planes (zb_encode_block),
                      * or one 64-bit word per sequence: LL | OF << 16 | ML << 32 (zb_encode_piece_sub) */
    uint32_t* out;   /* [ZB_OUT_BYTES / 4]  the encoded block */
} ZbScratch;

enum
{
    ZV_NBSEQ,
    ZV_NLIT,
    ZV_HUF_OK,
    ZV_HUF_MAXBITS,
    ZV_HUF_MAXSYM,
    ZV_HUF_NSYM,
    ZV_SKIP,
    ZV_LIT_MODE, /* 0 raw, 2 huffman */
    ZV_LIT_HDR,  /* bytes of the literals section header */
    ZV_TREE_BYTES,
    ZV_LIT_END, /* byte offset just after the literals section */
    ZV_SEQ_BITS0, /* byte offset of the sequence bit-stream */
    ZV_SEQ_TOTALBITS,
    ZV_OUT_SIZE,
    ZV_SRCMASK,      /* bit u: the literals of unit u are read from in->src */
    ZV_STREAM_BYTES, /* +0..3 */
    ZV_STREAM_BASE = ZV_STREAM_BYTES + 4, /* +0..3, byte offsets */
    ZV_FINAL_STATE = ZV_STREAM_BASE + 4,  /* +0..2 */
    ZV_COUNT = ZV_FINAL_STATE + 3
};

typedef struct ZbShared /* LDS on the GPU (about 9.5 KiB per wave) */
{
    uint32_t lit_hist[256];
    uint32_t sort_key[256]; /* Huffman construction: present symbols sorted by count.  sort_key + huf_w (2 KiB) are reused
                             * once the code lengths exist: symbol spread of the FSE table builds, then the FSE code tile */
    uint32_t huf_w[256];    /* ... weights / parent links / depths (Moffat-Katajainen, in place) */
    uint8_t huf_l[256];     /* ... code length per sorted position */
    uint8_t tree[160];      /* Huffman tree description */
    uint16_t cursor[3][64];
    uint16_t huf_code[256];
    uint8_t huf_len[256];
    uint32_t sym_hist[3][64]; /* LL / OF / ML code histograms */
    int16_t norm[3][64];
    uint16_t sym_start[3][64];
    uint16_t state_tab[3][512];
    uint8_t mode[4], table_log[4], rle_sym[4];
    uint32_t useq_base[ZB_MAX_UNITS + 1], ulit_base[ZB_MAX_UNITS + 1], carry[ZB_MAX_UNITS];
    uint32_t part2[4]; /* bits of the four literal streams */
    uint32_t small[2][16]; /* scratch the serial builders index by data (code lengths, weights): a local array indexed that way is
                            * private memory -- or a 16-way select chain -- on the GPU */
    uint32_t v[ZV_COUNT];
} ZbShared;

/* work space of the FSE table builds: 3 x 512 bytes over sort_key / huf_w */
#define ZB_SPREAD(sh, t) (((uint8_t*)(sh)->sort_key) + 512u * (uint32_t)(t))

/* table indices */
#define ZT_LL 0
#define ZT_OF 1
#define ZT_ML 2

/* ------------------------------------------------------------------------------------------------------------
 * constants of the format
 * ---------------------------------------------------------------------------------------------------------- */
ZB_FN uint32_t zb_highbit(uint32_t v) /* v > 0 */
{
    uint32_t r = 0;
    while (v >>= 1)
        ++r;
    return r;
}

/* Literals_Length_Code: zstd_internal.h:123-129 gives the number of extra bits per code, which fixes the baselines:
 * 0..15 direct, then 16,18,20,22 (1 bit), 24,28 (2), 32,40 (3), 48 (4), 64 (6), 128 (7) ... */
ZB_FN uint32_t zb_ll_code(uint32_t v)
{
    if (v < 16u)
        return v;
    if (v < 24u)
        return 16u + ((v - 16u) >> 1);
    if (v < 32u)
        return 20u + ((v - 24u) >> 2);
    if (v < 48u)
        return 22u + ((v - 32u) >> 3);
    if (v < 64u)
        return 24u;
    return zb_highbit(v) + 19u;
}
ZB_FN uint32_t zb_ll_bits(uint32_t code)
{
    if (code < 16u)
        return 0u;
    if (code < 20u)
        return 1u;
    if (code < 22u)
        return 2u;
    if (code < 24u)
        return 3u;
    if (code == 24u)
        return 4u;
    return code - 19u;
}
ZB_FN uint32_t zb_ll_base(uint32_t code)
{
    if (code < 16u)
        return code;
    if (code < 20u)
        return 16u + ((code - 16u) << 1);
    if (code < 22u)
        return 24u + ((code - 20u) << 2);
    if (code < 24u)
        return 32u + ((code - 22u) << 3);
    if (code == 24u)
        return 48u;
    return 1u << (code - 19u);
}
/* Match_Length_Code on mlBase = match length - 3 (zstd_internal.h:140-148): 0..31 direct, then 32,34,36,38 (1 bit),
 * 40,44 (2), 48,56 (3), 64,80 (4), 96 (5), 128 (7), 256 (8) ... */
ZB_FN uint32_t zb_ml_code(uint32_t m)
{
    if (m < 32u)
        return m;
    if (m < 40u)
        return 32u + ((m - 32u) >> 1);
    if (m < 48u)
        return 36u + ((m - 40u) >> 2);
    if (m < 64u)
        return 38u + ((m - 48u) >> 3);
    if (m < 96u)
        return 40u + ((m - 64u) >> 4);
    if (m < 128u)
        return 42u;
    return zb_highbit(m) + 36u;
}
ZB_FN uint32_t zb_ml_bits(uint32_t code)
{
    if (code < 32u)
        return 0u;
    if (code < 36u)
        return 1u;
    if (code < 38u)
        return 2u;
    if (code < 40u)
        return 3u;
    if (code < 42u)
        return 4u;
    if (code == 42u)
        return 5u;
    return code - 36u;
}
ZB_FN uint32_t zb_ml_base(uint32_t code)
{
    if (code < 32u)
        return code;
    if (code < 36u)
        return 32u + ((code - 32u) << 1);
    if (code < 38u)
        return 40u + ((code - 36u) << 2);
    if (code < 40u)
        return 48u + ((code - 38u) << 3);
    if (code < 42u)
        return 64u + ((code - 40u) << 4);
    if (code == 42u)
        return 96u;
    return 1u << (code - 36u);
}

/* predefined distributions, zstd_internal.h:130-136, 149-157, 161-166 */
ZB_FN int zb_default_norm(int t, uint32_t s)
{
    if (t == ZT_LL)
    {
        if (s == 0u)
            return 4;
        if (s == 1u || s == 25u)
            return 3;
        if (s >= 32u)
            return -1;
        if ((s >= 13u && s <= 15u) || s >= 27u)
            return 1;
        return 2;
    }
    if (t == ZT_ML)
    {
        if (s == 0u)
            return 1;
        if (s == 1u)
            return 4;
        if (s == 2u)
            return 3;
        if (s <= 8u)
            return 2;
        if (s >= 46u)
            return -1;
        return 1;
    }
    if (s >= 24u)
        return -1;
    if (s >= 6u && s <= 8u)
        return 2;
    return 1;
}
ZB_FN uint32_t zb_table_nsym(int t) { return t == ZT_LL ? 36u : t == ZT_ML ? 53u : 29u; }
ZB_FN uint32_t zb_table_default_log(int t) { return t == ZT_OF ? 5u : 6u; }
ZB_FN uint32_t zb_table_max_log(int t) { return t == ZT_OF ? 8u : 9u; } /* zstd_internal.h:112-114 */

/* ------------------------------------------------------------------------------------------------------------
 * little-endian bit writer on 32-bit words (the destination is zeroed first; neighbours share words -> atomics)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct ZbBits
{
    uint32_t* dst;
    uint64_t acc;
    uint32_t nacc; /* < 32 between calls */
    uint32_t word;
} ZbBits;

ZB_FN void zb_bits_open(ZbBits* b, uint32_t* dst, uint32_t bitpos)
{
    b->dst = dst;
    b->acc = 0;
    b->nacc = bitpos & 31u;
    b->word = bitpos >> 5;
}
ZB_FN void zb_bits_put(ZbBits* b, uint32_t value, uint32_t n) /* n <= 32, value < 2^n */
{
    b->acc |= (uint64_t)value << b->nacc;
    b->nacc += n;
    if (b->nacc >= 32u)
    {
        zb_atomic_or(b->dst + b->word, (uint32_t)b->acc);
        ++b->word;
        b->acc >>= 32;
        b->nacc -= 32u;
    }
}
ZB_FN void zb_bits_close(ZbBits* b)
{
    if (b->acc)
        zb_atomic_or(b->dst + b->word, (uint32_t)b->acc);
}

/* ------------------------------------------------------------------------------------------------------------
 * FSE: normalisation, table description, encoding table
 * ---------------------------------------------------------------------------------------------------------- */
/* Scale `hist` (nsym entries, sum `total` >= 2, at least two non-zero) to sum 2^tl with every present symbol >= 1. */
ZB_FN void zb_normalize(const uint32_t* hist, uint32_t nsym, uint32_t total, uint32_t tl, int16_t* norm)
{
    const uint32_t size = 1u << tl;
    uint32_t sum = 0;
    for (uint32_t s = 0; s < nsym; ++s)
    {
        uint32_t v = 0;
        if (hist[s])
        {
            v = (uint32_t)(((uint64_t)hist[s] << tl) / total);
            if (v == 0u)
                v = 1u;
        }
        norm[s] = (int16_t)v;
        sum += v;
    }
    while (sum != size)
    {
        uint32_t best = 0;
        for (uint32_t s = 1; s < nsym; ++s)
            if (norm[s] > norm[best])
                best = s;
        if (sum < size)
        {
            norm[best] = (int16_t)(norm[best] + (int)(size - sum));
            sum = size;
        }
        else
        {
            uint32_t take = sum - size;
            if (take > (uint32_t)norm[best] - 1u)
                take = (uint32_t)norm[best] - 1u;
            if (take == 0u)
                break; /* cannot happen: a largest count of 1 with sum > size needs more present symbols than cells, and the callers
                        * choose tl >= 5, and >= 6 for more than 32 of the at most 64 symbols (zb_build_seq_tables) -- but a loop
                        * that cannot end is the wrong way to find out */
            norm[best] = (int16_t)(norm[best] - (int)take);
            sum -= take;
        }
    }
}

/* NCount writer: the exact inverse of FSE_readNCount_body (entropy_common.c:42-187).  Returns bytes written. */
ZB_FN uint32_t zb_write_ncount(uint8_t* dst, const int16_t* norm, uint32_t nsym, uint32_t tl)
{
    uint64_t acc = 0;
    uint32_t nacc = 0, pos = 0;
    int remaining = (int)(1u << tl) + 1;
    int threshold = (int)(1u << tl);
    uint32_t nbits = tl + 1u;
    uint32_t last = nsym;
    while (last > 0u && norm[last - 1u] == 0)
        --last; /* symbols after the last present one are implied */
    acc = tl - 5u;
    nacc = 4;
    uint32_t s = 0;
    while (s < last && remaining > 1)
    {
        const int count = norm[s++];
        const int maxv = (2 * threshold - 1) - remaining;
        uint32_t v = (uint32_t)(count + 1);
        remaining -= count < 0 ? -count : count;
        if ((int)v >= threshold)
            v += (uint32_t)maxv;
        /* small values take nbits-1 bits */
        {
            const uint32_t nb = (int)v < maxv ? nbits - 1u : nbits;
            acc |= (uint64_t)v << nacc;
            nacc += nb;
        }
        if (count == 0)
        {
            /* run of further zero-probability symbols: 2-bit repeat codes, 3 = "three more and continue" */
            uint32_t run = 0;
            while (s + run < last && norm[s + run] == 0)
                ++run;
            s += run;
            for (;;)
            {
                while (nacc >= 8u)
                {
                    dst[pos++] = (uint8_t)acc;
                    acc >>= 8;
                    nacc -= 8u;
                }
                if (run >= 3u)
                {
                    acc |= (uint64_t)3u << nacc;
                    nacc += 2u;
                    run -= 3u;
                }
                else
                {
                    acc |= (uint64_t)run << nacc;
                    nacc += 2u;
                    break;
                }
            }
        }
        while (remaining < threshold && threshold > 1)
        {
            --nbits;
            threshold >>= 1;
        }
        while (nacc >= 8u)
        {
            dst[pos++] = (uint8_t)acc;
            acc >>= 8;
            nacc -= 8u;
        }
    }
    if (nacc)
        dst[pos++] = (uint8_t)acc;
    return pos;
}

/* Encoding table of one FSE distribution.  The decoder (zstd_decompress_block.c:484-603, fse_decompress.c:60-140)
 * spreads the symbols over the 2^tl states with step (size>>1)+(size>>3)+3, "less than one" symbols (-1) taking
 * the last cells, and gives the k-th cell of symbol s (in state order) nextState = count+k, nbBits = tl -
 * highbit(nextState).  Inverting that: from state x in [size, 2*size), emitting symbol s with `count` cells means
 * writing the low nb bits of x, nb chosen so that (x >> nb) lies in [count, 2*count), and moving to
 * size + cell[(x >> nb) - count].  state_tab lists the cells of every symbol in state order; sym_start[s] is the
 * first entry of s. */
ZB_FN void zb_build_enc_table(const int16_t* norm, uint32_t nsym, uint32_t tl, uint8_t* spread, uint16_t* state_tab,
                              uint16_t* sym_start, uint16_t* cursor)
{
    const uint32_t size = 1u << tl, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t high = size - 1u, pos = 0, cum = 0;
    for (uint32_t s = 0; s < nsym; ++s)
        if (norm[s] == -1)
            spread[high--] = (uint8_t)s;
    for (uint32_t s = 0; s < nsym; ++s)
    {
        sym_start[s] = (uint16_t)cum;
        cum += (uint32_t)(norm[s] < 0 ? 1 : norm[s]);
        for (int i = 0; i < norm[s]; ++i)
        {
            spread[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high)
                pos = (pos + step) & mask;
        }
    }
    /* cells in state order -> per-symbol lists */
    {
        for (uint32_t s = 0; s < nsym; ++s)
            cursor[s] = sym_start[s];
        for (uint32_t u = 0; u < size; ++u)
            state_tab[cursor[spread[u]]++] = (uint16_t)u;
    }
}

#if ZB_LANES > 1
/* zb_normalize by all lanes: lane s owns symbol s (nsym <= 64 <= ZB_LANES); (hist << tl) fits 32 bits (hist <= ZB_SEQ_MAX = 2^15,
 * tl <= 9).  One division per LANE instead of one per symbol on one lane; the correction loop (usually one round) finds the
 * first largest count with a wave maximum and a ballot. */
ZB_FN void zb_normalize_par(const uint32_t* hist, uint32_t nsym, uint32_t total, uint32_t tl, int16_t* norm, uint32_t zl)
{
    const uint32_t size = 1u << tl;
    const uint32_t h = zl < nsym ? hist[zl] : 0u;
    uint32_t v = 0, sum;
    if (h)
    {
        v = (h << tl) / total;
        if (v == 0u)
            v = 1u;
    }
    (void)zb_scan_excl(v, &sum);
    while (sum != size)
    {
        const uint32_t mx = zb_reduce_max(v);
        const uint32_t best = (uint32_t)__builtin_ctzll(zb_ballot(v == mx));
        if (sum < size)
        {
            if (zl == best)
                v += size - sum;
            sum = size;
        }
        else
        {
            uint32_t take = sum - size;
            if (take > mx - 1u)
                take = mx - 1u;
            if (take == 0u)
                break; /* (cannot happen: see zb_normalize) */
            if (zl == best)
                v -= take;
            sum -= take;
        }
    }
    if (zl < 64u)
        norm[zl] = (int16_t)v;
    ZB_SYNC_LDS();
}

/* zb_build_enc_table by all lanes (norm[0..63] in shared memory, 0 beyond nsym).  The serial walk -- symbol occurrence i goes to
 * the i-th cell of the walk 0, step, 2 step, ... (mod size) that is not one of the top cells the "less than one" symbols took --
 * is inverted per cell: cell u <= high is visit k(u) = u * step^-1 (mod size) of the walk, the top cells visited before it
 * (at most a handful: one per -1 symbol) are counted off, and the occurrence index that is left is looked up in the running
 * sums of the counts (incl[], 64 x u16 of scratch).  The cells of a symbol are then numbered in state order: lane s keeps how
 * many cells of symbol s the chunks of 64 cells before this one held, the cells of a chunk rank themselves among the equal
 * symbols of lower lanes with one ballot per DISTINCT symbol of the chunk. */
ZB_FN void zb_build_enc_table_par(const int16_t* norm, uint32_t nsym, uint32_t tl, uint16_t* state_tab, uint16_t* sym_start,
                                  uint16_t* incl, uint32_t zl)
{
    const uint32_t size = 1u << tl, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    const int nv = zl < nsym ? norm[zl] : 0;
    const uint32_t cnt = nv < 0 ? 1u : (uint32_t)nv, pcnt = nv > 0 ? (uint32_t)nv : 0u;
    uint32_t tot, inv = step, count_s = 0;
    const uint32_t start = zb_scan_excl(cnt, &tot);
    const uint32_t pex = zb_scan_excl(pcnt, &tot);
    const uint64_t low = zb_ballot(nv == -1);
    const uint32_t nlow = (uint32_t)__builtin_popcountll(low), high = size - 1u - nlow;
    for (int it = 0; it < 4; ++it) /* step^-1 mod 2^32 (Newton; step is odd: 3 correct bits to start with) */
        inv *= 2u - step * inv;
    if (zl < 64u)
    {
        sym_start[zl] = (uint16_t)start;
        incl[zl] = (uint16_t)(pex + pcnt);
    }
    ZB_SYNC_LDS();
    for (uint32_t u0 = 0; u0 < size; u0 += 64u)
    {
        const uint32_t u = u0 + zl;
        const int act = zl < 64u && u < size;
        uint32_t sym = 0, within = 0;
        if (act)
        {
            if (u > high)
            {
                /* the -1 symbols took the top cells in symbol order, the first one the last cell */
                uint64_t m = low;
                for (uint32_t j = size - 1u - u; j; --j)
                    m &= m - 1u;
                sym = (uint32_t)__builtin_ctzll(m);
            }
            else
            {
                const uint32_t k = (u * inv) & mask;
                uint32_t skipped = 0, lo = 0, hi = 64u;
                for (uint32_t j = 0; j < nlow; ++j)
                    skipped += (((size - 1u - j) * inv) & mask) < k ? 1u : 0u;
                {
                    const uint32_t i = k - skipped;
                    while (lo < hi) /* the first symbol whose running sum exceeds i */
                    {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((uint32_t)incl[mid] <= i)
                            lo = mid + 1u;
                        else
                            hi = mid;
                    }
                }
                sym = lo;
            }
        }
        {
            const uint32_t before = zb_shfl(count_s, sym), base = zb_shfl(start, sym);
            uint64_t rem = zb_ballot(act);
            while (rem)
            {
                const uint32_t s0 = zb_shfl(sym, (uint32_t)__builtin_ctzll(rem));
                const uint64_t m = zb_ballot(act && sym == s0);
                if (act && sym == s0)
                    within = (uint32_t)__builtin_popcountll(m & ((1ull << zl) - 1ull));
                if (zl == s0)
                    count_s += (uint32_t)__builtin_popcountll(m);
                rem &= ~m;
            }
            if (act)
                state_tab[base + before + within] = (uint16_t)u;
        }
    }
    ZB_SYNC_LDS();
}
#endif

ZB_FN uint32_t zb_sym_count(const int16_t* norm, uint32_t s) { return (uint32_t)(norm[s] < 0 ? 1 : norm[s]); }

/* One encoding step; returns nbBits << 10 | bits and updates *x. */
ZB_FN uint32_t zb_fse_step(uint32_t* x, uint32_t s, const int16_t* norm, const uint16_t* state_tab, const uint16_t* sym_start,
                           uint32_t tl)
{
    const uint32_t c = zb_sym_count(norm, s);
    uint32_t nb = tl - zb_highbit(c);
    if ((*x >> nb) < c)
        --nb; /* cannot underflow: x >= 2^tl >= ... see zb_build_enc_table */
    {
        const uint32_t bits = *x & ((1u << nb) - 1u);
        *x = (1u << tl) + state_tab[sym_start[s] + ((*x >> nb) - c)];
        return (nb << 10) | bits;
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Huffman code lengths (<= 11 bits) for the literals
 * ---------------------------------------------------------------------------------------------------------- */
/* Rank sort of the present literal symbols by (count, symbol) ascending into sh->sort_key; v[ZV_HUF_NSYM] = how
 * many.  All lanes (256 independent rank computations; the inner reads are wave-uniform LDS broadcasts). */
ZB_FN void zb_huffman_sort(ZbShared* sh, uint32_t zl)
{
    ZB_PAR_FOR(s, 256u)
    {
        const uint32_t c = sh->lit_hist[s];
        sh->huf_len[s] = 0;
        if (c)
        {
            uint32_t rank = 0;
            for (uint32_t t = 0; t < 256u; ++t)
            {
                const uint32_t ct = sh->lit_hist[t];
                rank += (ct != 0u) & ((ct < c) | ((ct == c) & (t < s)));
            }
            sh->sort_key[rank] = s;
            zb_atomic_add(&sh->v[ZV_HUF_NSYM], 1u);
        }
    }
}

/* In: sh->lit_hist, sh->sort_key (zb_huffman_sort).  Out: sh->huf_len / huf_code, v[ZV_HUF_*].  Serial (one lane). */
ZB_FN void zb_huffman_build(ZbShared* sh)
{
    uint32_t* A = sh->sort_key;
    const uint32_t n = sh->v[ZV_HUF_NSYM];
    sh->v[ZV_HUF_OK] = 0;
    if (n < 2u)
        return;
    sh->v[ZV_HUF_MAXSYM] = 0;
    for (uint32_t s = 256u; s-- > 0u;)
        if (sh->lit_hist[s])
        {
            sh->v[ZV_HUF_MAXSYM] = s;
            break;
        }
    /* minimum-redundancy code lengths in place (Moffat & Katajainen): W[i] starts as the sorted weights */
    {
        uint32_t* W = sh->huf_w;
        uint8_t* L = sh->huf_l;
        for (uint32_t i = 0; i < n; ++i)
            W[i] = sh->lit_hist[A[i]];
        if (n == 2u)
        {
            L[0] = L[1] = 1;
        }
        else
        {
            uint32_t root = 0, leaf = 2, next;
            W[0] += W[1];
            for (next = 1; next < n - 1u; ++next)
            {
                if (leaf >= n || W[root] < W[leaf])
                {
                    W[next] = W[root];
                    W[root++] = next;
                }
                else
                    W[next] = W[leaf++];
                if (leaf >= n || (root < next && W[root] < W[leaf]))
                {
                    W[next] += W[root];
                    W[root++] = next;
                }
                else
                    W[next] += W[leaf++];
            }
            W[n - 2u] = 0;
            for (int k = (int)n - 3; k >= 0; --k)
                W[k] = W[W[k]] + 1u;
            {
                int avbl = 1, used = 0, dpth = 0, r = (int)n - 2, nx = (int)n - 1;
                while (avbl > 0)
                {
                    while (r >= 0 && (int)W[r] == dpth)
                    {
                        ++used;
                        --r;
                    }
                    while (avbl > used)
                    {
                        W[nx--] = (uint32_t)dpth;
                        --avbl;
                    }
                    avbl = 2 * used;
                    ++dpth;
                    used = 0;
                }
            }
            for (uint32_t i = 0; i < n; ++i)
                L[i] = (uint8_t)(W[i] > 255u ? 255u : W[i]);
        }
        /* L is non-increasing (rarest symbol first).  Limit to 11 bits and restore Kraft equality. */
        if (L[0] > ZB_HUF_MAXBITS)
        {
            const uint32_t full = 1u << ZB_HUF_MAXBITS;
            uint32_t kraft = 0;
            for (uint32_t i = 0; i < n; ++i)
            {
                if (L[i] > ZB_HUF_MAXBITS)
                    L[i] = (uint8_t)ZB_HUF_MAXBITS;
                kraft += full >> L[i];
            }
            /* too full: lengthen the rarest symbols that are still shorter than 11 */
            for (uint32_t i = 0; i < n && kraft > full; ++i)
            {
                while (L[i] < ZB_HUF_MAXBITS && kraft > full)
                {
                    kraft -= full >> (L[i] + 1u);
                    ++L[i];
                }
            }
            /* slack left by the last step: shorten the most frequent symbols that fit exactly */
            for (int i = (int)n - 1; i >= 0 && kraft < full; --i)
            {
                while (L[i] > 1u && kraft + (full >> L[i]) <= full)
                {
                    kraft += full >> L[i];
                    --L[i];
                }
            }
            if (kraft != full)
                return; /* literals stay raw */
        }
        {
            uint32_t maxbits = 0;
            for (uint32_t i = 0; i < n; ++i)
            {
                sh->huf_len[A[i]] = L[i];
                if (L[i] > maxbits)
                    maxbits = L[i];
            }
            sh->v[ZV_HUF_MAXBITS] = maxbits;
        }
    }
    /* canonical codes as the decoder assigns them (huf_decompress.c HUF_readDTableX1 / RFC 8878 §4.2.1.3): the
     * longest codes get the smallest values, symbols of equal length in symbol order */
    {
        uint32_t* count = sh->small[0];
        uint32_t* start = sh->small[1];
        const uint32_t maxbits = sh->v[ZV_HUF_MAXBITS];
        for (uint32_t l = 0; l <= ZB_HUF_MAXBITS + 1u; ++l)
            count[l] = 0;
        for (uint32_t s = 0; s < 256u; ++s)
            ++count[sh->huf_len[s]];
        {
            uint32_t code = 0;
            for (uint32_t l = maxbits; l >= 1u; --l)
            {
                start[l] = code;
                code = (code + count[l]) >> 1;
            }
        }
        for (uint32_t s = 0; s < 256u; ++s)
        {
            const uint32_t l = sh->huf_len[s];
            sh->huf_code[s] = (uint16_t)(l ? start[l]++ : 0u);
        }
    }
    sh->v[ZV_HUF_OK] = 1;
}

#if ZB_LANES > 1
/* zb_huffman_build by all lanes -- the same code lengths and codes (the tree is the one the serial two-queue merge makes; only that
 * merge, 2 n dependent steps, stays on lane 0).  What goes to all lanes: the gather of the sorted weights, the depths of the internal
 * nodes (pointer jumping over the parent links instead of one node after the other), the leaves' depths (the internal nodes of a
 * depth are counted, the leaves fill what they leave free: one short serial pass over the DEPTHS, then every leaf looks its depth
 * up), the scatter to the symbols and the canonical codes (ranks among the symbols of equal length by ballots).  The serial form
 * made ~12 dependent LDS accesses per symbol on one lane: 9 % of the entropy kernel's wave time. */
ZB_FN void zb_huffman_build_par(ZbShared* sh, uint32_t zl)
{
    uint32_t* const A = sh->sort_key;
    uint32_t* const W = sh->huf_w;
    uint8_t* const L = sh->huf_l;
    uint32_t* const used = (uint32_t*)sh->cursor; /* [64]: internal nodes per depth, then leaves up to and including the depth */
    const uint32_t n = sh->v[ZV_HUF_NSYM];
    uint32_t maxsym = 0, overflow;
    ZB_SYNC_LDS();
    ZB_SERIAL(zl) { sh->v[ZV_HUF_OK] = 0; }
    if (n < 2u)
    {
        ZB_SYNC_LDS();
        return;
    }
    for (uint32_t c = 4u; c-- > 0u;)
    {
        const uint64_t m = zb_ballot(zl < 64u && sh->lit_hist[64u * c + (zl & 63u)] != 0u);
        if (m)
        {
            maxsym = 64u * c + 63u - (uint32_t)__builtin_clzll(m);
            break;
        }
    }
    ZB_PAR_FOR(i, n) W[i] = sh->lit_hist[A[i]];
    ZB_PAR_FOR(i, 64u) used[i] = 0;
    ZB_SYNC_LDS();
    if (n == 2u)
    {
        ZB_SERIAL(zl) { L[0] = L[1] = 1; }
    }
    else
    {
        /* minimum-redundancy code lengths (Moffat & Katajainen), phase 1 as in zb_huffman_build: parent links in W[0 .. n-3] */
        ZB_SERIAL(zl)
        {
            uint32_t root = 0, leaf = 2, next;
            W[0] += W[1];
            for (next = 1; next < n - 1u; ++next)
            {
                if (leaf >= n || W[root] < W[leaf])
                {
                    W[next] = W[root];
                    W[root++] = next;
                }
                else
                    W[next] = W[leaf++];
                if (leaf >= n || (root < next && W[root] < W[leaf]))
                {
                    W[next] += W[root];
                    W[root++] = next;
                }
                else
                    W[next] += W[leaf++];
            }
        }
        ZB_SYNC_LDS();
        /* phase 2, depths of the internal nodes 0 .. n-2 (the root is n-2): entry = link | distance to it << 16, doubled until
         * every link is the root */
        ZB_PAR_FOR(k, n - 1u) W[k] = k == n - 2u ? k : (W[k] | (1u << 16));
        ZB_SYNC_LDS();
        for (uint32_t round = 0; round < 8u; ++round)
        {
            uint32_t e[4], pending = 0;
            ZB_UNROLL
            for (uint32_t j = 0; j < 4u; ++j)
            {
                const uint32_t k = zl + j * ZB_LANES;
                e[j] = 0;
                if (k < n - 1u)
                {
                    const uint32_t mine = W[k], up = W[mine & 0xFFFFu];
                    e[j] = (up & 0xFFFFu) | ((mine & 0xFFFF0000u) + (up & 0xFFFF0000u));
                    pending |= (up & 0xFFFFu) != n - 2u;
                }
            }
            ZB_SYNC_LDS();
            ZB_UNROLL
            for (uint32_t j = 0; j < 4u; ++j)
                if (zl + j * ZB_LANES < n - 1u)
                    W[zl + j * ZB_LANES] = e[j];
            ZB_SYNC_LDS();
            if (!zb_ballot(pending != 0u))
                break;
        }
        overflow = 0;
        ZB_PAR_FOR(k, n - 1u)
        {
            const uint32_t d = W[k] >> 16;
            if (d < 64u)
                zb_atomic_add(&used[d], 1u);
            else
                overflow = 1;
        }
        ZB_SYNC_LDS();
        if (zb_ballot(overflow != 0u)) /* (a tree deeper than 63: more literals than a piece holds -- kept for completeness) */
        {
            ZB_SERIAL(zl)
            {
                for (uint32_t k = 0; k + 1u < n; ++k)
                    W[k] >>= 16;
                {
                    int avbl = 1, usedn = 0, dpth = 0, r = (int)n - 2, nx = (int)n - 1;
                    while (avbl > 0)
                    {
                        while (r >= 0 && (int)W[r] == dpth)
                        {
                            ++usedn;
                            --r;
                        }
                        while (avbl > usedn)
                        {
                            W[nx--] = (uint32_t)dpth;
                            --avbl;
                        }
                        avbl = 2 * usedn;
                        ++dpth;
                        usedn = 0;
                    }
                }
                for (uint32_t i = 0; i < n; ++i)
                    L[i] = (uint8_t)(W[i] > 255u ? 255u : W[i]);
            }
        }
        else
        {
            /* phase 3: a depth has avbl slots (1 at the root, twice the internal nodes of the depth above below it); what the
             * internal nodes leave free are leaves, handed out from the most frequent symbol (the last sorted position) */
            ZB_SERIAL(zl)
            {
                uint32_t avbl = 1, total = 0;
                for (uint32_t d = 0; d < 64u; ++d)
                {
                    const uint32_t un = used[d];
                    total += avbl > un ? avbl - un : 0u;
                    used[d] = total;
                    avbl = 2u * un;
                }
            }
            ZB_SYNC_LDS();
            ZB_PAR_FOR(i, n)
            {
                const uint32_t e = n - 1u - i; /* leaves handed out before this one */
                uint32_t d = 0;
                ZB_UNROLL
                for (uint32_t st = 32u; st; st >>= 1) /* the first depth whose running total exceeds e */
                    if (used[d + st - 1u] <= e)
                        d += st;
                L[i] = (uint8_t)d;
            }
        }
    }
    ZB_SYNC_LDS();
    /* L is non-increasing (rarest symbol first).  Limit to 11 bits and restore Kraft equality: as in zb_huffman_build */
    const uint32_t too_deep = L[0] > ZB_HUF_MAXBITS; /* (every lane has asked before lane 0 changes L below) */
    ZB_SYNC_LDS();
    if (too_deep)
    {
        ZB_SERIAL(zl)
        {
            const uint32_t full = 1u << ZB_HUF_MAXBITS;
            uint32_t kraft = 0;
            for (uint32_t i = 0; i < n; ++i)
            {
                if (L[i] > ZB_HUF_MAXBITS)
                    L[i] = (uint8_t)ZB_HUF_MAXBITS;
                kraft += full >> L[i];
            }
            for (uint32_t i = 0; i < n && kraft > full; ++i)
            {
                while (L[i] < ZB_HUF_MAXBITS && kraft > full)
                {
                    kraft -= full >> (L[i] + 1u);
                    ++L[i];
                }
            }
            for (int i = (int)n - 1; i >= 0 && kraft < full; --i)
            {
                while (L[i] > 1u && kraft + (full >> L[i]) <= full)
                {
                    kraft += full >> L[i];
                    --L[i];
                }
            }
            sh->v[ZV_SKIP] = kraft != full; /* (a flag both forms may use here: read back below) */
        }
        ZB_SYNC_LDS();
        if (sh->v[ZV_SKIP])
        {
            ZB_SYNC_LDS();
            ZB_SERIAL(zl) { sh->v[ZV_SKIP] = 0; }
            ZB_SYNC_LDS();
            return; /* literals stay raw */
        }
    }
    {
        uint32_t len[4], code[4], maxbits, base = 0;
        ZB_PAR_FOR(i, n) sh->huf_len[A[i]] = L[i];
        ZB_SYNC_LDS();
        maxbits = 0;
        ZB_UNROLL
        for (uint32_t c = 0; c < 4u; ++c)
        {
            len[c] = zl < 64u ? sh->huf_len[64u * c + (zl & 63u)] : 0u;
            code[c] = 0;
            maxbits = len[c] > maxbits ? len[c] : maxbits;
        }
        maxbits = zb_reduce_max(maxbits);
        /* canonical codes as the decoder assigns them: the longest codes get the smallest values, symbols of equal length in
         * symbol order -- start(l) = (start(l + 1) + count(l + 1)) >> 1, a symbol's code = start + its rank among its length */
        for (uint32_t l = maxbits; l >= 1u; --l)
        {
            uint32_t cnt = 0;
            ZB_UNROLL
            for (uint32_t c = 0; c < 4u; ++c)
            {
                const uint64_t m = zb_ballot(len[c] == l);
                if (len[c] == l)
                    code[c] = base + cnt + (uint32_t)__builtin_popcountll(m & ((1ull << (zl & 63u)) - 1ull));
                cnt += (uint32_t)__builtin_popcountll(m);
            }
            base = (base + cnt) >> 1;
        }
        ZB_UNROLL
        for (uint32_t c = 0; c < 4u; ++c)
            if (zl < 64u)
                sh->huf_code[64u * c + zl] = (uint16_t)code[c];
        ZB_SERIAL(zl)
        {
            sh->v[ZV_HUF_MAXSYM] = maxsym;
            sh->v[ZV_HUF_MAXBITS] = maxbits;
            sh->v[ZV_HUF_OK] = 1;
        }
    }
    ZB_SYNC_LDS();
}
#endif

/* Huffman tree description (RFC 8878 §4.2.1; HUF_readStats, entropy_common.c:236-327).  Weights of symbols
 * 0..maxsym-1; the last present symbol is implied.  Returns bytes written, 0 if it cannot be represented. */
ZB_FN uint32_t zb_write_huf_tree(ZbShared* sh, uint8_t* dst)
{
    const uint32_t maxbits = sh->v[ZV_HUF_MAXBITS], nw = sh->v[ZV_HUF_MAXSYM];
    if (nw <= 128u)
    {
        /* direct: header 127 + number of weights, two 4-bit weights per byte, first in the high nibble */
        dst[0] = (uint8_t)(127u + nw);
        for (uint32_t i = 0; i < nw; i += 2u)
        {
            const uint32_t l0 = sh->huf_len[i], l1 = i + 1u < nw ? sh->huf_len[i + 1u] : 0u;
            const uint32_t w0 = l0 ? maxbits + 1u - l0 : 0u, w1 = l1 ? maxbits + 1u - l1 : 0u;
            dst[1u + (i >> 1)] = (uint8_t)((w0 << 4) | w1);
        }
        return 1u + ((nw + 1u) >> 1);
    }
    /* FSE-compressed weights: table log <= 6, two interleaved states (fse_decompress.c:174-238) */
    {
        uint32_t* hist = sh->small[0];
        int16_t* norm = sh->norm[0];
        uint32_t distinct = 0;
        for (uint32_t w = 0; w < 16u; ++w)
            hist[w] = 0;
        for (uint32_t i = 0; i < nw; ++i)
        {
            const uint32_t l = sh->huf_len[i];
            ++hist[l ? maxbits + 1u - l : 0u];
        }
        for (uint32_t w = 0; w < 13u; ++w)
            distinct += hist[w] != 0u;
        if (distinct < 2u)
            return 0;
        {
            const uint32_t tl = 6u;
            uint32_t pos;
            zb_normalize(hist, 13u, nw, tl, norm);
            pos = 1u + zb_write_ncount(dst + 1, norm, 13u, tl);
            zb_build_enc_table(norm, 13u, tl, ZB_SPREAD(sh, 0), sh->state_tab[0], sh->sym_start[0], sh->cursor[0]);
            /* Weights are decoded alternately by state 1 (even indices) and state 2 (odd); the two last weights
             * are carried by the initial states (first cell of their symbol, so that the decoder's final state
             * update over-reads and stops, fse_decompress.c:214-236); the others are encoded from the end. */
            {
                uint32_t x[2];
                uint64_t acc = 0;
                uint32_t nacc = 0;
                int i = (int)nw - 1;
                for (int k = 0; k < 2; ++k, --i)
                {
                    const uint32_t l = sh->huf_len[i];
                    const uint32_t w = l ? maxbits + 1u - l : 0u;
                    x[i & 1] = (1u << tl) + sh->state_tab[0][sh->sym_start[0][w]];
                }
                for (; i >= 0; --i)
                {
                    const uint32_t l = sh->huf_len[i];
                    const uint32_t w = l ? maxbits + 1u - l : 0u;
                    const uint32_t r = zb_fse_step(&x[i & 1], w, norm, sh->state_tab[0], sh->sym_start[0], tl);
                    acc |= (uint64_t)(r & 1023u) << nacc;
                    nacc += r >> 10;
                    while (nacc >= 8u)
                    {
                        dst[pos++] = (uint8_t)acc;
                        acc >>= 8;
                        nacc -= 8u;
                    }
                }
                /* the decoder reads state 1 first: it is written last */
                acc |= (uint64_t)(x[1] - (1u << tl)) << nacc;
                nacc += tl;
                acc |= (uint64_t)(x[0] - (1u << tl)) << nacc;
                nacc += tl;
                acc |= (uint64_t)1u << nacc; /* end mark */
                nacc += 1u;
                while (nacc > 0u)
                {
                    dst[pos++] = (uint8_t)acc;
                    acc >>= 8;
                    nacc = nacc >= 8u ? nacc - 8u : 0u;
                }
            }
            if (pos - 1u >= 128u)
                return 0;
            dst[0] = (uint8_t)(pos - 1u);
            return pos;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * the block encoder
 * ---------------------------------------------------------------------------------------------------------- */
/* The block's literals are the units' literal buffers back to back (unit u holds literals ulit_base[u] ..
 * ulit_base[u+1]).  A lane walks its run of literal indices up or down; the reader keeps the current unit and one
 * cached 32-bit word. */
typedef struct ZbLitReader
{
    const ZbInput* in;
    const uint32_t* ulit_base; /* [nunits + 1] */
    uint32_t srcmask, nunits, u, lo, hi, cw, cwi;
} ZbLitReader;

/* 32-bit word wi of unit u's literals (nbytes of them; bytes at or past nbytes are unspecified).  Source-resident units
 * are read with aligned loads and a funnel shift, never touching a word that holds none of their bytes. */
ZB_FN uint32_t zb_unit_word(const ZbInput* in, uint32_t srcmask, uint32_t u, uint32_t wi, uint32_t nbytes)
{
    if ((srcmask >> u) & 1u)
    {
        const uint8_t* p = in->src + (size_t)u * ZB_UNIT + 4u * (size_t)wi;
        const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
        const uint32_t* q = (const uint32_t*)(p - mis);
        uint32_t w = q[0];
        if (mis)
        {
            w >>= 8u * mis;
            if (4u * wi + 4u - mis < nbytes)
                w |= q[1] << (32u - 8u * mis);
        }
        return w;
    }
    return ((const uint32_t*)(in->unit_lits + (size_t)u * ZB_UNIT))[wi];
}

ZB_FN void zb_lit_open(ZbLitReader* r, const ZbInput* in, uint32_t srcmask, const uint32_t* ulit_base, uint32_t nunits, uint32_t k)
{
    uint32_t lo = 0, hi = nunits;
    r->in = in;
    r->srcmask = srcmask;
    r->ulit_base = ulit_base;
    r->nunits = nunits;
    while (hi - lo > 1u)
    {
        const uint32_t mid = (lo + hi) >> 1;
        if (ulit_base[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    r->u = lo;
    r->lo = ulit_base[lo];
    r->hi = ulit_base[lo + 1u];
    r->cw = 0;
    r->cwi = 0xFFFFFFFFu;
}

ZB_FN uint32_t zb_lit_get(ZbLitReader* r, uint32_t k) /* k < total literals */
{
    while (k >= r->hi)
    {
        ++r->u;
        r->lo = r->hi;
        r->hi = r->ulit_base[r->u + 1u];
    }
    while (k < r->lo)
    {
        --r->u;
        r->hi = r->lo;
        r->lo = r->ulit_base[r->u];
    }
    {
        const uint32_t o = k - r->lo;
        const uint32_t wi = (r->u << 10) | (o >> 2); /* units hold at most 4096 literals = 1024 words */
        if (wi != r->cwi)
        {
            r->cw = zb_unit_word(r->in, r->srcmask, r->u, o >> 2, r->hi - r->lo);
            r->cwi = wi;
        }
        return (r->cw >> (8u * (o & 3u))) & 255u;
    }
}

ZB_FN uint32_t zb_of_code(uint32_t off) { return zb_highbit(off + 3u); }

/* Mode, table log, normalised counts and encoding table of the three sequence-symbol types from sh->sym_hist (both block layouts).
 * One lane per table in the one-lane form; with a wave every table is built by all lanes, one table after the other (the serial
 * builders were 28 % of the entropy kernel's wave time: a division per symbol, then ~4 dependent LDS accesses per table cell, on
 * one lane). */
ZB_FN void zb_build_seq_tables(ZbShared* sh, uint32_t nbseq, uint32_t zl)
{
#if ZB_LANES > 1
    for (uint32_t t = 0; t < 3u && nbseq; ++t)
    {
        const uint32_t nsym = zb_table_nsym((int)t);
        const uint64_t present = zb_ballot(zl < 64u && sh->sym_hist[t][zl & 63u] != 0u);
        const uint32_t distinct = (uint32_t)__builtin_popcountll(present), maxs = 63u - (uint32_t)__builtin_clzll(present | 1ull);
        if (distinct == 1u)
        {
            ZB_SERIAL(zl)
            {
                sh->mode[t] = 1; /* RLE_Mode */
                sh->rle_sym[t] = (uint8_t)maxs;
                sh->table_log[t] = 0;
            }
        }
        else if (((nbseq < 64u && !(ZB_DBG & 2u)) || (ZB_DBG & 4u)) && maxs < nsym)
        {
            const uint32_t tl = zb_table_default_log((int)t);
            ZB_SERIAL(zl)
            {
                sh->mode[t] = 0; /* Predefined_Mode */
                sh->table_log[t] = (uint8_t)tl;
            }
            if (zl < 64u)
                sh->norm[t][zl] = (int16_t)(zl < nsym ? zb_default_norm((int)t, zl) : 0);
            ZB_SYNC_LDS();
            zb_build_enc_table_par(sh->norm[t], nsym, tl, sh->state_tab[t], sh->sym_start[t], sh->cursor[t], zl);
        }
        else
        {
            uint32_t tl = zb_highbit(nbseq) - 1u;
            const uint32_t minlog = distinct > 32u ? 6u : 5u, maxlog = zb_table_max_log((int)t);
            if (tl < minlog)
                tl = minlog;
            if (tl > maxlog)
                tl = maxlog;
            ZB_SERIAL(zl)
            {
                sh->mode[t] = 2; /* FSE_Compressed_Mode */
                sh->table_log[t] = (uint8_t)tl;
                sh->rle_sym[t] = (uint8_t)maxs; /* highest present symbol, for the NCount writer */
            }
            zb_normalize_par(sh->sym_hist[t], maxs + 1u, nbseq, tl, sh->norm[t], zl);
            zb_build_enc_table_par(sh->norm[t], maxs + 1u, tl, sh->state_tab[t], sh->sym_start[t], sh->cursor[t], zl);
        }
    }
#else
    ZB_PAR_FOR(t, 3u)
    {
        if (nbseq)
        {
            const uint32_t nsym = zb_table_nsym((int)t);
            uint32_t distinct = 0, only = 0, maxs = 0;
            for (uint32_t s = 0; s < 64u; ++s)
                if (sh->sym_hist[t][s])
                {
                    ++distinct;
                    only = s;
                    maxs = s;
                }
            if (distinct == 1u)
            {
                sh->mode[t] = 1; /* RLE_Mode */
                sh->rle_sym[t] = (uint8_t)only;
                sh->table_log[t] = 0;
            }
            else if (((nbseq < 64u && !(ZB_DBG & 2u)) || (ZB_DBG & 4u)) && maxs < nsym)
            {
                sh->mode[t] = 0; /* Predefined_Mode */
                sh->table_log[t] = (uint8_t)zb_table_default_log((int)t);
                for (uint32_t s = 0; s < 64u; ++s)
                    sh->norm[t][s] = (int16_t)(s < nsym ? zb_default_norm((int)t, s) : 0);
                zb_build_enc_table(sh->norm[t], nsym, sh->table_log[t], ZB_SPREAD(sh, t), sh->state_tab[t], sh->sym_start[t], sh->cursor[t]);
            }
            else
            {
                uint32_t tl = zb_highbit(nbseq) - 1u;
                const uint32_t minlog = distinct > 32u ? 6u : 5u, maxlog = zb_table_max_log((int)t);
                if (tl < minlog)
                    tl = minlog;
                if (tl > maxlog)
                    tl = maxlog;
                sh->mode[t] = 2; /* FSE_Compressed_Mode */
                sh->table_log[t] = (uint8_t)tl;
                zb_normalize(sh->sym_hist[t], maxs + 1u, nbseq, tl, sh->norm[t]);
                zb_build_enc_table(sh->norm[t], maxs + 1u, tl, ZB_SPREAD(sh, t), sh->state_tab[t], sh->sym_start[t], sh->cursor[t]);
                sh->rle_sym[t] = (uint8_t)maxs; /* highest present symbol, for the NCount writer */
            }
        }
    }
#endif
}

/* Encodes one block.  Returns the size of the Compressed_Block content in sc->out, or 0 when it would not be
 * smaller than the raw bytes (the caller then stores a Raw_Block). */
ZB_FN uint32_t zb_encode_block(const ZbInput* in, const ZbScratch* sc, ZbShared* sh, uint32_t zl)
{
    uint8_t* const out8 = (uint8_t*)sc->out;

    /* ---- phase 0: unit bases; zero the histograms ---- */
    ZB_PAR_FOR(u, in->nunits)
    {
        const ZbUnitMeta m = in->meta[u];
        sh->useq_base[u] = m.nseq; /* counts now, bases after the scan below */
        sh->ulit_base[u] = m.nlit;
        sh->carry[u] = m.tail;
    }
    ZB_SYNC();
    ZB_SERIAL(zl)
    {
        uint32_t nseq = 0, nlit = 0, carry = 0, srcmask = 0;
        for (uint32_t u = 0; u < in->nunits; ++u)
        {
            const uint32_t un = sh->useq_base[u], ul = sh->ulit_base[u], ut = sh->carry[u];
            if (in->src && un == 0u)
                srcmask |= 1u << u;
            sh->useq_base[u] = nseq;
            sh->ulit_base[u] = nlit;
            sh->carry[u] = carry;
            nseq += un;
            nlit += ul;
            carry = un ? ut : carry + ul;
        }
        sh->useq_base[in->nunits] = nseq;
        sh->ulit_base[in->nunits] = nlit;
        sh->v[ZV_NBSEQ] = nseq;
        sh->v[ZV_NLIT] = nlit;
        sh->v[ZV_SRCMASK] = srcmask;
    }
    ZB_PAR_FOR(i, 256u) sh->lit_hist[i] = 0;
    ZB_PAR_FOR(i, 3u * 64u) sh->sym_hist[i >> 6][i & 63u] = 0;
    ZB_SYNC();
    const uint32_t nbseq = sh->v[ZV_NBSEQ], nlit = sh->v[ZV_NLIT], srcmask = sh->v[ZV_SRCMASK];

    ZB_MARK(1);
    /* ---- phase 1: merge the units: sequences (with their symbol histograms) and literals (with theirs) ---- */
    ZB_PAR_FOR(i, nbseq)
    {
        uint32_t lo = 0, hi = in->nunits;
        while (hi - lo > 1u)
        {
            const uint32_t mid = (lo + hi) >> 1;
            if (sh->useq_base[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        {
            const uint32_t k = i - sh->useq_base[lo];
            const uint64_t r = in->unit_recs[(uint64_t)lo * ZB_UNIT_SEQ_MAX + k];
            const uint32_t lit = (uint32_t)(r & 0xFFFFu) + (k == 0u ? sh->carry[lo] : 0u);
            const uint32_t ml = (uint32_t)((r >> 16) & 0xFFFFu), off = (uint32_t)(r >> 32);
            sc->seqs[i] = (uint64_t)lit | ((uint64_t)ml << 20) | ((uint64_t)off << 36);
            zb_atomic_add(&sh->sym_hist[ZT_LL][zb_ll_code(lit)], 1u);
            zb_atomic_add(&sh->sym_hist[ZT_ML][zb_ml_code(ml - 3u)], 1u);
            zb_atomic_add(&sh->sym_hist[ZT_OF][zb_of_code(off)], 1u);
        }
    }
    /* Plainly noise?  When the matches alone cannot pay for a compressed block (the second half of the test in phase 2),
     * the only open question is whether the literals deserve a Huffman table.  Every eighth unit's literals (4 KiB runs,
     * read exactly like the full pass below) answer that for blocks of noise, which stop here without the full
     * histogram; everything else goes on to the exact test. */
    if (nlit >= 32768u && in->raw_size - nlit < 3u * nbseq + 32u && !(ZB_DBG & 8u))
    {
        for (uint32_t u = (nlit >> 12) & 7u; u < in->nunits; u += 8u)
        {
            const uint32_t n = sh->ulit_base[u + 1u] - sh->ulit_base[u];
            ZB_PAR_FOR(j, n >> 2)
            {
                const uint32_t w = zb_unit_word(in, srcmask, u, j, n);
                zb_atomic_add(&sh->lit_hist[w & 255u], 1u);
                zb_atomic_add(&sh->lit_hist[(w >> 8) & 255u], 1u);
                zb_atomic_add(&sh->lit_hist[(w >> 16) & 255u], 1u);
                zb_atomic_add(&sh->lit_hist[w >> 24], 1u);
            }
        }
        ZB_SYNC();
        ZB_SERIAL(zl)
        {
            uint32_t largest = 0, ns = 0;
            for (uint32_t s2 = 0; s2 < 256u; ++s2)
            {
                ns += sh->lit_hist[s2];
                if (sh->lit_hist[s2] > largest)
                    largest = sh->lit_hist[s2];
            }
            sh->v[ZV_SKIP] = (ns >= 2048u && largest <= (ns >> 7) + 4u) ? 1u : 0u;
        }
        ZB_SYNC();
        if (sh->v[ZV_SKIP])
            return 0;
        ZB_PAR_FOR(i, 256u) sh->lit_hist[i] = 0;
        ZB_SYNC();
    }
    /* The literal histogram is taken PER HUFFMAN STREAM (the four quarters of the literals), two 16-bit counters to a word
     * (a stream has at most 32 768 literals): the streams' bit totals then follow from the code lengths without a second
     * pass over the literals.  The counters borrow the FSE state tables, which are not built before the totals are taken. */
    uint32_t* const hist4 = (uint32_t*)sh->state_tab; /* [2][256]: streams 0|1 and 2|3 */
    const uint32_t qseg = (nlit + 3u) >> 2;           /* literals per stream (the last one takes the rest) */
    ZB_PAR_FOR(i, 512u) hist4[i] = 0;
    ZB_SYNC();
    for (uint32_t u = 0; u < in->nunits; ++u)
    {
        /* (the unit buffers are 4 KiB aligned; bytes past nlit are masked off) */
        const uint32_t n = sh->ulit_base[u + 1u] - sh->ulit_base[u];
        const uint32_t kbase = sh->ulit_base[u];
        ZB_PAR_FOR(j, (n + 3u) >> 2)
        {
            const uint32_t w = zb_unit_word(in, srcmask, u, j, n);
            const uint32_t k = n - 4u * j; /* valid bytes in this word, >= 1 */
            const uint32_t k0 = kbase + 4u * j;
            uint32_t st0 = (k0 >= qseg) + (k0 >= 2u * qseg) + (k0 >= 3u * qseg);
            const uint32_t k3 = k0 + 3u;
            const uint32_t st3 = (k3 >= qseg) + (k3 >= 2u * qseg) + (k3 >= 3u * qseg);
            if (st0 == st3)
            {
                uint32_t* const h = hist4 + ((st0 >> 1) << 8);
                const uint32_t one = 1u << ((st0 & 1u) << 4);
                zb_atomic_add(&h[w & 255u], one);
                if (k > 1u)
                    zb_atomic_add(&h[(w >> 8) & 255u], one);
                if (k > 2u)
                    zb_atomic_add(&h[(w >> 16) & 255u], one);
                if (k > 3u)
                    zb_atomic_add(&h[w >> 24], one);
            }
            else /* a stream boundary inside the word */
                for (uint32_t b = 0; b < 4u && b < k; ++b)
                {
                    const uint32_t kb = k0 + b;
                    const uint32_t stb = (kb >= qseg) + (kb >= 2u * qseg) + (kb >= 3u * qseg);
                    zb_atomic_add(&hist4[((stb >> 1) << 8) + ((w >> (8u * b)) & 255u)], 1u << ((stb & 1u) << 4));
                }
        }
    }
    ZB_SYNC();
    ZB_PAR_FOR(i, 256u)
    {
        const uint32_t a = hist4[i], b = hist4[256u + i];
        sh->lit_hist[i] = (a & 0xFFFFu) + (a >> 16) + (b & 0xFFFFu) + (b >> 16);
    }
    ZB_SYNC();

    ZB_MARK(2);
    /* ---- phase 2: Huffman code for the literals, FSE tables for the three symbol types (with a wave: by all lanes) ---- */
    ZB_SERIAL(zl)
    {
        /* Is it worth going on?  Literals whose most frequent byte is as rare as in noise stay raw (the test zstd's
         * own HUF_compress uses, huf_compress.c "largest <= (srcSize >> 7)+4"), and then the block can only shrink
         * by what the matches remove minus about three bytes per sequence. */
        uint32_t largest = 0;
        for (uint32_t s2 = 0; s2 < 256u; ++s2)
            if (sh->lit_hist[s2] > largest)
                largest = sh->lit_hist[s2];
        sh->v[ZV_HUF_OK] = 0;
        sh->v[ZV_TREE_BYTES] = 0;
        sh->v[ZV_HUF_NSYM] = 0;
        sh->v[ZV_LIT_HDR] = (nlit >= 256u && !(ZB_DBG & 1u) && largest > (nlit >> 7) + 4u) ? 1u : 0u; /* try Huffman */
        sh->v[ZV_SKIP] = (!sh->v[ZV_LIT_HDR] && in->raw_size - nlit < 3u * nbseq + 32u) ? 1u : 0u;
    }
    ZB_SYNC();
    if (sh->v[ZV_SKIP])
        return 0;
    ZB_PAR_FOR(i, ZB_OUT_BYTES / 4u) sc->out[i] = 0;
    if (sh->v[ZV_LIT_HDR])
        zb_huffman_sort(sh, zl);
    ZB_SYNC();
#if ZB_LANES > 1
    if (sh->v[ZV_LIT_HDR])
        zb_huffman_build_par(sh, zl);
#else
    ZB_SERIAL(zl)
    {
        if (sh->v[ZV_LIT_HDR])
            zb_huffman_build(sh);
    }
#endif
    ZB_PAR_FOR(c, 4u) sh->part2[c] = 0;
    ZB_SYNC();
    /* bits of the four Huffman streams = per-stream symbol counts x code lengths (before the tree description and the FSE
     * tables, which reuse the counters' memory) */
    if (sh->v[ZV_HUF_OK])
    {
        ZB_PAR_FOR(i, 256u)
        {
            const uint32_t a = hist4[i], b = hist4[256u + i], l = sh->huf_len[i];
            if (a | b)
            {
                if (a & 0xFFFFu)
                    zb_atomic_add(&sh->part2[0], (a & 0xFFFFu) * l);
                if (a >> 16)
                    zb_atomic_add(&sh->part2[1], (a >> 16) * l);
                if (b & 0xFFFFu)
                    zb_atomic_add(&sh->part2[2], (b & 0xFFFFu) * l);
                if (b >> 16)
                    zb_atomic_add(&sh->part2[3], (b >> 16) * l);
            }
        }
    }
    ZB_SYNC();
    ZB_SERIAL(zl)
    {
        if (sh->v[ZV_HUF_OK]) /* uses table slot 0 as work space: must precede the FSE tables below */
            sh->v[ZV_TREE_BYTES] = zb_write_huf_tree(sh, sh->tree);
    }
    ZB_SYNC();
    ZB_MARK(9);
    zb_build_seq_tables(sh, nbseq, zl);
    ZB_SYNC();

    ZB_MARK(3);
    /* ---- (phase 3, the streams' bit totals, is folded into the histogram: see above) ---- */
    const uint32_t seg = (nlit + 3u) >> 2;

    ZB_MARK(4);
    /* ---- phase 4 (lane 0): decide the literals mode, write every header, lay out the bit streams ---- */
    ZB_SERIAL(zl)
    {
        uint32_t pos = 0, use_huf = 0;
        if (sh->v[ZV_HUF_OK])
        {
            const uint8_t* tree = sh->tree;
            const uint32_t tb = sh->v[ZV_TREE_BYTES];
            uint32_t csize = tb + 6u;
            for (uint32_t st = 0; st < 4u; ++st)
            {
                sh->v[ZV_STREAM_BYTES + st] = (sh->part2[st] + 1u + 7u) >> 3; /* + end mark */
                csize += sh->v[ZV_STREAM_BYTES + st];
            }
            {
                const uint32_t hdr = nlit < 1024u ? 3u : nlit < 16384u ? 4u : 5u;
                const uint32_t rawhdr = nlit < 32u ? 1u : nlit < 4096u ? 2u : 3u;
                /* every stream must hold at least its end mark plus one symbol for the decoder's 4-stream path */
                if (tb && csize + hdr < nlit + rawhdr && seg >= 1u && nlit >= 4u * 1u + 252u)
                {
                    const uint32_t sf = nlit < 1024u ? 1u : nlit < 16384u ? 2u : 3u;
                    const uint32_t nb = sf == 1u ? 10u : sf == 2u ? 14u : 18u;
                    const uint64_t h = 2u | (sf << 2) | ((uint64_t)nlit << 4) | ((uint64_t)csize << (4u + nb));
                    for (uint32_t k = 0; k < hdr; ++k)
                        out8[pos++] = (uint8_t)(h >> (8u * k));
                    for (uint32_t k = 0; k < tb; ++k)
                        out8[pos++] = tree[k];
                    for (uint32_t st = 0; st < 3u; ++st)
                    {
                        out8[pos++] = (uint8_t)sh->v[ZV_STREAM_BYTES + st];
                        out8[pos++] = (uint8_t)(sh->v[ZV_STREAM_BYTES + st] >> 8);
                    }
                    for (uint32_t st = 0; st < 4u; ++st)
                    {
                        sh->v[ZV_STREAM_BASE + st] = pos;
                        pos += sh->v[ZV_STREAM_BYTES + st];
                    }
                    use_huf = 1;
                }
            }
        }
        if (!use_huf)
        {
            /* Raw_Literals_Block: header then the bytes (copied below) */
            if (nlit < 32u)
                out8[pos++] = (uint8_t)(nlit << 3);
            else if (nlit < 4096u)
            {
                const uint32_t h = 4u | (nlit << 4);
                out8[pos++] = (uint8_t)h;
                out8[pos++] = (uint8_t)(h >> 8);
            }
            else
            {
                const uint32_t h = 12u | (nlit << 4);
                out8[pos++] = (uint8_t)h;
                out8[pos++] = (uint8_t)(h >> 8);
                out8[pos++] = (uint8_t)(h >> 16);
            }
            sh->v[ZV_STREAM_BASE] = pos;
            pos += nlit;
        }
        sh->v[ZV_LIT_MODE] = use_huf ? 2u : 0u;
        sh->v[ZV_LIT_END] = pos;
        /* sequences section header (zstd_decompress_block.c:700-760) */
        if (nbseq == 0u)
            out8[pos++] = 0;
        else
        {
            if (nbseq < 128u)
                out8[pos++] = (uint8_t)nbseq;
            else if (nbseq < 0x7F00u)
            {
                out8[pos++] = (uint8_t)((nbseq >> 8) + 128u);
                out8[pos++] = (uint8_t)nbseq;
            }
            else
            {
                out8[pos++] = 255;
                out8[pos++] = (uint8_t)(nbseq - 0x7F00u);
                out8[pos++] = (uint8_t)((nbseq - 0x7F00u) >> 8);
            }
            out8[pos++] = (uint8_t)((sh->mode[ZT_LL] << 6) | (sh->mode[ZT_OF] << 4) | (sh->mode[ZT_ML] << 2));
            for (uint32_t t = 0; t < 3u; ++t) /* LL, OF, ML in this order */
            {
                if (sh->mode[t] == 1u)
                    out8[pos++] = sh->rle_sym[t];
                else if (sh->mode[t] == 2u)
                    pos += zb_write_ncount(out8 + pos, sh->norm[t], (uint32_t)sh->rle_sym[t] + 1u, sh->table_log[t]);
            }
        }
        sh->v[ZV_SEQ_BITS0] = pos;
    }
    ZB_SYNC();

    ZB_MARK(5);
    /* ---- phase 5: literals ---- */
    if (sh->v[ZV_LIT_MODE] == 2u)
    {
        /* A stream is written from its LAST literal.  Per step every lane takes the next four literals (lane 0 the
         * last four), packs their codes, a wave prefix sum of the bit counts gives its position, and the <= 44 bits go
         * out with one or two atomicOr: no lane ever walks a long serial run. */
        for (uint32_t st = 0; st < 4u; ++st)
        {
            const uint32_t s0 = st * seg < nlit ? st * seg : nlit;
            const uint32_t s1 = st == 3u ? nlit : (s0 + seg < nlit ? s0 + seg : nlit);
            uint32_t running = sh->v[ZV_STREAM_BASE + st] * 8u;
            ZbLitReader lr;
            zb_lit_open(&lr, in, srcmask, sh->ulit_base, in->nunits, s1 ? s1 - 1u : 0u);
            for (uint32_t done = 0; done < s1 - s0; done += 4u * ZB_LANES)
            {
                /* my literals: indices s1-1 - (done + 4*zl + j), j = 0..3, as far as they exist */
                uint64_t acc = 0;
                uint32_t nb = 0;
                for (uint32_t j = 0; j < 4u; ++j)
                {
                    const uint32_t r = done + 4u * zl + j;
                    if (r < s1 - s0)
                    {
                        const uint32_t sy = zb_lit_get(&lr, s1 - 1u - r);
                        acc |= (uint64_t)sh->huf_code[sy] << nb;
                        nb += sh->huf_len[sy];
                    }
                }
                {
                    uint32_t total;
                    const uint32_t off = zb_scan_excl(nb, &total);
                    if (nb)
                    {
                        const uint32_t bp = running + off;
                        const uint64_t v = acc << (bp & 31u); /* nb <= 44, shift <= 31: fits 75 bits -> three words */
                        zb_atomic_or(sc->out + (bp >> 5), (uint32_t)v);
                        if ((bp & 31u) + nb > 32u)
                            zb_atomic_or(sc->out + (bp >> 5) + 1u, (uint32_t)(v >> 32));
                        if ((bp & 31u) + nb > 64u)
                            zb_atomic_or(sc->out + (bp >> 5) + 2u, (uint32_t)(acc >> (64u - (bp & 31u))));
                    }
                    running += total;
                }
            }
            ZB_SERIAL(zl)
            {
                zb_atomic_or(sc->out + (running >> 5), 1u << (running & 31u)); /* end mark after the stream's first symbol */
            }
        }
    }
    else
    {
        uint8_t* dst = out8 + sh->v[ZV_STREAM_BASE];
        for (uint32_t u = 0; u < in->nunits; ++u)
        {
            const uint8_t* src = ((srcmask >> u) & 1u) ? in->src + (size_t)u * ZB_UNIT : in->unit_lits + (uint64_t)u * ZB_UNIT;
            uint8_t* d2 = dst + sh->ulit_base[u];
            ZB_PAR_FOR(j, sh->ulit_base[u + 1u] - sh->ulit_base[u]) d2[j] = src[j];
        }
    }

    ZB_MARK(6);
    /* ---- phase 6: the three FSE state chains, last sequence first.  A chain is serial, so it must not wait on
     * global memory: tiles of 512 sequences get their three codes computed by all lanes into LDS (the Huffman work
     * arrays are free by now), then lanes 0..2 walk the tile. ---- */
    if (nbseq)
    {
        uint32_t* const codes = sh->sort_key; /* [512], spans sort_key + huf_w */
        const uint32_t ntiles = (nbseq + 511u) >> 9;
        ZB_PAR_FOR(t, 3u)
        {
            if (sh->mode[t] != 1u)
            {
                const uint64_t q = sc->seqs[nbseq - 1u];
                const uint32_t s2 = t == ZT_LL ? zb_ll_code(ZB_SEQ_LIT(q)) : t == ZT_ML ? zb_ml_code(ZB_SEQ_ML(q) - 3u) : zb_of_code(ZB_SEQ_OFF(q));
                sh->v[ZV_FINAL_STATE + t] = (1u << sh->table_log[t]) + sh->state_tab[t][sh->sym_start[t][s2]];
            }
        }
        for (uint32_t tile = ntiles; tile-- > 0u;)
        {
            const uint32_t n0 = tile << 9;
            const uint32_t n1 = n0 + 512u < nbseq - 1u ? n0 + 512u : nbseq - 1u; /* the last sequence has no transition */
            ZB_SYNC();
            ZB_PAR_FOR(k, n1 > n0 ? n1 - n0 : 0u)
            {
                const uint64_t q = sc->seqs[n0 + k];
                codes[k] = zb_ll_code(ZB_SEQ_LIT(q)) | (zb_of_code(ZB_SEQ_OFF(q)) << 8) | (zb_ml_code(ZB_SEQ_ML(q) - 3u) << 16);
            }
            ZB_SYNC();
            ZB_PAR_FOR(t, 3u)
            {
                if (sh->mode[t] != 1u)
                {
                    const uint32_t tl = sh->table_log[t];
                    uint16_t* sb = sc->sbits + (uint64_t)t * ZB_SEQ_MAX;
                    uint32_t x = sh->v[ZV_FINAL_STATE + t];
                    /* The only read of a step that depends on the state is the state table's; the symbol's cell count and
                     * first cell are fetched one step ahead and its code two steps ahead, so that a step costs ONE LDS round
                     * trip instead of three. */
                    const uint32_t sh8 = 8u * t;
                    const int16_t* const norm = sh->norm[t];
                    const uint16_t* const sym_start = sh->sym_start[t];
                    const uint16_t* const state_tab = sh->state_tab[t];
                    uint32_t n = n1;
                    uint32_t s_a = n > n0 ? (codes[n - 1u - n0] >> sh8) & 255u : 0u;
                    uint32_t c_a = zb_sym_count(norm, s_a), st_a = sym_start[s_a];
                    uint32_t s_b = n > n0 + 1u ? (codes[n - 2u - n0] >> sh8) & 255u : 0u;
                    while (n-- > n0)
                    {
                        const uint32_t c = c_a, st = st_a;
                        c_a = zb_sym_count(norm, s_b);
                        st_a = sym_start[s_b];
                        s_b = n > n0 + 1u ? (codes[n - 2u - n0] >> sh8) & 255u : 0u;
                        {
                            uint32_t nb = tl - zb_highbit(c);
                            if ((x >> nb) < c)
                                --nb;
                            sb[n] = (uint16_t)((nb << 10) | (x & ((1u << nb) - 1u)));
                            x = (1u << tl) + state_tab[st + ((x >> nb) - c)];
                        }
                    }
                    sh->v[ZV_FINAL_STATE + t] = x;
                }
            }
        }
        ZB_SYNC();
        ZB_PAR_FOR(t, 3u)
        {
            if (sh->mode[t] != 1u)
                sh->v[ZV_FINAL_STATE + t] -= 1u << sh->table_log[t];
        }
    }
    ZB_SYNC();

    ZB_MARK(7);
    /* ---- phase 7: sequence bit-stream, last sequence first: one sequence per lane and step, a wave prefix sum of the bit
     * counts places it (same scheme as the literal streams) ---- */
    {
        uint32_t running = sh->v[ZV_SEQ_BITS0] * 8u;
        for (uint32_t done = 0; done < nbseq; done += ZB_LANES)
        {
            const uint32_t r = done + zl;
            uint32_t bits = 0, lit = 0, ml = 0, ofv = 4, lc = 0, mc = 0, oc = 2, lb = 0, mb = 0, so = 0, sm = 0, sl = 0;
            if (r < nbseq)
            {
                const uint32_t n = nbseq - 1u - r;
                const uint64_t q = sc->seqs[n];
                lit = ZB_SEQ_LIT(q);
                ml = ZB_SEQ_ML(q) - 3u;
                ofv = ZB_SEQ_OFF(q) + 3u;
                lc = zb_ll_code(lit);
                mc = zb_ml_code(ml);
                oc = zb_highbit(ofv);
                lb = zb_ll_bits(lc);
                mb = zb_ml_bits(mc);
                if (n < nbseq - 1u)
                {
                    /* state updates of this sequence: OF, ML, LL (read back as LL, ML, OF) */
                    if (sh->mode[ZT_OF] != 1u)
                        so = sc->sbits[(uint64_t)ZT_OF * ZB_SEQ_MAX + n];
                    if (sh->mode[ZT_ML] != 1u)
                        sm = sc->sbits[(uint64_t)ZT_ML * ZB_SEQ_MAX + n];
                    if (sh->mode[ZT_LL] != 1u)
                        sl = sc->sbits[(uint64_t)ZT_LL * ZB_SEQ_MAX + n];
                }
                bits = (so >> 10) + (sm >> 10) + (sl >> 10) + lb + mb + oc;
            }
            {
                uint32_t total;
                const uint32_t off = zb_scan_excl(bits, &total);
                if (bits)
                {
                    ZbBits bw;
                    zb_bits_open(&bw, sc->out, running + off);
                    zb_bits_put(&bw, so & 1023u, so >> 10);
                    zb_bits_put(&bw, sm & 1023u, sm >> 10);
                    zb_bits_put(&bw, sl & 1023u, sl >> 10);
                    zb_bits_put(&bw, lit - zb_ll_base(lc), lb);
                    zb_bits_put(&bw, ml - zb_ml_base(mc), mb);
                    zb_bits_put(&bw, ofv - (1u << oc), oc);
                    zb_bits_close(&bw);
                }
                running += total;
            }
        }
        ZB_SERIAL(zl) { sh->v[ZV_SEQ_TOTALBITS] = running - sh->v[ZV_SEQ_BITS0] * 8u; }
    }
    ZB_SYNC();

    ZB_MARK(8);
    /* ---- phase 8 (lane 0): final states (ML, OF, LL: read back as LL, OF, ML), end mark, size ---- */
    ZB_SERIAL(zl)
    {
        uint32_t size = sh->v[ZV_SEQ_BITS0];
        if (nbseq)
        {
            ZbBits bw;
            uint32_t bits = sh->v[ZV_SEQ_TOTALBITS];
            zb_bits_open(&bw, sc->out, size * 8u + bits);
            if (sh->mode[ZT_ML] != 1u)
            {
                zb_bits_put(&bw, sh->v[ZV_FINAL_STATE + ZT_ML], sh->table_log[ZT_ML]);
                bits += sh->table_log[ZT_ML];
            }
            if (sh->mode[ZT_OF] != 1u)
            {
                zb_bits_put(&bw, sh->v[ZV_FINAL_STATE + ZT_OF], sh->table_log[ZT_OF]);
                bits += sh->table_log[ZT_OF];
            }
            if (sh->mode[ZT_LL] != 1u)
            {
                zb_bits_put(&bw, sh->v[ZV_FINAL_STATE + ZT_LL], sh->table_log[ZT_LL]);
                bits += sh->table_log[ZT_LL];
            }
            zb_bits_put(&bw, 1u, 1u);
            zb_bits_close(&bw);
            size += (bits + 1u + 7u) >> 3;
        }
        sh->v[ZV_OUT_SIZE] = size < in->raw_size ? size : 0u;
    }
    ZB_SYNC();
    ZB_MARK(10);
    return sh->v[ZV_OUT_SIZE];
}


/* ============================================================================================================
 * The same piece as a run of SUB-BLOCKS, one zstd block per match-finder unit (4 KiB of input).
 *
 * One FSE bit-stream per 128 KiB is one serial chain per piece, for the encoder (state chains) and for every decoder (one
 * sequence after the other); the literals' four Huffman streams are four serial chains.  A wave has 64 lanes.  So the piece
 * is written the way zstd's own target-block-size mode writes it (compress/zstd_compress_superblock.c): the entropy tables
 * are built once, from the statistics of the whole piece, and go out with the FIRST sub-block that needs them; the others
 * say Repeat_Mode / Treeless_Literals_Block (RFC 8878 3.1.1.3.1.1, 3.1.1.3.2.1.1) and carry only their own streams: up to 32
 * sequence streams and 128 literal streams per piece, every one decodable by its own lane.  Any zstd decoder reads the
 * result; the cost is about 20 bytes of headers per sub-block.
 *
 * Output: the sub-blocks back to back, each WITH its 3-byte Block_Header (Last_Block clear), and sub[u] = content size of
 * unit u's block (| 0x8000 when it is a Raw_Block) for the frame's directory.  Returns the total size, 0 when that would
 * not be smaller than one Raw_Block of the piece.
 * ========================================================================================================== */
typedef struct ZbSub /* per-unit values, over lit_hist (free once the Huffman code exists) */
{
    uint32_t seqbits[ZB_MAX_UNITS]; /* bits of the unit's sequence stream before the final states and the end mark */
    uint32_t upos[ZB_MAX_UNITS + 1]; /* byte offset of the unit's Block_Header */
    uint32_t litpos[ZB_MAX_UNITS];  /* ... of its first literal stream / raw literals / raw bytes */
    uint32_t seqpos[ZB_MAX_UNITS];  /* ... of its sequence bit-stream */
    uint16_t fstate[ZB_MAX_UNITS][4];
    uint8_t lmode[ZB_MAX_UNITS]; /* 0 raw literals, 2 Huffman with the tree, 3 treeless, 4 the whole unit is a Raw_Block */
    uint8_t nstr[ZB_MAX_UNITS];  /* Huffman streams: 1 or 4 */
} ZbSub;
#define ZB_SUB_RAW 0x8000u

ZB_FN uint32_t zb_unit_byte(const ZbInput* in, uint32_t srcmask, uint32_t u, uint32_t idx, uint32_t n, uint32_t* cw, uint32_t* cwi)
{
    if ((idx >> 2) != *cwi)
    {
        *cwi = idx >> 2;
        *cw = zb_unit_word(in, srcmask, u, idx >> 2, n);
    }
    return (*cw >> (8u * (idx & 3u))) & 255u;
}

/* ---- the sub-block encoder's merged sequence list: a sequence's three CODES and its extra bits, computed once (phase 1) ----
 * bits 0-5 LL code, 6-11 ML code, 12-16 OF code, 17-29 LL extra bits' value (a unit's literal run: <= 12 bits), 30-45 ML extra bits'
 * value, 46-61 OF extra bits' value (offset value < 2^17: <= 16 bits) */
#define ZP_LC(q) ((uint32_t)(q) & 63u)
#define ZP_MC(q) ((uint32_t)((q) >> 6) & 63u)
#define ZP_OC(q) ((uint32_t)((q) >> 12) & 31u)
#define ZP_LLX(q) ((uint32_t)((q) >> 17) & 0x1FFFu)
#define ZP_MLX(q) ((uint32_t)((q) >> 30) & 0xFFFFu)
#define ZP_OFX(q) ((uint32_t)((q) >> 46) & 0xFFFFu)
ZB_FN uint64_t zb_pack_seq(uint32_t lit, uint32_t ml, uint32_t ofv, uint32_t* lc_out, uint32_t* mc_out, uint32_t* oc_out, uint32_t* xbits)
{
    const uint32_t lc = zb_ll_code(lit), mc = zb_ml_code(ml - 3u), oc = zb_highbit(ofv);
    *lc_out = lc;
    *mc_out = mc;
    *oc_out = oc;
    *xbits = zb_ll_bits(lc) + zb_ml_bits(mc) + oc;
    return (uint64_t)lc | ((uint64_t)mc << 6) | ((uint64_t)oc << 12) | ((uint64_t)(lit - zb_ll_base(lc)) << 17) |
           ((uint64_t)(ml - 3u - zb_ml_base(mc)) << 30) | ((uint64_t)(ofv - (1u << oc)) << 46);
}
/* The same from tables in shared memory (zb_encode_piece_sub builds them in phase 0; the compare chains of zb_ll_code / zb_ml_code and
 * their bit counts and baselines are ~110 instructions a sequence): lcode[64] / mcode[128] the codes of small values, lbits / mbits
 * and lbase / mbase per code. */
typedef struct ZbCodeTabs
{
    uint32_t lbase[36], mbase[53];
    uint8_t lcode[64], mcode[128], lbits[36], mbits[53];
} ZbCodeTabs; /* 637 bytes, over huf_w until the Huffman code is built */
ZB_FN uint64_t zb_pack_seq_t(const ZbCodeTabs* ct, uint32_t lit, uint32_t ml, uint32_t ofv, uint32_t* lc_out, uint32_t* mc_out, uint32_t* oc_out,
                             uint32_t* xbits)
{
    const uint32_t m = ml - 3u;
    const uint32_t lc = lit < 64u ? ct->lcode[lit] : zb_highbit(lit) + 19u, mc = m < 128u ? ct->mcode[m] : zb_highbit(m) + 36u, oc = zb_highbit(ofv);
    *lc_out = lc;
    *mc_out = mc;
    *oc_out = oc;
    *xbits = (uint32_t)ct->lbits[lc] + (uint32_t)ct->mbits[mc] + oc;
    return (uint64_t)lc | ((uint64_t)mc << 6) | ((uint64_t)oc << 12) | ((uint64_t)(lit - ct->lbase[lc]) << 17) | ((uint64_t)(m - ct->mbase[mc]) << 30) |
           ((uint64_t)(ofv - (1u << oc)) << 46);
}
/* One encoding step from the per-symbol entry nb_hi << 22 | (count << nb_hi) << 11 | (sym_start - count + 512) (built once per table,
 * zb_encode_piece_sub phase 2): the same step as zb_fse_step without the symbol's count, its logarithm and sym_start being looked up
 * and derived again for every sequence. */
ZB_FN uint32_t zb_fse_step_tt(uint32_t* x, uint32_t e, const uint16_t* state_tab, uint32_t tl)
{
    const uint32_t nb = (e >> 22) - (*x < ((e >> 11) & 0x7FFu) ? 1u : 0u);
    const uint32_t bits = *x & ((1u << nb) - 1u);
    *x = (1u << tl) + state_tab[(*x >> nb) + (e & 0x7FFu) - 512u];
    return (nb << 10) | bits;
}

/* n bytes from src (any alignment) to dst (any alignment; the words at its two ends are shared with neighbours: bytes there), the
 * words in between four bytes at a time -- all lanes */
typedef uint32_t zb_u32_a1 __attribute__((aligned(1)));
ZB_FN void zb_copy_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t zl)
{
    const uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
    const uint32_t h = head < n ? head : n, nw = (n - h) >> 2, t0 = h + 4u * nw;
    ZB_PAR_FOR(j, h) dst[j] = src[j];
    ZB_PAR_FOR(w, nw) *(uint32_t*)(dst + h + 4u * w) = *(const zb_u32_a1*)(src + h + 4u * w);
    ZB_PAR_FOR(j, n - t0) dst[t0 + j] = src[t0 + j];
}

/* ---- staged bit output (zb_encode_piece_sub, phases 6 and 7) ----
 * The lanes of a step write one contiguous run of bits.  OR-ing every lane's two or three words into the output in global memory is
 * an atomic per word and lane (8 x 10^8 of them per 2 GiB of "tokens": as long as everything else the kernel does); here the lanes
 * OR into a tile in shared memory (`stg`, >= 152 words), and the run's whole words leave with plain stores, one lane a word.  The
 * word a run ends in stays in stg[0] for the next step; a stream's first word (it may hold the bytes in front of the stream) and
 * its last one (the bytes behind it) go out with an atomic OR.
 *   zb_stage_open: before a stream's first step.  zb_stage_begin / zb_stage_end around every step (`total` bits from bit position
 *   `running` of `out`); the lanes write at tile bit (running & 31) + their offset in between.  zb_stage_close: after the last step. */
ZB_FN void zb_stage_open(uint32_t* stg, uint32_t zl)
{
    ZB_SERIAL(zl) { stg[0] = 0; }
}
ZB_FN void zb_stage_begin(uint32_t* stg, uint32_t running, uint32_t total, uint32_t zl)
{
    const uint32_t nw = (((running & 31u) + total) >> 5) + 1u; /* (the word the run ends in, even when it ends on its boundary) */
    ZB_PAR_FOR(w, nw)
    {
        if (w)
            stg[w] = 0;
    }
    ZB_SYNC_LDS();
}
ZB_FN void zb_stage_end(uint32_t* stg, uint32_t* out, uint32_t running, uint32_t total, uint32_t first_bit, uint32_t zl)
{
    const uint32_t nfull = ((running & 31u) + total) >> 5, w0 = running >> 5;
    uint32_t tail;
    ZB_SYNC_LDS();
    tail = stg[nfull];
    ZB_PAR_FOR(w, nfull)
    {
        if (w0 + w == (first_bit >> 5) && (first_bit & 31u))
            zb_atomic_or(out + w0 + w, stg[w]);
        else
            out[w0 + w] = stg[w];
    }
    ZB_SYNC_LDS();
    ZB_SERIAL(zl) { stg[0] = tail; }
    ZB_SYNC_LDS();
}
ZB_FN void zb_stage_close(uint32_t* stg, uint32_t* out, uint32_t running, uint32_t zl)
{
    ZB_SERIAL(zl)
    {
        if (stg[0])
            zb_atomic_or(out + (running >> 5), stg[0]);
    }
    ZB_SYNC_LDS();
}

/* The output is NOT cleared as a whole (133 KiB per piece were: one byte written per byte of input, a fifth of the kernel's memory
 * instructions): every byte of a block is written by exactly one party -- the headers and raw bytes with byte stores, a bit-stream's
 * whole words with plain stores (zb_stage_end) -- except the words a stream shares with its neighbours and the ones its last bits are
 * OR-ed into (stream end, end mark, final states).  The stream's own bytes of THOSE words are cleared here, before the streams are
 * written: [s, e) = the stream's bytes, endbit = the bit (relative to the output) its staged bits end at. */
ZB_FN void zb_zero_edges(uint8_t* out8, uint32_t s, uint32_t e, uint32_t endbit)
{
    const uint32_t a = (s + 3u) & ~3u, z0 = (endbit >> 5) << 2;
    for (uint32_t j = s; j < a && j < e; ++j)
        out8[j] = 0;
    for (uint32_t j = z0 > s ? z0 : s; j < e; ++j)
        out8[j] = 0;
}

ZB_FN uint32_t zb_encode_piece_sub(const ZbInput* in, const ZbScratch* sc, ZbShared* sh, uint32_t zl, uint16_t* sub)
{
    uint8_t* const out8 = (uint8_t*)sc->out;
    ZbSub* const sb = (ZbSub*)sh->lit_hist;
    ZbCodeTabs* const ct = (ZbCodeTabs*)sh->huf_w; /* phase 1's code tables (the Huffman build takes the memory afterwards) */
    uint32_t* const stg = sh->sort_key; /* the staged bit output's tile (free once the FSE tables are built) */
    uint32_t* const strbits = sh->huf_w + 128; /* [ZB_MAX_UNITS][4] bits of every literal stream (past the FSE builders' spread area) */
    const uint32_t nunits = in->nunits;

    /* ---- phase 0: unit bases; zero the histograms ---- */
    ZB_PAR_FOR(u, nunits)
    {
        const ZbUnitMeta m = in->meta[u];
        sh->useq_base[u] = m.nseq;
        sh->ulit_base[u] = m.nlit;
        sh->carry[u] = 0; /* the unit's extra bits (phase 1 adds them up) */
    }
    ZB_SYNC();
    ZB_SERIAL(zl)
    {
        uint32_t nseq = 0, nlit = 0, srcmask = 0;
        for (uint32_t u = 0; u < nunits; ++u)
        {
            const uint32_t un = sh->useq_base[u], ul = sh->ulit_base[u];
            if (in->src && un == 0u)
                srcmask |= 1u << u;
            sh->useq_base[u] = nseq;
            sh->ulit_base[u] = nlit;
            nseq += un;
            nlit += ul;
        }
        sh->useq_base[nunits] = nseq;
        sh->ulit_base[nunits] = nlit;
        sh->v[ZV_NBSEQ] = nseq;
        sh->v[ZV_NLIT] = nlit;
        sh->v[ZV_SRCMASK] = srcmask;
    }
    ZB_PAR_FOR(i, 256u) sh->lit_hist[i] = 0;
    ZB_PAR_FOR(i, 3u * 64u) sh->sym_hist[i >> 6][i & 63u] = 0;
    ZB_PAR_FOR(i, 128u)
    {
        ct->mcode[i] = (uint8_t)zb_ml_code(i);
        if (i < 64u)
            ct->lcode[i] = (uint8_t)zb_ll_code(i);
        if (i < 36u)
        {
            ct->lbits[i] = (uint8_t)zb_ll_bits(i);
            ct->lbase[i] = zb_ll_base(i);
        }
        if (i < 53u)
        {
            ct->mbits[i] = (uint8_t)zb_ml_bits(i);
            ct->mbase[i] = zb_ml_base(i);
        }
    }
    ZB_SYNC();
    const uint32_t nbseq = sh->v[ZV_NBSEQ], nlit = sh->v[ZV_NLIT], srcmask = sh->v[ZV_SRCMASK];

    ZB_MARK(1);
    /* ---- phase 1: the sequences in block order (a unit's trailing literals stay with the unit: they are its block's last
     * literals), the three symbol histograms, the literal histogram ---- */
    if (!(in->flags & ZB_F_REPCODES))
    {
        /* eight sequences per lane and trip, their records loaded before the first is packed: the loop is bound by the round trips
         * to memory (one wave per piece), and this way eight of them are in flight (four: +0.7 % of the kernel in a same-box A/B) */
        ZB_PAR_FOR_K(t4, nbseq, 8u)
        {
            const uint32_t ibase = t4 * 8u * ZB_LANES + zl;
            uint64_t r4[8];
            uint32_t u4[8];
            ZB_UNROLL
            for (uint32_t q = 0; q < 8u; ++q)
            {
                const uint32_t i = ibase + q * ZB_LANES;
                uint32_t lo = 0;
                ZB_UNROLL
                for (uint32_t st = ZB_MAX_UNITS / 2u; st; st >>= 1) /* the last unit whose first sequence is at or before i */
                    if (lo + st < nunits && sh->useq_base[lo + st] <= i)
                        lo += st;
                u4[q] = lo;
                r4[q] = i < nbseq ? in->unit_recs[(uint64_t)lo * ZB_UNIT_SEQ_MAX + (i - sh->useq_base[lo])] : 0u;
            }
            ZB_UNROLL
            for (uint32_t q = 0; q < 8u; ++q)
            {
                const uint32_t i = ibase + q * ZB_LANES;
                if (i < nbseq)
                {
                    const uint64_t r = r4[q];
                    const uint32_t lit = (uint32_t)(r & 0xFFFFu), ml = (uint32_t)((r >> 16) & 0xFFFFu), off = (uint32_t)(r >> 32);
                    uint32_t lc, mc, oc, xb;
                    sc->seqs[i] = zb_pack_seq_t(ct, lit, ml, off + 3u, &lc, &mc, &oc, &xb);
                    zb_atomic_add(&sh->sym_hist[ZT_LL][lc], 1u);
                    zb_atomic_add(&sh->sym_hist[ZT_ML][mc], 1u);
                    zb_atomic_add(&sh->sym_hist[ZT_OF][oc], 1u);
                    zb_atomic_add(&sh->carry[u4[q]], xb);
                }
            }
        }
    }
    else
    {
        /* ---- with repeat-offset codes (zstd_compression_format.md "Repeat Offsets"; ZSTD_updateRep / ZSTD_storeSeq of the reference,
         * compress/zstd_compress_internal.h).  A zstd block starts with the three-entry offset history its predecessor left behind --
         * which is exactly what a decoder that gives every block a lane of its own does not have.  So a block here only ever refers to
         * history entries that were SET BY ITS OWN SEQUENCES: the history starts "unknown" in every block, an entry becomes known when
         * a sequence of the block writes it, and a repeat code is used only for a known entry.  Any zstd decoder reads such a block (it
         * simply never looks at what it inherited); the lane-parallel one needs nothing from the block before.  One lane per block, in
         * sequence order (the chain is serial by nature): the lane reads its unit's records eight at a time (the loads do not depend on
         * the history; one at a time the lane waited a memory round trip per sequence), writes the merged list and counts all three
         * code histograms. */
        ZB_PAR_FOR(u, nunits)
        {
            const uint32_t b0 = sh->useq_base[u], e0 = sh->useq_base[u + 1u];
            const uint64_t* recs = in->unit_recs + (uint64_t)u * ZB_UNIT_SEQ_MAX;
            uint32_t r1 = 0, r2 = 0, r3 = 0; /* 0 = unknown (an offset is never 0) */
            for (uint32_t i0 = b0; i0 < e0; i0 += 8u)
            {
                uint64_t q8[8];
                const uint32_t cnt = e0 - i0 < 8u ? e0 - i0 : 8u;
                for (uint32_t j = 0; j < 8u; ++j)
                    q8[j] = j < cnt ? recs[i0 - b0 + j] : 0u;
                for (uint32_t j = 0; j < 8u; ++j)
                    if (j < cnt)
                    {
                        const uint64_t r = q8[j];
                        const uint32_t lit = (uint32_t)(r & 0xFFFFu), ml = (uint32_t)((r >> 16) & 0xFFFFu), off = (uint32_t)(r >> 32);
                        uint32_t code = 0; /* 0: the offset itself */
                        if (lit != 0u)
                            code = off == r1 ? 1u : off == r2 ? 2u : off == r3 ? 3u : 0u;
                        else
                            code = off == r2 ? 1u : off == r3 ? 2u : (r1 > 1u && off == r1 - 1u) ? 3u : 0u;
                        /* which history entry was used (with literals: the code; without: one further, code 3 = r1 - 1 counts as "new") */
                        const uint32_t used = code == 0u ? 0u : (lit != 0u ? code : code + 1u);
                        if (used == 2u)
                        {
                            const uint32_t t = r2;
                            r2 = r1;
                            r1 = t;
                        }
                        else if (used != 1u) /* a new offset, entry 3, or r1 - 1: pushed in front */
                        {
                            const uint32_t v = used == 3u ? r3 : off;
                            r3 = r2;
                            r2 = r1;
                            r1 = v;
                        }
                        {
                            uint32_t lc, mc, oc, xb;
                            sc->seqs[i0 + j] = zb_pack_seq_t(ct, lit, ml, code ? code : off + 3u, &lc, &mc, &oc, &xb);
                            zb_atomic_add(&sh->sym_hist[ZT_LL][lc], 1u);
                            zb_atomic_add(&sh->sym_hist[ZT_ML][mc], 1u);
                            zb_atomic_add(&sh->sym_hist[ZT_OF][oc], 1u);
                            zb_atomic_add(&sh->carry[u], xb);
                        }
                    }
            }
        }
    }
    ZB_MARK(11);
    /* plainly noise?  (the sampled test of zb_encode_block) */
    if (nlit >= 32768u && in->raw_size - nlit < 3u * nbseq + 32u && !(ZB_DBG & 8u))
    {
        for (uint32_t u = (nlit >> 12) & 7u; u < nunits; u += 8u)
        {
            const uint32_t n = sh->ulit_base[u + 1u] - sh->ulit_base[u];
            ZB_PAR_FOR(j, n >> 2)
            {
                const uint32_t w = zb_unit_word(in, srcmask, u, j, n);
                zb_atomic_add(&sh->lit_hist[w & 255u], 1u);
                zb_atomic_add(&sh->lit_hist[(w >> 8) & 255u], 1u);
                zb_atomic_add(&sh->lit_hist[(w >> 16) & 255u], 1u);
                zb_atomic_add(&sh->lit_hist[w >> 24], 1u);
            }
        }
        ZB_SYNC();
        ZB_SERIAL(zl)
        {
            uint32_t largest = 0, ns = 0;
            for (uint32_t s2 = 0; s2 < 256u; ++s2)
            {
                ns += sh->lit_hist[s2];
                if (sh->lit_hist[s2] > largest)
                    largest = sh->lit_hist[s2];
            }
            sh->v[ZV_SKIP] = (ns >= 2048u && largest <= (ns >> 7) + 4u) ? 1u : 0u;
        }
        ZB_SYNC();
        if (sh->v[ZV_SKIP])
            return 0;
        ZB_PAR_FOR(i, 256u) sh->lit_hist[i] = 0;
        ZB_SYNC();
    }
    {
        /* The literal QUADS (four words, 16 bytes) of all units as one list (qbase[u] = quads of the units before u; a unit's last
         * quad may be partial): two quads per lane and trip, loaded before the first counter is touched.  The loop is bound by the
         * round trips to memory: unit by unit a piece took 64 of them one after the other (15 % of the kernel's time). */
        uint16_t* const qbase = sh->cursor[0]; /* [nunits + 1] <= 33 entries, <= 8192 (free until the table builds) */
        ZB_SERIAL(zl)
        {
            uint32_t acc = 0;
            for (uint32_t u = 0; u < nunits; ++u)
            {
                qbase[u] = (uint16_t)acc;
                acc += (sh->ulit_base[u + 1u] - sh->ulit_base[u] + 15u) >> 4;
            }
            qbase[nunits] = (uint16_t)acc;
        }
        ZB_SYNC_LDS();
        const uint32_t nquads = qbase[nunits];
        ZB_PAR_FOR_K(t2, nquads, 2u)
        {
            uint32_t w[2][4], nb[2];
            ZB_UNROLL
            for (uint32_t q = 0; q < 2u; ++q)
            {
                const uint32_t g = t2 * 2u * ZB_LANES + q * ZB_LANES + zl;
                uint32_t lo = 0;
                ZB_UNROLL
                for (uint32_t st = ZB_MAX_UNITS / 2u; st; st >>= 1) /* the last unit whose first quad is at or before g */
                    if (lo + st < nunits && (uint32_t)qbase[lo + st] <= g)
                        lo += st;
                nb[q] = 0;
                ZB_UNROLL
                for (uint32_t k = 0; k < 4u; ++k)
                    w[q][k] = 0;
                if (g < nquads)
                {
                    const uint32_t n = sh->ulit_base[lo + 1u] - sh->ulit_base[lo], w0 = 4u * (g - qbase[lo]);
                    nb[q] = n - 4u * w0; /* valid bytes from this quad's first word on, >= 1 */
                    ZB_UNROLL
                    for (uint32_t k = 0; k < 4u; ++k)
                        if (4u * k < nb[q])
                            w[q][k] = zb_unit_word(in, srcmask, lo, w0 + k, n);
                }
            }
            ZB_UNROLL
            for (uint32_t q = 0; q < 2u; ++q)
            {
                if (nb[q] >= 16u) /* a whole quad (all but a unit's last one): no byte is questioned */
                {
                    ZB_UNROLL
                    for (uint32_t k = 0; k < 4u; ++k)
                    {
                        const uint32_t v = w[q][k];
                        zb_atomic_add(&sh->lit_hist[v & 255u], 1u);
                        zb_atomic_add(&sh->lit_hist[(v >> 8) & 255u], 1u);
                        zb_atomic_add(&sh->lit_hist[(v >> 16) & 255u], 1u);
                        zb_atomic_add(&sh->lit_hist[v >> 24], 1u);
                    }
                    continue;
                }
                ZB_UNROLL
                for (uint32_t k = 0; k < 4u; ++k)
                {
                    const uint32_t v = w[q][k], left = nb[q] > 4u * k ? nb[q] - 4u * k : 0u; /* valid bytes in this word */
                    if (left > 0u)
                        zb_atomic_add(&sh->lit_hist[v & 255u], 1u);
                    if (left > 1u)
                        zb_atomic_add(&sh->lit_hist[(v >> 8) & 255u], 1u);
                    if (left > 2u)
                        zb_atomic_add(&sh->lit_hist[(v >> 16) & 255u], 1u);
                    if (left > 3u)
                        zb_atomic_add(&sh->lit_hist[v >> 24], 1u);
                }
            }
        }
    }
    ZB_SYNC();

    ZB_MARK(2);
    /* ---- phase 2: Huffman code for the literals, FSE tables for the three symbol types: as in zb_encode_block, from the statistics
     * of the whole piece (with a wave: by all lanes -- zb_huffman_build_par, zb_build_seq_tables; the one-lane build runs the serial
     * builders and must produce the same tables) ---- */
    ZB_SERIAL(zl)
    {
        uint32_t largest = 0;
        for (uint32_t s2 = 0; s2 < 256u; ++s2)
            if (sh->lit_hist[s2] > largest)
                largest = sh->lit_hist[s2];
        sh->v[ZV_HUF_OK] = 0;
        sh->v[ZV_TREE_BYTES] = 0;
        sh->v[ZV_HUF_NSYM] = 0;
        sh->v[ZV_LIT_HDR] = (nlit >= 256u && !(ZB_DBG & 1u) && largest > (nlit >> 7) + 4u) ? 1u : 0u; /* try Huffman */
        sh->v[ZV_SKIP] = (!sh->v[ZV_LIT_HDR] && in->raw_size - nlit < 3u * nbseq + 32u) ? 1u : 0u;
    }
    ZB_SYNC();
    if (sh->v[ZV_SKIP])
        return 0;
    ZB_MARK(12);
    if (sh->v[ZV_LIT_HDR])
        zb_huffman_sort(sh, zl);
    ZB_SYNC();
    ZB_MARK(13);
#if ZB_LANES > 1
    if (sh->v[ZV_LIT_HDR])
        zb_huffman_build_par(sh, zl);
#else
    ZB_SERIAL(zl)
    {
        if (sh->v[ZV_LIT_HDR])
            zb_huffman_build(sh);
    }
#endif
    ZB_SYNC();
    ZB_MARK(14);
    ZB_SERIAL(zl)
    {
        if (sh->v[ZV_HUF_OK]) /* uses table slot 0 as work space: must precede the FSE tables below */
        {
            sh->v[ZV_TREE_BYTES] = zb_write_huf_tree(sh, sh->tree);
            if (!sh->v[ZV_TREE_BYTES])
                sh->v[ZV_HUF_OK] = 0;
        }
    }
    ZB_PAR_FOR(i, 4u * ZB_MAX_UNITS) strbits[i] = 0;
    ZB_SYNC();
    ZB_MARK(9);
    zb_build_seq_tables(sh, nbseq, zl);
    /* the literal histogram is dead: its memory holds the per-unit values from here on */
    ZB_SYNC();
    /* ... and so are the code histograms: per symbol, what an encoding step needs of it (zb_fse_step_tt); the work arrays of the
     * table builds become the extra-bit counts of the LL and ML codes */
    ZB_PAR_FOR(i, 3u * 64u)
    {
        const uint32_t t = i >> 6, s2 = i & 63u;
        const int16_t nv = sh->norm[t][s2];
        uint32_t e = 0;
        if (sh->mode[t] != 1u && nv != 0)
        {
            const uint32_t c = (uint32_t)(nv < 0 ? 1 : nv), nbh = (uint32_t)sh->table_log[t] - zb_highbit(c);
            e = (nbh << 22) | ((c << nbh) << 11) | ((uint32_t)sh->sym_start[t][s2] + 512u - c);
        }
        sh->sym_hist[t][s2] = e;
    }
    ZB_PAR_FOR(i, 64u)
    {
        ((uint8_t*)sh->cursor[0])[i] = (uint8_t)(i < 36u ? zb_ll_bits(i) : 0u);
        ((uint8_t*)sh->cursor[1])[i] = (uint8_t)(i < 53u ? zb_ml_bits(i) : 0u);
    }
    ZB_PAR_FOR(u, ZB_MAX_UNITS)
    {
        sb->seqbits[u] = 0;
        sb->lmode[u] = 0;
        sb->nstr[u] = 0;
    }
    ZB_SYNC();

    ZB_MARK(3);
    /* ---- phase 3: bits of every literal stream: a unit's literals are one stream below 256 of them, else four (three of
     * ceil(n / 4), the last takes the rest) ---- */
    if (sh->v[ZV_HUF_OK])
    {
        /* the literal quads of all units as one list, two quads per lane and trip (as in the histogram of phase 1: word by word and
         * unit by unit a piece made 224 round trips to memory here, one after the other) */
        uint16_t* const qbase = (uint16_t*)sh->small; /* [nunits + 1] <= 33 of the 64 entries (the serial builders' scratch) */
        ZB_SERIAL(zl)
        {
            uint32_t acc = 0;
            for (uint32_t u = 0; u < nunits; ++u)
            {
                qbase[u] = (uint16_t)acc;
                acc += (sh->ulit_base[u + 1u] - sh->ulit_base[u] + 15u) >> 4;
            }
            qbase[nunits] = (uint16_t)acc;
        }
        ZB_SYNC_LDS();
        const uint32_t nquads = qbase[nunits];
        ZB_PAR_FOR_K(t2, nquads, 2u)
        {
            uint32_t w[2][4], un[2], uu[2], w0s[2];
            ZB_UNROLL
            for (uint32_t qd = 0; qd < 2u; ++qd)
            {
                const uint32_t g = t2 * 2u * ZB_LANES + qd * ZB_LANES + zl;
                uint32_t lo = 0;
                ZB_UNROLL
                for (uint32_t st = ZB_MAX_UNITS / 2u; st; st >>= 1)
                    if (lo + st < nunits && (uint32_t)qbase[lo + st] <= g)
                        lo += st;
                un[qd] = 0;
                uu[qd] = lo;
                w0s[qd] = 0;
                ZB_UNROLL
                for (uint32_t k = 0; k < 4u; ++k)
                    w[qd][k] = 0;
                if (g < nquads)
                {
                    const uint32_t n = sh->ulit_base[lo + 1u] - sh->ulit_base[lo], w0 = 4u * (g - qbase[lo]);
                    un[qd] = n;
                    w0s[qd] = w0;
                    ZB_UNROLL
                    for (uint32_t k = 0; k < 4u; ++k)
                        if (4u * (w0 + k) < n)
                            w[qd][k] = zb_unit_word(in, srcmask, lo, w0 + k, n);
                }
            }
            ZB_UNROLL
            for (uint32_t qd = 0; qd < 2u; ++qd)
            {
                const uint32_t n = un[qd], u = uu[qd];
                const uint32_t seg = n < 256u ? n : (n + 3u) >> 2;
                {
                    /* a whole quad inside ONE stream (nearly all of them): sixteen code lengths, one addition to the stream's total */
                    const uint32_t b0 = 4u * w0s[qd], b1 = b0 + 15u;
                    const uint32_t q0 = (b0 >= seg) + (b0 >= 2u * seg) + (b0 >= 3u * seg), q1 = (b1 >= seg) + (b1 >= 2u * seg) + (b1 >= 3u * seg);
                    if (b1 < n && q0 == q1)
                    {
                        uint32_t bits = 0;
                        ZB_UNROLL
                        for (uint32_t kw = 0; kw < 4u; ++kw)
                        {
                            const uint32_t wv = w[qd][kw];
                            bits += (uint32_t)sh->huf_len[wv & 255u] + sh->huf_len[(wv >> 8) & 255u] + sh->huf_len[(wv >> 16) & 255u] + sh->huf_len[wv >> 24];
                        }
                        zb_atomic_add(&strbits[4u * u + q0], bits);
                        continue;
                    }
                }
                ZB_UNROLL
                for (uint32_t kw = 0; kw < 4u; ++kw)
                {
                    const uint32_t j = w0s[qd] + kw;
                    if (4u * j < n)
                    {
                        const uint32_t wv = w[qd][kw];
                        const uint32_t k = n - 4u * j < 4u ? n - 4u * j : 4u;
                        uint32_t q = (4u * j >= seg) + (4u * j >= 2u * seg) + (4u * j >= 3u * seg), bits = 0;
                        for (uint32_t b = 0; b < k; ++b)
                        {
                            const uint32_t ib = 4u * j + b;
                            const uint32_t qb = (ib >= seg) + (ib >= 2u * seg) + (ib >= 3u * seg);
                            if (qb != q)
                            {
                                zb_atomic_add(&strbits[4u * u + q], bits);
                                bits = 0;
                                q = qb;
                            }
                            bits += sh->huf_len[(wv >> (8u * b)) & 255u];
                        }
                        zb_atomic_add(&strbits[4u * u + q], bits);
                    }
                }
            }
        }
    }

    ZB_MARK(4);
    /* ---- phase 4: the FSE state chains, last sequence first.  A chain is serial: every unit's three chains run on the unit's
     * own lane, interleaved (three independent LDS round trips per step instead of one), eight sequences loaded ahead of the
     * eight steps.  The lane also adds up the sequences' extra bits. ---- */
    ZB_PAR_FOR(u, nunits)
    {
        const uint32_t b0 = sh->useq_base[u], e0 = sh->useq_base[u + 1u];
        if (e0 > b0)
        {
            const uint32_t tl_l = sh->table_log[ZT_LL], tl_o = sh->table_log[ZT_OF], tl_m = sh->table_log[ZT_ML];
            const uint32_t c_l = sh->mode[ZT_LL] != 1u, c_o = sh->mode[ZT_OF] != 1u, c_m = sh->mode[ZT_ML] != 1u;
            uint32_t x_l = 0, x_o = 0, x_m = 0, bits = sh->carry[u], n = e0; /* (the extra bits were added up in phase 1) */
            uint64_t qn[8]; /* the eight sequences after the ones being worked on: loaded a batch ahead (the lane waited a round trip
                             * to memory per batch) */
            ZB_UNROLL
            for (uint32_t j = 0; j < 8u; ++j)
                qn[j] = j < n - b0 ? sc->seqs[n - 1u - j] : 0u;
            while (n > b0)
            {
                uint64_t q[8], tr[8]; /* the steps' transition bits: stored eight at a time, one 64-byte run per lane (three 2-byte
                                       * stores per step and lane were 96 partial cache lines per wave and step: half of "tokens"' time) */
                const uint32_t cnt = n - b0 < 8u ? n - b0 : 8u;
                ZB_UNROLL
                for (uint32_t j = 0; j < 8u; ++j)
                    q[j] = qn[j];
                ZB_UNROLL
                for (uint32_t j = 0; j < 8u; ++j)
                    qn[j] = cnt + j < n - b0 ? sc->seqs[n - cnt - 1u - j] : 0u;
                ZB_UNROLL
                for (uint32_t j = 0; j < 8u; ++j)
                {
                    tr[j] = 0;
                    if (j < cnt)
                    {
                        const uint32_t i = n - 1u - j;
                        const uint32_t lc = ZP_LC(q[j]), mc = ZP_MC(q[j]), oc = ZP_OC(q[j]);
                        if (i == e0 - 1u) /* the block's last sequence: the states the decoder starts from */
                        {
                            x_l = (1u << tl_l) + (c_l ? sh->state_tab[ZT_LL][sh->sym_start[ZT_LL][lc]] : 0u);
                            x_o = (1u << tl_o) + (c_o ? sh->state_tab[ZT_OF][sh->sym_start[ZT_OF][oc]] : 0u);
                            x_m = (1u << tl_m) + (c_m ? sh->state_tab[ZT_ML][sh->sym_start[ZT_ML][mc]] : 0u);
                        }
                        else
                        {
                            /* the three entries first: they depend on the codes alone, the steps on the states */
                            const uint32_t e_l = sh->sym_hist[ZT_LL][lc], e_o = sh->sym_hist[ZT_OF][oc], e_m = sh->sym_hist[ZT_ML][mc];
                            uint32_t r_l = 0, r_o = 0, r_m = 0;
                            if (c_l)
                                r_l = zb_fse_step_tt(&x_l, e_l, sh->state_tab[ZT_LL], tl_l);
                            if (c_o)
                                r_o = zb_fse_step_tt(&x_o, e_o, sh->state_tab[ZT_OF], tl_o);
                            if (c_m)
                                r_m = zb_fse_step_tt(&x_m, e_m, sh->state_tab[ZT_ML], tl_m);
                            bits += (r_l >> 10) + (r_o >> 10) + (r_m >> 10);
                            tr[j] = (uint64_t)r_l | ((uint64_t)r_o << 16) | ((uint64_t)r_m << 32);
                        }
                    }
                }
                for (uint32_t j = 0; j < 8u; ++j)
                    if (j < cnt)
                        ((uint64_t*)sc->sbits)[n - 1u - j] = tr[j];
                n -= cnt;
            }
            sb->fstate[u][ZT_LL] = (uint16_t)(c_l ? x_l - (1u << tl_l) : 0u);
            sb->fstate[u][ZT_OF] = (uint16_t)(c_o ? x_o - (1u << tl_o) : 0u);
            sb->fstate[u][ZT_ML] = (uint16_t)(c_m ? x_m - (1u << tl_m) : 0u);
            sb->seqbits[u] = bits;
        }
    }
    ZB_SYNC();

    ZB_MARK(5);
    /* ---- phase 5 (lane 0): what every unit becomes, all headers, where its streams go ---- */
    ZB_SERIAL(zl)
    {
        uint32_t pos = 0, tree_due = sh->v[ZV_HUF_OK], tables_due = 1;
        const uint32_t tb = sh->v[ZV_TREE_BYTES];
        for (uint32_t u = 0; u < nunits; ++u)
        {
            const uint32_t n = sh->ulit_base[u + 1u] - sh->ulit_base[u];
            const uint32_t ns = sh->useq_base[u + 1u] - sh->useq_base[u];
            const uint32_t ubytes = in->raw_size - u * ZB_UNIT < ZB_UNIT ? in->raw_size - u * ZB_UNIT : ZB_UNIT;
            const uint32_t rawhdr = n < 32u ? 1u : n < 4096u ? 2u : 3u;
            uint32_t lmode = 0, nstr = 0, lsize = rawhdr + n, lhdr = rawhdr, cs = 0;
            if (sh->v[ZV_HUF_OK] && n)
            {
                nstr = n < 256u ? 1u : 4u;
                for (uint32_t q = 0; q < nstr; ++q)
                    cs += (strbits[4u * u + q] + 1u + 7u) >> 3; /* + end mark */
                cs += (nstr == 4u ? 6u : 0u) + (tree_due ? tb : 0u);
                {
                    const uint32_t hdr = (n < 1024u && cs < 1024u) ? 3u : 4u;
                    if (cs + hdr < lsize)
                    {
                        lmode = tree_due ? 2u : 3u;
                        lsize = cs + hdr;
                        lhdr = hdr;
                    }
                }
            }
            /* sequences section: count, modes and (first time) the table descriptions, the bit-stream */
            uint32_t shdr = 1, sbytes = 0;
            if (ns)
            {
                uint32_t bits = sb->seqbits[u] + 1u;
                for (uint32_t t = 0; t < 3u; ++t)
                    if (sh->mode[t] != 1u)
                        bits += sh->table_log[t];
                sbytes = (bits + 7u) >> 3;
                shdr = (ns < 128u ? 1u : 2u) + 1u;
            }
            /* (the table descriptions are written in place below, their size is known only then: the test leaves them out) */
            {
                if (in->src && lsize + shdr + sbytes >= ubytes)
                {
                    /* does not pay: the unit's bytes as a Raw_Block (entropy tables live on across it) */
                    const uint32_t h = (0u << 1) | (ubytes << 3);
                    out8[pos] = (uint8_t)h;
                    out8[pos + 1u] = (uint8_t)(h >> 8);
                    out8[pos + 2u] = (uint8_t)(h >> 16);
                    sb->upos[u] = pos;
                    sb->litpos[u] = pos + 3u;
                    sb->lmode[u] = 4;
                    sub[u] = (uint16_t)(ubytes | ZB_SUB_RAW);
                    pos += 3u + ubytes;
                    continue;
                }
            }
            sb->upos[u] = pos;
            {
                uint32_t p = pos + 3u;
                if (lmode >= 2u)
                {
                    const uint32_t sf = nstr == 1u ? 0u : lhdr == 3u ? 1u : 2u;
                    const uint32_t nb = lhdr == 3u ? 10u : 14u;
                    const uint32_t h = lmode | (sf << 2) | (n << 4) | (cs << (4u + nb));
                    for (uint32_t k = 0; k < lhdr; ++k)
                        out8[p++] = (uint8_t)(h >> (8u * k));
                    if (lmode == 2u)
                    {
                        for (uint32_t k = 0; k < tb; ++k)
                            out8[p++] = sh->tree[k];
                        tree_due = 0;
                    }
                    if (nstr == 4u)
                        for (uint32_t q = 0; q < 3u; ++q)
                        {
                            const uint32_t by = (strbits[4u * u + q] + 1u + 7u) >> 3;
                            out8[p++] = (uint8_t)by;
                            out8[p++] = (uint8_t)(by >> 8);
                        }
                    sb->litpos[u] = p;
                    p = pos + 3u + lsize;
                }
                else
                {
                    if (n < 32u)
                        out8[p++] = (uint8_t)(n << 3);
                    else if (n < 4096u)
                    {
                        const uint32_t h = 4u | (n << 4);
                        out8[p++] = (uint8_t)h;
                        out8[p++] = (uint8_t)(h >> 8);
                    }
                    else
                    {
                        const uint32_t h = 12u | (n << 4);
                        out8[p++] = (uint8_t)h;
                        out8[p++] = (uint8_t)(h >> 8);
                        out8[p++] = (uint8_t)(h >> 16);
                    }
                    sb->litpos[u] = p;
                    p += n;
                }
                if (ns == 0u)
                    out8[p++] = 0;
                else
                {
                    if (ns < 128u)
                        out8[p++] = (uint8_t)ns;
                    else
                    {
                        out8[p++] = (uint8_t)((ns >> 8) + 128u);
                        out8[p++] = (uint8_t)ns;
                    }
                    if (tables_due)
                    {
                        out8[p++] = (uint8_t)((sh->mode[ZT_LL] << 6) | (sh->mode[ZT_OF] << 4) | (sh->mode[ZT_ML] << 2));
                        for (uint32_t t = 0; t < 3u; ++t) /* LL, OF, ML in this order */
                        {
                            if (sh->mode[t] == 1u)
                                out8[p++] = sh->rle_sym[t];
                            else if (sh->mode[t] == 2u)
                                p += zb_write_ncount(out8 + p, sh->norm[t], (uint32_t)sh->rle_sym[t] + 1u, sh->table_log[t]);
                        }
                        tables_due = 0;
                    }
                    else /* what the first one said: Predefined again, anything else by Repeat_Mode */
                        out8[p++] = (uint8_t)(((sh->mode[ZT_LL] ? 3u : 0u) << 6) | ((sh->mode[ZT_OF] ? 3u : 0u) << 4) |
                                              ((sh->mode[ZT_ML] ? 3u : 0u) << 2));
                    sb->seqpos[u] = p;
                    p += sbytes;
                }
                {
                    const uint32_t content = p - (pos + 3u);
                    const uint32_t h = (2u << 1) | (content << 3);
                    out8[pos] = (uint8_t)h;
                    out8[pos + 1u] = (uint8_t)(h >> 8);
                    out8[pos + 2u] = (uint8_t)(h >> 16);
                    sub[u] = (uint16_t)content;
                }
                sb->lmode[u] = (uint8_t)lmode;
                sb->nstr[u] = (uint8_t)nstr;
                pos = p;
            }
        }
        sb->upos[nunits] = pos;
        sh->v[ZV_OUT_SIZE] = pos < in->raw_size + 3u ? pos : 0u;
    }
    ZB_SYNC();
    if (!sh->v[ZV_OUT_SIZE])
        return 0;
    ZB_PAR_FOR(u, nunits)
    {
        const uint32_t lmode = sb->lmode[u];
        if (lmode == 2u || lmode == 3u)
        {
            uint32_t s0 = sb->litpos[u];
            for (uint32_t q = 0; q < sb->nstr[u]; ++q)
            {
                const uint32_t bits = strbits[4u * u + q], by = (bits + 1u + 7u) >> 3;
                zb_zero_edges(out8, s0, s0 + by, s0 * 8u + bits);
                s0 += by;
            }
        }
        if (sh->useq_base[u + 1u] > sh->useq_base[u] && lmode != 4u)
        {
            uint32_t bits = sb->seqbits[u] + 1u;
            for (uint32_t t = 0; t < 3u; ++t)
                if (sh->mode[t] != 1u)
                    bits += sh->table_log[t];
            zb_zero_edges(out8, sb->seqpos[u], sb->seqpos[u] + ((bits + 7u) >> 3), sb->seqpos[u] * 8u + sb->seqbits[u]);
        }
    }
    ZB_SYNC();

    ZB_MARK(6);
    /* ---- phase 6: literals.  A Huffman stream is written from its LAST literal: per step every lane takes the next four
     * literals (lane 0 the last four), a wave prefix sum of the bit counts places them (as in zb_encode_block). ---- */
    for (uint32_t u = 0; u < nunits; ++u)
    {
        const uint32_t n = sh->ulit_base[u + 1u] - sh->ulit_base[u];
        const uint32_t lmode = sb->lmode[u];
        if (lmode == 2u || lmode == 3u)
        {
            const uint32_t nstr = sb->nstr[u], seg = nstr == 1u ? n : (n + 3u) >> 2;
            uint32_t base = sb->litpos[u];
            for (uint32_t st = 0; st < nstr; ++st)
            {
                const uint32_t s0 = st * seg, s1 = st + 1u == nstr ? n : s0 + seg;
                const uint32_t first_bit = base * 8u;
                uint32_t running = first_bit;
                zb_stage_open(stg, zl);
                for (uint32_t done = 0; done < s1 - s0; done += 4u * ZB_LANES)
                {
                    uint64_t acc = 0;
                    uint32_t nb = 0;
                    const uint32_t r0 = done + 4u * zl;
                    if (r0 < s1 - s0)
                    {
                        /* my (up to) four literals, the highest index first: they lie in one or two words, both loaded at once (a
                         * load per byte that leaves the cached word made two dependent round trips of them) */
                        const uint32_t k4 = s1 - s0 - r0 < 4u ? s1 - s0 - r0 : 4u;
                        const uint32_t hi_idx = s1 - 1u - r0, lo_idx = hi_idx + 1u - k4;
                        const uint32_t whi = zb_unit_word(in, srcmask, u, hi_idx >> 2, n);
                        const uint32_t wlo = (lo_idx >> 2) != (hi_idx >> 2) ? zb_unit_word(in, srcmask, u, lo_idx >> 2, n) : whi;
                        ZB_UNROLL
                        for (uint32_t j = 0; j < 4u; ++j)
                            if (j < k4)
                            {
                                const uint32_t idx = hi_idx - j;
                                const uint32_t sy = (((idx >> 2) == (hi_idx >> 2) ? whi : wlo) >> (8u * (idx & 3u))) & 255u;
                                acc |= (uint64_t)sh->huf_code[sy] << nb;
                                nb += sh->huf_len[sy];
                            }
                    }
                    {
                        uint32_t total;
                        const uint32_t off = zb_scan_excl(nb, &total);
                        zb_stage_begin(stg, running, total, zl);
                        if (nb)
                        {
                            const uint32_t bp = (running & 31u) + off;
                            const uint64_t v = acc << (bp & 31u); /* nb <= 44, shift <= 31: fits 75 bits -> three words */
                            zb_atomic_or(stg + (bp >> 5), (uint32_t)v);
                            if ((bp & 31u) + nb > 32u)
                                zb_atomic_or(stg + (bp >> 5) + 1u, (uint32_t)(v >> 32));
                            if ((bp & 31u) + nb > 64u)
                                zb_atomic_or(stg + (bp >> 5) + 2u, (uint32_t)(acc >> (64u - (bp & 31u))));
                        }
                        zb_stage_end(stg, sc->out, running, total, first_bit, zl);
                        running += total;
                    }
                }
                zb_stage_close(stg, sc->out, running, zl);
                ZB_SERIAL(zl) { zb_atomic_or(sc->out + (running >> 5), 1u << (running & 31u)); } /* end mark */
                base += (strbits[4u * u + st] + 1u + 7u) >> 3;
            }
        }
        else
        {
            /* raw literals, or the whole unit raw: bytes (the destination shares words with its neighbours: byte stores) */
            const uint8_t* src = ((srcmask >> u) & 1u) || lmode == 4u ? in->src + (size_t)u * ZB_UNIT : in->unit_lits + (uint64_t)u * ZB_UNIT;
            const uint32_t cnt = lmode == 4u ? (uint32_t)(sub[u] & 0x7FFFu) : n;
            zb_copy_bytes(out8 + sb->litpos[u], src, cnt, zl);
        }
    }

    ZB_MARK(7);
    /* ---- phase 7: sequence bit-streams, last sequence first: one sequence per lane and step.  The steps of ALL units form one list
     * and the records of the step after the current one are loaded before the current one is worked on: a step is a round trip to
     * memory, and with one wave per piece nothing else hides it (17 % of the kernel's wave time were these loads, one step at a
     * time). ---- */
    {
        uint32_t nu = 0, ndone = 0; /* the step whose records are on their way */
        uint64_t nq = 0, ntr = 0;
        uint32_t first_bit = 0, running = 0;
        while (nu < nunits && !(sh->useq_base[nu + 1u] > sh->useq_base[nu] && sb->lmode[nu] != 4u))
            ++nu;
        if (nu < nunits)
        {
            const uint32_t b0 = sh->useq_base[nu], ns = sh->useq_base[nu + 1u] - b0;
            if (zl < ns)
            {
                nq = sc->seqs[b0 + ns - 1u - zl];
                ntr = zl ? ((const uint64_t*)sc->sbits)[b0 + ns - 1u - zl] : 0u;
            }
        }
        while (nu < nunits)
        {
            const uint32_t u = nu, done = ndone;
            const uint32_t b0 = sh->useq_base[u], ns = sh->useq_base[u + 1u] - b0;
            const uint64_t q = nq, tr = ntr;
            /* the next step: of this unit, or the first of the next unit that has a sequence stream */
            ndone += ZB_LANES;
            if (ndone >= ns)
            {
                ndone = 0;
                ++nu;
                while (nu < nunits && !(sh->useq_base[nu + 1u] > sh->useq_base[nu] && sb->lmode[nu] != 4u))
                    ++nu;
            }
            nq = 0;
            ntr = 0;
            if (nu < nunits)
            {
                const uint32_t nb0 = sh->useq_base[nu], nns = sh->useq_base[nu + 1u] - nb0, nr = ndone + zl;
                if (nr < nns)
                {
                    nq = sc->seqs[nb0 + nns - 1u - nr];
                    ntr = nr ? ((const uint64_t*)sc->sbits)[nb0 + nns - 1u - nr] : 0u; /* (zero for a table in RLE mode) */
                }
            }
            if (done == 0u)
            {
                first_bit = sb->seqpos[u] * 8u;
                running = first_bit;
                zb_stage_open(stg, zl);
            }
            {
                const uint32_t r = done + zl;
                uint32_t bits = 0, llx = 0, mlx = 0, ofx = 0, oc = 0, lb = 0, mb = 0, so = 0, sm = 0, sl = 0;
                if (r < ns)
                {
                    llx = ZP_LLX(q);
                    mlx = ZP_MLX(q);
                    ofx = ZP_OFX(q);
                    oc = ZP_OC(q);
                    lb = ((const uint8_t*)sh->cursor[0])[ZP_LC(q)];
                    mb = ((const uint8_t*)sh->cursor[1])[ZP_MC(q)];
                    /* every sequence but the block's last one updates the states: OF, ML, LL (read back as LL, ML, OF) */
                    sl = (uint32_t)tr & 0xFFFFu;
                    so = (uint32_t)(tr >> 16) & 0xFFFFu;
                    sm = (uint32_t)(tr >> 32) & 0xFFFFu;
                    bits = (so >> 10) + (sm >> 10) + (sl >> 10) + lb + mb + oc;
                }
                {
                    uint32_t total;
                    const uint32_t off = zb_scan_excl(bits, &total);
                    zb_stage_begin(stg, running, total, zl);
                    if (bits)
                    {
                        ZbBits bw;
                        /* three puts instead of six: the transition bits of the three states (<= 27 bits), the two lengths' extra
                         * bits (a unit's lengths: <= 12 + 12), the offset's */
                        const uint32_t no = so >> 10, nm = sm >> 10, nl = sl >> 10;
                        zb_bits_open(&bw, stg, (running & 31u) + off);
                        zb_bits_put(&bw, (so & 1023u) | ((sm & 1023u) << no) | ((sl & 1023u) << (no + nm)), no + nm + nl);
                        zb_bits_put(&bw, llx | (mlx << lb), lb + mb);
                        zb_bits_put(&bw, ofx, oc);
                        zb_bits_close(&bw);
                    }
                    zb_stage_end(stg, sc->out, running, total, first_bit, zl);
                    running += total;
                }
            }
            if (done + ZB_LANES >= ns)
                zb_stage_close(stg, sc->out, running, zl);
        }
    }
    ZB_SYNC();

    ZB_MARK(8);
    /* ---- phase 8: final states (ML, OF, LL: read back as LL, OF, ML) and the end mark of every stream ---- */
    ZB_PAR_FOR(u, nunits)
    {
        if (sh->useq_base[u + 1u] > sh->useq_base[u] && sb->lmode[u] != 4u)
        {
            ZbBits bw;
            zb_bits_open(&bw, sc->out, sb->seqpos[u] * 8u + sb->seqbits[u]);
            if (sh->mode[ZT_ML] != 1u)
                zb_bits_put(&bw, sb->fstate[u][ZT_ML], sh->table_log[ZT_ML]);
            if (sh->mode[ZT_OF] != 1u)
                zb_bits_put(&bw, sb->fstate[u][ZT_OF], sh->table_log[ZT_OF]);
            if (sh->mode[ZT_LL] != 1u)
                zb_bits_put(&bw, sb->fstate[u][ZT_LL], sh->table_log[ZT_LL]);
            zb_bits_put(&bw, 1u, 1u);
            zb_bits_close(&bw);
        }
    }
    ZB_SYNC();
    ZB_MARK(10);
    return sh->v[ZV_OUT_SIZE];
}

#endif /* ZSTD_BLOCK_CORE_H */
