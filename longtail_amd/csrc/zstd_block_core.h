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

#include "zstd/zb_fse.h" /* FSE -- normalisation, table description, encoding table, state step */
#include "zstd/zb_huffman.h" /* Huffman code lengths (<= 11 bits) for the literals, the tree description */
#include "zstd/zb_piece.h" /* literal reader, sequence tables, one Compressed_Block per 128 KiB piece (zb_encode_block) */
#include "zstd/zb_sub.h" /* the same piece as a run of sub-blocks, one zstd block per 4 KiB match-finder unit (zb_encode_piece_sub) */
#endif /* ZSTD_BLOCK_CORE_H */
