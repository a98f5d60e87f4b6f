/* zstd_model.c -- TEST INFRASTRUCTURE (see oracle/oracle.h): host build of the zstd block encoder that the GPU runs.
 *
 * longtail_amd/csrc/zstd_block_core.h is written once for two execution models; this file instantiates it with ONE
 * lane, which makes it a bit-exact, GPU-free model of k_zstd.hip's entropy stage.  Tests use it two ways:
 *   - here (no GPU): model frames are fed to the REFERENCE decoder (oracle/_ref, ZSTD_decompressDCtx behind
 *     ZStdCompressionAPI_Decompress, lib/zstd/longtail_zstd.c:144-177) on many inputs, which pins the format work;
 *   - on the GPU box: the match-finder output of the kernel is pushed through ltz_model_encode_block and must give
 *     the kernel's bytes exactly.
 * The sequences of ltz_model_compress come from a small greedy matcher below (any valid parse does; it is not a
 * model of the GPU match finder).  Nothing in the product links this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ZB_LANES 1u
#define ZB_FN static inline
#define ZB_SYNC() ((void)0)
static inline void zb_atomic_add(uint32_t* p, uint32_t v) { *p += v; }
static inline void zb_atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
static inline uint32_t zb_scan_excl(uint32_t v, uint32_t* total)
{
    *total = v;
    return 0;
}

static uint32_t g_ltz_dbg;
#define ZB_DBG g_ltz_dbg
#include "../longtail_amd/csrc/zstd_decode_core.h" /* includes zstd_block_core.h */
void ltz_model_debug(uint32_t flags) { g_ltz_dbg = flags; }
static int g_ltz_sub; /* 1: pieces are written as runs of sub-blocks (zb_encode_piece_sub) */
static uint32_t g_ltz_flags = 0; /* ZbInput.flags of the next encodes (the kernel's default: plain offsets; ZB_F_REPCODES = LTHIP_ZSTD_REP=1) */
void ltz_model_flags(uint32_t flags) { g_ltz_flags = flags; }
static uint16_t g_ltz_last_sub[ZB_MAX_UNITS];
void ltz_model_sub_blocks(int on) { g_ltz_sub = on; }
const uint16_t* ltz_model_last_sub(void) { return g_ltz_last_sub; }

/* 1: the sub-block encoder runs with 64 lanes (zstd_model_lanes.c: the kernel's all-lanes code on the host) instead of one */
static int g_ltz_lanes64;
void ltz_lanes_order(int descending);
void ltz_model_lanes64(int on) /* 0: one lane; 1: 64 lanes, run in ascending order between meeting points; 2: descending */
{
    g_ltz_lanes64 = on;
    ltz_lanes_order(on == 2);
}
uint32_t ltz_lanes_encode_piece_sub(const void* meta, const uint8_t* unit_lits, const uint64_t* unit_recs, uint32_t nunits, uint32_t raw_size,
                                    const uint8_t* src, uint32_t flags, uint8_t* out, uint16_t* sub);

#include "oracle.h"

/* src != NULL: units without a sequence have no literal buffer, their literals are src + u * 4096 (ZbInput.src) */
uint32_t ltz_model_encode_block_src(const void* meta, const uint8_t* unit_lits, const uint64_t* unit_recs, uint32_t nunits,
                                    uint32_t raw_size, const uint8_t* src, uint8_t* out)
{
    ZbInput in;
    ZbScratch sc;
    ZbShared* sh;
    uint32_t n;
    if (g_ltz_sub && g_ltz_lanes64)
        return ltz_lanes_encode_piece_sub(meta, unit_lits, unit_recs, nunits, raw_size, src, g_ltz_flags, out, g_ltz_last_sub);
    sh = (ZbShared*)calloc(1, sizeof(ZbShared));
    in.meta = (const ZbUnitMeta*)meta;
    in.unit_lits = unit_lits;
    in.unit_recs = unit_recs;
    in.nunits = nunits;
    in.raw_size = raw_size;
    in.src = src;
    in.flags = g_ltz_flags;
    sc.seqs = (uint64_t*)malloc(sizeof(uint64_t) * ZB_SEQ_MAX);
    sc.sbits = (uint16_t*)malloc(sizeof(uint16_t) * 4 * ZB_SEQ_MAX);
    sc.out = (uint32_t*)malloc(ZB_OUT_BYTES);
    memset(sc.out, 0xA5, ZB_OUT_BYTES); /* the encoders must not rely on a cleared output (the kernel's work area is reused) */
    n = g_ltz_sub ? zb_encode_piece_sub(&in, &sc, sh, 0, g_ltz_last_sub) : zb_encode_block(&in, &sc, sh, 0);
    if (n)
        memcpy(out, sc.out, n);
    free(sc.seqs);
    free(sc.sbits);
    free(sc.out);
    free(sh);
    return n;
}

uint32_t ltz_model_encode_block(const void* meta, const uint8_t* unit_lits, const uint64_t* unit_recs, uint32_t nunits,
                                uint32_t raw_size, uint8_t* out)
{
    return ltz_model_encode_block_src(meta, unit_lits, unit_recs, nunits, raw_size, NULL, out);
}

/* greedy matcher over one block: 4 KiB units, matches never cross a unit end, offsets < 65536 within the frame */
static void model_match_block(const uint8_t* base, size_t block_off, uint32_t size, uint32_t* table, ZbUnitMeta* meta,
                              uint8_t* unit_lits, uint64_t* unit_recs)
{
    const uint8_t* blk = base + block_off;
    const uint32_t nunits = (size + ZB_UNIT - 1u) / ZB_UNIT;
    for (uint32_t u = 0; u < nunits; ++u)
    {
        const uint32_t start = u * ZB_UNIT, end = start + ZB_UNIT < size ? start + ZB_UNIT : size;
        uint32_t p = start, anchor = start, nseq = 0, nlit = 0;
        uint8_t* lits = unit_lits + (size_t)u * ZB_UNIT;
        uint64_t* recs = unit_recs + (size_t)u * ZB_UNIT_SEQ_MAX;
        while (p + 4u <= end)
        {
            uint32_t v, h;
            size_t abs_p = block_off + p, cand;
            memcpy(&v, blk + p, 4);
            h = (v * 2654435761u) >> 18;
            cand = table[h];
            table[h] = (uint32_t)abs_p + 1u;
            if (cand && abs_p - (cand - 1u) < 65536u && abs_p - (cand - 1u) > 0 && memcmp(base + cand - 1u, blk + p, 4) == 0)
            {
                const uint8_t* m = base + cand - 1u;
                uint32_t len = 4;
                while (p + len < end && m[len] == blk[p + len])
                    ++len;
                memcpy(lits + nlit, blk + anchor, p - anchor);
                nlit += p - anchor;
                recs[nseq++] = ZB_REC(p - anchor, len, abs_p - (cand - 1u));
                p += len;
                anchor = p;
            }
            else
                ++p;
        }
        if (nseq)
            memcpy(lits + nlit, blk + anchor, end - anchor);
        else
            memset(lits, 0xA5, ZB_UNIT); /* like the GPU match finder: no copy for a unit without a match (ZbInput.src) */
        nlit += end - anchor;
        meta[u].nseq = nseq;
        meta[u].nlit = nlit;
        meta[u].tail = end - anchor;
        meta[u].uniform = 0;
    }
}

size_t ltz_model_bound(size_t n) { return n + (n >> 8) + 64 + 3 * (n / ZB_BLOCK_MAX + 1); }

/* One frame: magic | FHD 0xE0 | u64 content size | blocks {last:1,type:2,size:21} (same container as k_zstd.hip). */
int ltz_model_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n)
{
    size_t pos = 0, off = 0;
    uint32_t* table;
    ZbUnitMeta meta[ZB_MAX_UNITS];
    uint8_t* unit_lits;
    uint64_t* unit_recs;
    uint8_t* enc;
    if (cap < ltz_model_bound(n))
        return -1;
    table = (uint32_t*)calloc(1u << 14, sizeof(uint32_t));
    unit_lits = (uint8_t*)malloc(ZB_BLOCK_MAX);
    unit_recs = (uint64_t*)malloc(sizeof(uint64_t) * ZB_SEQ_MAX);
    enc = (uint8_t*)malloc(ZB_OUT_BYTES);
    dst[pos++] = 0x28;
    dst[pos++] = 0xB5;
    dst[pos++] = 0x2F;
    dst[pos++] = 0xFD;
    dst[pos++] = 0xE0;
    for (int i = 0; i < 8; ++i)
        dst[pos++] = (uint8_t)((uint64_t)n >> (8 * i));
    if (n == 0)
    {
        dst[pos++] = 1;
        dst[pos++] = 0;
        dst[pos++] = 0;
    }
    while (off < n)
    {
        const uint32_t size = n - off < ZB_BLOCK_MAX ? (uint32_t)(n - off) : ZB_BLOCK_MAX;
        const uint32_t last = off + size == n;
        uint32_t csize, rle = 1;
        for (uint32_t i = 1; i < size && rle; ++i)
            rle = src[off + i] == src[off];
        if (rle)
        {
            const uint32_t h = last | (1u << 1) | (size << 3);
            dst[pos++] = (uint8_t)h;
            dst[pos++] = (uint8_t)(h >> 8);
            dst[pos++] = (uint8_t)(h >> 16);
            dst[pos++] = src[off];
            off += size;
            continue;
        }
        model_match_block(src, off, size, table, meta, unit_lits, unit_recs);
        csize = ltz_model_encode_block_src(meta, unit_lits, unit_recs, (size + ZB_UNIT - 1u) / ZB_UNIT, size, src + off, enc);
        if (csize == 0xFFFFFFFFu) /* (64-lane execution: the lanes did not keep step, zstd_model_lanes.c) */
        {
            free(table);
            free(unit_lits);
            free(unit_recs);
            free(enc);
            return -2;
        }
        if (g_ltz_sub && csize)
        {
            /* the sub-blocks carry their own headers: only Last_Block is left to set */
            const uint32_t lastu = (size + ZB_UNIT - 1u) / ZB_UNIT - 1u;
            memcpy(dst + pos, enc, csize);
            if (last)
                dst[pos + csize - 3u - (g_ltz_last_sub[lastu] & 0x7FFFu)] |= 1u;
            pos += csize;
        }
        else
        {
            const uint32_t h = last | ((csize ? 2u : 0u) << 1) | ((csize ? csize : size) << 3);
            dst[pos++] = (uint8_t)h;
            dst[pos++] = (uint8_t)(h >> 8);
            dst[pos++] = (uint8_t)(h >> 16);
            memcpy(dst + pos, csize ? enc : src + off, csize ? csize : size);
            pos += csize ? csize : size;
        }
        off += size;
    }
    free(table);
    free(unit_lits);
    free(unit_recs);
    free(enc);
    *out_n = pos;
    return 0;
}

static uint32_t g_ltz_fail_line;
uint32_t ltz_model_fail_line(void) { return g_ltz_fail_line; }

/* Host instantiation of the GPU's zstd decoder (zstd_decode_core.h).  Returns 0 and *out_n, or -1 on malformed input. */
int ltz_model_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n)
{
    ZdShared* sh = (ZdShared*)calloc(1, sizeof(ZdShared));
    uint8_t* lits = (uint8_t*)malloc(ZD_LIT_MAX + 32);
    uint32_t r = ZD_ERROR;
    if (n <= 0xFFFFFFF0u && cap <= 0xFFFFFFF0u)
        r = zd_decode_payload(src, (uint32_t)n, dst, (uint32_t)cap, lits, sh, 0);
    g_ltz_fail_line = r == ZD_ERROR ? sh->v[ZDV_SRC] : 0;
    free(lits);
    free(sh);
    if (r == ZD_ERROR)
        return -1;
    *out_n = r;
    return 0;
}

/* Both FSE table builders of the decoder core on one normalised distribution (norm[0..maxsym], -1 = low probability): the serial
 * one into a, the all-lanes one (run here with one lane) into b; each table as size * {sym, nb, base lo, base hi}.  Returns
 * serial status | cooperative status << 1. */
int ltz_model_fse_tables(const int16_t* norm, uint32_t maxsym, uint32_t tl, uint8_t* a, uint8_t* b)
{
    ZdFse* t = (ZdFse*)calloc(2, sizeof(ZdFse));
    uint16_t next[64], cum[66], pre[ZD_FSE_PAR_MASK_WORDS];
    uint32_t masks[ZD_FSE_PAR_MASK_WORDS];
    const int ra = zd_build_fse(&t[0], norm, maxsym, tl, next);
    const int rb = zd_build_fse_par(&t[1], norm, maxsym, tl, cum, masks, pre, 0);
    for (int k = 0; k < 2; ++k)
    {
        uint8_t* o = k ? b : a;
        if ((k ? rb : ra) == 0)
            for (uint32_t u = 0; u < (1u << tl); ++u)
            {
                o[4 * u] = t[k].sym[u];
                o[4 * u + 1] = t[k].nb[u];
                o[4 * u + 2] = (uint8_t)t[k].base[u];
                o[4 * u + 3] = (uint8_t)(t[k].base[u] >> 8);
            }
    }
    free(t);
    return ra | rb << 1;
}
