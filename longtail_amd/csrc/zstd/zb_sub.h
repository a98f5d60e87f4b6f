/* zb_sub.h -- part of zstd_block_core.h (included there, in this order; not a header of its own): the same piece as a run of sub-blocks, one zstd block per 4 KiB match-finder unit (zb_encode_piece_sub). */

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

