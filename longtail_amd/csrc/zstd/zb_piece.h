/* zb_piece.h -- part of zstd_block_core.h (included there, in this order; not a header of its own): literal reader, sequence tables, one Compressed_Block per 128 KiB piece (zb_encode_block). */
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

