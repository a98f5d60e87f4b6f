/* zb_huffman.h -- part of zstd_block_core.h (included there, in this order; not a header of its own): Huffman code lengths (<= 11 bits) for the literals, the tree description. */
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

