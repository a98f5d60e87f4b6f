/* zb_fse.h -- part of zstd_block_core.h (included there, in this order; not a header of its own): FSE -- normalisation, table description, encoding table, state step. */
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

