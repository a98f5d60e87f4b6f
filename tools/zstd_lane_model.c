/* zstd_lane_model.c -- design tool (not product, not oracle): what the CANDIDATES of the lane parser cost in offset bits.
 *
 * tools/zstd_parse_stats.py (profiles/r05_zstd_parse_stats.txt) says where the zstd frames of word-soup text lose against the
 * reference encoder: two thirds of the frame are sequences, 19 bits each, 13.6 of them the offset -- the median offset is 2.3 KiB where
 * the nearest occurrence is a word or two back.  The private table keeps, of the positions written in one step, the LOWEST (the lanes
 * of a wave stand 64 bytes apart and a candidate must lie below every reader), and the shared table the EARLIEST occurrence in the 64 KiB
 * group: both are far away by construction.  This model restates the parse (64 lanes in lock step over the 64-byte sub-units of a 4 KiB
 * unit, private table of 1280 entries pre-seeded with every fourth position of the 4 KiB before the unit, shared table of the group's
 * earliest aligned occurrences, four one-byte steps after a hit and then aligned dwords, backward extension to the anchor, at most 8
 * records per lane, later lanes drop what earlier lanes cover) and prices what it finds the way the Python tool prices the kernel's
 * output: literals by their order-0 entropy, a sequence by the entropies of its three codes plus extra bits, an offset that repeats
 * one of the last three as a repeat code.  Variants change where candidates come from.
 *
 *   gcc -O2 -o /tmp/zstd_lane_model tools/zstd_lane_model.c -Iinclude -lm && /tmp/zstd_lane_model
 */
#include "../include/longtail_synth.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EMPTY 0xFFFFFFFFu
enum { UNIT = 4096, LANES = 64, SUB = 64, GROUP = 65536, TAB = 1280, SHARED = 8192, MAXREC = 8 };

struct variant
{
    const char* name;
    int own;      /* n > 0: every lane has a table of n entries of its OWN positions (asked first) */
    int nearest;  /* 1: the candidate is the nearest earlier occurrence in the group (what a serial parser's table would hold) */
    int highest;  /* 1: the highest lane wins a same-step conflict */
    int rep;      /* 1: the lane's last offset is tried first */
    int prev;     /* n > 0: own table also pre-seeded with the n bytes before the sub-unit (positions of the lane below) */
    int nearwin;  /* 1: when both the private and another candidate verify, the NEARER one is taken unless the other is 4+ bytes longer */
};

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t hidx(uint32_t v, uint32_t tab) { return (uint32_t)(((uint64_t)(v * 2654435761u) * tab) >> 32); }
struct seq { uint32_t start, len, off; };

static double entropy_bits(const uint64_t* c, int n)
{
    double tot = 0, h = 0;
    for (int i = 0; i < n; ++i) tot += (double)c[i];
    for (int i = 0; i < n; ++i) if (c[i]) h += (double)c[i] * log2(tot / (double)c[i]);
    return h;
}
static int hb(uint32_t v) { return 31 - __builtin_clz(v | 1u); }

static uint32_t mlen_at(const uint8_t* src, uint32_t p, uint32_t c, uint32_t lim)
{
    uint32_t m = 0;
    while (p + m < lim && src[p + m] == src[c + m]) ++m;
    return m;
}

/* returns priced bytes; fills stats */
static double model_block(const uint8_t* src, uint32_t n, const struct variant* V, double* seq_bits_each, double* off_bits_each, uint64_t* nseq_out, uint64_t* nlit_out)
{
    struct seq* seqs = (struct seq*)malloc(sizeof(struct seq) * (n / 4 + 16));
    uint64_t ns = 0;
    static uint32_t tab[TAB], shared[SHARED], own[LANES][64], p[LANES], lend[LANES], anchor[LANES], nrec[LANES], nmiss[LANES], lastoff[LANES];
    static struct seq lseq[LANES][MAXREC];
    uint32_t* nearest_tab = (uint32_t*)malloc(4u << 16);
    for (uint32_t g0 = 0; g0 < n; g0 += GROUP)
    {
        const uint32_t glen = n - g0 < GROUP ? n - g0 : GROUP;
        for (uint32_t i = 0; i < SHARED; ++i) shared[i] = EMPTY;
        for (uint32_t q = g0; q + 4 <= g0 + glen; q += 4)
        {
            const uint32_t hs = hidx(rd32(src + q) ^ 0x9E3779B9u, SHARED);
            if (shared[hs] == EMPTY) shared[hs] = q;
        }
        for (uint32_t w = 0; w * UNIT < glen; ++w)
        {
            const uint32_t ustart = g0 + w * UNIT, ulen = glen - w * UNIT < UNIT ? glen - w * UNIT : UNIT;
            const int64_t start_limit = (int64_t)(ustart + ulen) - 4;
            const uint32_t end_limit = ustart + ulen;
            for (uint32_t i = 0; i < TAB; ++i) tab[i] = EMPTY;
            for (uint32_t q = ustart >= UNIT && ustart - UNIT >= g0 ? ustart - UNIT : g0; q + 4 <= ustart; q += 4)
                tab[hidx(rd32(src + q), TAB)] = q;
            for (uint32_t l = 0; l < LANES; ++l)
            {
                p[l] = ustart + l * SUB;
                lend[l] = p[l] + SUB < ustart + ulen ? p[l] + SUB : ustart + ulen;
                anchor[l] = p[l];
                nrec[l] = nmiss[l] = 0;
                lastoff[l] = 0;
                for (int i = 0; i < 64; ++i) own[l][i] = EMPTY;
                if (V->own && V->prev)
                    for (uint32_t q = p[l] >= g0 + (uint32_t)V->prev ? p[l] - (uint32_t)V->prev : g0; q + 4 <= p[l] && q < p[l]; ++q)
                        own[l][hidx(rd32(src + q), (uint32_t)V->own)] = q;
            }
            for (;;)
            {
                int any = 0;
                static uint32_t cand[LANES], hh[LANES], vv[LANES];
                for (uint32_t l = 0; l < LANES; ++l)
                {
                    hh[l] = EMPTY;
                    if (p[l] < lend[l] && (int64_t)p[l] <= start_limit && nrec[l] < MAXREC)
                    {
                        vv[l] = rd32(src + p[l]);
                        hh[l] = hidx(vv[l], TAB);
                        cand[l] = tab[hh[l]];
                        any = 1;
                    }
                }
                if (!any) break;
                if (V->highest) { for (uint32_t l = 0; l < LANES; ++l) if (hh[l] != EMPTY) tab[hh[l]] = p[l]; }
                else { for (uint32_t l = LANES; l-- > 0;) if (hh[l] != EMPTY) tab[hh[l]] = p[l]; }
                for (uint32_t l = 0; l < LANES; ++l)
                {
                    if (hh[l] == EMPTY) continue;
                    const uint32_t P = p[l], v = vv[l];
                    uint32_t c = EMPTY;
#define OK(C) ((C) != EMPTY && (C) < P && (C) >= g0 && rd32(src + (C)) == v)
                    if (V->nearest)
                    {
                        for (uint32_t q = P; q-- > g0;)
                            if (rd32(src + q) == v) { c = q; break; }
                    }
                    else
                    {
                        uint32_t c_near = EMPTY;
                        if (V->rep && lastoff[l] && P >= g0 + lastoff[l] && OK(P - lastoff[l])) c_near = P - lastoff[l];
                        if (c_near == EMPTY && V->own) { const uint32_t o = own[l][hidx(v, (uint32_t)V->own)]; if (OK(o)) c_near = o; }
                        uint32_t c_far = EMPTY;
                        if (OK(cand[l])) c_far = cand[l];
                        else { const uint32_t s = shared[hidx(v ^ 0x9E3779B9u, SHARED)]; if (OK(s)) c_far = s; }
                        if (c_near != EMPTY && c_far != EMPTY && V->nearwin)
                            c = mlen_at(src, P, c_far, end_limit) >= mlen_at(src, P, c_near, end_limit) + 4u ? c_far : c_near;
                        else
                            c = c_near != EMPTY ? c_near : c_far;
                    }
                    if (V->own) own[l][hidx(v, (uint32_t)V->own)] = P;
                    if (c != EMPTY)
                    {
                        uint32_t s = P, cs = c, ml = mlen_at(src, P, c, end_limit);
                        while (s > anchor[l] && cs > g0 && src[s - 1] == src[cs - 1] && P - s < 8) { --s; --cs; ++ml; }
                        struct seq* q = &lseq[l][nrec[l]++];
                        q->start = s; q->len = ml; q->off = s - cs;
                        lastoff[l] = q->off;
                        p[l] = s + ml;
                        anchor[l] = p[l];
                        nmiss[l] = 0;
                    }
                    else
                    {
                        p[l] = nmiss[l] < 4 ? P + 1 : (P | 3u) + 1u;
                        ++nmiss[l];
                    }
                }
            }
            uint32_t cover = ustart;
            for (uint32_t l = 0; l < LANES; ++l)
                for (uint32_t k = 0; k < nrec[l]; ++k)
                {
                    const struct seq q = lseq[l][k];
                    if (q.start < cover) continue;
                    seqs[ns++] = q;
                    cover = q.start + q.len;
                }
        }
    }
    /* price */
    uint64_t lit_hist[256] = {0}, oc[40] = {0}, lc[64] = {0}, mc[64] = {0};
    double extra = 0, off_extra = 0;
    uint64_t nlit = 0;
    uint32_t anchor0 = 0, rep[3] = {1, 4, 8};
    for (uint64_t i = 0; i < ns; ++i)
    {
        const uint32_t lit = seqs[i].start - anchor0;
        for (uint32_t b = 0; b < lit; ++b) ++lit_hist[src[anchor0 + b]];
        nlit += lit;
        const uint32_t llc = lit < 16 ? lit : 16 + hb(lit), mlv = seqs[i].len - 3, mlc = mlv < 32 ? mlv : 32 + hb(mlv);
        ++lc[llc < 63 ? llc : 63];
        ++mc[mlc < 63 ? mlc : 63];
        extra += (lit >= 16 ? hb(lit) : 0) + (mlv >= 32 ? hb(mlv) : 0);
        const uint32_t off = seqs[i].off;
        /* sub-block layout: the repeat history starts anew in every 4 KiB unit */
        if ((seqs[i].start / UNIT) != (anchor0 ? (anchor0 - 1) / UNIT : 0xFFFFFFFFu) && i) { rep[0] = 1; rep[1] = 4; rep[2] = 8; }
        int r = off == rep[0] ? 0 : off == rep[1] ? 1 : off == rep[2] ? 2 : -1;
        if (lit == 0 && r >= 0) r = -1; /* (the shifted meaning with no literals: not modelled, priced as a new offset) */
        if (r >= 0)
        {
            ++oc[r];
            if (r == 1) { const uint32_t t = rep[0]; rep[0] = rep[1]; rep[1] = t; }
            else if (r == 2) { const uint32_t t = rep[2]; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = t; }
        }
        else
        {
            const int code = hb(off + 3);
            ++oc[code < 39 ? code : 39];
            off_extra += code;
            rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off;
        }
        anchor0 = seqs[i].start + seqs[i].len;
    }
    for (uint32_t b = anchor0; b < n; ++b) ++lit_hist[src[b]];
    nlit += n - anchor0;
    const double lit_bits = entropy_bits(lit_hist, 256), off_bits = entropy_bits(oc, 40) + off_extra;
    const double seq_bits = off_bits + entropy_bits(lc, 64) + entropy_bits(mc, 64) + extra;
    *seq_bits_each = ns ? seq_bits / (double)ns : 0;
    *off_bits_each = ns ? off_bits / (double)ns : 0;
    *nseq_out = ns;
    *nlit_out = nlit;
    free(seqs);
    free(nearest_tab);
    return (lit_bits + seq_bits) / 8.0;
}

static uint64_t g_s = 88172645463325252ull;
static double urand(void) { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (double)(g_s >> 11) / 9007199254740992.0; }

int main(int argc, char** argv)
{
    const uint32_t block = argc > 1 ? (uint32_t)atoi(argv[1]) << 20 : 2u << 20;
    uint8_t* buf = (uint8_t*)malloc(block + 64);
    const struct variant variants[] = {
        {"as built (lowest lane wins, earliest shared)", 0, 0, 0, 0, 0, 0},
        {"highest lane wins", 0, 0, 1, 0, 0, 0},
        {"+ the lane's last offset first", 0, 0, 0, 1, 0, 0},
        {"+ own table 16", 16, 0, 0, 0, 0, 0},
        {"+ own table 16, last offset", 16, 0, 0, 1, 0, 0},
        {"+ own table 16 seeded with 64 B below", 16, 0, 0, 0, 64, 0},
        {"+ own table 32 seeded with 128 B below, last offset", 32, 0, 0, 1, 128, 0},
        {"+ own table 32 seeded 128 B, last offset, nearer wins", 32, 0, 0, 1, 128, 1},
        {"nearest occurrence in the group (serial table)", 0, 1, 0, 0, 0, 0},
    };
    const char* names[] = {"text", "tokens", "records", "mixed"};
    const int kinds[] = {-1, 12, 11, 1};
    for (int k = 0; k < 4; ++k)
    {
        if (kinds[k] < 0)
        {
            /* word soup as tools/zstd_ratio_table.py makes it: 4096 words of 3..11 letters, index = min(floor(pareto(1.1)), 4095) */
            static char words[4096][12];
            static int wl[4096];
            for (int i = 0; i < 4096; ++i)
            {
                wl[i] = 3 + (int)(urand() * 9);
                for (int j = 0; j < wl[i]; ++j) words[i][j] = (char)('a' + (int)(urand() * 26));
            }
            uint32_t at = 0;
            while (at < block)
            {
                const double x = pow(1.0 - urand(), -1.0 / 1.1) - 1.0;
                const int i = x >= 4095.0 ? 4095 : (int)x;
                for (int j = 0; j < wl[i] && at < block; ++j) buf[at++] = (uint8_t)words[i][j];
                if (at < block) buf[at++] = ' ';
            }
        }
        else
            for (uint64_t w = 0; w < block / 8; ++w)
            {
                const uint64_t x = lt_synth_word(1000, w, kinds[k]);
                memcpy(buf + 8 * w, &x, 8);
            }
        printf("%s (%u MiB)\n", names[k], block >> 20);
        for (size_t v = 0; v < sizeof variants / sizeof variants[0]; ++v)
        {
            double sb, ob;
            uint64_t ns, nl;
            const double bytes = model_block(buf, block, &variants[v], &sb, &ob, &ns, &nl);
            printf("  %-56s priced ratio %6.3f  seqs/unit %6.1f  lits/unit %7.1f  bits/seq %5.2f (offset %5.2f)\n", variants[v].name, (double)block / bytes,
                   (double)ns / (block / 4096.0), (double)nl / (block / 4096.0), sb, ob);
        }
    }
    return 0;
}
