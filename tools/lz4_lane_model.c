/* lz4_lane_model.c -- design tool (not product, not oracle): CPU model of a LANE-SEQUENTIAL LZ4 parse for K5.
 *
 * Today's K5 probes 64 consecutive positions per step and then selects among the hits with a scalar walk (~400 wave
 * instructions per 64 positions on compressible data).  The alternative modelled here gives every lane of the wave its own
 * SUB-UNIT (unit / 64 bytes) and lets it run the CPU's greedy loop on it: probe, on a hit extend and jump, else step -- 64
 * independent parsers in lock step, one probe per lane per iteration.  This program measures what that does to the RATIO
 * (window group, private table, pre-seeding, sub-unit size, matches crossing sub-unit ends, insert policy) against the
 * reference's LZ4_compress_fast (oracle restatement) on the synthetic kinds of include/longtail_synth.h.
 *
 *   gcc -O2 -o /tmp/lz4_lane_model tools/lz4_lane_model.c oracle/lz4_oracle.c -Iinclude -Ioracle && /tmp/lz4_lane_model
 */
#include "../include/longtail_synth.h"
#include "../oracle/oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EMPTY 0xFFFFFFFFu

struct params
{
    uint32_t unit;      /* bytes per wave unit */
    uint32_t group;     /* units per window group */
    uint32_t tab;       /* table entries per wave */
    uint32_t lanes;     /* 64 */
    int cross;          /* 1: a lane's match may run past its sub-unit end (later lanes drop what it covers) */
    int seed_stride;    /* pre-seed every n-th position of the group history (0 = none) */
    int insert_in_match;/* 1: also insert position p + mlen - 2 after a match (lz4.c:1231) */
    int back;           /* 1: backward extension down to the lane's anchor */
    int history_groups; /* extra earlier groups visible as history (pre-seeded) */
    int trim;           /* 1: a covered sequence is trimmed to start at the cover instead of dropped */
    int ways;           /* 1 or 2 entries per bucket (2: newest + previous, one 32-bit word) */
    int policy;         /* same-step write conflicts: 0 highest lane wins, 1 lowest lane wins */
    int maxrec;         /* sequences a lane may record for its sub-unit */
    uint32_t near_bytes;/* history older than this is pre-seeded at far_stride instead of seed_stride (0 = all at seed_stride) */
    int far_stride;     /* 0 = the far history is not inserted at all */
    int accel;          /* a lane steps 1 + (consecutive misses >> accel) bytes after a miss (0 = always 1) */
    int raw;            /* 1: the table is read again AFTER the step's writes: the surviving entry of a write conflict is a candidate at once */
    int best2;          /* 1: when the private AND the shared candidate verify, the one with the longer match is taken (a deeper search) */
    uint32_t shared;    /* > 0: slots of a table shared by the group's waves that keeps the EARLIEST aligned occurrence of a key (looked at when the private table has nothing) */
};

static uint32_t rd32(const uint8_t* p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static uint32_t hidx(uint32_t v, uint32_t tab) { return (uint32_t)(((uint64_t)(v * 2654435761u) * tab) >> 32); }
static uint32_t lenbytes(uint32_t l) { return l >= 15 ? (l - 15) / 255 + 1 : 0; }

struct seq
{
    uint32_t start, len, off;
};

/* returns the LZ4 block size of `n` bytes at `src` compressed by the modelled parse */
static uint64_t g_far;
static uint64_t model_block(const uint8_t* src, uint32_t n, const struct params* P, uint64_t* nseq_out, uint64_t* iters_out)
{
    const uint32_t U = P->unit, G = P->group, L = P->lanes, sub = U / L;
    const uint32_t gbytes = U * G;
    struct seq* seqs = (struct seq*)malloc(sizeof(struct seq) * (n / 4 + 16));
    uint64_t ns = 0, iters = 0;
    uint32_t* tab = (uint32_t*)malloc(sizeof(uint32_t) * P->tab * 2);
    const uint32_t NB = P->tab / (uint32_t)P->ways; /* buckets */
    uint32_t* shared = (uint32_t*)malloc(4 * (P->shared + 1));
    uint32_t* p = (uint32_t*)malloc(4 * L);
    uint32_t* lane_end = (uint32_t*)malloc(4 * L);
    uint32_t* lane_anchor = (uint32_t*)malloc(4 * L);
    struct seq* lane_seqs = (struct seq*)malloc(sizeof(struct seq) * L * (sub / 4 + 2));
    uint32_t* lane_ns = (uint32_t*)malloc(4 * L);
    uint32_t* lane_miss = (uint32_t*)malloc(4 * L);
    for (uint32_t g0 = 0; g0 < n; g0 += gbytes)
    {
        const uint32_t glen = n - g0 < gbytes ? n - g0 : gbytes;
        const uint32_t hist0 = g0 >= (uint32_t)P->history_groups * gbytes ? g0 - (uint32_t)P->history_groups * gbytes : 0;
        if (P->shared)
        {
            for (uint32_t i = 0; i < P->shared; ++i)
                shared[i] = EMPTY;
            for (uint32_t q = g0; q + 4 <= g0 + glen; q += 4)
            {
                const uint32_t hs = hidx(rd32(src + q) ^ 0x9E3779B9u, P->shared);
                if (shared[hs] == EMPTY)
                    shared[hs] = q;
            }
        }
        for (uint32_t w = 0; w * U < glen; ++w)
        {
            const uint32_t ustart = g0 + w * U;
            const uint32_t ulen = glen - w * U < U ? glen - w * U : U;
            /* block-end rules: a match starts <= n - 12 and ends <= n - 5 */
            const int64_t start_limit = (int64_t)(ustart + ulen) - 4 < (int64_t)n - 12 ? (int64_t)(ustart + ulen) - 4 : (int64_t)n - 12;
            const uint32_t end_limit = ustart + ulen < n - 5 ? ustart + ulen : (n >= 5 ? n - 5 : 0);
            for (uint32_t i = 0; i < P->tab * 2; ++i)
                tab[i] = EMPTY;
#define TAB_INSERT(H, POS)                                   \
    do                                                       \
    {                                                        \
        if (P->ways == 2)                                    \
            tab[2 * (H) + 1] = tab[2 * (H)];                 \
        tab[2 * (H)] = (POS);                                \
    } while (0)
            if (P->seed_stride)
            {
                uint32_t q = hist0;
                if (P->near_bytes && ustart - hist0 > P->near_bytes)
                {
                    const uint32_t near0 = ustart - P->near_bytes;
                    if (P->far_stride)
                        for (; q < near0; q += (uint32_t)P->far_stride)
                            TAB_INSERT(hidx(rd32(src + q), NB), q);
                    q = near0;
                }
                for (; q + 4 <= ustart; q += (uint32_t)P->seed_stride)
                    TAB_INSERT(hidx(rd32(src + q), NB), q);
            }
            uint32_t active = 0;
            for (uint32_t l = 0; l < L; ++l)
            {
                p[l] = ustart + l * sub;
                lane_end[l] = ustart + (l + 1) * sub < ustart + ulen ? ustart + (l + 1) * sub : ustart + ulen;
                lane_anchor[l] = p[l];
                lane_ns[l] = 0;
                lane_miss[l] = 0;
                if (p[l] < ustart + ulen)
                    ++active;
            }
            for (;;)
            {
                int any = 0;
                /* all lanes read the table, then all write (ascending lane order: the highest lane wins a conflict) */
                static uint32_t cand[256], cand2[256], hh[256], vv[256];
                for (uint32_t l = 0; l < L; ++l)
                {
                    cand[l] = EMPTY;
                    hh[l] = EMPTY;
                    if (p[l] < lane_end[l] && (int64_t)p[l] <= start_limit && lane_ns[l] < (uint32_t)P->maxrec)
                    {
                        vv[l] = rd32(src + p[l]);
                        hh[l] = hidx(vv[l], NB);
                        cand[l] = tab[2 * hh[l]];
                        cand2[l] = P->ways == 2 ? tab[2 * hh[l] + 1] : EMPTY;
                        any = 1;
                    }
                }
                if (!any)
                    break;
                ++iters;
                /* one store instruction: every lane writes (its position, the newest entry it READ) -- on a conflict one lane's
                 * whole word survives */
                if (P->policy == 0)
                {
                    for (uint32_t l = 0; l < L; ++l)
                        if (hh[l] != EMPTY)
                        {
                            tab[2 * hh[l] + 1] = P->ways == 2 ? cand[l] : EMPTY;
                            tab[2 * hh[l]] = p[l];
                        }
                }
                else
                {
                    for (uint32_t l = L; l-- > 0;)
                        if (hh[l] != EMPTY)
                        {
                            tab[2 * hh[l] + 1] = P->ways == 2 ? cand[l] : EMPTY;
                            tab[2 * hh[l]] = p[l];
                        }
                }
                for (uint32_t l = 0; l < L; ++l)
                {
                    if (hh[l] == EMPTY)
                        continue;
                    uint32_t c = cand[l];
                    if (P->raw)
                    {
                        const uint32_t fresh = tab[2 * hh[l]];
                        if (fresh != EMPTY && fresh < p[l])
                            c = fresh;
                    }
                    if (!(c != EMPTY && c < p[l] && p[l] - c <= 65535 && rd32(src + c) == vv[l]))
                        c = cand2[l];
                    if (P->shared && !(c != EMPTY && c < p[l] && p[l] - c <= 65535 && rd32(src + c) == vv[l]))
                        c = shared[hidx(vv[l] ^ 0x9E3779B9u, P->shared)];
                    else if (P->shared && P->best2)
                    {
                        const uint32_t c2 = shared[hidx(vv[l] ^ 0x9E3779B9u, P->shared)];
                        if (c2 != EMPTY && c2 < p[l] && c2 != c && rd32(src + c2) == vv[l])
                        {
                            uint32_t m1 = 4, m2 = 4;
                            while (p[l] + m1 < end_limit && src[p[l] + m1] == src[c + m1])
                                ++m1;
                            while (p[l] + m2 < end_limit && src[p[l] + m2] == src[c2 + m2])
                                ++m2;
                            if (m2 > m1)
                                c = c2;
                        }
                    }
                    if (c != EMPTY && c < p[l] && p[l] - c <= 65535 && rd32(src + c) == vv[l])
                    {
                        uint32_t s = p[l], cs = c, ml = 4;
                        const uint32_t lim = P->cross ? end_limit : (lane_end[l] < end_limit ? lane_end[l] : end_limit);
                        while (s + ml < lim && src[s + ml] == src[cs + ml])
                            ++ml;
                        if (s + ml > lim)
                            ml = lim > s ? lim - s : 0;
                        if (ml < 4)
                        {
                            p[l] += 1;
                            continue;
                        }
                        lane_miss[l] = 0;
                        if (P->back)
                            while (s > lane_anchor[l] && cs > 0 && src[s - 1] == src[cs - 1])
                            {
                                --s;
                                --cs;
                                ++ml;
                            }
                        struct seq* q = &lane_seqs[l * (sub / 4 + 2) + lane_ns[l]++];
                        q->start = s;
                        q->len = ml;
                        q->off = s - cs;
                        p[l] = s + ml;
                        lane_anchor[l] = p[l];
                        if (P->insert_in_match && p[l] >= 2 && p[l] - 2 + 4 <= n)
                            TAB_INSERT(hidx(rd32(src + p[l] - 2), NB), p[l] - 2);
                    }
                    else
                    {
                        p[l] += 1 + (P->accel ? (lane_miss[l] >> P->accel) : 0);
                        ++lane_miss[l];
                    }
                }
            }
            /* cover: a lane drops (or trims) what earlier lanes' matches already cover */
            uint32_t cover = ustart;
            for (uint32_t l = 0; l < L; ++l)
            {
                for (uint32_t k = 0; k < lane_ns[l]; ++k)
                {
                    struct seq q = lane_seqs[l * (sub / 4 + 2) + k];
                    if (q.start < cover)
                    {
                        if (!P->trim || q.start + q.len < cover + 4)
                            continue;
                        const uint32_t cut = cover - q.start;
                        q.start += cut;
                        q.len -= cut;
                    }
                    seqs[ns++] = q;
                    cover = q.start + q.len;
                }
            }
        }
    }
    /* LZ4 size of the whole block as ONE stream (what the stitch produces) */
    uint64_t out = 0;
    uint32_t anchor = 0;
    for (uint64_t i = 0; i < ns; ++i)
    {
        const uint32_t lit = seqs[i].start - anchor;
        out += 1 + lenbytes(lit) + lit + 2 + lenbytes(seqs[i].len - 4);
        if (seqs[i].off > 6400)
            ++g_far;
        anchor = seqs[i].start + seqs[i].len;
    }
    out += 1 + lenbytes(n - anchor) + (n - anchor);
    *nseq_out += ns;
    *iters_out += iters;
    free(seqs);
    free(tab);
    free(shared);
    free(p);
    free(lane_end);
    free(lane_anchor);
    free(lane_seqs);
    free(lane_ns);
    free(lane_miss);
    return out;
}

int main(int argc, char** argv)
{
    const uint32_t block = 8u << 20;
    const int nblocks = argc > 1 ? atoi(argv[1]) : 2;
    const int kinds[] = {1, 11, 12, 13};
    const char* names[] = {"mixed", "records", "tokens", "lines"};
    uint8_t* buf = (uint8_t*)malloc(block);
    uint8_t* dst = (uint8_t*)malloc(lto_lz4_bound(block));
    struct params variants[] = {
        /* unit group tab lanes cross seed inm back hist trim ways policy maxrec near far accel raw best2 shared */
        {4096, 16, 2560, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 16384, 16, 2, 1, 0, 0},
        {4096, 16, 1536, 64, 1, 0, 0, 1, 0, 0, 1, 1, 8, 0, 0, 2, 1, 0, 8192},
        {4096, 16, 1536, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 8192},
        {4096, 16, 1536, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 1, 8192},
        {4096, 16, 1536, 64, 1, 4, 0, 1, 0, 0, 2, 1, 8, 4096, 0, 2, 1, 1, 8192},
        /* round 4: geometries that put 32 waves on a CU (two workgroups of 72 KiB) */
        {2048, 16, 768, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 2048, 0, 2, 1, 0, 4096},
        {2048, 16, 768, 64, 1, 4, 0, 1, 0, 0, 1, 1, 4, 2048, 0, 2, 1, 0, 4096},
        {2048, 16, 1024, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 2048, 0, 2, 1, 0, 4096},
        {4096, 8, 1536, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 4096},
        {2048, 16, 768, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 2048, 0, 2, 1, 0, 8192},
        {2048, 32, 768, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 2048, 0, 2, 1, 0, 8192},
        {4096, 12, 512, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 4096},
        {4096, 12, 768, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 4096},
        {4096, 12, 1536, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 8192},
        {4096, 16, 512, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 8192},
        {4096, 16, 512, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 4096},
        {4096, 16, 256, 64, 1, 4, 0, 1, 0, 0, 1, 1, 8, 4096, 0, 2, 1, 0, 2048},
    };
    printf("%-58s", "variant (unit group tab lanes cross seed inm back hist trim)");
    for (int k = 0; k < 4; ++k)
        printf(" %8s", names[k]);
    printf("   iters/unit (mixed)\n");
    printf("%-58s", "reference LZ4_compress_fast(acc 1)");
    for (int k = 0; k < 4; ++k)
    {
        uint64_t in = 0, out = 0;
        for (int b = 0; b < nblocks; ++b)
        {
            for (uint64_t w = 0; w < block / 8; ++w)
            {
                const uint64_t x = lt_synth_word(1000 + b, w, kinds[k]);
                memcpy(buf + 8 * w, &x, 8);
            }
            out += (uint64_t)lto_lz4_compress(buf, (int)block, dst, (int)lto_lz4_bound(block));
            in += block;
        }
        printf(" %8.4f", (double)in / (double)out);
    }
    printf("\n");
    for (size_t v = 0; v < sizeof variants / sizeof variants[0]; ++v)
    {
        const struct params* P = &variants[v];
        char label[128];
        snprintf(label, sizeof label, "%5u %3u %5u %3u   %d    %d    %d    %d    %d    %d  w%d p%d", P->unit, P->group, P->tab, P->lanes, P->cross, P->seed_stride,
                 P->insert_in_match, P->back, P->history_groups, P->trim, P->ways, P->policy);
        snprintf(label + strlen(label), sizeof label - strlen(label), " m%d n%u f%d a%d s%u", P->maxrec, P->near_bytes, P->far_stride, P->accel, P->shared);
        printf("%-58s", label);
        double it_mixed = 0;
        for (int k = 0; k < 4; ++k)
        {
            uint64_t in = 0, out = 0, nseq = 0, iters = 0;
            g_far = 0;
            for (int b = 0; b < nblocks; ++b)
            {
                for (uint64_t w = 0; w < block / 8; ++w)
                {
                    const uint64_t x = lt_synth_word(1000 + b, w, kinds[k]);
                    memcpy(buf + 8 * w, &x, 8);
                }
                out += model_block(buf, block, P, &nseq, &iters);
                in += block;
            }
            printf(" %8.4f", (double)in / (double)out);
            if (k == 0)
                it_mixed = (double)iters / ((double)in / P->unit);
            printf("(%4.1f,%2.0f%%)", (double)iters / ((double)in / P->unit), 100.0 * (double)g_far / (double)(nseq ? nseq : 1));
        }
        printf("\n");
        (void)it_mixed;
    }
    return 0;
}
