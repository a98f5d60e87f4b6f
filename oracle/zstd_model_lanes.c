/* zstd_model_lanes.c -- TEST INFRASTRUCTURE (see oracle/oracle.h): the zstd entropy stage run with SIXTY-FOUR lanes on the host.
 *
 * longtail_amd/csrc/zstd_block_core.h is written once for two execution models.  zstd_model.c instantiates it with ONE lane; the
 * kernel (k_zstd.hip) with 64, and since round 4 the 64-lane build takes code of its own where a wave can do better than a lane
 * (the "#if ZB_LANES > 1" sections: zb_normalize_par, zb_build_enc_table_par, zb_huffman_build_par, the table builds of the
 * sub-block layout) -- code the one-lane model never runs.  This file instantiates the header with ZB_LANES = 64 for the host: 64
 * FIBERS (ucontext) in one thread, each a lane with its own `zl`, that meet where the wave meets -- ZB_SYNC / ZB_SYNC_LDS and the
 * wave collectives (zb_scan_excl, zb_ballot, zb_shfl, zb_reduce_max) are barriers over all 64.  Between two such points a lane runs
 * alone, the lanes one after the other (ascending or descending: ltz_lanes_order).  That is a WEAKER model than the wave's lock step
 * -- there a lane's read at statement k precedes another lane's write at statement k + 1 -- so the shared code is held to the
 * stricter contract "correct under any order of the lanes between two meeting points": every value a lane reads and another lane
 * writes in the same stretch needs a ZB_SYNC_LDS between them (round 5 found one such place, the depth test in front of the
 * Huffman length limiter, and gave it its fence).  Under that contract the bytes must be the one-lane model's (and the kernel's).  tests/test_zstd_model.py compares the two on the CPU: a divergence of the all-lanes code no longer
 * needs a GPU to show.  A lane that leaves the function while others wait, or lanes that disagree about the number of meeting
 * points, is a defect of the shared code: the scheduler reports it (return value 0xFFFFFFFF) instead of hanging.
 * Nothing in the product links this file.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#define LANES 64u
static ucontext_t g_main, g_ctx[LANES];
static char* g_stacks;
static uint32_t g_lane;     /* the lane that is running */
static uint32_t g_gen;      /* meeting points passed */
static uint32_t g_arrived;  /* lanes waiting at the current one */
static uint32_t g_finished; /* lanes that have left the function */
static int g_broken;        /* a lane left while others waited */
static uint32_t g_x[LANES];
static int g_descending;    /* the order in which the lanes run between two meeting points: 0 .. 63, or 63 .. 0 */
void ltz_lanes_order(int descending) { g_descending = descending; }

static uint32_t g_site, g_bad_site[2]; /* where the lanes of the current meeting stand (source line); the first disagreement */
uint32_t ltz_lanes_bad_site(int k) { return g_bad_site[k & 1]; }

static void lanes_meet_at(uint32_t site)
{
    const uint32_t gen = g_gen;
    if (g_arrived == 0u)
        g_site = site;
    else if (g_site != site && !g_broken)
    {
        g_broken = 1; /* the lanes are not at the same point of the program: a collective inside divergent control flow */
        g_bad_site[0] = g_site;
        g_bad_site[1] = site;
    }
    if (g_finished && !g_broken)
    {
        g_broken = 1; /* somebody will never arrive */
        g_bad_site[0] = site;
        g_bad_site[1] = 0xFFFFFFFFu;
    }
    if (!g_broken && ++g_arrived == LANES)
    {
        g_arrived = 0;
        ++g_gen;
        return;
    }
    while (g_gen == gen || g_broken) /* (once the lanes are out of step nobody runs on: the scheduler gives up) */
        swapcontext(&g_ctx[g_lane], &g_main);
}

#define ZB_LANES 64u
#define ZB_FN static
#define lanes_meet() lanes_meet_at(__LINE__)
#define ZB_SYNC() lanes_meet_at(__LINE__)
#define ZB_SYNC_LDS() lanes_meet_at(__LINE__)
static void zb_atomic_add(uint32_t* p, uint32_t v) { *p += v; } /* (one thread: lanes never run at the same time) */
static void zb_atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
static uint32_t zb_scan_excl_at(uint32_t v, uint32_t* total, uint32_t site)
{
    uint32_t s = 0, ex = 0;
    g_x[g_lane] = v;
    lanes_meet_at(site);
    for (uint32_t i = 0; i < LANES; ++i)
    {
        if (i == g_lane)
            ex = s;
        s += g_x[i];
    }
    *total = s;
    lanes_meet_at(site | 0x80000000u); /* (nobody overwrites g_x before everybody has read it) */
    return ex;
}
static uint64_t zb_ballot_at(int pred, uint32_t site)
{
    uint64_t m = 0;
    g_x[g_lane] = pred ? 1u : 0u;
    lanes_meet_at(site);
    for (uint32_t i = 0; i < LANES; ++i)
    {
        m |= (uint64_t)g_x[i] << i;
    }
    lanes_meet_at(site | 0x80000000u);
    return m;
}
static uint32_t zb_shfl_at(uint32_t v, uint32_t lane, uint32_t site)
{
    uint32_t r;
    g_x[g_lane] = v;
    lanes_meet_at(site);
    r = g_x[lane & (LANES - 1u)];
    lanes_meet_at(site | 0x80000000u);
    return r;
}
static uint32_t zb_reduce_max_at(uint32_t v, uint32_t site)
{
    uint32_t m = 0;
    g_x[g_lane] = v;
    lanes_meet_at(site);
    for (uint32_t i = 0; i < LANES; ++i)
        m = g_x[i] > m ? g_x[i] : m;
    lanes_meet_at(site | 0x80000000u);
    return m;
}

#define zb_scan_excl(v, total) zb_scan_excl_at((v), (total), __LINE__)
#define zb_ballot(pred) zb_ballot_at((pred), __LINE__)
#define zb_shfl(v, lane) zb_shfl_at((v), (lane), __LINE__)
#define zb_reduce_max(v) zb_reduce_max_at((v), __LINE__)

#include "../longtail_amd/csrc/zstd_block_core.h"

static const ZbInput* g_in;
static const ZbScratch* g_sc;
static ZbShared* g_sh;
static uint16_t* g_sub;
static uint32_t g_ret[LANES];

static void lane_main(void)
{
    const uint32_t lane = g_lane;
    g_ret[lane] = zb_encode_piece_sub(g_in, g_sc, g_sh, lane, g_sub);
    ++g_finished;
    if (g_arrived && !g_broken)
    {
        g_broken = 1; /* the others wait for a lane that is gone */
        g_bad_site[0] = g_site;
        g_bad_site[1] = 0xFFFFFFFFu;
    }
    /* uc_link: back to the scheduler */
}

/* zb_encode_piece_sub by 64 lanes: same arguments and result as ltz_model_encode_block_src with the sub-block layout; `sub` receives
 * the directory entries (ZB_MAX_UNITS).  0xFFFFFFFF: the lanes did not keep step (see the header of this file). */
uint32_t ltz_lanes_encode_piece_sub(const void* meta, const uint8_t* unit_lits, const uint64_t* unit_recs, uint32_t nunits, uint32_t raw_size,
                                    const uint8_t* src, uint32_t flags, uint8_t* out, uint16_t* sub)
{
    enum { STACK = 1 << 20 };
    ZbInput in;
    ZbScratch sc;
    uint32_t n;
    volatile uint32_t alive = LANES; /* (lives across swapcontext) */
    in.meta = (const ZbUnitMeta*)meta;
    in.unit_lits = unit_lits;
    in.unit_recs = unit_recs;
    in.nunits = nunits;
    in.raw_size = raw_size;
    in.src = src;
    in.flags = flags;
    sc.seqs = (uint64_t*)malloc(sizeof(uint64_t) * ZB_SEQ_MAX);
    sc.sbits = (uint16_t*)malloc(sizeof(uint16_t) * 4 * ZB_SEQ_MAX);
    sc.out = (uint32_t*)malloc(ZB_OUT_BYTES);
    memset(sc.out, 0xA5, ZB_OUT_BYTES); /* the encoders must not rely on a cleared output */
    g_sh = (ZbShared*)calloc(1, sizeof(ZbShared));
    g_in = &in;
    g_sc = &sc;
    g_sub = sub;
    g_gen = g_arrived = g_finished = 0;
    g_bad_site[0] = g_bad_site[1] = 0;
    g_broken = 0;
    if (!g_stacks)
        g_stacks = (char*)malloc((size_t)STACK * LANES);
    for (uint32_t i = 0; i < LANES; ++i)
    {
        getcontext(&g_ctx[i]);
        g_ctx[i].uc_stack.ss_sp = g_stacks + (size_t)STACK * i;
        g_ctx[i].uc_stack.ss_size = STACK;
        g_ctx[i].uc_link = &g_main;
        makecontext(&g_ctx[i], lane_main, 0);
        g_ret[i] = 0xFFFFFFFFu;
    }
    {
        unsigned char done[LANES];
        uint32_t idle_rounds = 0;
        memset(done, 0, sizeof done);
        while (alive && !g_broken && idle_rounds < 4u)
        {
            const uint32_t gen0 = g_gen, fin0 = g_finished;
            for (uint32_t k = 0; k < LANES && !g_broken; ++k)
            {
                const uint32_t i = g_descending ? LANES - 1u - k : k;
                if (!done[i])
                {
                    const uint32_t fin = g_finished;
                    g_lane = i;
                    swapcontext(&g_main, &g_ctx[i]);
                    if (g_finished != fin)
                    {
                        done[i] = 1;
                        --alive;
                    }
                }
            }
            idle_rounds = (g_gen == gen0 && g_finished == fin0) ? idle_rounds + 1u : 0u;
        }
    }
    n = (alive || g_broken) ? 0xFFFFFFFFu : g_ret[0]; /* (lane 0's result is the one the kernel stores) */
    if (n != 0xFFFFFFFFu && n)
        memcpy(out, sc.out, n);
    free(sc.seqs);
    free(sc.sbits);
    free(sc.out);
    free(g_sh);
    g_sh = NULL;
    return n;
}
