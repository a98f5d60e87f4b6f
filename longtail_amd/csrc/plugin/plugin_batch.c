/* plugin_batch.c -- one GPU submission for the small windows of MANY chunkers.
 *
 * The reference drives one chunker per (asset, part) job from a bikeshed worker (src/longtail.c:2039-2296); on a tree of small
 * files (BASELINE.json configs[2]: 65 536 files of 1 MiB) every worker used to pay, per file, one upload, ~15 kernel launches, three
 * downloads and a stream synchronisation of its own: 32 workers kept the GPU's command processor busy with ~100 k tiny operations
 * per 4 GiB and the HIP plugins ran at HALF the speed of the reference's CPU plugins (6.1 against 12.4 GB/s, round 3 measurement).
 * Here a worker hands its filled pinned window to a dispatcher thread and sleeps; the dispatcher takes WHATEVER IS QUEUED (up to 64
 * windows with the same chunk parameters), aims one 64-slot plan at them (lthip_plan_reaim: no allocation, no synchronisation),
 * uploads them into the slots of one device arena, runs ONE lthip_chunk_hash over all of them, downloads the dense result tables
 * once and hands every worker its slice.  No timer: while a submission runs the other workers refill and queue up, so the batch
 * grows with the load by itself, and a single-threaded caller gets a batch of one (the old behaviour plus a thread hop).
 *
 * Large windows (one 64 MiB part of a big asset) keep the direct path of plugin_chunker.c: they fill the GPU by themselves. */
#include "plugin_common.h"

#include <stdio.h>

#define LTB_SLOTS 64                 /* windows per submission */
#define LTB_ARENA_SLOTS 160          /* slots of the device arena: a submission in flight + the uploads of the next one + spare */
#define LTB_SLOT_BYTES ((uint64_t)LTP_WINDOW_SMALL)

#ifdef LTP_TIMING /* debug build only (tools/dropin_scaling.py with a library built -DLTP_TIMING): where the submitters and the dispatcher spend their time */
#include <stdio.h>
#include <time.h>
static double ltb_now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static double g_t[8]; /* 0 slot wait, 1 upload + sync, 2 queue wait (submitters, summed over threads); 3 dispatcher idle, 4 run_batch, 5 of it: device, 6 wake */
static uint64_t g_tn[8];
#define LTB_T0 double t0__ = ltb_now()
#define LTB_T(i)                                 \
    do                                           \
    {                                            \
        const double n__ = ltb_now();            \
        __atomic_fetch_add(&g_tn[i], 1, __ATOMIC_RELAXED); \
        pthread_mutex_lock(&g_tlock);            \
        g_t[i] += n__ - t0__;                    \
        pthread_mutex_unlock(&g_tlock);          \
        t0__ = n__;                              \
    } while (0)
static pthread_mutex_t g_tlock = PTHREAD_MUTEX_INITIALIZER;
#else
#define LTB_T0 ((void)0)
#define LTB_T(i) ((void)0)
#endif

struct ltb_req
{
    struct ltp_chunk_window* w;
    uint64_t have;
    uint32_t min_chunk, avg_chunk, max_chunk;
    int done, err;
    uint64_t total;
    int slot; /* arena slot the submitting thread has uploaded the window into */
    /* the submitter sleeps on a condition variable of ITS OWN (with g_lock): the dispatcher wakes exactly the threads whose windows are
     * done.  One shared variable + broadcast woke every waiting worker after every submission, each to take g_lock, see "not mine" and
     * sleep again -- a herd that grows with the job system's worker count: at 128 bikeshed workers a submission of 20 windows took
     * 2 ms instead of 0.3 and CreateVersionIndex fell from 32 to 10 GB/s (round 6, tools/dropin_scaling.py). */
    pthread_cond_t cv;
    struct ltb_req* next;
};

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_work = PTHREAD_COND_INITIALIZER;
static pthread_cond_t g_done = PTHREAD_COND_INITIALIZER;
static struct ltb_req *g_head, *g_tail;
static pthread_t g_thread;
static int g_running, g_stop, g_joining;
static uint64_t g_stat_batches, g_stat_windows; /* diagnostics: Longtail_Hip_BatchStats */
/* The arena is written by the SUBMITTING threads (each uploads its window on its own stream, all of them in parallel, and waits
 * for its own copy) and read by the dispatcher's kernels; a slot goes back to the free list when its submission is done. */
static void* g_arena;
static int g_arena_err;
static int g_free_slots[LTB_ARENA_SLOTS], g_free_count;
static pthread_cond_t g_slot_cv = PTHREAD_COND_INITIALIZER;

static void memo_put(const uint64_t* digests, uint32_t count, uint64_t content_hash);
static void memo_shutdown(void);

struct ltb_state /* owned by the dispatcher thread */
{
    lthip_ctx* ctx;
    lthip_plan* plan;
    uint32_t mn, av, mx;
    struct ltp_buf d_off, d_len, d_hash, d_first, d_content, h_res;
};

static void state_free(struct ltb_state* st)
{
    if (!st->ctx)
        return;
    if (st->plan)
        lthip_plan_destroy(st->ctx, st->plan);
    lthip_free_device(st->ctx, st->d_off.p);
    lthip_free_device(st->ctx, st->d_len.p);
    lthip_free_device(st->ctx, st->d_hash.p);
    lthip_free_device(st->ctx, st->d_first.p);
    lthip_free_device(st->ctx, st->d_content.p);
    lthip_free_pinned(st->ctx, st->h_res.p);
    lthip_ctx_destroy(st->ctx);
    memset(st, 0, sizeof *st);
}

static int state_prepare(struct ltb_state* st, uint32_t mn, uint32_t av, uint32_t mx)
{
    int err = 0;
    if (!st->ctx)
        return ENODEV;
    if (!st->plan || st->mn != mn || st->av != av || st->mx != mx)
    {
        if (st->plan)
            lthip_plan_destroy(st->ctx, st->plan);
        st->plan = 0;
        uint64_t offs[LTB_SLOTS], sizes[LTB_SLOTS];
        for (int i = 0; i < LTB_SLOTS; ++i)
        {
            offs[i] = (uint64_t)i * LTB_SLOT_BYTES;
            sizes[i] = LTB_SLOT_BYTES;
        }
        err = lthip_plan_create(st->ctx, LTB_SLOTS, offs, sizes, mn, av, mx, &st->plan);
        if (err)
            return err;
        st->mn = mn;
        st->av = av;
        st->mx = mx;
    }
    return 0;
}

/* one submission: reqs[0..n) have the same chunk parameters */
static int run_batch(struct ltb_state* st, struct ltb_req** reqs, uint32_t n)
{
    int err = state_prepare(st, reqs[0]->min_chunk, reqs[0]->avg_chunk, reqs[0]->max_chunk);
    if (err)
        return err;
    lthip_ctx* ctx = st->ctx;
    uint64_t offs[LTB_SLOTS], sizes[LTB_SLOTS];
    for (uint32_t i = 0; i < n; ++i)
    {
        offs[i] = (uint64_t)reqs[i]->slot * LTB_SLOT_BYTES;
        sizes[i] = reqs[i]->have;
    }
    err = lthip_plan_reaim(ctx, st->plan, n, offs, sizes);
    if (err)
        return err;
    const uint64_t cap = lthip_plan_chunk_capacity(st->plan);
    /* what THIS submission can produce at most: a window of `have` bytes has at most have / min + 1 chunks.  The downloads and the
     * content-hash launch are sized by it, not by the plan's capacity for 64 full windows (a single window at a small target chunk
     * size paid tens of MB of D2H per submission) */
    uint64_t ub = 0, run_ub = 0;
    for (uint32_t i = 0; i < n; ++i)
    {
        const uint64_t c = sizes[i] / reqs[0]->min_chunk + 2u;
        ub += c;
        if (c > run_ub)
            run_ub = c;
    }
    if (ub > cap)
        ub = cap;
    if (!err) err = ltp_dev_reserve(ctx, &st->d_off, (size_t)cap * 8);
    if (!err) err = ltp_dev_reserve(ctx, &st->d_len, (size_t)cap * 4);
    if (!err) err = ltp_dev_reserve(ctx, &st->d_hash, (size_t)cap * 8);
    if (!err) err = ltp_dev_reserve(ctx, &st->d_first, (size_t)(LTB_SLOTS + 1) * 4);
    if (!err) err = ltp_dev_reserve(ctx, &st->d_content, (size_t)LTB_SLOTS * 8);
    /* pinned results: [first: 65 u32][content hashes: 64 u64 at 512][offsets at 1024][hashes][lengths] */
    const size_t o_off = 1024, o_hash = o_off + (size_t)cap * 8, o_len = o_hash + (size_t)cap * 8;
    if (!err) err = ltp_pin_reserve(ctx, &st->h_res, o_len + (size_t)cap * 4);
    if (err)
        return err;
    LTB_T0;
    if (!err)
        err = lthip_chunk_hash(ctx, st->plan, g_arena, (uint64_t*)st->d_off.p, (uint32_t*)st->d_len.p, (uint64_t*)st->d_hash.p,
                               (uint32_t*)st->d_first.p, 0);
    if (!err) /* every window's digest array hashed as ChunkAssets will hash it (the content-hash memo below) */
        err = lthip_hash_runs_u64_bounded(ctx, (const uint64_t*)st->d_hash.p, (const uint32_t*)st->d_first.p, n, ub, run_ub,
                                          (uint64_t*)st->d_content.p);
    uint8_t* h = (uint8_t*)st->h_res.p;
    if (!err) err = lthip_copy_d2h(ctx, h, st->d_first.p, (size_t)(n + 1) * 4);
    if (!err) err = lthip_copy_d2h(ctx, h + 512, st->d_content.p, (size_t)n * 8);
    if (!err) err = lthip_copy_d2h(ctx, h + o_off, st->d_off.p, (size_t)ub * 8);
    if (!err) err = lthip_copy_d2h(ctx, h + o_hash, st->d_hash.p, (size_t)ub * 8);
    if (!err) err = lthip_copy_d2h(ctx, h + o_len, st->d_len.p, (size_t)ub * 4);
    if (!err) err = lthip_ctx_sync(ctx);
    LTB_T(5);
    if (err)
        return err;
    const uint32_t* first = (const uint32_t*)h;
    const uint64_t* r_off = (const uint64_t*)(h + o_off);
    const uint64_t* r_hash = (const uint64_t*)(h + o_hash);
    const uint32_t* r_len = (const uint32_t*)(h + o_len);
    for (uint32_t i = 0; i < n; ++i)
    {
        struct ltb_req* r = reqs[i];
        const uint32_t a = first[i], b = first[i + 1];
        if (b < a || b > ub || (uint64_t)(b - a) > r->w->ccap)
        {
            r->err = EIO;
            continue;
        }
        const uint32_t cnt = b - a;
        for (uint32_t k = 0; k < cnt; ++k)
            r->w->h_off[k] = r_off[a + k] - offs[i]; /* window relative, as the direct path delivers them */
        memcpy(r->w->h_len, r_len + a, (size_t)cnt * 4);
        memcpy(r->w->h_hash, r_hash + a, (size_t)cnt * 8);
        r->total = cnt;
        memo_put(r_hash + a, cnt, ((const uint64_t*)(h + 512))[i]);
    }
    return 0;
}

static void* dispatcher(void* arg)
{
    (void)arg;
    struct ltb_state st;
    memset(&st, 0, sizeof st);
    {
        /* the arena: allocated here, on the dispatcher's context, before the first request is looked at */
        void* a = 0;
        int err = lthip_ctx_create(ltp_device(), LTHIP_STREAM_PRIVATE, &st.ctx) != 0 ? ENODEV : 0;
        if (!err)
            err = lthip_malloc_device(st.ctx, (size_t)(LTB_ARENA_SLOTS * LTB_SLOT_BYTES), &a);
        else
            st.ctx = 0;
        pthread_mutex_lock(&g_lock);
        g_arena = a;
        g_arena_err = err ? err : 0;
        g_free_count = 0;
        if (!err)
            for (int i = LTB_ARENA_SLOTS - 1; i >= 0; --i)
                g_free_slots[g_free_count++] = i;
        else
            g_arena_err = err;
        pthread_cond_broadcast(&g_slot_cv);
        pthread_mutex_unlock(&g_lock);
        if (err)
        {
            /* no arena, no slots, so no request can ever be queued: this dispatcher is done.  The requester that reads g_arena_err
             * collects the thread, and the NEXT request starts another dispatcher that tries again -- running out of memory once is
             * the error of the calls that met it, not a state the ChunkerAPI objects stay in (tests/test_gpu_alloc_failures.py) */
            if (st.ctx)
                lthip_ctx_destroy(st.ctx);
            return 0;
        }
    }
    for (;;)
    {
        struct ltb_req* reqs[LTB_SLOTS];
        uint32_t n = 0;
        LTB_T0;
        pthread_mutex_lock(&g_lock);
        while (!g_head && !g_stop)
            pthread_cond_wait(&g_work, &g_lock);
        LTB_T(3);
        if (!g_head && g_stop)
        {
            pthread_mutex_unlock(&g_lock);
            break;
        }
        /* everything queued with the parameters of the oldest request, in arrival order; the rest waits for the next round */
        struct ltb_req *keep_head = 0, *keep_tail = 0, *r = g_head;
        const uint32_t mn = r->min_chunk, av = r->avg_chunk, mx = r->max_chunk;
        while (r)
        {
            struct ltb_req* nx = r->next;
            r->next = 0;
            if (n < LTB_SLOTS && r->min_chunk == mn && r->avg_chunk == av && r->max_chunk == mx)
                reqs[n++] = r;
            else
            {
                if (keep_tail)
                    keep_tail->next = r;
                else
                    keep_head = r;
                keep_tail = r;
            }
            r = nx;
        }
        g_head = keep_head;
        g_tail = keep_tail;
        pthread_mutex_unlock(&g_lock);

        const int err = run_batch(&st, reqs, n);
        LTB_T(4);

        pthread_mutex_lock(&g_lock);
        g_stat_batches += 1;
        g_stat_windows += n;
        for (uint32_t i = 0; i < n; ++i)
        {
            if (err && !reqs[i]->err)
                reqs[i]->err = err;
            g_free_slots[g_free_count++] = reqs[i]->slot;
            reqs[i]->done = 1;
            pthread_cond_signal(&reqs[i]->cv);
            pthread_cond_signal(&g_slot_cv); /* one freed slot, one waiter (if any) */
        }
        pthread_mutex_unlock(&g_lock);
        LTB_T(6);
    }
#ifdef LTP_TIMING
    fprintf(stderr, "ltb timing: submitters slot-wait %.3f s (%llu)  upload+sync %.3f s  queue-wait %.3f s | dispatcher idle %.3f s  run_batch %.3f s (%llu; device part %.3f s)  wake %.3f s\n",
            g_t[0], (unsigned long long)g_tn[0], g_t[1], g_t[2], g_t[3], g_t[4], (unsigned long long)g_tn[4], g_t[5], g_t[6]);
#endif
    pthread_mutex_lock(&g_lock);
    void* a = g_arena;
    g_arena = 0;
    g_free_count = 0;
    pthread_mutex_unlock(&g_lock);
    if (st.ctx)
        lthip_free_device(st.ctx, a);
    state_free(&st);
    return 0;
}

int ltp_batch_chunk_hash(struct ltp_chunk_window* w, uint64_t have, uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk,
                         uint64_t* out_total)
{
    if (!w || !out_total || have == 0 || have > LTB_SLOT_BYTES)
        return EINVAL;
    struct ltb_req r;
    memset(&r, 0, sizeof r);
    r.w = w;
    r.have = have;
    r.min_chunk = min_chunk;
    r.avg_chunk = avg_chunk;
    r.max_chunk = max_chunk;
    lthip_ctx* my = ltp_thread_ctx();
    if (!my)
        return ENODEV;
    pthread_cond_init(&r.cv, 0);
    LTB_T0;
    pthread_mutex_lock(&g_lock);
    while (g_joining) /* a shutdown is collecting the previous dispatcher: start the next one only when it is gone */
        pthread_cond_wait(&g_done, &g_lock);
    if (!g_running)
    {
        g_stop = 0;
        g_arena = 0;
        g_arena_err = 0;
        g_free_count = 0;
        if (pthread_create(&g_thread, 0, dispatcher, 0) != 0)
        {
            pthread_mutex_unlock(&g_lock);
            pthread_cond_destroy(&r.cv);
            return EAGAIN;
        }
        g_running = 1;
    }
    while (!g_arena_err && (!g_arena || g_free_count == 0)) /* the arena is being made, or every slot is in flight */
        pthread_cond_wait(&g_slot_cv, &g_lock);
    if (g_arena_err)
    {
        const int e = g_arena_err;
        if (g_running && !g_joining)
        {
            /* the dispatcher that could not make its arena has exited (above): collect it; the next request starts a fresh one */
            g_joining = 1;
            g_running = 0;
            const pthread_t t = g_thread;
            pthread_mutex_unlock(&g_lock);
            pthread_join(t, 0);
            pthread_mutex_lock(&g_lock);
            g_joining = 0;
            pthread_cond_broadcast(&g_done);
        }
        pthread_mutex_unlock(&g_lock);
        pthread_cond_destroy(&r.cv);
        return e;
    }
    r.slot = g_free_slots[--g_free_count];
    uint8_t* dst = (uint8_t*)g_arena + (uint64_t)r.slot * LTB_SLOT_BYTES;
    pthread_mutex_unlock(&g_lock);
    LTB_T(0);

    /* my window goes up on MY stream, next to the other threads' uploads and under the dispatcher's kernels */
    int err = lthip_copy_h2d(my, dst, w->h_win, (size_t)have);
    if (!err)
        err = lthip_ctx_sync(my);
    LTB_T(1);

    pthread_mutex_lock(&g_lock);
    if (err)
    {
        g_free_slots[g_free_count++] = r.slot;
        pthread_cond_signal(&g_slot_cv);
        pthread_mutex_unlock(&g_lock);
        pthread_cond_destroy(&r.cv);
        return err;
    }
    if (g_tail)
        g_tail->next = &r;
    else
        g_head = &r;
    g_tail = &r;
    pthread_cond_signal(&g_work);
    while (!r.done)
        pthread_cond_wait(&r.cv, &g_lock);
    pthread_mutex_unlock(&g_lock);
    LTB_T(2);
    pthread_cond_destroy(&r.cv);
    *out_total = r.total;
    return r.err;
}

/* stops the dispatcher and frees its arena (called when the last HIP ChunkerAPI is disposed: no request can be pending) */
void ltp_batch_shutdown(void)
{
    pthread_mutex_lock(&g_lock);
    if (!g_running)
    {
        pthread_mutex_unlock(&g_lock);
        return;
    }
    g_stop = 1;
    g_joining = 1;
    pthread_cond_signal(&g_work);
    pthread_t t = g_thread;
    g_running = 0;
    pthread_mutex_unlock(&g_lock);
    pthread_join(t, 0);
    memo_shutdown();
    pthread_mutex_lock(&g_lock);
    g_joining = 0;
    pthread_cond_broadcast(&g_done);
    pthread_mutex_unlock(&g_lock);
}

/* ---------------------------------------------------------------------------------------------------
 * Content-hash memo.  ChunkAssets hashes every asset's array of chunk digests AFTER all chunking, one asset after the other on the
 * calling thread (src/longtail.c:2518-2537): one GPU round trip per asset, serial -- as much time as the chunking itself on a tree of
 * small files.  The batcher already has those digests on the device when a window is chunked, so it hashes each window's digest
 * array in the same submission (lthip_hash_runs_u64) and remembers {digests -> hash}; HashBuffer looks here before it goes to the
 * GPU.  A hit is verified byte for byte against the stored digests, so it is exactly the value the GPU computed for exactly these
 * bytes.  Bounded (LTP_MEMO_MAX_BYTES; cleared when full and when the last HIP ChunkerAPI goes away).
 * ------------------------------------------------------------------------------------------------- */
#define LTP_MEMO_BUCKETS 65536u
#define LTP_MEMO_MAX_BYTES (128u << 20)
struct ltp_memo_entry
{
    struct ltp_memo_entry* next;
    uint64_t content_hash;
    uint32_t count;
    uint64_t digests[];
};
static pthread_mutex_t g_memo_lock = PTHREAD_MUTEX_INITIALIZER;
static struct ltp_memo_entry** g_memo;
static size_t g_memo_bytes;
static uint64_t g_memo_hits, g_memo_puts;

static uint32_t memo_bucket(const uint64_t* d, uint32_t count)
{
    uint64_t x = d[0] ^ (d[count - 1] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)count << 48);
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return (uint32_t)x & (LTP_MEMO_BUCKETS - 1u);
}

static void memo_clear_locked(void)
{
    if (!g_memo)
        return;
    for (uint32_t b = 0; b < LTP_MEMO_BUCKETS; ++b)
    {
        struct ltp_memo_entry* e = g_memo[b];
        while (e)
        {
            struct ltp_memo_entry* n = e->next;
            free(e);
            e = n;
        }
        g_memo[b] = 0;
    }
    g_memo_bytes = 0;
}

static void memo_put(const uint64_t* digests, uint32_t count, uint64_t content_hash)
{
    if (!count)
        return;
    const size_t bytes = sizeof(struct ltp_memo_entry) + (size_t)count * 8;
    pthread_mutex_lock(&g_memo_lock);
    if (!g_memo)
        g_memo = (struct ltp_memo_entry**)calloc(LTP_MEMO_BUCKETS, sizeof *g_memo); /* library bookkeeping: plain malloc */
    if (g_memo)
    {
        if (g_memo_bytes + bytes > LTP_MEMO_MAX_BYTES)
            memo_clear_locked();
        const uint32_t b = memo_bucket(digests, count);
        struct ltp_memo_entry* e = g_memo[b];
        while (e && !(e->count == count && memcmp(e->digests, digests, (size_t)count * 8) == 0))
            e = e->next;
        if (!e)
        {
            e = (struct ltp_memo_entry*)malloc(bytes);
            if (e)
            {
                e->content_hash = content_hash;
                e->count = count;
                memcpy(e->digests, digests, (size_t)count * 8);
                e->next = g_memo[b];
                g_memo[b] = e;
                g_memo_bytes += bytes;
                ++g_memo_puts;
            }
        }
    }
    pthread_mutex_unlock(&g_memo_lock);
}

int ltp_memo_get(const void* data, uint32_t length, uint64_t* out_hash)
{
    if (length < 8u || (length & 7u) || !g_memo) /* (g_memo only ever goes from 0 to its table: a stale 0 is a miss) */
        return 0;
    const uint32_t count = length / 8u;
    uint64_t first, last;
    memcpy(&first, data, 8);
    memcpy(&last, (const uint8_t*)data + (size_t)(count - 1u) * 8, 8);
    int hit = 0;
    pthread_mutex_lock(&g_memo_lock);
    if (g_memo)
    {
        uint64_t x = first ^ (last * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)count << 48);
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        const struct ltp_memo_entry* e = g_memo[(uint32_t)x & (LTP_MEMO_BUCKETS - 1u)];
        while (e && !(e->count == count && memcmp(e->digests, data, length) == 0))
            e = e->next;
        if (e)
        {
            *out_hash = e->content_hash;
            hit = 1;
            ++g_memo_hits;
        }
    }
    pthread_mutex_unlock(&g_memo_lock);
    return hit;
}

static void memo_shutdown(void)
{
    pthread_mutex_lock(&g_memo_lock);
    memo_clear_locked();
    free(g_memo);
    g_memo = 0;
    pthread_mutex_unlock(&g_memo_lock);
}

void Longtail_Hip_BatchStats(uint64_t* out_batches, uint64_t* out_windows)
{
    pthread_mutex_lock(&g_lock);
    if (out_batches)
        *out_batches = g_stat_batches;
    if (out_windows)
        *out_windows = g_stat_windows;
    pthread_mutex_unlock(&g_lock);
}

void Longtail_Hip_MemoStats(uint64_t* out_puts, uint64_t* out_hits)
{
    pthread_mutex_lock(&g_memo_lock);
    if (out_puts)
        *out_puts = g_memo_puts;
    if (out_hits)
        *out_hits = g_memo_hits;
    pthread_mutex_unlock(&g_memo_lock);
}
