/* plugin_common.c -- allocator hooks, per-thread GPU contexts and the chunk-window registry shared by
 * the HIP ChunkerAPI / HashAPI / CompressionAPI objects.  Plain C99 over the lthip_* C ABI. */
#include "plugin_common.h"

#include <sched.h>
#include <stdatomic.h>

#include <stdio.h>

/* ---------------------------------------------------------------------------------------------------
 * allocator
 * ------------------------------------------------------------------------------------------------- */
static Longtail_Hip_AllocFunc g_alloc;
static Longtail_Hip_FreeFunc g_free;

void Longtail_Hip_SetAllocator(Longtail_Hip_AllocFunc alloc_func, Longtail_Hip_FreeFunc free_func)
{
    g_alloc = alloc_func;
    g_free = free_func;
}

void* ltp_alloc(const char* context, size_t size) { return g_alloc ? g_alloc(context, size) : malloc(size); }

void ltp_free(void* p)
{
    if (!p)
        return;
    if (g_free)
        g_free(p);
    else
        free(p);
}

/* ---------------------------------------------------------------------------------------------------
 * device selection + per-thread context
 * ------------------------------------------------------------------------------------------------- */
static int g_device = -1; /* read and written with atomics: plugin objects are created from any thread */

int Longtail_Hip_SetDevice(int device)
{
    if (device < 0 || device >= lthip_device_count())
        return EINVAL;
    __atomic_store_n(&g_device, device, __ATOMIC_RELEASE);
    return 0;
}

int ltp_device(void)
{
    int d = __atomic_load_n(&g_device, __ATOMIC_ACQUIRE);
    if (d < 0)
    {
        const char* e = getenv("LONGTAIL_HIP_DEVICE");
        int want = e ? atoi(e) : 0;
        int expected = -1;
        __atomic_compare_exchange_n(&g_device, &expected, want, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
        d = __atomic_load_n(&g_device, __ATOMIC_ACQUIRE);
    }
    return d;
}

/* ---------------------------------------------------------------------------------------------------
 * error latch
 * ------------------------------------------------------------------------------------------------- */
static __thread int t_last_error;

void ltp_latch_error(int err)
{
    if (err && !t_last_error)
        t_last_error = err;
}

int Longtail_Hip_GetLastError(void)
{
    const int e = t_last_error;
    t_last_error = 0;
    return e;
}

int Longtail_Hip_SetBlockingWaits(int on) { return lthip_set_blocking_waits(ltp_device(), on); } /* (include/longtail_hip.h: first call, or never) */

static pthread_key_t g_key;
static pthread_once_t g_key_once = PTHREAD_ONCE_INIT;

static void thread_state_destroy(void* p)
{
    struct ltp_thread_state* ts = (struct ltp_thread_state*)p;
    if (!ts)
        return;
    if (ts->ctx)
    {
        lthip_free_device(ts->ctx, ts->d_in.p);
        lthip_free_device(ts->ctx, ts->d_out.p);
        lthip_free_device(ts->ctx, ts->d_aux.p);
        lthip_free_pinned(ts->ctx, ts->h_pin.p);
        lthip_ctx_destroy(ts->ctx);
    }
    free(ts); /* bookkeeping of the library itself, not of a plugin object: plain malloc/free */
}

static void make_key(void) { pthread_key_create(&g_key, thread_state_destroy); }

struct ltp_thread_state* ltp_thread_state_get(void)
{
    pthread_once(&g_key_once, make_key);
    struct ltp_thread_state* ts = (struct ltp_thread_state*)pthread_getspecific(g_key);
    if (ts)
        return ts;
    ts = (struct ltp_thread_state*)calloc(1, sizeof *ts);
    if (!ts)
        return 0;
    if (lthip_ctx_create(ltp_device(), LTHIP_STREAM_PRIVATE, &ts->ctx) != 0)
    {
        free(ts);
        return 0;
    }
    pthread_setspecific(g_key, ts);
    return ts;
}

lthip_ctx* ltp_thread_ctx(void)
{
    struct ltp_thread_state* ts = ltp_thread_state_get();
    return ts ? ts->ctx : 0;
}

int ltp_dev_reserve(lthip_ctx* ctx, struct ltp_buf* b, size_t bytes)
{
    if (b->cap >= bytes && b->p)
        return 0;
    lthip_free_device(ctx, b->p);
    b->p = 0;
    b->cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    int err = lthip_malloc_device(ctx, cap, &b->p);
    if (err)
        return err;
    b->cap = cap;
    return 0;
}

int ltp_pin_reserve(lthip_ctx* ctx, struct ltp_buf* b, size_t bytes)
{
    if (b->cap >= bytes && b->p)
        return 0;
    lthip_free_pinned(ctx, b->p);
    b->p = 0;
    b->cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    int err = lthip_malloc_pinned(ctx, cap, &b->p);
    if (err)
        return err;
    b->cap = cap;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * window registry
 * ------------------------------------------------------------------------------------------------- */
#define LTP_MAX_WINDOWS 1024
static pthread_rwlock_t g_win_lock = PTHREAD_RWLOCK_INITIALIZER;
static struct ltp_window g_windows[LTP_MAX_WINDOWS];
static unsigned char g_window_used[LTP_MAX_WINDOWS];
static int g_window_high; /* highest used slot + 1 */
/* per-slot sequence number: odd while a writer (always under g_win_lock) is changing the slot, bumped twice per change.  The
 * lock-free look-up of a thread's own window (below) is valid only while the number it noted at ltp_window_set_current stands.
 *
 * A sequence number protects against torn VALUES, not against unmapped MEMORY: the slot holds pointers to a window's pinned tables,
 * and whoever changes the slot (the chunker's next window, DisposeChunker -- possibly from another thread --, the pool's trim) goes
 * on to recycle or free those tables.  So the fast path also COUNTS itself in (g_window_readers[slot]) before it looks at the
 * sequence number, and a writer, having made the number odd, waits for the count to drain before it touches the slot: a reader that
 * saw the old number is finished with the tables before they can go away, a reader that comes later sees the odd / new number and
 * takes the locked path.  Both counters are sequentially consistent (store-buffering pattern: reader = count++ then load seq, writer
 * = seq++ then load count -- at least one of them sees the other).  The count's cache line is the slot's own: the thread that owns
 * the chunker is the only regular visitor, so the two atomic RMWs cost a few nanoseconds, not a contended lock. */
static _Atomic uint64_t g_window_seq[LTP_MAX_WINDOWS];
static struct
{
    _Atomic uint32_t n;
    char pad[60];
} g_window_readers[LTP_MAX_WINDOWS];

static void slot_write_begin(int slot)
{
    atomic_fetch_add_explicit(&g_window_seq[slot], 1, memory_order_seq_cst);
    for (unsigned spins = 0; atomic_load_explicit(&g_window_readers[slot].n, memory_order_seq_cst) != 0; ++spins)
        if (spins > 64)
            sched_yield(); /* a reader is inside one binary search: nanoseconds */
}
static void slot_write_end(int slot) { atomic_fetch_add_explicit(&g_window_seq[slot], 1, memory_order_release); }

int ltp_window_register(void)
{
    int slot = -1;
    pthread_rwlock_wrlock(&g_win_lock);
    for (int i = 0; i < LTP_MAX_WINDOWS; ++i)
    {
        if (!g_window_used[i])
        {
            slot_write_begin(i);
            g_window_used[i] = 1;
            memset(&g_windows[i], 0, sizeof g_windows[i]);
            slot_write_end(i);
            if (i + 1 > g_window_high)
                g_window_high = i + 1;
            slot = i;
            break;
        }
    }
    pthread_rwlock_unlock(&g_win_lock);
    return slot;
}

void ltp_window_publish(int slot, const struct ltp_window* w)
{
    if (slot < 0)
        return;
    pthread_rwlock_wrlock(&g_win_lock);
    slot_write_begin(slot);
    g_windows[slot] = *w;
    slot_write_end(slot);
    pthread_rwlock_unlock(&g_win_lock);
}

void ltp_window_unregister(int slot)
{
    if (slot < 0)
        return;
    pthread_rwlock_wrlock(&g_win_lock);
    slot_write_begin(slot);
    g_window_used[slot] = 0;
    memset(&g_windows[slot], 0, sizeof g_windows[slot]);
    slot_write_end(slot);
    while (g_window_high > 0 && !g_window_used[g_window_high - 1])
        --g_window_high;
    pthread_rwlock_unlock(&g_win_lock);
}

/* The calling thread's own window: slot + the slot's sequence number when the chunker published it.  Any later change of the slot
 * (the chunker's next window, its disposal by another thread, reuse by another chunker) changes the number, and the look-up
 * falls back to the locked scan: a thread never reads a slot that is being written and never trusts one that has changed. */
static __thread int t_current_slot = -1;
static __thread uint64_t t_current_seq;

void ltp_window_set_current(int slot)
{
    t_current_slot = slot;
    if (slot >= 0)
        t_current_seq = atomic_load_explicit(&g_window_seq[slot], memory_order_acquire);
}

static int window_find(const struct ltp_window* w, const uint8_t* p, uint32_t len, uint64_t* out_hash)
{
    const uint64_t rel = (uint64_t)(p - w->base);
    uint32_t lo = 0, hi = w->count;
    while (lo < hi)
    {
        uint32_t mid = lo + (hi - lo) / 2;
        if (w->offsets[mid] < rel)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < w->count && w->offsets[lo] == rel && w->lens[lo] == len)
    {
        *out_hash = w->hashes[lo];
        return 1;
    }
    return 0;
}

int ltp_window_lookup(const void* data, uint32_t len, uint64_t* out_hash)
{
    const uint8_t* p = (const uint8_t*)data;
    /* the calling thread's own chunker first: no lock, no scan (the common case by far, src/longtail.c:2231-2296) */
    const int slot = t_current_slot;
    if (slot >= 0 && !(t_current_seq & 1u))
    {
        atomic_fetch_add_explicit(&g_window_readers[slot].n, 1, memory_order_seq_cst);
        if (atomic_load_explicit(&g_window_seq[slot], memory_order_seq_cst) == t_current_seq)
        {
            /* no writer has begun on this slot, and none can get past slot_write_begin while the count stands: the slot and the
             * tables it points to are stable until the decrement below */
            const struct ltp_window* cur = &g_windows[slot];
            int hit = -1;
            uint64_t h = 0;
            if (cur->base && p >= cur->base && p < cur->base + cur->size)
                hit = window_find(cur, p, len, &h);
            atomic_fetch_sub_explicit(&g_window_readers[slot].n, 1, memory_order_release);
            if (hit >= 0)
            {
                if (hit)
                    *out_hash = h;
                return hit;
            }
        }
        else
            atomic_fetch_sub_explicit(&g_window_readers[slot].n, 1, memory_order_release);
    }
    int found = 0;
    pthread_rwlock_rdlock(&g_win_lock);
    for (int i = 0; i < g_window_high && !found; ++i)
    {
        const struct ltp_window* w = &g_windows[i];
        if (!g_window_used[i] || !w->base || p < w->base || p >= w->base + w->size)
            continue;
        const uint64_t rel = (uint64_t)(p - w->base);
        uint32_t lo = 0, hi = w->count;
        while (lo < hi)
        {
            uint32_t mid = lo + (hi - lo) / 2;
            if (w->offsets[mid] < rel)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < w->count && w->offsets[lo] == rel && w->lens[lo] == len)
        {
            *out_hash = w->hashes[lo];
            found = 1;
        }
        break; /* the pointer lies inside this window: no other window can own it */
    }
    pthread_rwlock_unlock(&g_win_lock);
    return found;
}

/* ---------------------------------------------------------------------------------------------------
 * window pool
 * ------------------------------------------------------------------------------------------------- */
static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_pool_cond = PTHREAD_COND_INITIALIZER;
static struct ltp_chunk_window* g_pool_free[2];
static uint32_t g_pool_alive[2]; /* windows of the class in existence (idle + handed out) */
static uint64_t g_pool_pinned;   /* bytes of pinned memory held by windows */

static uint32_t pool_cap(int cls)
{
    static uint32_t caps[2];
    if (!caps[cls])
    {
        const char* e = getenv(cls ? "LONGTAIL_HIP_LARGE_WINDOWS" : "LONGTAIL_HIP_SMALL_WINDOWS");
        int v = e ? atoi(e) : 0;
        caps[cls] = v > 0 ? (uint32_t)v : (cls ? 32u : 256u);
    }
    return caps[cls];
}

static void window_destroy(struct ltp_chunk_window* w)
{
    if (!w)
        return;
    if (w->plan)
        lthip_plan_destroy(0, w->plan);
    lthip_free_pinned(0, w->h_win);
    lthip_free_device(0, w->d_win);
    lthip_free_device(0, w->d_off);
    lthip_free_device(0, w->d_len);
    lthip_free_device(0, w->d_hash);
    lthip_free_device(0, w->d_first);
    lthip_free_pinned(0, w->h_off);
    lthip_free_pinned(0, w->h_len);
    lthip_free_pinned(0, w->h_hash);
    free(w);
}

static uint64_t window_pinned_bytes(const struct ltp_chunk_window* w) { return w->cap + w->ccap * 20u; }

static struct ltp_chunk_window* window_create(lthip_ctx* ctx, uint64_t cap, uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk, int cls,
                                              int* out_err)
{
    struct ltp_chunk_window* w = (struct ltp_chunk_window*)calloc(1, sizeof *w);
    if (!w)
    {
        *out_err = ENOMEM;
        return 0;
    }
    /* result tables for the smallest chunks the API allows (48 bytes, hpcdcchunker.c:143) would be a third of the window: size them
     * for this chunker's minimum and rebuild them when a later user of the window needs more */
    const uint64_t ccap = cap / min_chunk + 2;
    int err = 0;
    if (!err) err = lthip_malloc_pinned(ctx, cap, (void**)&w->h_win);
    if (!err) err = lthip_malloc_device(ctx, cap + 64, &w->d_win);
    if (!err) err = lthip_malloc_device(ctx, ccap * 8, (void**)&w->d_off);
    if (!err) err = lthip_malloc_device(ctx, ccap * 4, (void**)&w->d_len);
    if (!err) err = lthip_malloc_device(ctx, ccap * 8, (void**)&w->d_hash);
    if (!err) err = lthip_malloc_device(ctx, 16, (void**)&w->d_first);
    if (!err) err = lthip_malloc_pinned(ctx, ccap * 8, (void**)&w->h_off);
    if (!err) err = lthip_malloc_pinned(ctx, ccap * 4, (void**)&w->h_len);
    if (!err) err = lthip_malloc_pinned(ctx, ccap * 8, (void**)&w->h_hash);
    if (!err)
    {
        const uint64_t off0 = 0;
        err = lthip_plan_create(ctx, 1, &off0, &cap, min_chunk, avg_chunk, max_chunk, &w->plan);
    }
    if (err)
    {
        window_destroy(w);
        *out_err = err;
        return 0;
    }
    w->cap = cap;
    w->ccap = ccap;
    w->min_chunk = min_chunk;
    w->avg_chunk = avg_chunk;
    w->max_chunk = max_chunk;
    w->cls = cls;
    return w;
}

struct ltp_chunk_window* ltp_window_acquire(lthip_ctx* ctx, uint64_t bytes, uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk, int* out_err)
{
    *out_err = 0;
    if (bytes > LTP_WINDOW_LARGE)
    {
        struct ltp_chunk_window* w = window_create(ctx, bytes, min_chunk, avg_chunk, max_chunk, 2, out_err);
        if (w)
        {
            pthread_mutex_lock(&g_pool_lock);
            g_pool_pinned += window_pinned_bytes(w);
            pthread_mutex_unlock(&g_pool_lock);
        }
        return w;
    }
    const int cls = bytes > LTP_WINDOW_SMALL ? 1 : 0;
    const uint64_t cap = cls ? LTP_WINDOW_LARGE : LTP_WINDOW_SMALL;
    struct ltp_chunk_window* w = 0;
    int create = 0;
    pthread_mutex_lock(&g_pool_lock);
    for (;;)
    {
        /* an idle window made for the same chunk parameters (the plan holds the discriminator and the table capacities) */
        struct ltp_chunk_window** pp = &g_pool_free[cls];
        while (*pp && !((*pp)->min_chunk == min_chunk && (*pp)->avg_chunk == avg_chunk && (*pp)->max_chunk == max_chunk))
            pp = &(*pp)->next;
        if (*pp)
        {
            w = *pp;
            *pp = w->next;
            w->next = 0;
            break;
        }
        if (g_pool_alive[cls] < pool_cap(cls))
        {
            ++g_pool_alive[cls];
            create = 1;
            break;
        }
        if (g_pool_free[cls])
        {
            /* at the cap, but an idle window with other parameters exists: replace it */
            struct ltp_chunk_window* victim = g_pool_free[cls];
            g_pool_free[cls] = victim->next;
            g_pool_pinned -= window_pinned_bytes(victim);
            pthread_mutex_unlock(&g_pool_lock);
            window_destroy(victim);
            pthread_mutex_lock(&g_pool_lock);
            create = 1;
            break;
        }
        pthread_cond_wait(&g_pool_cond, &g_pool_lock);
    }
    pthread_mutex_unlock(&g_pool_lock);
    if (create)
    {
        w = window_create(ctx, cap, min_chunk, avg_chunk, max_chunk, cls, out_err);
        pthread_mutex_lock(&g_pool_lock);
        if (w)
            g_pool_pinned += window_pinned_bytes(w);
        else
        {
            --g_pool_alive[cls];
            pthread_cond_signal(&g_pool_cond);
        }
        pthread_mutex_unlock(&g_pool_lock);
    }
    return w;
}

void ltp_window_release(struct ltp_chunk_window* w)
{
    if (!w)
        return;
    if (w->cls == 2)
    {
        pthread_mutex_lock(&g_pool_lock);
        g_pool_pinned -= window_pinned_bytes(w);
        pthread_mutex_unlock(&g_pool_lock);
        window_destroy(w);
        return;
    }
    pthread_mutex_lock(&g_pool_lock);
    w->next = g_pool_free[w->cls];
    g_pool_free[w->cls] = w;
    pthread_cond_signal(&g_pool_cond);
    pthread_mutex_unlock(&g_pool_lock);
}

void ltp_window_pool_trim(void)
{
    for (int cls = 0; cls < 2; ++cls)
        for (;;)
        {
            pthread_mutex_lock(&g_pool_lock);
            struct ltp_chunk_window* w = g_pool_free[cls];
            if (w)
            {
                g_pool_free[cls] = w->next;
                --g_pool_alive[cls];
                g_pool_pinned -= window_pinned_bytes(w);
            }
            pthread_mutex_unlock(&g_pool_lock);
            if (!w)
                break;
            window_destroy(w);
        }
}

uint64_t ltp_window_pool_pinned_bytes(void)
{
    pthread_mutex_lock(&g_pool_lock);
    const uint64_t v = g_pool_pinned;
    pthread_mutex_unlock(&g_pool_lock);
    return v;
}

uint64_t Longtail_Hip_PinnedBytes(void) { return ltp_window_pool_pinned_bytes(); }
