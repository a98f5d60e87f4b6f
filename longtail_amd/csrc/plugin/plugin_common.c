/* plugin_common.c -- allocator hooks, per-thread GPU contexts and the chunk-window registry shared by
 * the HIP ChunkerAPI / HashAPI / CompressionAPI objects.  Plain C99 over the lthip_* C ABI. */
#include "plugin_common.h"

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
static int g_device = -1;

int Longtail_Hip_SetDevice(int device)
{
    if (device < 0 || device >= lthip_device_count())
        return EINVAL;
    g_device = device;
    return 0;
}

int ltp_device(void)
{
    if (g_device < 0)
    {
        const char* e = getenv("LONGTAIL_HIP_DEVICE");
        g_device = e ? atoi(e) : 0;
    }
    return g_device;
}

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

int ltp_window_register(void)
{
    int slot = -1;
    pthread_rwlock_wrlock(&g_win_lock);
    for (int i = 0; i < LTP_MAX_WINDOWS; ++i)
    {
        if (!g_window_used[i])
        {
            g_window_used[i] = 1;
            memset(&g_windows[i], 0, sizeof g_windows[i]);
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
    g_windows[slot] = *w;
    pthread_rwlock_unlock(&g_win_lock);
}

void ltp_window_unregister(int slot)
{
    if (slot < 0)
        return;
    pthread_rwlock_wrlock(&g_win_lock);
    g_window_used[slot] = 0;
    memset(&g_windows[slot], 0, sizeof g_windows[slot]);
    while (g_window_high > 0 && !g_window_used[g_window_high - 1])
        --g_window_high;
    pthread_rwlock_unlock(&g_win_lock);
}

int ltp_window_lookup(const void* data, uint32_t len, uint64_t* out_hash)
{
    const uint8_t* p = (const uint8_t*)data;
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
