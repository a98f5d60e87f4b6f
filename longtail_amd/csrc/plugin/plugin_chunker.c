/* plugin_chunker.c -- Longtail_ChunkerAPI on the GPU (C99 host code over the lthip_* C ABI).
 *
 * Mirrors lib/hpcdcchunker/longtail_hpcdcchunker.c of the reference:
 *   Longtail_CreateHipChunkerAPI  <-> Longtail_CreateHPCDCChunkerAPI   (:563-589)
 *   GetMinChunkSize               <-> HPCDCChunker_GetMinChunkSize     (:332-346)  -> 48
 *   CreateChunker / DisposeChunker<-> HPCDCChunker_CreateChunker/_DisposeChunker (:348-388, 427-450), handles pooled
 *   NextChunk                     <-> HPCDCChunker_NextChunk + Longtail_HPCDCNextChunk (:390-425, 225-310)
 *   NextChunkFromBuffer           <-> HPCDCChunker_NextChunkFromBuffer (:452-523), window-seed quirk included
 *
 * The pull-style, one-chunk-per-call API is turned into bulk GPU work: the first NextChunk of a window
 * drains the feeder (StorageChunkFeederFunc, src/longtail.c:1923-1960, serves any request size) into pinned
 * memory, the GPU computes every cut AND every chunk hash of the window, and later calls only hand out
 * ranges.  A cut depends only on the chunk start and the next `max` bytes, so windows reproduce the
 * reference's stream semantics exactly: chunks that start less than `max` bytes before the end of a
 * non-final window are recomputed in the next window.
 *
 * Memory: a chunker owns no buffers.  It borrows a WINDOW (pinned host + device memory, result tables, a plan of the window's
 * capacity) from the bounded pool of plugin_common.c when its stream starts -- a 2 MiB one; a stream that fills it moves to a
 * 64 MiB one (= one reference part at target 65536, src/longtail.c:2396) -- and returns it when it is disposed.  However many
 * chunkers longtail's job system keeps alive, pinned memory stays below the pool's cap (include/longtail_hip.h,
 * Longtail_Hip_PinnedBytes).
 */
#include "plugin_common.h"

#define HIP_CHUNKER_POOL 232 /* HPCDCCHUNKER_MAX_CACHED_CHUNKER_COUNT, hpcdcchunker.c:102 (handles only: a few hundred bytes each) */

struct HipChunker
{
    uint32_t min, avg, max;
    struct ltp_chunk_window* w; /* borrowed while a stream is in progress */
    uint64_t have;              /* valid bytes in the window */
    uint64_t base;              /* stream offset of window byte 0 */
    int eof;
    uint32_t ntotal, nfinal, next; /* chunks of the current window: computed / final / handed out */
    struct ltp_window pub;         /* what the HashAPI may look up */
    int slot;                      /* window registry slot */
};

struct HipChunkerAPI
{
    struct Longtail_ChunkerAPI api;
    pthread_mutex_t lock;
    struct HipChunker* pool[HIP_CHUNKER_POOL];
    uint32_t pool_count;
};

static uint32_t g_api_count; /* live HipChunkerAPI objects: the last one to go trims the window pool */

static void chunker_unpublish(struct HipChunker* c)
{
    memset(&c->pub, 0, sizeof c->pub);
    ltp_window_publish(c->slot, &c->pub); /* ranges of the old window are no longer valid */
    ltp_window_set_current(-1);
}

static void chunker_drop_window(struct HipChunker* c)
{
    if (c->w)
    {
        chunker_unpublish(c);
        ltp_window_release(c->w);
        c->w = 0;
    }
}

static void chunker_free(struct HipChunker* c)
{
    if (!c)
        return;
    chunker_drop_window(c);
    ltp_window_unregister(c->slot);
    ltp_free(c);
}

/* LONGTAIL_HIP_BATCH=0 keeps every window on its thread's own stream (the round-1/2 behaviour; ablation) */
static int batching_enabled(void)
{
    static int v = -1;
    if (v < 0)
    {
        const char* e = getenv("LONGTAIL_HIP_BATCH");
        v = !(e && e[0] == '0');
    }
    return v;
}

/* fill the window from the feeder and run the GPU over it */
static int chunker_refill(struct HipChunker* c, Longtail_Chunker_Feeder feeder, void* feeder_context, int* feeder_failed)
{
    lthip_ctx* ctx = ltp_thread_ctx();
    if (!ctx)
        return ENODEV;
    int err = 0;
    const uint64_t need_min = (uint64_t)c->max * 4u; /* the reference buffers 4 * max (hpcdcchunker.c:148) */
    if (!c->w)
    {
        c->w = ltp_window_acquire(ctx, need_min > LTP_WINDOW_SMALL ? need_min : LTP_WINDOW_SMALL, c->min, c->avg, c->max, &err);
        if (!c->w)
            return err ? err : ENOMEM;
    }
    struct ltp_chunk_window* w = c->w;

    /* keep the bytes whose chunking is not final yet */
    const uint64_t keep_from = c->nfinal < c->ntotal ? w->h_off[c->nfinal] : c->have;
    chunker_unpublish(c);
    if (keep_from < c->have && keep_from > 0)
        memmove(w->h_win, w->h_win + keep_from, (size_t)(c->have - keep_from));
    c->base += keep_from;
    c->have -= keep_from;
    c->ntotal = c->nfinal = c->next = 0;

    for (;;)
    {
        while (c->have < w->cap && !c->eof)
        {
            uint64_t want = w->cap - c->have;
            if (want > 0x7FFFFFFFu)
                want = 0x7FFFFFFFu;
            uint32_t got = 0;
            err = feeder(feeder_context, (Longtail_ChunkerAPI_HChunker)c, (uint32_t)want, (char*)w->h_win + c->have, &got);
            if (err)
            {
                *feeder_failed = 1;
                return err;
            }
            if (got == 0)
                c->eof = 1;
            c->have += got;
        }
        if (c->eof || w->cls != 0)
            break;
        /* the small window is full and the stream goes on: move to a large one (its bytes come along) */
        struct ltp_chunk_window* big = ltp_window_acquire(ctx, (uint64_t)LTP_WINDOW_SMALL + 1u, c->min, c->avg, c->max, &err);
        if (!big)
            return err ? err : ENOMEM;
        memcpy(big->h_win, w->h_win, (size_t)c->have);
        ltp_window_release(w);
        c->w = w = big;
    }
    if (c->have == 0)
        return 0;

    uint64_t total = 0;
    if (w->cls == 0 && c->have <= LTP_WINDOW_SMALL && batching_enabled())
    {
        /* a small window: submitted together with the other threads' (plugin_batch.c) */
        err = ltp_batch_chunk_hash(w, c->have, c->min, c->avg, c->max, &total);
        if (err)
            return err;
        if (total > w->ccap)
            return EIO;
    }
    else
    {
        err = lthip_plan_resize_single(ctx, w->plan, c->have);
        if (!err)
            err = lthip_copy_h2d(ctx, w->d_win, w->h_win, (size_t)c->have);
        if (!err)
            err = lthip_chunk_hash(ctx, w->plan, w->d_win, w->d_off, w->d_len, w->d_hash, w->d_first, &total);
        if (err)
            return err;
        if (total > w->ccap)
            return EIO;
        err = lthip_copy_d2h(ctx, w->h_off, w->d_off, (size_t)total * 8);
        if (!err) err = lthip_copy_d2h(ctx, w->h_len, w->d_len, (size_t)total * 4);
        if (!err) err = lthip_copy_d2h(ctx, w->h_hash, w->d_hash, (size_t)total * 8);
        if (!err) err = lthip_ctx_sync(ctx);
        if (err)
            return err;
    }
    c->ntotal = (uint32_t)total;
    if (c->eof)
        c->nfinal = c->ntotal;
    else
    {
        /* a cut decision looks at most `max` bytes ahead of the chunk start */
        uint32_t n = 0;
        while (n < c->ntotal && w->h_off[n] + c->max <= c->have)
            ++n;
        c->nfinal = n;
        if (n == 0)
            return EIO; /* cannot happen while the window holds 4 * max bytes: never hand out an early end of stream instead */
    }
    c->pub.base = w->h_win;
    c->pub.size = c->have;
    c->pub.offsets = w->h_off;
    c->pub.lens = w->h_len;
    c->pub.hashes = w->h_hash;
    c->pub.count = c->nfinal;
    ltp_window_publish(c->slot, &c->pub);
    ltp_window_set_current(c->slot);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * API functions
 * ------------------------------------------------------------------------------------------------- */
static void HipChunker_Dispose(struct Longtail_API* base_api)
{
    struct HipChunkerAPI* api = (struct HipChunkerAPI*)base_api;
    if (!api)
        return;
    pthread_mutex_lock(&api->lock);
    while (api->pool_count)
        chunker_free(api->pool[--api->pool_count]);
    pthread_mutex_unlock(&api->lock);
    pthread_mutex_destroy(&api->lock);
    ltp_free(api);
    if (__atomic_sub_fetch(&g_api_count, 1, __ATOMIC_ACQ_REL) == 0)
    {
        ltp_batch_shutdown();
        ltp_window_pool_trim();
    }
}

static int HipChunker_GetMinChunkSize(struct Longtail_ChunkerAPI* chunker_api, uint32_t* out_min_chunk_size)
{
    if (!chunker_api || !out_min_chunk_size)
        return EINVAL;
    *out_min_chunk_size = 48u; /* ChunkerWindowSize, hpcdcchunker.c:12,343 */
    return 0;
}

static int HipChunker_CreateChunker(struct Longtail_ChunkerAPI* chunker_api, uint32_t min_chunk_size,
                                    uint32_t avg_chunk_size, uint32_t max_chunk_size,
                                    Longtail_ChunkerAPI_HChunker* out_chunker)
{
    if (!chunker_api || !out_chunker)
        return EINVAL;
    /* hpcdcchunker.c:143-146 */
    if (min_chunk_size < 48u || min_chunk_size > avg_chunk_size || avg_chunk_size > max_chunk_size)
        return EINVAL;
    /* a window holds 4 * max bytes and a part handed to the kernels is below 4 GiB: larger maxima cannot be served (the reference's
     * own buffer arithmetic is 32-bit as well, hpcdcchunker.c:148-171) */
    if ((uint64_t)max_chunk_size * 4u > 0xF0000000ull)
        return EINVAL;
    struct HipChunkerAPI* api = (struct HipChunkerAPI*)chunker_api;
    struct HipChunker* c = 0;
    pthread_mutex_lock(&api->lock);
    if (api->pool_count)
        c = api->pool[--api->pool_count];
    pthread_mutex_unlock(&api->lock);
    if (!c)
    {
        c = (struct HipChunker*)ltp_alloc("HipChunker", sizeof *c);
        if (!c)
            return ENOMEM;
        memset(c, 0, sizeof *c);
        c->slot = ltp_window_register();
    }
    c->min = min_chunk_size;
    c->avg = avg_chunk_size;
    c->max = max_chunk_size;
    c->have = 0;
    c->base = 0;
    c->eof = 0;
    c->ntotal = c->nfinal = c->next = 0;
    *out_chunker = (Longtail_ChunkerAPI_HChunker)c;
    return 0;
}

static int HipChunker_NextChunk(struct Longtail_ChunkerAPI* chunker_api, Longtail_ChunkerAPI_HChunker chunker,
                                Longtail_Chunker_Feeder feeder, void* feeder_context,
                                struct Longtail_Chunker_ChunkRange* out_chunk_range)
{
    if (!chunker_api || !chunker || !feeder || !feeder_context || !out_chunk_range)
        return EINVAL; /* hpcdcchunker.c:409-412 */
    struct HipChunker* c = (struct HipChunker*)chunker;
    if (c->next == c->nfinal && !(c->eof && c->nfinal == c->ntotal))
    {
        int feeder_failed = 0;
        int err = chunker_refill(c, feeder, feeder_context, &feeder_failed);
        if (err)
        {
            /* the reference turns a failing feeder into an empty range + ESPIPE (hpcdcchunker.c:244-248,
             * 420-423) -- mirrored; failures of the GPU path keep their own errno */
            out_chunk_range->buf = 0;
            out_chunk_range->offset = 0;
            out_chunk_range->len = 0;
            return feeder_failed ? ESPIPE : err;
        }
    }
    if (c->next == c->nfinal)
    {
        /* end of stream: {0, total, 0} + ESPIPE (hpcdcchunker.c:250-255) */
        out_chunk_range->buf = 0;
        out_chunk_range->offset = c->base + c->have;
        out_chunk_range->len = 0;
        return ESPIPE;
    }
    const uint32_t i = c->next++;
    out_chunk_range->buf = c->w->h_win + c->w->h_off[i];
    out_chunk_range->offset = c->base + c->w->h_off[i];
    out_chunk_range->len = c->w->h_len[i];
    return 0;
}

static int HipChunker_DisposeChunker(struct Longtail_ChunkerAPI* chunker_api, Longtail_ChunkerAPI_HChunker chunker)
{
    if (!chunker_api || !chunker)
        return EINVAL;
    struct HipChunkerAPI* api = (struct HipChunkerAPI*)chunker_api;
    struct HipChunker* c = (struct HipChunker*)chunker;
    chunker_drop_window(c); /* the window goes back to the pool at once: only the handle is cached */
    pthread_mutex_lock(&api->lock);
    if (api->pool_count < HIP_CHUNKER_POOL)
    {
        api->pool[api->pool_count++] = c;
        c = 0;
    }
    pthread_mutex_unlock(&api->lock);
    chunker_free(c);
    return 0;
}

static int HipChunker_NextChunkFromBuffer(struct Longtail_ChunkerAPI* chunker_api, Longtail_ChunkerAPI_HChunker chunker,
                                          const void* buffer, uint64_t buffer_size, const void** out_next_chunk_start)
{
    if (!chunker_api || !chunker || !buffer || buffer_size == 0 || !out_next_chunk_start)
        return EINVAL; /* hpcdcchunker.c:470-474 */
    struct HipChunker* c = (struct HipChunker*)chunker;
    if (buffer_size <= c->min)
    {
        *out_next_chunk_start = (const uint8_t*)buffer + buffer_size; /* :479-484 */
        return 0;
    }
    /* only the first min(size, max) bytes can influence the cut (:499) */
    struct ltp_thread_state* ts = ltp_thread_state_get();
    if (!ts)
        return ENODEV;
    const uint64_t n = buffer_size > c->max ? c->max : buffer_size;
    int err = ltp_dev_reserve(ts->ctx, &ts->d_in, (size_t)n + 64);
    if (!err)
        err = lthip_copy_h2d(ts->ctx, ts->d_in.p, buffer, (size_t)n);
    uint64_t len = 0;
    if (!err)
        err = lthip_chunk_from_buffer(ts->ctx, ts->d_in.p, n, c->min, c->avg, c->max, &len);
    if (err)
        return err;
    *out_next_chunk_start = (const uint8_t*)buffer + len;
    return 0;
}

struct Longtail_ChunkerAPI* Longtail_CreateHipChunkerAPI(void)
{
    if (lthip_device_count() <= 0)
        return 0; /* no GPU: fail loudly instead of silently chunking on the CPU */
    struct HipChunkerAPI* api = (struct HipChunkerAPI*)ltp_alloc("HipChunkerAPI", sizeof *api);
    if (!api)
        return 0;
    memset(api, 0, sizeof *api);
    api->api.m_API.Dispose = HipChunker_Dispose;
    api->api.GetMinChunkSize = HipChunker_GetMinChunkSize;
    api->api.CreateChunker = HipChunker_CreateChunker;
    api->api.NextChunk = HipChunker_NextChunk;
    api->api.DisposeChunker = HipChunker_DisposeChunker;
    api->api.NextChunkFromBuffer = HipChunker_NextChunkFromBuffer;
    pthread_mutex_init(&api->lock, 0);
    __atomic_add_fetch(&g_api_count, 1, __ATOMIC_ACQ_REL);
    return &api->api;
}
