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
 */
#include "plugin_common.h"

#define HIP_CHUNKER_WINDOW_BYTES (64u << 20) /* one reference part at target_chunk_size 65536 (src/longtail.c:2396) */
#define HIP_CHUNKER_POOL 232                 /* HPCDCCHUNKER_MAX_CACHED_CHUNKER_COUNT, hpcdcchunker.c:102 */

struct HipChunker
{
    uint32_t min, avg, max;
    /* window */
    uint8_t* h_win;   /* pinned */
    void* d_win;
    uint64_t cap;     /* bytes */
    uint64_t have;    /* valid bytes in the window */
    uint64_t base;    /* stream offset of window byte 0 */
    int eof;
    /* results of the current window */
    uint64_t* d_off;
    uint32_t* d_len;
    uint64_t* d_hash;
    uint32_t* d_first;
    uint64_t* h_off;
    uint32_t* h_len;
    uint64_t* h_hash;
    uint64_t ccap;
    uint32_t ntotal, nfinal, next;
    /* cached plan */
    lthip_plan* plan;
    uint64_t plan_size;
    uint32_t plan_min, plan_avg, plan_max;
    int slot; /* window registry slot */
};

struct HipChunkerAPI
{
    struct Longtail_ChunkerAPI api;
    pthread_mutex_t lock;
    struct HipChunker* pool[HIP_CHUNKER_POOL];
    uint32_t pool_count;
};

static void chunker_release_buffers(struct HipChunker* c)
{
    if (c->plan)
        lthip_plan_destroy(0, c->plan); /* the creating thread (and its context) may be gone */
    c->plan = 0;
    lthip_free_pinned(0, c->h_win);
    lthip_free_device(0, c->d_win);
    lthip_free_device(0, c->d_off);
    lthip_free_device(0, c->d_len);
    lthip_free_device(0, c->d_hash);
    lthip_free_device(0, c->d_first);
    lthip_free_pinned(0, c->h_off);
    lthip_free_pinned(0, c->h_len);
    lthip_free_pinned(0, c->h_hash);
    c->h_win = 0;
    c->d_win = 0;
    c->d_off = 0;
    c->d_len = 0;
    c->d_hash = 0;
    c->d_first = 0;
    c->h_off = 0;
    c->h_len = 0;
    c->h_hash = 0;
    c->cap = 0;
    c->ccap = 0;
}

static void chunker_free(struct HipChunker* c)
{
    if (!c)
        return;
    ltp_window_unregister(c->slot);
    chunker_release_buffers(c);
    ltp_free(c);
}

static int chunker_reserve(struct HipChunker* c, lthip_ctx* ctx)
{
    uint64_t cap = HIP_CHUNKER_WINDOW_BYTES;
    if (cap < (uint64_t)c->max * 4u)
        cap = (uint64_t)c->max * 4u; /* the reference buffers 4*max (hpcdcchunker.c:148) */
    if (cap > 0xF0000000ull)
        cap = 0xF0000000ull;
    const uint64_t ccap = cap / c->min + 2;
    if (c->cap >= cap && c->ccap >= ccap)
        return 0;
    chunker_release_buffers(c);
    int err = 0;
    if (!err) err = lthip_malloc_pinned(ctx, cap, (void**)&c->h_win);
    if (!err) err = lthip_malloc_device(ctx, cap + 64, &c->d_win);
    if (!err) err = lthip_malloc_device(ctx, ccap * 8, (void**)&c->d_off);
    if (!err) err = lthip_malloc_device(ctx, ccap * 4, (void**)&c->d_len);
    if (!err) err = lthip_malloc_device(ctx, ccap * 8, (void**)&c->d_hash);
    if (!err) err = lthip_malloc_device(ctx, 16, (void**)&c->d_first);
    if (!err) err = lthip_malloc_pinned(ctx, ccap * 8, (void**)&c->h_off);
    if (!err) err = lthip_malloc_pinned(ctx, ccap * 4, (void**)&c->h_len);
    if (!err) err = lthip_malloc_pinned(ctx, ccap * 8, (void**)&c->h_hash);
    if (err)
    {
        chunker_release_buffers(c);
        return err;
    }
    c->cap = cap;
    c->ccap = ccap;
    return 0;
}

/* fill the window from the feeder and run the GPU over it */
static int chunker_refill(struct HipChunker* c, Longtail_Chunker_Feeder feeder, void* feeder_context, int* feeder_failed)
{
    lthip_ctx* ctx = ltp_thread_ctx();
    if (!ctx)
        return ENODEV;
    int err = chunker_reserve(c, ctx);
    if (err)
        return err;

    /* keep the bytes whose chunking is not final yet */
    const uint64_t keep_from = c->nfinal < c->ntotal ? c->h_off[c->nfinal] : c->have;
    {
        struct ltp_window none;
        memset(&none, 0, sizeof none);
        ltp_window_publish(c->slot, &none); /* ranges of the old window are no longer valid */
    }
    if (keep_from < c->have && keep_from > 0)
        memmove(c->h_win, c->h_win + keep_from, (size_t)(c->have - keep_from));
    c->base += keep_from;
    c->have -= keep_from;
    c->ntotal = c->nfinal = c->next = 0;

    while (c->have < c->cap && !c->eof)
    {
        uint64_t want = c->cap - c->have;
        if (want > 0x7FFFFFFFu)
            want = 0x7FFFFFFFu;
        uint32_t got = 0;
        err = feeder(feeder_context, (Longtail_ChunkerAPI_HChunker)c, (uint32_t)want, (char*)c->h_win + c->have, &got);
        if (err)
        {
            *feeder_failed = 1;
            return err;
        }
        if (got == 0)
            c->eof = 1;
        c->have += got;
    }
    if (c->have == 0)
        return 0;

    if (!c->plan || c->plan_size != c->have || c->plan_min != c->min || c->plan_avg != c->avg || c->plan_max != c->max)
    {
        if (c->plan)
            lthip_plan_destroy(0, c->plan);
        c->plan = 0;
        const uint64_t off0 = 0, sz = c->have;
        err = lthip_plan_create(ctx, 1, &off0, &sz, c->min, c->avg, c->max, &c->plan);
        if (err)
            return err;
        c->plan_size = c->have;
        c->plan_min = c->min;
        c->plan_avg = c->avg;
        c->plan_max = c->max;
    }
    uint64_t total = 0;
    err = lthip_copy_h2d(ctx, c->d_win, c->h_win, (size_t)c->have);
    if (!err)
        err = lthip_chunk_hash(ctx, c->plan, c->d_win, c->d_off, c->d_len, c->d_hash, c->d_first, &total);
    if (err)
        return err;
    if (total > c->ccap)
        return EIO;
    err = lthip_copy_d2h(ctx, c->h_off, c->d_off, (size_t)total * 8);
    if (!err) err = lthip_copy_d2h(ctx, c->h_len, c->d_len, (size_t)total * 4);
    if (!err) err = lthip_copy_d2h(ctx, c->h_hash, c->d_hash, (size_t)total * 8);
    if (!err) err = lthip_ctx_sync(ctx);
    if (err)
        return err;
    c->ntotal = (uint32_t)total;
    if (c->eof)
        c->nfinal = c->ntotal;
    else
    {
        /* a cut decision looks at most `max` bytes ahead of the chunk start */
        uint32_t n = 0;
        while (n < c->ntotal && c->h_off[n] + c->max <= c->have)
            ++n;
        c->nfinal = n;
    }
    struct ltp_window w;
    w.base = c->h_win;
    w.size = c->have;
    w.offsets = c->h_off;
    w.lens = c->h_len;
    w.hashes = c->h_hash;
    w.count = c->nfinal;
    ltp_window_publish(c->slot, &w);
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
    out_chunk_range->buf = c->h_win + c->h_off[i];
    out_chunk_range->offset = c->base + c->h_off[i];
    out_chunk_range->len = c->h_len[i];
    return 0;
}

static int HipChunker_DisposeChunker(struct Longtail_ChunkerAPI* chunker_api, Longtail_ChunkerAPI_HChunker chunker)
{
    if (!chunker_api || !chunker)
        return EINVAL;
    struct HipChunkerAPI* api = (struct HipChunkerAPI*)chunker_api;
    struct HipChunker* c = (struct HipChunker*)chunker;
    struct ltp_window none;
    memset(&none, 0, sizeof none);
    ltp_window_publish(c->slot, &none);
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
    return &api->api;
}
