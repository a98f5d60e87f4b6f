/* plugin_codec_batch.c -- one GPU submission for the blocks of MANY concurrent Compress / Decompress calls.
 *
 * CompressBlock / DecompressBlock (lib/compressblockstore/longtail_compressblockstore.c:67-141, 271-338) hand the CompressionAPI ONE
 * stored block per call, from as many bikeshed workers as the job system has.  One block is a poor unit of work for the device: the
 * match finder's lane parser is one persistent workgroup per CU (an 8 MiB block is 128 groups: half the chip), every call costs a
 * dozen launches -- 32 workers calling the bulk entry points with one block each got 7-9 GB/s out of a codec that does 240 GB/s on a
 * thousand blocks per call (profiles/r03_plugin_rate.txt: WriteContent with the HIP LZ4 object SLOWER than the reference's own).
 * Here a worker uploads its block on its own stream into its own device buffer, queues {where the block is, where the payload goes}
 * and sleeps; a dispatcher thread takes WHATEVER IS QUEUED for the same operation (up to 64 blocks), calls the bulk entry point once --
 * its offsets are relative to the lowest of the queued addresses, the blocks live in the workers' own allocations -- reads the sizes
 * back and wakes the workers, which fetch their payloads on their own streams.  No timer: while a submission runs the other workers
 * upload and queue up, so the batch grows with the load; a single caller gets a batch of one.  LONGTAIL_HIP_CODEC_BATCH=0: every call
 * on its own (rounds 1-2).  */
#include "plugin_common.h"

#define LTC_MAX 64
#ifndef LTC_MIN
#define LTC_MIN 16 /* up to this many queued blocks a submission takes them all */
#endif

struct ltc_req
{
    int kind; /* codec * 2 + decompress */
    const void* d_in;
    void* d_out;
    uint32_t n, cap, produced;
    int done, err;
    pthread_cond_t cv; /* the caller's own (with g_lock): a finished submission wakes its callers, not every waiting worker (plugin_batch.c) */
    struct ltc_req* next;
};

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_work = PTHREAD_COND_INITIALIZER;
static pthread_cond_t g_done = PTHREAD_COND_INITIALIZER;
static struct ltc_req *g_head, *g_tail;
static pthread_t g_thread;
static int g_running, g_stop, g_joining;
static uint64_t g_stat_batches, g_stat_blocks;

static int run_batch(lthip_ctx* ctx, void* d_sizes, uint32_t* h_sizes, struct ltc_req** reqs, uint32_t n)
{
    uint64_t s_off[LTC_MAX], d_off[LTC_MAX];
    uint32_t s_size[LTC_MAX], d_cap[LTC_MAX];
    uintptr_t s_base = (uintptr_t)reqs[0]->d_in, d_base = (uintptr_t)reqs[0]->d_out;
    for (uint32_t i = 1; i < n; ++i)
    {
        if ((uintptr_t)reqs[i]->d_in < s_base)
            s_base = (uintptr_t)reqs[i]->d_in;
        if ((uintptr_t)reqs[i]->d_out < d_base)
            d_base = (uintptr_t)reqs[i]->d_out;
    }
    for (uint32_t i = 0; i < n; ++i)
    {
        s_off[i] = (uint64_t)((uintptr_t)reqs[i]->d_in - s_base);
        d_off[i] = (uint64_t)((uintptr_t)reqs[i]->d_out - d_base);
        s_size[i] = reqs[i]->n;
        d_cap[i] = reqs[i]->cap;
    }
    int err;
    switch (reqs[0]->kind)
    {
    case 0:
        err = lthip_lz4_compress_blocks(ctx, (const void*)s_base, n, s_off, s_size, (void*)d_base, d_off, d_cap, (uint32_t*)d_sizes, 0);
        break;
    case 1:
        err = lthip_lz4_decompress_blocks(ctx, (const void*)s_base, n, s_off, s_size, (void*)d_base, d_off, d_cap, (uint32_t*)d_sizes);
        break;
    case 2:
    case 4:
    case 5:
        err = lthip_zstd_compress_blocks_q(ctx, (const void*)s_base, n, s_off, s_size, (void*)d_base, d_off, d_cap, (uint32_t*)d_sizes,
                                           reqs[0]->kind == 2 ? 0 : reqs[0]->kind - 3);
        break;
    default:
        err = lthip_zstd_decompress_blocks(ctx, (const void*)s_base, n, s_off, s_size, (void*)d_base, d_off, d_cap, (uint32_t*)d_sizes);
        break;
    }
    if (!err)
        err = lthip_copy_d2h(ctx, h_sizes, d_sizes, (size_t)n * 4);
    if (!err)
        err = lthip_ctx_sync(ctx);
    if (err)
        return err;
    for (uint32_t i = 0; i < n; ++i)
        reqs[i]->produced = h_sizes[i];
    return 0;
}

static void* dispatcher(void* arg)
{
    (void)arg;
    lthip_ctx* ctx = 0;
    void *d_sizes = 0, *h_sizes = 0;
    for (;;)
    {
        struct ltc_req* reqs[LTC_MAX];
        uint32_t n = 0;
        pthread_mutex_lock(&g_lock);
        while (!g_head && !g_stop)
            pthread_cond_wait(&g_work, &g_lock);
        if (!g_head && g_stop)
        {
            pthread_mutex_unlock(&g_lock);
            break;
        }
        /* What is queued for the operation of the oldest request, in arrival order -- but only the older HALF of it once more than
         * LTC_MIN are waiting; the rest waits for the next round.  Compress is a blocking call: a worker cannot overlap its own
         * upload, kernel and download, only different workers can overlap theirs.  When one submission takes everybody, the W
         * workers move in lock step -- all upload, all wait for the kernels, all download -- and the link runs one direction at a
         * time (round 5: 21 GB/s of WriteContent at W = 32 on a link that does 55 each way).  Taking half leaves the other half
         * uploading while these compute and downloading while the next ones upload: the workers fall into staggered groups and both
         * directions of the link and the kernels run at once. */
        uint32_t queued = 0;
        for (const struct ltc_req* q = g_head; q; q = q->next)
            ++queued;
        const uint32_t take = queued <= LTC_MIN ? queued : (queued + 1) / 2 > LTC_MIN ? (queued + 1) / 2 : LTC_MIN;
        struct ltc_req *keep_head = 0, *keep_tail = 0, *r = g_head;
        const int kind = r->kind;
        while (r)
        {
            struct ltc_req* nx = r->next;
            r->next = 0;
            if (n < LTC_MAX && n < take && r->kind == kind)
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

        /* the dispatcher's own context and its two small buffers: made for the first submission and, should that fail (out of
         * memory at that moment), made AGAIN for the next one -- a failed set-up is the error of the requests that met it, not a
         * state the CompressionAPI objects stay in (tests/test_gpu_alloc_failures.py) */
        int init_err = 0;
        if (!ctx && lthip_ctx_create(ltp_device(), LTHIP_STREAM_PRIVATE, &ctx) != 0)
        {
            ctx = 0;
            init_err = ENODEV;
        }
        if (!init_err && !d_sizes)
            init_err = lthip_malloc_device(ctx, LTC_MAX * 4 + 64, &d_sizes);
        if (!init_err && !h_sizes)
            init_err = lthip_malloc_pinned(ctx, LTC_MAX * 4 + 64, &h_sizes);
        int errs[LTC_MAX];
        uint32_t submissions = 1;
        const int err = init_err ? init_err : run_batch(ctx, d_sizes, (uint32_t*)h_sizes, reqs, n);
        for (uint32_t i = 0; i < n; ++i)
            errs[i] = err;
        if (err && !init_err && n > 1)
        {
            /* A library-level failure of a shared submission (scratch ENOMEM, an EINVAL one request's arguments caused) is not every
             * caller's failure: the up to 64 requests are unrelated blocks of unrelated threads.  Run them again one at a time, so
             * that only the offending request reports the error (a transient ENOMEM may well pass at a 64th of the size). */
            for (uint32_t i = 0; i < n; ++i)
                errs[i] = run_batch(ctx, d_sizes, (uint32_t*)h_sizes, reqs + i, 1);
            submissions += n;
        }

        pthread_mutex_lock(&g_lock);
        g_stat_batches += submissions;
        g_stat_blocks += n;
        for (uint32_t i = 0; i < n; ++i)
        {
            reqs[i]->err = errs[i];
            reqs[i]->done = 1;
            pthread_cond_signal(&reqs[i]->cv);
        }
        pthread_mutex_unlock(&g_lock);
    }
    if (ctx)
    {
        lthip_free_device(ctx, d_sizes);
        lthip_free_pinned(ctx, h_sizes);
        lthip_ctx_destroy(ctx);
    }
    return 0;
}

/* The block at d_in (n bytes, uploaded and synchronised by the caller) through the codec into d_out (cap bytes), together with
 * whatever the other threads have queued; blocks until done.  *produced as the bulk entry points report it. */
int ltp_codec_batch(int codec, int decompress, int quality, const void* d_in, uint32_t n, void* d_out, uint32_t cap, uint32_t* produced)
{
    struct ltc_req r;
    memset(&r, 0, sizeof r);
    r.kind = codec * 2 + (decompress ? 1 : 0); /* 0 lz4 compress, 1 lz4 decompress, 2 zstd compress, 3 zstd decompress ... */
    if (codec == 1 && !decompress && quality > 0)
        r.kind = 3 + quality; /* ... 4 / 5 zstd compress at LTHIP_ZSTD_Q_HIGH / _MAX: a submission carries one kind */
    r.d_in = d_in;
    r.d_out = d_out;
    r.n = n;
    r.cap = cap;
    pthread_cond_init(&r.cv, 0);
    pthread_mutex_lock(&g_lock);
    while (g_joining) /* a shutdown is collecting the previous dispatcher: start the next one only when it is gone */
        pthread_cond_wait(&g_done, &g_lock);
    if (!g_running)
    {
        g_stop = 0;
        if (pthread_create(&g_thread, 0, dispatcher, 0) != 0)
        {
            pthread_mutex_unlock(&g_lock);
            pthread_cond_destroy(&r.cv);
            return EAGAIN;
        }
        g_running = 1;
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
    pthread_cond_destroy(&r.cv);
    *produced = r.produced;
    return r.err;
}

/* stops the dispatcher (called when the last HIP CompressionAPI is disposed: no request can be pending) */
void ltp_codec_batch_shutdown(void)
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
    pthread_mutex_lock(&g_lock);
    g_joining = 0;
    pthread_cond_broadcast(&g_done);
    pthread_mutex_unlock(&g_lock);
}

void Longtail_Hip_CodecBatchStats(uint64_t* out_submissions, uint64_t* out_blocks)
{
    pthread_mutex_lock(&g_lock);
    if (out_submissions)
        *out_submissions = g_stat_batches;
    if (out_blocks)
        *out_blocks = g_stat_blocks;
    pthread_mutex_unlock(&g_lock);
}
