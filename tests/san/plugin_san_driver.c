/* plugin_san_driver.c -- TEST INFRASTRUCTURE: drives the plain-C plugin layer (longtail_amd/csrc/plugin/ sources, compiled with
 * -fsanitize=address,undefined against tests/san/mock_lthip.c) the way the reference core does -- NextChunk / HashBuffer
 * alternately (src/longtail.c:2231-2296), many threads on one API object, streaming hash contexts, per-block Compress /
 * Decompress from unaligned host buffers -- and checks results against the oracle.  What this run is for: the host logic
 * (window pool and its blocking, window moves, refill bookkeeping, registry, error paths) under the sanitizers; the kernels are
 * covered by the -m gpu tests.  Exit code 0 = all checks passed and nothing leaked (ASan's leak check runs at exit). */
#include "../../include/longtail_hip.h"
#include "../../oracle/oracle.h"

#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void mock_fail_alloc_after(int n);

#define CHECK(c)                                                         \
    do                                                                   \
    {                                                                    \
        if (!(c))                                                        \
        {                                                                \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
            exit(2);                                                     \
        }                                                                \
    } while (0)

struct feed
{
    const uint8_t* data;
    uint64_t size, pos, fail_at;
    int fail_errno;
};

static int feeder(void* ctx, Longtail_ChunkerAPI_HChunker ch, uint32_t want, char* buf, uint32_t* got)
{
    struct feed* f = (struct feed*)ctx;
    (void)ch;
    if (f->fail_errno && f->pos >= f->fail_at)
        return f->fail_errno;
    uint64_t n = f->size - f->pos;
    if (n > want)
        n = want;
    if (f->fail_errno && f->pos + n > f->fail_at)
        n = f->fail_at - f->pos;
    memcpy(buf, f->data + f->pos, (size_t)n);
    f->pos += n;
    *got = (uint32_t)n;
    return 0;
}

static struct Longtail_ChunkerAPI* g_chunker;
static struct Longtail_HashAPI* g_hash;

/* chunk + hash `size` bytes like DynamicChunking and compare with the oracle */
static void chunk_and_check(const uint8_t* data, uint64_t size, uint32_t mn, uint32_t av, uint32_t mx)
{
    const uint64_t cap = size / mn + 8;
    uint32_t* lens = (uint32_t*)malloc(4 * cap);
    const uint64_t n = lto_hpcdc_chunk_stream(data, size, mn, av, mx, lens, cap);
    Longtail_ChunkerAPI_HChunker c = 0;
    CHECK(g_chunker->CreateChunker(g_chunker, mn, av, mx, &c) == 0);
    struct feed f = {data, size, 0, 0, 0};
    uint64_t i = 0, off = 0;
    for (;;)
    {
        struct Longtail_Chunker_ChunkRange r;
        const int err = g_chunker->NextChunk(g_chunker, c, feeder, &f, &r);
        if (err == ESPIPE)
        {
            CHECK(r.len == 0 && r.offset == size && r.buf == 0);
            break;
        }
        CHECK(err == 0 && i < n && r.len == lens[i] && r.offset == off && memcmp(r.buf, data + off, r.len) == 0);
        uint64_t h = 0;
        CHECK(g_hash->HashBuffer(g_hash, r.len, r.buf, &h) == 0);
        CHECK(h == lto_blake3_u64(data + off, r.len));
        off += r.len;
        ++i;
    }
    CHECK(i == n);
    CHECK(g_chunker->DisposeChunker(g_chunker, c) == 0);
    free(lens);
}

struct job
{
    uint64_t seed, size;
    uint32_t target;
};

static void* worker(void* arg)
{
    const struct job* j = (const struct job*)arg;
    uint8_t* d = (uint8_t*)malloc(j->size + 1);
    lto_synth_fill(d, j->size, j->seed, 0, (int)(j->seed % 3));
    const uint32_t t = j->target;
    chunk_and_check(d, j->size, t / 8 > 48 ? t / 8 : 48, t / 2 > 48 ? t / 2 : 48, t * 2 > 48 ? t * 2 : 48);
    free(d);
    return 0;
}

struct codec_job
{
    struct Longtail_CompressionAPI* api;
    uint32_t tag;
    uint64_t seed;
    size_t size;
    int rounds, failed;
};

static void* codec_worker(void* arg)
{
    struct codec_job* j = (struct codec_job*)arg;
    uint8_t* d = (uint8_t*)malloc(j->size);
    const size_t cap = j->api->GetMaxCompressedSize(j->api, j->tag, j->size);
    uint8_t* out = (uint8_t*)malloc(cap + 8);
    uint8_t* back = (uint8_t*)malloc(j->size + 8);
    for (int r = 0; r < j->rounds; ++r)
    {
        lto_synth_fill(d, j->size, j->seed + (uint64_t)r, 0, 1);
        size_t got = 0, got2 = 0;
        if (j->api->Compress(j->api, j->tag, (const char*)d, (char*)out + 4, j->size, cap, &got) != 0 || got == 0 ||
            j->api->Decompress(j->api, (const char*)out + 4, (char*)back + 4, got, j->size, &got2) != 0 || got2 != j->size ||
            memcmp(back + 4, d, j->size) != 0)
            j->failed = 1;
    }
    free(d);
    free(out);
    free(back);
    return 0;
}

/* 8. a chunker DISPOSED FROM ANOTHER THREAD while the thread that drew its last chunk keeps calling HashBuffer (on memory of its
 * own: the chunk's bytes died with the chunker, reading them would be the caller's bug).  The hasher's lock-free look-up of "my
 * thread's window" (plugin_common.c ltp_window_lookup) must not read the slot -- or the pinned tables it points to -- while the
 * disposer recycles or frees them.  Run under ASan and, `make run-tsan`, under ThreadSanitizer. */
struct handoff
{
    pthread_mutex_t m;
    pthread_cond_t cv;
    Longtail_ChunkerAPI_HChunker pending[4];
    int n, done;
};

static void* disposer(void* arg)
{
    struct handoff* h = (struct handoff*)arg;
    for (;;)
    {
        pthread_mutex_lock(&h->m);
        while (h->n == 0 && !h->done)
            pthread_cond_wait(&h->cv, &h->m);
        if (h->n == 0 && h->done)
        {
            pthread_mutex_unlock(&h->m);
            return 0;
        }
        Longtail_ChunkerAPI_HChunker c = h->pending[--h->n];
        pthread_cond_broadcast(&h->cv);
        pthread_mutex_unlock(&h->m);
        CHECK(g_chunker->DisposeChunker(g_chunker, c) == 0);
    }
}

static void dispose_from_another_thread(void)
{
    struct handoff h;
    pthread_mutex_init(&h.m, 0);
    pthread_cond_init(&h.cv, 0);
    h.n = 0;
    h.done = 0;
    pthread_t th;
    CHECK(pthread_create(&th, 0, disposer, &h) == 0);
    const uint64_t size = 400000;
    uint8_t* d = (uint8_t*)malloc(size);
    uint8_t* mine = (uint8_t*)malloc(131072);
    lto_synth_fill(d, size, 4242, 0, 1);
    for (int round = 0; round < 200; ++round)
    {
        Longtail_ChunkerAPI_HChunker c = 0;
        CHECK(g_chunker->CreateChunker(g_chunker, 8192, 32768, 131072, &c) == 0);
        struct feed f = {d, size, 0, 0, 0};
        struct Longtail_Chunker_ChunkRange r;
        CHECK(g_chunker->NextChunk(g_chunker, c, feeder, &f, &r) == 0 && r.len > 0);
        uint64_t want = 0, got = 0;
        CHECK(g_hash->HashBuffer(g_hash, r.len, r.buf, &want) == 0); /* from the window's tables */
        const uint32_t len = r.len;
        memcpy(mine, r.buf, len);
        pthread_mutex_lock(&h.m);
        while (h.n == 4)
            pthread_cond_wait(&h.cv, &h.m);
        h.pending[h.n++] = c; /* the other thread disposes it ... */
        pthread_cond_broadcast(&h.cv);
        pthread_mutex_unlock(&h.m);
        for (int k = 0; k < 3; ++k) /* ... while this one goes on hashing: its "current window" is being torn down */
        {
            CHECK(g_hash->HashBuffer(g_hash, len, mine, &got) == 0 && got == want);
        }
    }
    pthread_mutex_lock(&h.m);
    h.done = 1;
    pthread_cond_broadcast(&h.cv);
    pthread_mutex_unlock(&h.m);
    pthread_join(th, 0);
    free(d);
    free(mine);
    pthread_mutex_destroy(&h.m);
    pthread_cond_destroy(&h.cv);
}

int main(void)
{
    g_chunker = Longtail_CreateHipChunkerAPI();
    g_hash = Longtail_CreateHipBlake3HashAPI();
    struct Longtail_CompressionAPI* lz4 = Longtail_CreateHipLZ4CompressionAPI();
    struct Longtail_CompressionAPI* zstd = Longtail_CreateHipZStdCompressionAPI();
    CHECK(g_chunker && g_hash && lz4 && zstd);
    uint32_t mn = 0;
    CHECK(g_chunker->GetMinChunkSize(g_chunker, &mn) == 0 && mn == 48);
    CHECK(g_hash->GetIdentifier(g_hash) == 0x626c6b33u);

    /* 1. sizes around the window classes (2 MiB small, 64 MiB large: a stream that fills the small window moves), several windows */
    {
        const uint64_t sizes[] = {0, 1, 47, 48, 8192, 100000, (2u << 20) - 1, 2u << 20, (2u << 20) + 1, 5u << 20, (64u << 20) + 12345, (130u << 20) + 7};
        uint8_t* d = (uint8_t*)malloc((130u << 20) + 8);
        lto_synth_fill(d, (130u << 20) + 7, 99, 0, 1);
        for (unsigned i = 0; i < sizeof sizes / sizeof sizes[0]; ++i)
            chunk_and_check(d, sizes[i], 8192, 32768, 131072);
        chunk_and_check(d, 300000, 48, 48, 48);       /* smallest parameters: tables sized for 48-byte chunks */
        chunk_and_check(d, 3u << 20, 48, 100, 300);
        chunk_and_check(d, 40u << 20, 1u << 20, 4u << 20, 16u << 20); /* 4 * max = 64 MiB: exactly a large window */
        chunk_and_check(d, 100u << 20, 2u << 20, 8u << 20, 20u << 20); /* 4 * max = 80 MiB: a private window */
        free(d);
    }
    /* 2. parameter contract (hpcdcchunker.c:143-146) */
    {
        Longtail_ChunkerAPI_HChunker c = 0;
        CHECK(g_chunker->CreateChunker(g_chunker, 47, 100, 200, &c) == EINVAL);
        CHECK(g_chunker->CreateChunker(g_chunker, 100, 99, 200, &c) == EINVAL);
        CHECK(g_chunker->CreateChunker(g_chunker, 100, 200, 199, &c) == EINVAL);
        CHECK(g_chunker->CreateChunker(g_chunker, 1u << 20, 1u << 29, 0x40000000u, &c) == EINVAL); /* 4 * max does not fit a window */
    }
    /* 3. a failing feeder: empty range + ESPIPE, the handle stays disposable (hpcdcchunker.c:244-248, 420-423) */
    {
        uint8_t* d = (uint8_t*)malloc(70u << 20);
        lto_synth_fill(d, 70u << 20, 5, 0, 0);
        const uint64_t fails[] = {0, 1, 100000, (2u << 20) + 5, (64u << 20) + 4096};
        for (unsigned i = 0; i < 5; ++i)
        {
            Longtail_ChunkerAPI_HChunker c = 0;
            CHECK(g_chunker->CreateChunker(g_chunker, 8192, 32768, 131072, &c) == 0);
            struct feed f = {d, 70u << 20, 0, fails[i], EIO};
            uint64_t total = 0;
            for (;;)
            {
                struct Longtail_Chunker_ChunkRange r = {(const uint8_t*)1, 7, 7};
                const int err = g_chunker->NextChunk(g_chunker, c, feeder, &f, &r);
                if (err)
                {
                    CHECK(err == ESPIPE && r.buf == 0 && r.offset == 0 && r.len == 0);
                    break;
                }
                total += r.len;
            }
            CHECK(total <= fails[i]);
            CHECK(g_chunker->DisposeChunker(g_chunker, c) == 0);
        }
        free(d);
    }
    /* 4. many threads on one API object with a tiny pool (LONGTAIL_HIP_*_WINDOWS set by the test): windows are waited for */
    {
        enum { T = 12 };
        pthread_t th[T];
        struct job jobs[T];
        for (int i = 0; i < T; ++i)
        {
            jobs[i].seed = 1000 + (uint64_t)i;
            jobs[i].size = i % 3 == 0 ? (3u << 20) + 17u * (uint64_t)i : 200000u + 4099u * (uint64_t)i;
            jobs[i].target = i % 2 ? 65536 : 4096;
            CHECK(pthread_create(&th[i], 0, worker, &jobs[i]) == 0);
        }
        for (int i = 0; i < T; ++i)
            pthread_join(th[i], 0);
        CHECK(Longtail_Hip_PinnedBytes() <= (3u * (2u << 20) + 2u * (64u << 20)) * 2u); /* the caps the test set, with table slack */
    }
    /* 5. foreign buffers, empty input, the streaming trio, the error latch */
    {
        uint8_t* d = (uint8_t*)malloc(300001);
        lto_synth_fill(d, 300001, 8, 0, 1);
        uint64_t h = 1;
        CHECK(g_hash->HashBuffer(g_hash, 0, d, &h) == 0 && h == lto_blake3_u64(0, 0));
        for (uint32_t n = 1; n <= 300001; n = n * 3 + 1)
        {
            CHECK(g_hash->HashBuffer(g_hash, n, d + 1, &h) == 0 && h == lto_blake3_u64(d + 1, n)); /* unaligned source */
        }
        Longtail_HashAPI_HContext hc = 0;
        CHECK(g_hash->BeginContext(g_hash, &hc) == 0);
        g_hash->Hash(g_hash, hc, 1000, d);
        g_hash->Hash(g_hash, hc, 1, d + 1000);
        g_hash->Hash(g_hash, hc, 299000, d + 1001);
        CHECK(g_hash->EndContext(g_hash, hc) == lto_blake3_u64(d, 300001));
        CHECK(Longtail_Hip_GetLastError() == 0);
        {
            /* a stream of several batches (1 MiB each) in pieces that straddle them, ending exactly on a batch boundary and not */
            for (int round = 0; round < 2; ++round)
            {
                const size_t n = round ? (3u << 20) : (3u << 20) + 12345u;
                uint8_t* s = (uint8_t*)malloc(n);
                lto_synth_fill(s, n, 77 + round, 0, 1);
                CHECK(g_hash->BeginContext(g_hash, &hc) == 0);
                size_t o = 0, step = 700001;
                while (o < n)
                {
                    const size_t k = n - o < step ? n - o : step;
                    g_hash->Hash(g_hash, hc, (uint32_t)k, s + o);
                    o += k;
                }
                CHECK(g_hash->EndContext(g_hash, hc) == lto_blake3_u64(s, (uint32_t)n));
                CHECK(Longtail_Hip_GetLastError() == 0);
                free(s);
            }
        }
        CHECK(g_hash->BeginContext(g_hash, &hc) == 0);
        g_hash->Hash(g_hash, hc, 100, d);
        mock_fail_alloc_after(0); /* the device buffer of the final hash cannot be had */
        {
            /* force a fresh device allocation: a larger input than any before on this thread */
            uint8_t* big = (uint8_t*)calloc(1, 4u << 20);
            g_hash->Hash(g_hash, hc, 4u << 20, big);
            free(big);
        }
        CHECK(g_hash->EndContext(g_hash, hc) == 0);
        CHECK(Longtail_Hip_GetLastError() == ENOMEM && Longtail_Hip_GetLastError() == 0);
        mock_fail_alloc_after(-1);
        free(d);
    }
    /* 6. Compress / Decompress on 4-byte aligned destinations (compressblockstore.c:117-125), error mapping */
    {
        const size_t n = 700001;
        uint8_t* d = (uint8_t*)malloc(n);
        lto_synth_fill(d, n, 21, 0, 1);
        struct Longtail_CompressionAPI* apis[2] = {lz4, zstd};
        const uint32_t tags[2] = {0x6c7a3432u, 0x7a746432u};
        for (int a = 0; a < 2; ++a)
        {
            const size_t cap = apis[a]->GetMaxCompressedSize(apis[a], tags[a], n);
            uint8_t* out = (uint8_t*)malloc(cap + 8);
            uint8_t* back = (uint8_t*)malloc(n + 8);
            size_t got = 0, got2 = 0;
            CHECK(apis[a]->Compress(apis[a], tags[a], (const char*)d, (char*)out + 4, n, cap, &got) == 0 && got > 0 && got < n);
            CHECK(apis[a]->Decompress(apis[a], (const char*)out + 4, (char*)back + 4, got, n, &got2) == 0 && got2 == n);
            CHECK(memcmp(back + 4, d, n) == 0);
            CHECK(apis[a]->Compress(apis[a], tags[a] ^ 0xFF000000u, (const char*)d, (char*)out, n, cap, &got) == EINVAL);
            memset(out, 0xFF, 100);
            CHECK(apis[a]->Decompress(apis[a], (const char*)out, (char*)back, 100, n, &got2) == (a == 0 ? EBADF : EINVAL));
            free(out);
            free(back);
        }
        size_t got = 0;
        uint8_t small[64];
        CHECK(lz4->Compress(lz4, tags[0], (const char*)d, (char*)small, 50000, 60, &got) == ENOMEM); /* longtail_lz4.c:70-74 */
        free(d);
    }
    /* 7. many threads in Compress / Decompress of both codecs at once: their blocks go through the codec dispatcher together
     * (plugin_codec_batch.c: offsets relative to the lowest queued address, mixed operations sorted into submissions) */
    {
        enum { T = 10 };
        pthread_t th[T];
        struct codec_job jobs[T];
        for (int i = 0; i < T; ++i)
        {
            jobs[i].api = i % 2 ? zstd : lz4;
            jobs[i].tag = i % 2 ? 0x7a746432u : 0x6c7a3432u;
            jobs[i].seed = 300 + (uint64_t)i;
            jobs[i].size = 50000u + 77777u * (size_t)i;
            jobs[i].rounds = 4;
            jobs[i].failed = 0;
            CHECK(pthread_create(&th[i], 0, codec_worker, &jobs[i]) == 0);
        }
        for (int i = 0; i < T; ++i)
        {
            pthread_join(th[i], 0);
            CHECK(jobs[i].failed == 0);
        }
        uint64_t subs = 0, blocks = 0;
        Longtail_Hip_CodecBatchStats(&subs, &blocks);
        CHECK(blocks >= (uint64_t)T * 4u * 2u && subs >= 1 && subs <= blocks);
    }
    dispose_from_another_thread();
    lz4->m_API.Dispose(&lz4->m_API);
    zstd->m_API.Dispose(&zstd->m_API);
    g_hash->m_API.Dispose(&g_hash->m_API);
    g_chunker->m_API.Dispose(&g_chunker->m_API);
    CHECK(Longtail_Hip_PinnedBytes() == 0); /* the last chunker API trims the pool */
    printf("plugin_san: all checks passed\n");
    return 0;
}
