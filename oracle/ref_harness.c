/* ref_harness.c -- TEST INFRASTRUCTURE. Thin C entry points around the REFERENCE library, compiled
 * against the reference's own headers where they lie (oracle/Makefile target `ref`) and linked into
 * oracle/_ref/liblongtail_ref.so.  Own code, no reference source text; everything it calls is the
 * reference's public API.  Lets Python (ctypes) tests
 *   - probe the reference algorithms through the reference's plugin structs (pins oracle/*.c),
 *   - run Longtail_CreateVersionIndex / Longtail_WriteContent with ANY plugin pointers (reference or
 *     the HIP ones from liblongtail_hip.so) and compare serialized results byte for byte,
 *   - time the reference's bikeshed-threaded CPU path (bench.py cpu_baseline kind "reference").
 */
#include "src/longtail.h"
#include "lib/longtail_platform.h"
#include "lib/atomiccancel/longtail_atomiccancel.h"
#include "lib/bikeshed/longtail_bikeshed.h"
#include "lib/blake3/longtail_blake3.h"
#include "lib/compressblockstore/longtail_compressblockstore.h"
#include "lib/compressionregistry/longtail_compression_registry.h"
#include "lib/fsblockstore/longtail_fsblockstore.h"
#include "lib/hashregistry/longtail_hash_registry.h"
#include "lib/hpcdcchunker/longtail_hpcdcchunker.h"
#include "lib/lz4/longtail_lz4.h"
#include "lib/memstorage/longtail_memstorage.h"
#include "lib/memtracer/longtail_memtracer.h"
#include "lib/filestorage/longtail_filestorage.h"
#include "lib/zstd/longtail_zstd.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

int refh_version(void) { return 4; }

/* ------------------------------------------------------------------------------------------------
 * algorithm probes through the reference's plugin structs
 * ---------------------------------------------------------------------------------------------- */
struct mem_feed
{
    const uint8_t* data;
    uint64_t size;
    uint64_t pos;
};

static int mem_feeder(void* context, Longtail_ChunkerAPI_HChunker chunker, uint32_t requested_size, char* buffer,
                      uint32_t* out_size)
{
    struct mem_feed* f = (struct mem_feed*)context;
    uint64_t n = f->size - f->pos;
    (void)chunker;
    if (n > requested_size)
        n = requested_size;
    if (n)
        memcpy(buffer, f->data + f->pos, (size_t)n);
    f->pos += n;
    *out_size = (uint32_t)n;
    return 0;
}

/* Chunk data[0..size) with ANY ChunkerAPI (0 => reference hpcdc) through NextChunk and hash every
 * range with ANY HashAPI (0 => reference blake3).  Returns chunk count, or -(errno) on error.
 * Also checks the API contract the core relies on: ranges are contiguous, buf bytes equal the input. */
int64_t refh_chunk_stream(struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api,
                          const uint8_t* data, uint64_t size, uint32_t min, uint32_t avg, uint32_t max,
                          uint64_t* out_offsets, uint32_t* out_lens, uint64_t* out_hashes, uint64_t cap)
{
    struct Longtail_ChunkerAPI* own_c = 0;
    struct Longtail_HashAPI* own_h = 0;
    int64_t count = 0;
    uint64_t expect = 0;
    if (!chunker_api)
        chunker_api = own_c = Longtail_CreateHPCDCChunkerAPI();
    if (!hash_api)
        hash_api = own_h = Longtail_CreateBlake3HashAPI();
    Longtail_ChunkerAPI_HChunker c = 0;
    int err = chunker_api->CreateChunker(chunker_api, min, avg, max, &c);
    if (err)
    {
        count = -err;
        goto done;
    }
    struct mem_feed f = {data, size, 0};
    for (;;)
    {
        struct Longtail_Chunker_ChunkRange r;
        err = chunker_api->NextChunk(chunker_api, c, mem_feeder, &f, &r);
        if (err == ESPIPE)
        {
            if (r.len != 0 || r.offset != size)
                count = -1000;
            break;
        }
        if (err)
        {
            count = -err;
            break;
        }
        if (r.offset != expect || r.len == 0 || memcmp(r.buf, data + r.offset, r.len) != 0)
        {
            count = -1001;
            break;
        }
        uint64_t h = 0;
        err = hash_api->HashBuffer(hash_api, r.len, r.buf, &h);
        if (err)
        {
            count = -err;
            break;
        }
        if ((uint64_t)count < cap)
        {
            if (out_offsets)
                out_offsets[count] = r.offset;
            if (out_lens)
                out_lens[count] = r.len;
            if (out_hashes)
                out_hashes[count] = h;
        }
        expect += r.len;
        ++count;
    }
    chunker_api->DisposeChunker(chunker_api, c);
done:
    SAFE_DISPOSE_API(own_c);
    SAFE_DISPOSE_API(own_h);
    return count;
}

/* A feeder that serves `fail_at` bytes and then fails with `fail_errno` (a storage Read error, src/longtail.c:1950-1954).
 * Drives NextChunk like DynamicChunking does and reports what the failing call returned: the reference turns a feeder
 * failure into an empty range + ESPIPE (hpcdcchunker.c:244-248, 420-423).  Returns the number of chunks handed out before
 * the failing call (their lengths in out_lens), or -1001 when a range's bytes are not the input's.
 * out_fail[0] = errno of the failing call, [1] = range.len, [2] = range.offset, [3] = (range.buf != 0). */
struct failing_feed
{
    struct mem_feed mem;
    uint64_t fail_at;
    int fail_errno;
    uint32_t calls_after_failure;
};

static int failing_feeder(void* context, Longtail_ChunkerAPI_HChunker chunker, uint32_t requested_size, char* buffer,
                          uint32_t* out_size)
{
    struct failing_feed* f = (struct failing_feed*)context;
    if (f->mem.pos >= f->fail_at)
    {
        ++f->calls_after_failure;
        return f->fail_errno;
    }
    if (f->mem.pos + requested_size > f->fail_at)
        requested_size = (uint32_t)(f->fail_at - f->mem.pos);
    return mem_feeder(&f->mem, chunker, requested_size, buffer, out_size);
}

int64_t refh_chunk_stream_failing_feeder(struct Longtail_ChunkerAPI* chunker_api, const uint8_t* data, uint64_t size,
                                         uint32_t min, uint32_t avg, uint32_t max, uint64_t fail_at, int fail_errno,
                                         uint32_t* out_lens, uint64_t cap, uint64_t* out_fail)
{
    struct Longtail_ChunkerAPI* own_c = 0;
    int64_t count = 0;
    uint64_t expect = 0;
    if (!chunker_api)
        chunker_api = own_c = Longtail_CreateHPCDCChunkerAPI();
    Longtail_ChunkerAPI_HChunker c = 0;
    int err = chunker_api->CreateChunker(chunker_api, min, avg, max, &c);
    if (err)
    {
        SAFE_DISPOSE_API(own_c);
        return -err;
    }
    struct failing_feed f = {{data, size, 0}, fail_at, fail_errno, 0};
    for (;;)
    {
        struct Longtail_Chunker_ChunkRange r = {(const uint8_t*)1, 77, 77};
        err = chunker_api->NextChunk(chunker_api, c, failing_feeder, &f, &r);
        if (err)
        {
            out_fail[0] = (uint64_t)err;
            out_fail[1] = r.len;
            out_fail[2] = r.offset;
            out_fail[3] = r.buf != 0;
            break;
        }
        if (r.offset != expect || r.len == 0 || memcmp(r.buf, data + r.offset, r.len) != 0)
        {
            count = -1001;
            break;
        }
        if ((uint64_t)count < cap)
            out_lens[count] = r.len;
        expect += r.len;
        ++count;
    }
    /* the handle must stay usable for DisposeChunker after a failure (src/longtail.c:2296) */
    chunker_api->DisposeChunker(chunker_api, c);
    SAFE_DISPOSE_API(own_c);
    return count;
}

/* Same through NextChunkFromBuffer (the mmap-style entry point, dead in the core but in the struct). */
int64_t refh_chunk_from_buffer(struct Longtail_ChunkerAPI* chunker_api, const uint8_t* data, uint64_t size,
                               uint32_t min, uint32_t avg, uint32_t max, uint32_t* out_lens, uint64_t cap)
{
    struct Longtail_ChunkerAPI* own_c = 0;
    int64_t count = 0;
    if (!chunker_api)
        chunker_api = own_c = Longtail_CreateHPCDCChunkerAPI();
    Longtail_ChunkerAPI_HChunker c = 0;
    int err = chunker_api->CreateChunker(chunker_api, min, avg, max, &c);
    if (err)
    {
        SAFE_DISPOSE_API(own_c);
        return -err;
    }
    const uint8_t* p = data;
    const uint8_t* end = data + size;
    while (p != end)
    {
        const void* next = 0;
        err = chunker_api->NextChunkFromBuffer(chunker_api, c, p, (uint64_t)(end - p), &next);
        if (err)
        {
            count = -err;
            break;
        }
        if ((uint64_t)count < cap)
            out_lens[count] = (uint32_t)((const uint8_t*)next - p);
        ++count;
        p = (const uint8_t*)next;
    }
    chunker_api->DisposeChunker(chunker_api, c);
    SAFE_DISPOSE_API(own_c);
    return count;
}

uint64_t refh_blake3(const void* data, uint32_t len)
{
    struct Longtail_HashAPI* h = Longtail_CreateBlake3HashAPI();
    uint64_t out = 0;
    static const char empty = 0;
    h->HashBuffer(h, len, data ? data : &empty, &out);
    SAFE_DISPOSE_API(h);
    return out;
}

uint32_t refh_blake3_id(void) { return Longtail_GetBlake3HashType(); }
uint32_t refh_lz4_type(void) { return Longtail_GetLZ4DefaultQuality(); }
uint32_t refh_zstd_type(int which)
{
    switch (which)
    {
    case 0: return Longtail_GetZStdMinQuality();
    case 1: return Longtail_GetZStdDefaultQuality();
    case 2: return Longtail_GetZStdMaxQuality();
    case 3: return Longtail_GetZStdHighQuality();
    default: return Longtail_GetZStdLowQuality();
    }
}

/* codec = 0 LZ4, 1 ZStd ; `api` may be a foreign (HIP) CompressionAPI, 0 => reference */
static struct Longtail_CompressionAPI* make_codec(int codec)
{
    return codec == 0 ? Longtail_CreateLZ4CompressionAPI() : Longtail_CreateZStdCompressionAPI();
}

size_t refh_codec_bound(int codec, uint32_t settings, size_t n)
{
    struct Longtail_CompressionAPI* a = make_codec(codec);
    size_t r = a->GetMaxCompressedSize(a, settings, n);
    SAFE_DISPOSE_API(a);
    return r;
}

int refh_codec_compress(int codec, uint32_t settings, const char* src, size_t n, char* dst, size_t cap, size_t* out_n)
{
    struct Longtail_CompressionAPI* a = make_codec(codec);
    int err = a->Compress(a, settings, src, dst, n, cap, out_n);
    SAFE_DISPOSE_API(a);
    return err;
}

int refh_codec_decompress(int codec, const char* src, size_t n, char* dst, size_t cap, size_t* out_n)
{
    struct Longtail_CompressionAPI* a = make_codec(codec);
    int err = a->Decompress(a, src, dst, n, cap, out_n);
    SAFE_DISPOSE_API(a);
    return err;
}

void refh_free(void* p) { Longtail_Free(p); }

/* ------------------------------------------------------------------------------------------------
 * end-to-end: in-memory tree -> Longtail_CreateVersionIndex (-> Longtail_WriteContent)
 * ---------------------------------------------------------------------------------------------- */
static int make_parent_dirs(struct Longtail_StorageAPI* s, const char* path)
{
    char* tmp = Longtail_Strdup(path);
    int err = 0;
    for (char* p = tmp; *p && !err; ++p)
    {
        if (*p == '/' && p != tmp)
        {
            *p = 0;
            if (!s->IsDir(s, tmp))
                err = s->CreateDir(s, tmp);
            *p = '/';
        }
    }
    Longtail_Free(tmp);
    return err;
}

static int fill_storage(struct Longtail_StorageAPI* s, const char* root, uint32_t nfiles, const char* const* names,
                        const uint8_t* const* datas, const uint64_t* sizes)
{
    int err = 0;
    if (!s->IsDir(s, root))
        err = s->CreateDir(s, root);
    for (uint32_t i = 0; i < nfiles && !err; ++i)
    {
        char* path = s->ConcatPath(s, root, names[i]);
        err = make_parent_dirs(s, path);
        if (!err)
        {
            Longtail_StorageAPI_HOpenFile w = 0;
            err = s->OpenWriteFile(s, path, 0, &w);
            if (!err)
            {
                if (sizes[i])
                    err = s->Write(s, w, 0, sizes[i], datas[i]);
                s->CloseFile(s, w);
            }
        }
        Longtail_Free(path);
    }
    return err;
}

/* Where the source tree lives: 0 (default) = the reference's in-memory storage; a directory (use tmpfs, e.g. /dev/shm)
 * = the reference's file storage below it, which is what SURVEY.md §8(d) asks the CPU baseline to read from
 * ("page-cache-warm, from tmpfs"): the in-memory storage serialises every read behind one lock. */
static char g_tree_dir[512];
void refh_set_tree_dir(const char* dir)
{
    g_tree_dir[0] = 0;
    if (dir && strlen(dir) < sizeof(g_tree_dir) - 64)
        strcpy(g_tree_dir, dir);
}

struct refh_tree
{
    char root[600];
    struct Longtail_StorageAPI* storage;
    struct Longtail_JobAPI* jobs;
    struct Longtail_FileInfos* files;
    uint32_t* tags;
};

static void tree_free(struct refh_tree* t)
{
    Longtail_Free(t->tags);
    Longtail_Free(t->files);
    SAFE_DISPOSE_API(t->jobs);
    SAFE_DISPOSE_API(t->storage);
    if (g_tree_dir[0] && strncmp(t->root, g_tree_dir, strlen(g_tree_dir)) == 0 && strstr(t->root, "/refh_tree_"))
    {
        char cmd[700];
        snprintf(cmd, sizeof cmd, "rm -rf '%s'", t->root); /* our own scratch directory */
        if (system(cmd) != 0)
            fprintf(stderr, "refh: could not remove %s\n", t->root);
    }
}

static int tree_make(struct refh_tree* t, uint32_t nfiles, const char* const* names, const uint8_t* const* datas,
                     const uint64_t* sizes, int workers, uint32_t tag)
{
    memset(t, 0, sizeof *t);
    if (g_tree_dir[0])
    {
        static unsigned serial; /* (a tree of its own per call: bench.py keeps one alive across several sweeps while others come and go) */
        snprintf(t->root, sizeof t->root, "%s/refh_tree_%d_%u", g_tree_dir, (int)getpid(), __atomic_add_fetch(&serial, 1u, __ATOMIC_RELAXED));
        t->storage = Longtail_CreateFSStorageAPI();
    }
    else
    {
        strcpy(t->root, "root");
        t->storage = Longtail_CreateInMemStorageAPI();
    }
    t->jobs = Longtail_CreateBikeshedJobAPI((uint32_t)workers, 0);
    int err = fill_storage(t->storage, t->root, nfiles, names, datas, sizes);
    if (!err)
        err = Longtail_GetFilesRecursively2(t->storage, t->jobs, 0, 0, 0, t->root, &t->files);
    if (!err)
    {
        t->tags = (uint32_t*)Longtail_Alloc("refh", sizeof(uint32_t) * (t->files->m_Count + 1));
        for (uint32_t i = 0; i < t->files->m_Count; ++i)
            t->tags[i] = tag;
    }
    if (err)
        tree_free(t);
    return err;
}

/* Build the VersionIndex of an in-memory tree with the given plugins (0 => reference plugin) and
 * return its serialized form (Longtail_WriteVersionIndexToBuffer) -- free with refh_free(). */
int refh_version_index(struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api, uint32_t nfiles,
                       const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                       uint32_t target_chunk_size, int workers, uint32_t tag, void** out_buf, uint64_t* out_size,
                       double* out_seconds)
{
    struct refh_tree t;
    struct Longtail_ChunkerAPI* own_c = 0;
    struct Longtail_HashAPI* own_h = 0;
    struct Longtail_VersionIndex* vi = 0;
    int err = tree_make(&t, nfiles, names, datas, sizes, workers, tag);
    if (err)
        return err;
    if (!chunker_api)
        chunker_api = own_c = Longtail_CreateHPCDCChunkerAPI();
    if (!hash_api)
        hash_api = own_h = Longtail_CreateBlake3HashAPI();
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    err = Longtail_CreateVersionIndex(t.storage, hash_api, chunker_api, t.jobs, 0, 0, 0, t.root, t.files, t.tags,
                                      target_chunk_size, 0, &vi);
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (out_seconds)
        *out_seconds = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    if (!err)
    {
        size_t sz = 0;
        err = Longtail_WriteVersionIndexToBuffer(vi, out_buf, &sz);
        *out_size = sz;
    }
    Longtail_Free(vi);
    SAFE_DISPOSE_API(own_c);
    SAFE_DISPOSE_API(own_h);
    tree_free(&t);
    return err;
}

/* Longtail_CreateVersionIndex under cancellation (test/test.cpp:4733-4837 TestCreateVersionCancelOperation): the token is
 * cancelled before the call (cancel_after_progress == 0) or from the progress callback after that many OnProgress calls,
 * i.e. while chunking jobs are in flight on the bikeshed workers.  Returns the error of Longtail_CreateVersionIndex
 * (ECANCELED expected); *out_index_is_null tells whether the output pointer stayed 0. */
struct cancel_progress
{
    struct Longtail_ProgressAPI api;
    struct Longtail_CancelAPI* cancel_api;
    Longtail_CancelAPI_HCancelToken token;
    uint32_t calls, cancel_after;
};

static void cancel_progress_dispose(struct Longtail_API* a) { (void)a; }
static void cancel_progress_on(struct Longtail_ProgressAPI* a, uint32_t total, uint32_t done)
{
    struct cancel_progress* p = (struct cancel_progress*)a;
    (void)total;
    (void)done;
    if (++p->calls == p->cancel_after)
        p->cancel_api->Cancel(p->cancel_api, p->token);
}

int refh_version_index_cancel(struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api, uint32_t nfiles,
                              const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                              uint32_t target_chunk_size, int workers, uint32_t cancel_after_progress,
                              int* out_index_is_null, uint32_t* out_progress_calls)
{
    struct refh_tree t;
    struct Longtail_ChunkerAPI* own_c = 0;
    struct Longtail_HashAPI* own_h = 0;
    struct Longtail_VersionIndex* vi = 0;
    int err = tree_make(&t, nfiles, names, datas, sizes, workers, 0);
    if (err)
        return -err;
    if (!chunker_api)
        chunker_api = own_c = Longtail_CreateHPCDCChunkerAPI();
    if (!hash_api)
        hash_api = own_h = Longtail_CreateBlake3HashAPI();
    struct Longtail_CancelAPI* cancel_api = Longtail_CreateAtomicCancelAPI();
    Longtail_CancelAPI_HCancelToken token = 0;
    cancel_api->CreateToken(cancel_api, &token);
    struct cancel_progress prog;
    memset(&prog, 0, sizeof prog);
    prog.api.m_API.Dispose = cancel_progress_dispose;
    prog.api.OnProgress = cancel_progress_on;
    prog.cancel_api = cancel_api;
    prog.token = token;
    prog.cancel_after = cancel_after_progress;
    if (cancel_after_progress == 0)
        cancel_api->Cancel(cancel_api, token);
    err = Longtail_CreateVersionIndex(t.storage, hash_api, chunker_api, t.jobs, cancel_after_progress ? &prog.api : 0,
                                      cancel_api, token, t.root, t.files, t.tags, target_chunk_size, 0, &vi);
    *out_index_is_null = vi == 0;
    *out_progress_calls = prog.calls;
    Longtail_Free(vi);
    cancel_api->DisposeToken(cancel_api, token);
    SAFE_DISPOSE_API(cancel_api);
    SAFE_DISPOSE_API(own_c);
    SAFE_DISPOSE_API(own_h);
    tree_free(&t);
    return err;
}

/* The struct Longtail_FileInfos the reference builds for a tree (Longtail_GetFilesRecursively2), flattened:
 * [u32 count][u32 path_data_size][u64 sizes[count]][u32 path_start_offsets[count]][u16 permissions[count]][path data].
 * Free with refh_free().  Lets the tests lay a tree out on the GPU in the reference's asset order. */
int refh_tree_file_infos(uint32_t nfiles, const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                         void** out_buf, uint64_t* out_size)
{
    struct refh_tree t;
    int err = tree_make(&t, nfiles, names, datas, sizes, 0, 0);
    if (err)
        return err;
    const uint32_t n = t.files->m_Count, pd = t.files->m_PathDataSize;
    const size_t size = 8 + (size_t)n * (8 + 4 + 2) + pd;
    uint8_t* b = (uint8_t*)Longtail_Alloc("refh", size);
    uint8_t* w = b;
    memcpy(w, &n, 4);
    memcpy(w + 4, &pd, 4);
    w += 8;
    memcpy(w, t.files->m_Sizes, (size_t)n * 8);
    w += (size_t)n * 8;
    memcpy(w, t.files->m_PathStartOffsets, (size_t)n * 4);
    w += (size_t)n * 4;
    memcpy(w, t.files->m_Permissions, (size_t)n * 2);
    w += (size_t)n * 2;
    memcpy(w, t.files->m_PathData, pd);
    *out_buf = b;
    *out_size = size;
    tree_free(&t);
    return 0;
}

/* Leak accounting with the reference's own tracer (lib/memtracer): everything allocated through Longtail_Alloc --
 * including the HIP plugins' objects once Longtail_Hip_SetAllocator(refh_alloc_ptr(), refh_free_ptr()) is in effect --
 * is counted; after every API object is disposed the outstanding count must be back to zero. */
void refh_memtrace_begin(void)
{
    Longtail_MemTracer_Init();
    Longtail_SetReAllocAndFree(Longtail_MemTracer_ReAlloc, Longtail_MemTracer_Free);
}
uint64_t refh_memtrace_outstanding(void) { return Longtail_MemTracer_GetAllocationCount(0); }
void refh_memtrace_end(void)
{
    Longtail_SetReAllocAndFree(0, 0);
    Longtail_MemTracer_Dispose();
}
void* refh_alloc_ptr(void) { return (void*)Longtail_Alloc; }
void* refh_free_ptr(void) { return (void*)Longtail_Free; }

static struct Longtail_CompressionRegistryAPI* make_registry(struct Longtail_CompressionAPI* foreign, uint32_t type);

/* A stored block image (the bytes of a .lsb file) through the reference: Longtail_ReadStoredBlockFromBuffer, then its
 * BlockIndex is compared field by field with Longtail_CreateBlockIndex over the expected chunks (same block hash) and
 * the [raw][compressed] payload is decompressed with the reference codec its tag names.  Returns 0 and the raw bytes. */
int refh_open_stored_block(const void* image, uint64_t size, uint32_t chunk_count, const uint64_t* chunk_hashes,
                           const uint32_t* chunk_sizes, uint32_t tag, uint8_t* out_raw, uint64_t out_cap, uint64_t* out_raw_size)
{
    struct Longtail_StoredBlock* sb = 0;
    int err = Longtail_ReadStoredBlockFromBuffer(image, (size_t)size, &sb);
    if (err)
        return err;
    struct Longtail_HashAPI* h = Longtail_CreateBlake3HashAPI();
    struct Longtail_BlockIndex* expect = 0;
    uint32_t* idx = (uint32_t*)Longtail_Alloc("refh", sizeof(uint32_t) * (chunk_count + 1));
    for (uint32_t i = 0; i < chunk_count; ++i)
        idx[i] = i;
    err = Longtail_CreateBlockIndex(h, tag, chunk_count, idx, chunk_hashes, chunk_sizes, &expect);
    if (!err)
    {
        const struct Longtail_BlockIndex* bi = sb->m_BlockIndex;
        if (*bi->m_BlockHash != *expect->m_BlockHash || *bi->m_HashIdentifier != *expect->m_HashIdentifier ||
            *bi->m_ChunkCount != chunk_count || *bi->m_Tag != tag ||
            memcmp(bi->m_ChunkHashes, expect->m_ChunkHashes, sizeof(uint64_t) * chunk_count) != 0 ||
            memcmp(bi->m_ChunkSizes, expect->m_ChunkSizes, sizeof(uint32_t) * chunk_count) != 0)
            err = 3000;
    }
    if (!err)
    {
        const uint32_t* hdr = (const uint32_t*)sb->m_BlockData;
        if (sb->m_BlockChunksDataSize < 8 || hdr[1] != sb->m_BlockChunksDataSize - 8 || hdr[0] > out_cap)
            err = 3001;
        else
        {
            struct Longtail_CompressionRegistryAPI* reg = make_registry(0, 0);
            struct Longtail_CompressionAPI* api = 0;
            uint32_t settings = 0;
            err = reg->GetCompressionAPI(reg, tag, &api, &settings);
            if (!err)
            {
                size_t n = 0;
                err = api->Decompress(api, (const char*)&hdr[2], (char*)out_raw, hdr[1], hdr[0], &n);
                if (!err && n != hdr[0])
                    err = 3002;
                *out_raw_size = n;
            }
            SAFE_DISPOSE_API(reg);
        }
    }
    Longtail_Free(expect);
    Longtail_Free(idx);
    SAFE_DISPOSE_API(h);
    sb->Dispose(sb);
    return err;
}

/* Longtail_CreateMissingContent on bare arrays: a store that has `existing` chunk hashes, a version with the unique chunk
 * list (hashes, sizes, tags).  Returns the serialized StoreIndex (Longtail_WriteStoreIndexToBuffer); free with refh_free. */
int refh_missing_content(const uint64_t* existing, uint32_t existing_count, const uint64_t* chunk_hashes, const uint32_t* chunk_sizes,
                         const uint32_t* chunk_tags, uint32_t chunk_count, uint32_t max_block_size, uint32_t max_chunks_per_block,
                         void** out_buf, uint64_t* out_size)
{
    struct Longtail_StoreIndex si;
    struct Longtail_VersionIndex vi;
    struct Longtail_StoreIndex* missing = 0;
    struct Longtail_HashAPI* h = Longtail_CreateBlake3HashAPI();
    memset(&si, 0, sizeof si);
    memset(&vi, 0, sizeof vi);
    si.m_ChunkCount = &existing_count;
    si.m_ChunkHashes = (TLongtail_Hash*)existing;
    vi.m_ChunkCount = &chunk_count;
    vi.m_ChunkHashes = (TLongtail_Hash*)chunk_hashes;
    vi.m_ChunkSizes = (uint32_t*)chunk_sizes;
    vi.m_ChunkTags = (uint32_t*)chunk_tags;
    int err = Longtail_CreateMissingContent(h, &si, &vi, max_block_size, max_chunks_per_block, &missing);
    if (!err)
    {
        size_t sz = 0;
        err = Longtail_WriteStoreIndexToBuffer(missing, out_buf, &sz);
        *out_size = sz;
    }
    Longtail_Free(missing);
    SAFE_DISPOSE_API(h);
    return err;
}

/* Longtail_GetExistingStoreIndex (src/longtail.c:7087-7325) on a SERIALIZED store index: which blocks of the store cover the
 * given chunk hashes (usage filter, most-used blocks first).  Returns the serialized result; free with refh_free. */
int refh_get_existing_store_index(const void* store_index_buf, uint64_t store_index_size, const uint64_t* chunks, uint32_t chunk_count,
                                  uint32_t min_block_usage_percent, void** out_buf, uint64_t* out_size)
{
    struct Longtail_StoreIndex* si = 0;
    struct Longtail_StoreIndex* existing = 0;
    int err = Longtail_ReadStoreIndexFromBuffer(store_index_buf, (size_t)store_index_size, &si);
    if (err)
        return err;
    err = Longtail_GetExistingStoreIndex(si, chunk_count, (const TLongtail_Hash*)chunks, min_block_usage_percent, &existing);
    if (!err)
    {
        size_t sz = 0;
        err = Longtail_WriteStoreIndexToBuffer(existing, out_buf, &sz);
        *out_size = sz;
    }
    Longtail_Free(existing);
    Longtail_Free(si);
    return err;
}

/* ---- synchronous wrappers for the async block-store calls ---- */
struct sync_existing
{
    struct Longtail_AsyncGetExistingContentAPI api;
    HLongtail_Sema sema;
    struct Longtail_StoreIndex* index;
    int err;
};
static void sync_existing_done(struct Longtail_AsyncGetExistingContentAPI* a, struct Longtail_StoreIndex* si, int err)
{
    struct sync_existing* s = (struct sync_existing*)a;
    s->index = si;
    s->err = err;
    Longtail_PostSema(s->sema, 1);
}
static int get_existing(struct Longtail_BlockStoreAPI* bs, uint32_t n, const TLongtail_Hash* hashes,
                        struct Longtail_StoreIndex** out)
{
    struct sync_existing s;
    memset(&s, 0, sizeof s);
    s.api.OnComplete = sync_existing_done;
    void* mem = Longtail_Alloc("refh", Longtail_GetSemaSize());
    Longtail_CreateSema(mem, 0, &s.sema);
    int err = bs->GetExistingContent(bs, n, hashes, 0, &s.api);
    if (!err)
    {
        Longtail_WaitSema(s.sema, LONGTAIL_TIMEOUT_INFINITE);
        err = s.err;
        *out = s.index;
    }
    Longtail_DeleteSema(s.sema);
    Longtail_Free(mem);
    return err;
}

/* ---- null backing block store for timing runs ("null / in-memory backing block store") ---- */
struct null_store
{
    struct Longtail_BlockStoreAPI api;
    TLongtail_Atomic64 bytes;
    TLongtail_Atomic64 blocks;
};
static void null_dispose(struct Longtail_API* a) { Longtail_Free(a); }
static int null_put(struct Longtail_BlockStoreAPI* a, struct Longtail_StoredBlock* b, struct Longtail_AsyncPutStoredBlockAPI* done)
{
    struct null_store* s = (struct null_store*)a;
    Longtail_AtomicAdd64(&s->bytes, (int64_t)b->m_BlockChunksDataSize);
    Longtail_AtomicAdd64(&s->blocks, 1);
    done->OnComplete(done, 0);
    return 0;
}
static int null_preflight(struct Longtail_BlockStoreAPI* a, uint32_t n, const TLongtail_Hash* h, struct Longtail_AsyncPreflightStartedAPI* done)
{
    (void)a; (void)n; (void)h;
    if (done)
        done->OnComplete(done, 0, 0, 0);
    return 0;
}
static int null_get(struct Longtail_BlockStoreAPI* a, uint64_t h, struct Longtail_AsyncGetStoredBlockAPI* done)
{
    (void)a; (void)h; (void)done;
    return ENOENT;
}
static int null_existing(struct Longtail_BlockStoreAPI* a, uint32_t n, const TLongtail_Hash* h, uint32_t pct, struct Longtail_AsyncGetExistingContentAPI* done)
{
    (void)a; (void)n; (void)h; (void)pct;
    struct Longtail_StoreIndex* idx = 0;
    int err = Longtail_CreateStoreIndexFromBlocks(0, 0, &idx);
    if (err)
        return err;
    done->OnComplete(done, idx, 0);
    return 0;
}
static int null_prune(struct Longtail_BlockStoreAPI* a, uint32_t n, const TLongtail_Hash* h, struct Longtail_AsyncPruneBlocksAPI* done)
{
    (void)a; (void)n; (void)h; (void)done;
    return ENOTSUP;
}
static int null_stats(struct Longtail_BlockStoreAPI* a, struct Longtail_BlockStore_Stats* st)
{
    struct null_store* s = (struct null_store*)a;
    memset(st, 0, sizeof *st);
    st->m_StatU64[Longtail_BlockStoreAPI_StatU64_PutStoredBlock_Byte_Count] = (uint64_t)s->bytes;
    st->m_StatU64[Longtail_BlockStoreAPI_StatU64_PutStoredBlock_Count] = (uint64_t)s->blocks;
    return 0;
}
static int null_flush(struct Longtail_BlockStoreAPI* a, struct Longtail_AsyncFlushAPI* done)
{
    (void)a;
    if (done)
        done->OnComplete(done, 0);
    return 0;
}
static struct Longtail_BlockStoreAPI* make_null_store(void)
{
    struct null_store* s = (struct null_store*)Longtail_Alloc("refh", sizeof *s);
    memset(s, 0, sizeof *s);
    return Longtail_MakeBlockStoreAPI(s, null_dispose, null_put, null_preflight, null_get, null_existing, null_prune,
                                      null_stats, null_flush);
}

/* one compression type -> caller-supplied CompressionAPI (kept alive by the caller) */
static struct Longtail_CompressionAPI* g_foreign_codec;
static uint32_t g_foreign_type;
static struct Longtail_CompressionAPI* foreign_for_type(uint32_t type, uint32_t* out_settings)
{
    if (type != g_foreign_type || !g_foreign_codec)
        return 0;
    *out_settings = type;
    return g_foreign_codec;
}

/* Non-owning wrapper: the registry disposes the APIs it created, the foreign one belongs to the caller. */
struct borrowed_codec
{
    struct Longtail_CompressionAPI api;
    struct Longtail_CompressionAPI* inner;
};
static void borrowed_dispose(struct Longtail_API* a) { Longtail_Free(a); }
static size_t borrowed_bound(struct Longtail_CompressionAPI* a, uint32_t s, size_t n)
{
    struct borrowed_codec* b = (struct borrowed_codec*)a;
    return b->inner->GetMaxCompressedSize(b->inner, s, n);
}
static int borrowed_compress(struct Longtail_CompressionAPI* a, uint32_t s, const char* src, char* dst, size_t n,
                             size_t cap, size_t* out_n)
{
    struct borrowed_codec* b = (struct borrowed_codec*)a;
    return b->inner->Compress(b->inner, s, src, dst, n, cap, out_n);
}
static int borrowed_decompress(struct Longtail_CompressionAPI* a, const char* src, char* dst, size_t n, size_t cap,
                               size_t* out_n)
{
    struct borrowed_codec* b = (struct borrowed_codec*)a;
    return b->inner->Decompress(b->inner, src, dst, n, cap, out_n);
}
static struct Longtail_CompressionAPI* borrowed_for_type(uint32_t type, uint32_t* out_settings)
{
    if (!foreign_for_type(type, out_settings))
        return 0;
    struct borrowed_codec* b = (struct borrowed_codec*)Longtail_Alloc("refh", sizeof *b);
    b->api.m_API.Dispose = borrowed_dispose;
    b->api.GetMaxCompressedSize = borrowed_bound;
    b->api.Compress = borrowed_compress;
    b->api.Decompress = borrowed_decompress;
    b->inner = g_foreign_codec;
    return &b->api;
}

static struct Longtail_CompressionRegistryAPI* make_registry(struct Longtail_CompressionAPI* foreign, uint32_t type)
{
    Longtail_CompressionRegistry_CreateForTypeFunc funcs[3];
    uint32_t n = 0;
    g_foreign_codec = foreign;
    g_foreign_type = type;
    if (foreign)
        funcs[n++] = borrowed_for_type; /* consulted first: overrides the reference codec for `type` */
    funcs[n++] = Longtail_CompressionRegistry_CreateForLZ4;
    funcs[n++] = Longtail_CompressionRegistry_CreateForZstd;
    return Longtail_CreateDefaultCompressionRegistry(n, funcs);
}

/* Ingest an in-memory tree end to end (UpSync sequence, cmd/main.c:972-1153):
 *   CreateVersionIndex -> GetExistingContent(empty store) -> CreateMissingContent -> WriteContent
 * through compressblockstore(fsblockstore(in-mem target)) with block tag `tag`, where the codec for
 * `tag` is `codec_api` (0 => reference), then RESTORE the tree with a reference-only registry
 * (Longtail_WriteVersion) and compare every file byte for byte.
 * Returns 0 when everything round-trips; stats in out_*. */
static int ingest_core(struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api,
                       struct Longtail_CompressionRegistryAPI* reg_w /* disposed here */, uint32_t tag, uint32_t nfiles,
                       const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                       uint32_t target_chunk_size, uint32_t max_block_size, uint32_t max_chunks_per_block,
                       int workers, int verify, uint64_t* out_chunk_count, uint64_t* out_block_count,
                       uint64_t* out_stored_bytes, double* out_seconds_index, double* out_seconds_write)
{
    struct refh_tree t;
    struct Longtail_ChunkerAPI* own_c = 0;
    struct Longtail_HashAPI* own_h = 0;
    struct Longtail_VersionIndex* vi = 0;
    struct Longtail_StoreIndex* existing = 0;
    struct Longtail_StoreIndex* missing = 0;
    struct timespec a, b;
    int err = reg_w ? tree_make(&t, nfiles, names, datas, sizes, workers, tag) : ENOMEM;
    if (err)
    {
        SAFE_DISPOSE_API(reg_w);
        return err;
    }
    if (!chunker_api)
        chunker_api = own_c = Longtail_CreateHPCDCChunkerAPI();
    if (!hash_api)
        hash_api = own_h = Longtail_CreateBlake3HashAPI();

    struct Longtail_StorageAPI* target = Longtail_CreateInMemStorageAPI();
    struct Longtail_BlockStoreAPI* fs = verify ? Longtail_CreateFSBlockStoreAPI(t.jobs, target, "store", 0, 0) : make_null_store();
    struct Longtail_BlockStoreAPI* cbs = Longtail_CreateCompressBlockStoreAPI(fs, reg_w);

    clock_gettime(CLOCK_MONOTONIC, &a);
    err = Longtail_CreateVersionIndex(t.storage, hash_api, chunker_api, t.jobs, 0, 0, 0, t.root, t.files, t.tags,
                                      target_chunk_size, 0, &vi);
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (out_seconds_index)
        *out_seconds_index = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    if (!err)
        err = get_existing(cbs, *vi->m_ChunkCount, vi->m_ChunkHashes, &existing);
    if (!err)
        err = Longtail_CreateMissingContent(hash_api, existing, vi, max_block_size, max_chunks_per_block, &missing);
    if (!err)
    {
        clock_gettime(CLOCK_MONOTONIC, &a);
        err = Longtail_WriteContent(t.storage, cbs, t.jobs, 0, 0, 0, missing, vi, t.root);
        clock_gettime(CLOCK_MONOTONIC, &b);
        if (out_seconds_write)
            *out_seconds_write = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    }
    if (!err)
    {
        struct Longtail_BlockStore_Stats st;
        memset(&st, 0, sizeof st);
        fs->GetStats(fs, &st);
        if (out_chunk_count)
            *out_chunk_count = *vi->m_ChunkCount;
        if (out_block_count)
            *out_block_count = *missing->m_BlockCount;
        if (out_stored_bytes)
            *out_stored_bytes = st.m_StatU64[Longtail_BlockStoreAPI_StatU64_PutStoredBlock_Byte_Count];
    }
    SAFE_DISPOSE_API(cbs);
    SAFE_DISPOSE_API(reg_w);

    /* restore through a REFERENCE-ONLY registry */
    if (!err && verify)
    {
        struct Longtail_CompressionRegistryAPI* reg_r = make_registry(0, 0);
        struct Longtail_BlockStoreAPI* cbs_r = Longtail_CreateCompressBlockStoreAPI(fs, reg_r);
        struct Longtail_StorageAPI* restored = Longtail_CreateInMemStorageAPI();
        struct Longtail_StoreIndex* have = 0;
        err = get_existing(cbs_r, *vi->m_ChunkCount, vi->m_ChunkHashes, &have);
        if (!err)
            err = Longtail_WriteVersion(cbs_r, restored, t.jobs, 0, 0, 0, have, vi, "out", 1);
        for (uint32_t i = 0; i < nfiles && !err; ++i)
        {
            char* path = restored->ConcatPath(restored, "out", names[i]);
            Longtail_StorageAPI_HOpenFile r = 0;
            uint64_t sz = 0;
            err = restored->OpenReadFile(restored, path, &r);
            if (!err)
            {
                err = restored->GetSize(restored, r, &sz);
                if (!err && sz != sizes[i])
                    err = 2000;
                if (!err && sz)
                {
                    void* buf = Longtail_Alloc("refh", (size_t)sz);
                    err = restored->Read(restored, r, 0, sz, buf);
                    if (!err && memcmp(buf, datas[i], (size_t)sz) != 0)
                        err = 2001;
                    Longtail_Free(buf);
                }
                restored->CloseFile(restored, r);
            }
            Longtail_Free(path);
        }
        Longtail_Free(have);
        SAFE_DISPOSE_API(cbs_r);
        SAFE_DISPOSE_API(reg_r);
        SAFE_DISPOSE_API(restored);
    }
    Longtail_Free(missing);
    Longtail_Free(existing);
    Longtail_Free(vi);
    SAFE_DISPOSE_API(fs);
    SAFE_DISPOSE_API(target);
    SAFE_DISPOSE_API(own_c);
    SAFE_DISPOSE_API(own_h);
    tree_free(&t);
    return err;
}

static int ingest_impl(struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api,
                       struct Longtail_CompressionAPI* codec_api, uint32_t tag, uint32_t nfiles,
                       const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                       uint32_t target_chunk_size, uint32_t max_block_size, uint32_t max_chunks_per_block,
                       int workers, int verify, uint64_t* out_chunk_count, uint64_t* out_block_count,
                       uint64_t* out_stored_bytes, double* out_seconds_index, double* out_seconds_write)
{
    return ingest_core(chunker_api, hash_api, make_registry(codec_api, tag), tag, nfiles, names, datas, sizes, target_chunk_size,
                       max_block_size, max_chunks_per_block, workers, verify, out_chunk_count, out_block_count, out_stored_bytes,
                       out_seconds_index, out_seconds_write);
}

/* The embedding exactly as INTEGRATION.md prints it (the registries OWN the plugin objects, created lazily through the exported
 * factories and disposed with the registry -- lib/compressionregistry/longtail_compression_registry.c:50-146,
 * lib/hashregistry/longtail_hash_registry.c:41-67):
 *     Longtail_CreateDefaultCompressionRegistry(3, {CreateForHipLZ4, CreateForHipZstd, Longtail_CompressionRegistry_CreateForLZ4})
 *     Longtail_CreateDefaultHashRegistry(1, {Longtail_GetBlake3HashType()}, {Longtail_CreateHipBlake3HashAPI()})
 * The function pointers come from liblongtail_hip.so (this library is not linked against it); the hash API is looked up
 * through GetHashAPI with the reference's type id.  UpSync with those, restore through a REFERENCE-ONLY registry, compare the files.
 * out_apis_created: how many CompressionAPI objects the registry's factories were asked to make that succeeded (one per type USED). */
typedef struct Longtail_CompressionAPI* (*refh_create_for_type)(uint32_t, uint32_t*);
typedef struct Longtail_HashAPI* (*refh_create_hash)(void);
typedef struct Longtail_ChunkerAPI* (*refh_create_chunker)(void);
static refh_create_for_type g_emb_lz4, g_emb_zstd;
static int g_emb_created;
static struct Longtail_CompressionAPI* emb_for_lz4(uint32_t type, uint32_t* out_settings)
{
    struct Longtail_CompressionAPI* a = g_emb_lz4(type, out_settings);
    g_emb_created += a != 0;
    return a;
}
static struct Longtail_CompressionAPI* emb_for_zstd(uint32_t type, uint32_t* out_settings)
{
    struct Longtail_CompressionAPI* a = g_emb_zstd(type, out_settings);
    g_emb_created += a != 0;
    return a;
}

int refh_ingest_registry_embedding(void* create_for_hip_lz4, void* create_for_hip_zstd, void* create_hip_hash,
                                   void* create_hip_chunker, uint32_t tag, uint32_t nfiles, const char* const* names,
                                   const uint8_t* const* datas, const uint64_t* sizes, uint32_t target_chunk_size,
                                   uint32_t max_block_size, uint32_t max_chunks_per_block, int workers, uint64_t* out_chunk_count,
                                   uint64_t* out_block_count, uint64_t* out_stored_bytes, int* out_apis_created)
{
    if (!create_for_hip_lz4 || !create_for_hip_zstd || !create_hip_hash || !create_hip_chunker)
        return EINVAL;
    g_emb_lz4 = (refh_create_for_type)create_for_hip_lz4;
    g_emb_zstd = (refh_create_for_type)create_for_hip_zstd;
    g_emb_created = 0;
    Longtail_CompressionRegistry_CreateForTypeFunc funcs[3] = {emb_for_lz4, emb_for_zstd, Longtail_CompressionRegistry_CreateForLZ4};
    struct Longtail_CompressionRegistryAPI* reg = Longtail_CreateDefaultCompressionRegistry(3, funcs);
    const uint32_t hash_types[1] = {Longtail_GetBlake3HashType()};
    const struct Longtail_HashAPI* hash_apis[1] = {((refh_create_hash)create_hip_hash)()};
    if (!hash_apis[0])
    {
        SAFE_DISPOSE_API(reg);
        return ENODEV;
    }
    struct Longtail_HashRegistryAPI* hreg = Longtail_CreateDefaultHashRegistry(1, hash_types, hash_apis);
    struct Longtail_HashAPI* hash_api = 0;
    int err = hreg ? hreg->GetHashAPI(hreg, Longtail_GetBlake3HashType(), &hash_api) : ENOMEM;
    struct Longtail_ChunkerAPI* chunker_api = err ? 0 : ((refh_create_chunker)create_hip_chunker)();
    if (!err && !chunker_api)
        err = ENODEV;
    if (!err && hash_api->GetIdentifier(hash_api) != Longtail_GetBlake3HashType())
        err = 2100;
    if (!err) /* (ingest_core disposes the compression registry, and with it the HIP CompressionAPI objects it created) */
        err = ingest_core(chunker_api, hash_api, reg, tag, nfiles, names, datas, sizes, target_chunk_size, max_block_size,
                          max_chunks_per_block, workers, 1, out_chunk_count, out_block_count, out_stored_bytes, 0, 0);
    else
        SAFE_DISPOSE_API(reg);
    if (out_apis_created)
        *out_apis_created = g_emb_created;
    SAFE_DISPOSE_API(chunker_api);
    SAFE_DISPOSE_API(hreg); /* disposes the HIP HashAPI */
    return err;
}

int refh_ingest_roundtrip(struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api,
                          struct Longtail_CompressionAPI* codec_api, uint32_t tag, uint32_t nfiles,
                          const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                          uint32_t target_chunk_size, uint32_t max_block_size, uint32_t max_chunks_per_block,
                          int workers, uint64_t* out_chunk_count, uint64_t* out_block_count,
                          uint64_t* out_stored_bytes, double* out_seconds_index, double* out_seconds_write)
{
    return ingest_impl(chunker_api, hash_api, codec_api, tag, nfiles, names, datas, sizes, target_chunk_size,
                       max_block_size, max_chunks_per_block, workers, 1, out_chunk_count, out_block_count,
                       out_stored_bytes, out_seconds_index, out_seconds_write);
}

/* bench.py cpu_baseline leg: the same UpSync sequence, reference plugins, no restore pass */
int refh_ingest_time(uint32_t tag, uint32_t nfiles, const char* const* names, const uint8_t* const* datas,
                     const uint64_t* sizes, uint32_t target_chunk_size, uint32_t max_block_size,
                     uint32_t max_chunks_per_block, int workers, uint64_t* out_chunk_count,
                     uint64_t* out_block_count, uint64_t* out_stored_bytes, double* out_seconds_index,
                     double* out_seconds_write)
{
    return ingest_impl(0, 0, 0, tag, nfiles, names, datas, sizes, target_chunk_size, max_block_size,
                       max_chunks_per_block, workers, 0, out_chunk_count, out_block_count, out_stored_bytes,
                       out_seconds_index, out_seconds_write);
}

/* bench.py cpu_baseline leg, SURVEY.md §8(d) protocol: ONE source tree (tmpfs file storage when refh_set_tree_dir was called),
 * then for every worker count W of `workers` and every repetition the timed sequence
 *     Longtail_CreateVersionIndex -> [GetExistingContent on the empty null store, untimed] -> Longtail_CreateMissingContent
 *     -> Longtail_WriteContent (compressblockstore over a null block sink)
 * with the reference's plugins and Longtail_CreateBikeshedJobAPI(W, 0).  out_seconds[(w * reps + r) * 3 + {0,1,2}] = seconds of
 * the three calls. */
static uint64_t g_last_raw_bytes;
uint64_t refh_last_raw_bytes(void) { return g_last_raw_bytes; } /* of the last refh_ingest_sweep*: bytes of the chunks written */

/* the timed part, on a tree that exists: for every W of `workers` and every repetition the three calls */
static int ingest_sweep_on_tree(struct refh_tree* t, struct Longtail_ChunkerAPI* foreign_chunker, struct Longtail_HashAPI* foreign_hash,
                                struct Longtail_CompressionAPI* foreign_codec, uint32_t tag, uint32_t target_chunk_size, uint32_t max_block_size,
                                uint32_t max_chunks_per_block, uint32_t n_workers, const int* workers, uint32_t reps, double* out_seconds,
                                uint64_t* out_chunk_count, uint64_t* out_block_count, uint64_t* out_stored_bytes)
{
    int err = 0;
    /* a plugin object handed in (the embedder's: liblongtail_hip.so's constructors) replaces the reference's; it stays the caller's */
    struct Longtail_ChunkerAPI* chunker_api = foreign_chunker ? foreign_chunker : Longtail_CreateHPCDCChunkerAPI();
    struct Longtail_HashAPI* hash_api = foreign_hash ? foreign_hash : Longtail_CreateBlake3HashAPI();
    for (uint32_t w = 0; w < n_workers && !err; ++w)
    {
        struct Longtail_JobAPI* jobs = Longtail_CreateBikeshedJobAPI((uint32_t)workers[w], 0);
        for (uint32_t r = 0; r < reps && !err; ++r)
        {
            struct Longtail_VersionIndex* vi = 0;
            struct Longtail_StoreIndex* existing = 0;
            struct Longtail_StoreIndex* missing = 0;
            struct Longtail_CompressionRegistryAPI* reg = make_registry(foreign_codec, tag);
            struct Longtail_BlockStoreAPI* sink = make_null_store();
            struct Longtail_BlockStoreAPI* cbs = Longtail_CreateCompressBlockStoreAPI(sink, reg);
            double* secs = out_seconds + ((size_t)w * reps + r) * 3;
            struct timespec a, b;
            clock_gettime(CLOCK_MONOTONIC, &a);
            err = Longtail_CreateVersionIndex(t->storage, hash_api, chunker_api, jobs, 0, 0, 0, t->root, t->files, t->tags,
                                              target_chunk_size, 0, &vi);
            clock_gettime(CLOCK_MONOTONIC, &b);
            secs[0] = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
            if (!err)
                err = get_existing(cbs, *vi->m_ChunkCount, vi->m_ChunkHashes, &existing);
            if (!err)
            {
                clock_gettime(CLOCK_MONOTONIC, &a);
                err = Longtail_CreateMissingContent(hash_api, existing, vi, max_block_size, max_chunks_per_block, &missing);
                clock_gettime(CLOCK_MONOTONIC, &b);
                secs[1] = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
            }
            if (!err)
            {
                clock_gettime(CLOCK_MONOTONIC, &a);
                err = Longtail_WriteContent(t->storage, cbs, jobs, 0, 0, 0, missing, vi, t->root);
                clock_gettime(CLOCK_MONOTONIC, &b);
                secs[2] = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
            }
            if (!err)
            {
                struct Longtail_BlockStore_Stats st;
                memset(&st, 0, sizeof st);
                sink->GetStats(sink, &st);
                if (out_chunk_count)
                    *out_chunk_count = *vi->m_ChunkCount;
                if (out_block_count)
                    *out_block_count = *missing->m_BlockCount;
                if (out_stored_bytes)
                    *out_stored_bytes = st.m_StatU64[Longtail_BlockStoreAPI_StatU64_PutStoredBlock_Byte_Count];
                /* what reached the codec: the unique chunks' bytes (the stored bytes above are [raw][compressed] + payload per block) */
                uint64_t raw = 0;
                for (uint32_t c = 0; c < *missing->m_ChunkCount; ++c)
                    raw += missing->m_ChunkSizes[c];
                g_last_raw_bytes = raw;
            }
            SAFE_DISPOSE_API(cbs);
            SAFE_DISPOSE_API(reg);
            SAFE_DISPOSE_API(sink);
            Longtail_Free(missing);
            Longtail_Free(existing);
            Longtail_Free(vi);
        }
        SAFE_DISPOSE_API(jobs);
    }
    if (!foreign_chunker)
        SAFE_DISPOSE_API(chunker_api);
    if (!foreign_hash)
        SAFE_DISPOSE_API(hash_api);
    return err;
}

static int ingest_sweep_impl(struct Longtail_ChunkerAPI* foreign_chunker, struct Longtail_HashAPI* foreign_hash,
                             struct Longtail_CompressionAPI* foreign_codec, uint32_t tag, uint32_t nfiles, const char* const* names,
                             const uint8_t* const* datas, const uint64_t* sizes, uint32_t target_chunk_size, uint32_t max_block_size,
                             uint32_t max_chunks_per_block, uint32_t n_workers, const int* workers, uint32_t reps, double* out_seconds,
                             uint64_t* out_chunk_count, uint64_t* out_block_count, uint64_t* out_stored_bytes)
{
    struct refh_tree t;
    int err = tree_make(&t, nfiles, names, datas, sizes, 1, tag);
    if (err)
        return err;
    err = ingest_sweep_on_tree(&t, foreign_chunker, foreign_hash, foreign_codec, tag, target_chunk_size, max_block_size, max_chunks_per_block,
                               n_workers, workers, reps, out_seconds, out_chunk_count, out_block_count, out_stored_bytes);
    tree_free(&t);
    return err;
}

/* One source tree kept across several sweeps (bench.py: the reference's plugins, then the embedder's, on the SAME files without
 * writing them to tmpfs again): refh_tree_create -> refh_ingest_sweep_tree ... -> refh_tree_destroy. */
void* refh_tree_create(uint32_t tag, uint32_t nfiles, const char* const* names, const uint8_t* const* datas, const uint64_t* sizes)
{
    struct refh_tree* t = (struct refh_tree*)malloc(sizeof *t);
    if (!t)
        return 0;
    if (tree_make(t, nfiles, names, datas, sizes, 1, tag))
    {
        free(t);
        return 0;
    }
    return t;
}
void refh_tree_destroy(void* tree)
{
    if (!tree)
        return;
    tree_free((struct refh_tree*)tree);
    free(tree);
}
int refh_ingest_sweep_tree(void* tree, struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api,
                           struct Longtail_CompressionAPI* codec_api, uint32_t tag, uint32_t target_chunk_size, uint32_t max_block_size,
                           uint32_t max_chunks_per_block, uint32_t n_workers, const int* workers, uint32_t reps, double* out_seconds,
                           uint64_t* out_chunk_count, uint64_t* out_block_count, uint64_t* out_stored_bytes)
{
    if (!tree)
        return EINVAL;
    return ingest_sweep_on_tree((struct refh_tree*)tree, chunker_api, hash_api, codec_api, tag, target_chunk_size, max_block_size,
                                max_chunks_per_block, n_workers, workers, reps, out_seconds, out_chunk_count, out_block_count, out_stored_bytes);
}

int refh_ingest_sweep(uint32_t tag, uint32_t nfiles, const char* const* names, const uint8_t* const* datas, const uint64_t* sizes,
                      uint32_t target_chunk_size, uint32_t max_block_size, uint32_t max_chunks_per_block, uint32_t n_workers,
                      const int* workers, uint32_t reps, double* out_seconds, uint64_t* out_chunk_count, uint64_t* out_block_count,
                      uint64_t* out_stored_bytes)
{
    return ingest_sweep_impl(0, 0, 0, tag, nfiles, names, datas, sizes, target_chunk_size, max_block_size, max_chunks_per_block, n_workers,
                             workers, reps, out_seconds, out_chunk_count, out_block_count, out_stored_bytes);
}

/* The same measurement with the EMBEDDER'S plugin objects in the unmodified core (any of them 0 = the reference's): what a longtail
 * user gets by switching constructors and nothing else (bench.py secondary.drop_in). */
int refh_ingest_sweep_apis(struct Longtail_ChunkerAPI* chunker_api, struct Longtail_HashAPI* hash_api,
                           struct Longtail_CompressionAPI* codec_api, uint32_t tag, uint32_t nfiles, const char* const* names,
                           const uint8_t* const* datas, const uint64_t* sizes, uint32_t target_chunk_size, uint32_t max_block_size,
                           uint32_t max_chunks_per_block, uint32_t n_workers, const int* workers, uint32_t reps, double* out_seconds,
                           uint64_t* out_chunk_count, uint64_t* out_block_count, uint64_t* out_stored_bytes)
{
    return ingest_sweep_impl(chunker_api, hash_api, codec_api, tag, nfiles, names, datas, sizes, target_chunk_size, max_block_size,
                             max_chunks_per_block, n_workers, workers, reps, out_seconds, out_chunk_count, out_block_count,
                             out_stored_bytes);
}

int refh_cpu_count(void) { return (int)Longtail_GetCPUCount(); }
