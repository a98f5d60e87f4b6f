/* longtail_hip.h -- C ABI of liblongtail_hip.so: MI355X (gfx950) implementations of longtail's
 * chunk -> hash -> compress hot path.
 *
 * Two layers, both plain C (pointers + sizes, errno-style int results, no C++/torch types):
 *
 *  A. PLUGIN CONSTRUCTORS -- what a longtail embedder binds instead of the CPU constructors.  They
 *     return longtail's own struct-of-function-pointer objects (include/longtail_abi.h), so they drop
 *     into Longtail_CreateVersionIndex / Longtail_WriteContent / the registries unchanged.
 *
 *  B. BULK DEVICE API (lthip_*) -- the thin shim over the HIP kernels that layer A is written on, for
 *     callers that already hold asset bytes in HBM (bench.py, the multi-GPU driver, tests).
 *
 * Every entry point cites the reference interface it replaces.
 */
#ifndef LONGTAIL_HIP_H
#define LONGTAIL_HIP_H

#include "longtail_abi.h"

#if defined(_WIN32)
#define LTHIP_EXPORT
#else
#define LTHIP_EXPORT __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* =====================================================================================================
 * A. plugin constructors
 * =================================================================================================== */

/* Replaces Longtail_CreateHPCDCChunkerAPI() (lib/hpcdcchunker/longtail_hpcdcchunker.h:10,
 * implementation longtail_hpcdcchunker.c:332-589).  GetMinChunkSize -> 48.  NextChunk drains the
 * feeder into pinned host memory (windows of up to 64 MiB), computes ALL cut points and ALL chunk
 * hashes of the window on the GPU, then hands out ranges whose `buf` points into the pinned window;
 * ESPIPE + {0,total,0} at end of stream exactly like the reference. */
LTHIP_EXPORT struct Longtail_ChunkerAPI* Longtail_CreateHipChunkerAPI(void);

/* Replaces Longtail_CreateBlake3HashAPI() (lib/blake3/longtail_blake3.h, implementation
 * longtail_blake3.c:6-141).  GetIdentifier -> 0x626c6b33 ('blk3') so indexes are interchangeable with
 * CPU-built ones.  HashBuffer on a range previously handed out by a HIP chunker returns the digest the
 * GPU already computed; any other buffer is hashed on the GPU on demand. */
LTHIP_EXPORT struct Longtail_HashAPI* Longtail_CreateHipBlake3HashAPI(void);

/* WHEN TO CONSTRUCT THE CODEC OBJECTS.  One stored block per Compress call crosses the link twice: the HIP codec objects pay on data that
 * compresses -- WriteContent at 32 bikeshed workers: LZ4 31-47 GB/s against the reference codec's 22-25, ZStd 33-35 against 7-14 -- and
 * lose on incompressible bytes, where the CPU's LZ4 is a memcpy (25-29 against 56-105 GB/s): there bind the reference's own LZ4 beside
 * the HIP chunker + hash, or use the bulk session (lthip_ingest_*), which never brings payload bytes back through Compress
 * (INTEGRATION.md "Which codec object to bind"; numbers: the bench line's secondary.*.drop_in, profiles/r06q_*).
 *
 * Replaces Longtail_CreateLZ4CompressionAPI() / Longtail_CompressionRegistry_CreateForLZ4()
 * (lib/lz4/longtail_lz4.h:10-12, longtail_lz4.c:12-23,47-123).  Same type id 'lz42', same bound
 * (n + n/255 + 16), payload = one LZ4 *block* that LZ4_decompress_safe decodes. */
LTHIP_EXPORT struct Longtail_CompressionAPI* Longtail_CreateHipLZ4CompressionAPI(void);
LTHIP_EXPORT struct Longtail_CompressionAPI* Longtail_CompressionRegistry_CreateForHipLZ4(uint32_t compression_type, uint32_t* out_settings);
LTHIP_EXPORT uint32_t Longtail_GetHipLZ4DefaultQuality(void);

/* Replaces Longtail_CreateZStdCompressionAPI() / Longtail_CompressionRegistry_CreateForZstd()
 * (lib/zstd/longtail_zstd.h:10-16, longtail_zstd.c:30-41,72-177).  Type ids 'ztd1'..'ztd5'; the setting selects one of three
 * parses (lthip_zstd_quality_of_settings: 'ztd1' / 'ztd2' default, 'ztd4' high, 'ztd3' / 'ztd5' max; unknown ids -> default, exactly
 * like longtail_zstd.c:43-60).  Compress: one zstd frame per block -- RLE / Compressed (LZ sequences, Huffman literals, FSE sequences) / Raw
 * blocks -- that ZSTD_decompressDCtx decodes.  Decompress: any zstd frame(s) without a dictionary, e.g. the reference
 * encoder's; malformed input -> EINVAL (longtail_zstd.c:168-172). */
LTHIP_EXPORT struct Longtail_CompressionAPI* Longtail_CreateHipZStdCompressionAPI(void);
LTHIP_EXPORT struct Longtail_CompressionAPI* Longtail_CompressionRegistry_CreateForHipZstd(uint32_t compression_type, uint32_t* out_settings);

/* Route the plugins' host allocations through the embedder's allocator, e.g. Longtail_Alloc /
 * Longtail_Free (src/longtail.h:967-974) so memtracer leak checks see them.  Default malloc/free. */
typedef void* (*Longtail_Hip_AllocFunc)(const char* context, size_t size);
typedef void (*Longtail_Hip_FreeFunc)(void* p);
LTHIP_EXPORT void Longtail_Hip_SetAllocator(Longtail_Hip_AllocFunc alloc_func, Longtail_Hip_FreeFunc free_func);

/* GPU used by plugin objects created afterwards (default: $LONGTAIL_HIP_DEVICE or 0). */
LTHIP_EXPORT int Longtail_Hip_SetDevice(int device);
/* OPTIONAL, and if used the embedder's FIRST call into this library, before the process has touched the GPU: host threads waiting
 * for the device SLEEP until the interrupt (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) on the plugin device) instead of polling.
 * The job system calls the blocking Longtail_*API functions from 32-256 threads at once; polling was a third of the drop-in path's
 * CPU time, and under a container's CPU quota that is throughput: CreateVersionIndex through the plugins 30 -> 40 GB/s at 32 workers,
 * UpSync 13.9 -> 16.5 (profiles/r06_dropin_scaling.txt; bench.py's drop_in legs run in a process of their own that makes this call).
 * It is the device's policy, PROCESS-WIDE, and must not be switched in a process that already ran GPU work: waits on completion signals
 * made before the switch may never return (observed).  Returns 0 or an errno; the library never makes this call on its own. */
LTHIP_EXPORT int Longtail_Hip_SetBlockingWaits(int on);
LTHIP_EXPORT int lthip_set_blocking_waits(int device, int on); /* the call behind it */

/* HashAPI.Hash / EndContext return no error code (longtail_blake3.c:43-79 cannot fail; a GPU path can: allocation, copy, no
 * device).  The first errno of such a call on the calling thread is latched; this returns and clears it (0 = none). */
LTHIP_EXPORT int Longtail_Hip_GetLastError(void);

/* Pinned host memory currently held by the chunker windows.  Windows are pooled in two classes -- 2 MiB (at most
 * $LONGTAIL_HIP_SMALL_WINDOWS, default 256) and 64 MiB (at most $LONGTAIL_HIP_LARGE_WINDOWS, default 32) -- so the bound is
 * about 0.5 + 2 GiB (and as much HBM) however many chunkers longtail's job system keeps alive; a thread that needs a window beyond
 * the cap waits for one to be released.  The reference pools 4 * max_chunk bytes per chunker (hpcdcchunker.c:148-171). */
LTHIP_EXPORT uint64_t Longtail_Hip_PinnedBytes(void);
/* Diagnostics of the small-window batcher (plugin_batch.c): GPU submissions made and windows carried by them since the library
 * was loaded; windows / batches = how many chunkers shared a submission on average. */
LTHIP_EXPORT void Longtail_Hip_BatchStats(uint64_t* out_batches, uint64_t* out_windows);
/* ... and of the codec objects: submissions of the dispatcher that runs concurrent Compress / Decompress calls together, blocks in them */
LTHIP_EXPORT void Longtail_Hip_CodecBatchStats(uint64_t* out_submissions, uint64_t* out_blocks);
/* ... and of its content-hash memo: digest arrays remembered, HashBuffer calls answered from them */
LTHIP_EXPORT void Longtail_Hip_MemoStats(uint64_t* out_puts, uint64_t* out_hits);

/* =====================================================================================================
 * B. bulk device API
 * =================================================================================================== */

typedef struct lthip_ctx lthip_ctx;   /* one GPU + one stream + scratch pools; NOT thread-safe: one per host thread */
typedef struct lthip_plan lthip_plan; /* device-resident description of a batch of parts */

/* hip_stream: the hipStream_t to launch on, e.g. torch.cuda.current_stream().cuda_stream; NULL is HIP's default
 * (null) stream, exactly as in every HIP API; LTHIP_STREAM_PRIVATE asks for a private non-blocking stream owned
 * by the context (what the plugin layer uses: one per host thread). */
#define LTHIP_STREAM_PRIVATE ((void*)(intptr_t)-1)
LTHIP_EXPORT int lthip_ctx_create(int device, void* hip_stream, lthip_ctx** out_ctx);
LTHIP_EXPORT void lthip_ctx_destroy(lthip_ctx* ctx);
LTHIP_EXPORT int lthip_ctx_sync(lthip_ctx* ctx);
LTHIP_EXPORT const char* lthip_ctx_error(const lthip_ctx* ctx); /* text of the last failure */
LTHIP_EXPORT int lthip_device_count(void);
/* Identity of the sources the library was built from: the first 16 hex digits of the sha256 tools/build_id.py computes over
 * longtail_amd/csrc/ and include/.  A test recomputes it from the tree, so a stale binary cannot pass for HEAD. */
LTHIP_EXPORT const char* lthip_build_id(void);
/* Version of the BINARY interface of section B: bumped whenever a struct of this header grows or changes layout, an enum value
 * moves, or an entry point changes its signature (new entry points alone do not bump it).  An embedder built against this header
 * checks  lthip_abi_version() == LTHIP_ABI_VERSION  once after loading the library.
 *   1  rounds 1-3      2  round 4: lthip_ingest_result.gathered_bytes appended, LTHIP_K_COUNT 9 -> 10
 *   3  round 5: lthip_ingest_result starts with struct_size (set by the caller; the library writes no more than that) */
#define LTHIP_ABI_VERSION 3
LTHIP_EXPORT int lthip_abi_version(void);

/* Memory helpers so that plain-C callers (the plugin layer) need no HIP headers.  Copies are
 * asynchronous on the context's stream: lthip_ctx_sync() before reading a d2h destination. */
LTHIP_EXPORT int lthip_malloc_device(lthip_ctx* ctx, size_t bytes, void** out);
LTHIP_EXPORT void lthip_free_device(lthip_ctx* ctx, void* p);
LTHIP_EXPORT int lthip_malloc_pinned(lthip_ctx* ctx, size_t bytes, void** out);
LTHIP_EXPORT void lthip_free_pinned(lthip_ctx* ctx, void* p);
LTHIP_EXPORT int lthip_copy_h2d(lthip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
LTHIP_EXPORT int lthip_copy_d2h(lthip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* The same copy made by the compute units instead of the copy engines, either direction, `dst` and `src` 16-byte aligned, the
 * host side PINNED (lthip_malloc_pinned / hipHostMalloc).  On the MI355X boxes measured (profiles/r05_pcie_duplex.json) the copy
 * engines serve one direction at a time -- h2d and d2h queued on two streams take the sum of their times -- while kernel copies
 * in both directions overlap (46 GB/s each against 57 GB/s alone): a loop that streams slices in and block images out uses this
 * (and lthip_gather_ranges, which accepts pinned host memory as source or destination too) on contexts of their own. */
LTHIP_EXPORT int lthip_link_copy(lthip_ctx* ctx, void* dst, const void* src, size_t bytes);

/* Per-kernel timing with HIP events on the context's stream (bench.py roofline leg). */
enum lthip_kernel_id
{
    LTHIP_K_BUZHASH = 0,  /* candidate scan                     (hpcdcchunker.c:266-306) */
    LTHIP_K_SELECT = 1,   /* cut selection                      (hpcdcchunker.c:250-264,281-309) */
    LTHIP_K_COMPACT = 2,  /* scans + compaction of chunk lists  (src/longtail.c:2499-2517) */
    LTHIP_K_B3_LEAF = 3,  /* BLAKE3 leaves                      (blake3.c:118-151) */
    LTHIP_K_B3_PARENT = 4,/* BLAKE3 parents + root              (blake3.c:216-249,576-618) */
    LTHIP_K_LZ4_SEG = 5,  /* LZ4 segment encoder                (lz4.c:930-1338) */
    LTHIP_K_LZ4_STITCH = 6,/* LZ4 stitch scan + compaction copy */
    LTHIP_K_OTHER = 7,
    LTHIP_K_ZSTD_ENC = 8, /* zstd entropy stage (Huffman literals, FSE sequences), one wave per 128 KiB piece */
    LTHIP_K_GATHER = 9,   /* device block assembly: chunk ranges -> contiguous block images (src/longtail.c:4640-4721) */
    LTHIP_K_COUNT = 10
};
LTHIP_EXPORT int lthip_timing_enable(lthip_ctx* ctx, int on);
LTHIP_EXPORT int lthip_timing_reset(lthip_ctx* ctx);
/* resolves pending events (synchronises the stream); total milliseconds and launch count per kernel id */
LTHIP_EXPORT int lthip_timing_get(lthip_ctx* ctx, int kernel_id, double* out_total_ms, uint64_t* out_launches);

/* ---- phase 1: chunk + hash ------------------------------------------------------------------------
 * A *part* is what the reference chunks with one chunker: one (asset, target_chunk_size*1024-byte
 * segment) job of ChunkAssets (src/longtail.c:2396-2458).  Parts live in one device buffer at
 * 16-byte aligned offsets.  min/avg/max as computed by src/longtail.c:1985-1987. */
LTHIP_EXPORT int lthip_plan_create(lthip_ctx* ctx, uint32_t part_count, const uint64_t* part_offsets /*host*/,
                                   const uint64_t* part_sizes /*host*/, uint32_t min_chunk, uint32_t avg_chunk,
                                   uint32_t max_chunk, lthip_plan** out_plan);
/* A plan of ONE part (created with part_count 1, offset 0, size = capacity) aimed at `size` <= capacity bytes: no allocation, no
 * kernel, no synchronisation -- the plugin chunker keeps one plan per window and re-aims it at every refill. */
LTHIP_EXPORT int lthip_plan_resize_single(lthip_ctx* ctx, lthip_plan* plan, uint64_t size);
/* The plan aimed at ANOTHER set of parts: at most as many as it was created with, and no more 16 KiB tiles in total; no
 * allocation, no synchronisation (the tables are rewritten on the context's stream).  The plugin layer's batcher keeps one plan of
 * N window slots and aims it at the windows of every submission. */
LTHIP_EXPORT int lthip_plan_reaim(lthip_ctx* ctx, lthip_plan* plan, uint32_t part_count, const uint64_t* part_offsets /*host*/,
                                  const uint64_t* part_sizes /*host*/);
/* ctx may be NULL (e.g. the creating thread's context is gone): the device is synchronised instead of the stream */
LTHIP_EXPORT void lthip_plan_destroy(lthip_ctx* ctx, lthip_plan* plan);
/* upper bound on the number of chunks the plan can produce (size the output arrays with it) */
LTHIP_EXPORT uint64_t lthip_plan_chunk_capacity(const lthip_plan* plan);
/* 2 when lthip_chunk_hash runs this plan as two slices on two streams (plans of >= 1 GiB in >= 2 parts: the candidate scan of the
 * second half of the parts beside the leaf hashing of the first; results identical to the single pass), else 1.  Per-kernel timings
 * (lthip_timing_get) of the scan and the leaf hashing then OVERLAP: their sum exceeds the wall time of the call. */
LTHIP_EXPORT uint32_t lthip_plan_slices(const lthip_plan* plan);

/* Runs buzhash scan -> cut selection -> compaction -> (if d_chunk_hashes) BLAKE3 on the stream.
 * Outputs (device pointers, capacity >= lthip_plan_chunk_capacity):
 *   d_chunk_offsets[i]  byte offset of chunk i in d_data            (dense, parts in order)
 *   d_chunk_lens[i]     its length
 *   d_chunk_hashes[i]   first 8 BLAKE3 bytes as LE u64              (may be NULL: chunk only)
 *   d_part_first[p]     index of the first chunk of part p, d_part_first[part_count] = total
 * Asynchronous; *out_total (host, may be NULL) is filled after an internal stream sync when given. */
LTHIP_EXPORT int lthip_chunk_hash(lthip_ctx* ctx, const lthip_plan* plan, const void* d_data, uint64_t* d_chunk_offsets,
                                  uint32_t* d_chunk_lens, uint64_t* d_chunk_hashes, uint32_t* d_part_first,
                                  uint64_t* out_total);

/* HPCDCChunker_NextChunkFromBuffer (hpcdcchunker.c:452-523) on a device buffer of `size` > min_chunk bytes:
 * *out_len = length of the chunk at its front.  Reproduces the reference's window-seeding quirk (the rolling
 * window starts from buf[0..48) instead of buf[min-48..min), :488-494), which NextChunk does not have.
 * Synchronous (one small kernel + read-back). */
LTHIP_EXPORT int lthip_chunk_from_buffer(lthip_ctx* ctx, const void* d_data, uint64_t size, uint32_t min_chunk,
                                         uint32_t avg_chunk, uint32_t max_chunk, uint64_t* out_len);

/* Host-side evaluation of the division-free cut test the kernels use (1 when h % d == d-1 for the discriminator
 * d): exported so the arithmetic can be checked exhaustively without a GPU. */
LTHIP_EXPORT int lthip_divtest_eval(uint32_t discriminator, uint32_t hash);

/* BLAKE3-64 of arbitrary device ranges: d_hashes[i] = blake3(d_data + d_offsets[i], d_lens[i]).
 * Same contract as Blake3Hash_HashBuffer (longtail_blake3.c:81-102); len 0 is legal. */
LTHIP_EXPORT int lthip_hash_ranges(lthip_ctx* ctx, const void* d_data, uint64_t range_count, const uint64_t* d_offsets,
                                   const uint32_t* d_lens, uint32_t max_len /*upper bound of d_lens[], 0 = unknown*/,
                                   uint64_t* d_hashes);

/* BLAKE3-64 of ONE input of at most 64 KiB in one launch, read where it lies and answered where `out` points: both must be device
 * accessible (pinned host memory from lthip_malloc_pinned, or device memory).  What the plugin layer's HashBuffer uses for path
 * strings and hash arrays.  Asynchronous on the context's stream; EINVAL above 64 KiB. */
LTHIP_EXPORT int lthip_hash_one(lthip_ctx* ctx, const void* in, uint32_t len, uint64_t* out);

/* Streaming BLAKE3-64 with O(1) state (Blake3Hash_BeginContext/_Hash/_EndContext, longtail_blake3.c:24-79): the caller cuts the
 * stream into batches of LTHIP_B3_STREAM_BATCH bytes (device memory, 16-byte aligned), calls lthip_b3_stream_batch for the batches
 * in order -- a batch only when at least one byte follows it -- and lthip_b3_stream_final with the rest (1 .. one batch of bytes, or
 * 0 bytes after 0 batches: the empty stream).  d_stack: LTHIP_B3_STREAM_STACK_BYTES of device memory per stream; d_out: device or
 * pinned host memory.  Asynchronous on the context's stream.  Streams below 4 TiB. */
#define LTHIP_B3_STREAM_BATCH (1u << 20)
#define LTHIP_B3_STREAM_STACK_BYTES 2048u
LTHIP_EXPORT int lthip_b3_stream_batch(lthip_ctx* ctx, const void* d_data, uint64_t batch_index, void* d_stack);
LTHIP_EXPORT int lthip_b3_stream_final(lthip_ctx* ctx, const void* d_tail, uint32_t tail_len, uint64_t batch_count, const void* d_stack,
                                       uint64_t* d_out);

/* BLAKE3-64 of runs of 64-bit values: d_out[i] = blake3(bytes of d_values[d_first[i] .. d_first[i+1])), i < run_count.  Over the
 * chunk hashes and the part table of lthip_chunk_hash: every part's content hash (src/longtail.c:2518-2537 for a one-part asset). */
/* ..._bounded: the caller knows upper bounds of d_first[run_count] - d_first[0] (all values) and of the longest run: with runs of at
 * most 32768 values the call then never waits for the device (no read-back of the counts).  0 = unknown. */
LTHIP_EXPORT int lthip_hash_runs_u64_bounded(lthip_ctx* ctx, const uint64_t* d_values, const uint32_t* d_first, uint32_t run_count,
                                             uint64_t total_values_bound, uint64_t run_values_bound, uint64_t* d_out);
LTHIP_EXPORT int lthip_hash_runs_u64(lthip_ctx* ctx, const uint64_t* d_values, const uint32_t* d_first, uint32_t run_count,
                                     uint64_t* d_out);

/* ---- phase 2: per-block compression ----------------------------------------------------------------
 * One call compresses a batch of stored blocks (the unit of CompressBlock, compressblockstore.c:67-141).
 * Block b = d_src[src_offsets[b] .. +src_sizes[b]) -> d_dst[dst_offsets[b] ..) with capacity dst_caps[b];
 * d_out_sizes[b] = payload size, or 0 when it does not fit (LZ4CompressionAPI_Compress -> ENOMEM).
 * The offset/size tables are HOST arrays (copied to the device by the call).  The calls queue their work on the context's
 * stream and return (results are ready after lthip_ctx_sync or any later work on the stream); a call of any size is cut
 * into internal batches of LTHIP_BATCH_BYTES of input (environment, default 8 GiB) so that the scratch stays bounded. */
LTHIP_EXPORT size_t lthip_lz4_bound(size_t size); /* LZ4_COMPRESSBOUND, lib/lz4/ext/lz4.h:215 */
LTHIP_EXPORT int lthip_lz4_compress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count,
                                           const uint64_t* src_offsets, const uint32_t* src_sizes, void* d_dst,
                                           const uint64_t* dst_offsets, const uint32_t* dst_caps,
                                           uint32_t* d_out_sizes, int segment_log2 /*0 = default*/);
/* LZ4_decompress_safe semantics per block; d_out_sizes[b] = decoded size or 0xFFFFFFFF on malformed input */
LTHIP_EXPORT int lthip_lz4_decompress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count,
                                             const uint64_t* src_offsets, const uint32_t* src_sizes, void* d_dst,
                                             const uint64_t* dst_offsets, const uint32_t* dst_caps,
                                             uint32_t* d_out_sizes);

/* Greedy packing of unique chunks into stored blocks exactly as Longtail_CreateStoreIndex does it
 * (src/longtail.c:6801-6860: same tag, <= max_chunks_per_block chunks, size <= max_block_size * 1.1).  Host arrays;
 * block_starts[0..*out_block_count] are chunk indices (last entry = chunk_count). */
LTHIP_EXPORT int lthip_pack_blocks(uint64_t chunk_count, const uint32_t* chunk_lens, uint32_t max_block_size,
                                   uint32_t max_chunks_per_block, uint64_t* block_starts, uint64_t capacity,
                                   uint64_t* out_block_count);

/* The same greedy rule, resumable: packs from chunk `first_chunk` until the batch holds max_batch_bytes of chunk data or
 * the blocks' codec bounds (size + size / bound_div + bound_add, rounded up to 64) fill arena_bytes -- always at least one
 * block.  block_starts[0..n] (n = *out_block_count, last entry = *out_next_chunk) and block_sizes[0..n) are host arrays of
 * `capacity` entries.  Lets the caller pack batch k+1 while the device compresses batch k. */
LTHIP_EXPORT int lthip_pack_blocks_batch(uint64_t chunk_count, const uint32_t* chunk_lens, uint64_t first_chunk,
                                         uint32_t max_block_size, uint32_t max_chunks_per_block, uint64_t max_batch_bytes,
                                         uint64_t arena_bytes, uint32_t bound_div, uint32_t bound_add,
                                         uint64_t* block_starts, uint64_t* block_sizes, uint64_t capacity,
                                         uint64_t* out_block_count, uint64_t* out_next_chunk);

/* Block assembly (WriteContentBlockJob, src/longtail.c:4640-4721) as a device gather:
 * d_dst[d_dst_offsets[i] ..) = d_src[d_src_offsets[i] .. + d_lens[i]) for every range (all tables on the device). */
LTHIP_EXPORT int lthip_gather_ranges(lthip_ctx* ctx, const void* d_src, uint64_t range_count, const uint64_t* d_src_offsets,
                                     const uint32_t* d_lens, void* d_dst, const uint64_t* d_dst_offsets);

/* ZStd (ZStdCompressionAPI_Compress, lib/zstd/longtail_zstd.c:105-142): one zstd frame per block, 128 KiB pieces
 * stored as RLE / Compressed (LZ sequences + Huffman literals + FSE) / Raw blocks; decodable by the reference's
 * ZSTD_decompressDCtx.  A compressed piece is a run of small zstd blocks (one per 4 KiB of content, one set of entropy
 * tables per piece) and the frame ends with a skippable frame holding the directory of block sizes, which lets
 * lthip_zstd_decompress_blocks decode every block on a lane of its own (LTHIP_ZSTD_SUB=0: one block per
 * piece).  Same calling convention as lthip_lz4_compress_blocks. */
LTHIP_EXPORT size_t lthip_zstd_bound(size_t size); /* ZSTD_COMPRESSBOUND, lib/zstd/ext/zstd.h:232 */
LTHIP_EXPORT int lthip_zstd_compress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count,
                                            const uint64_t* src_offsets, const uint32_t* src_sizes, void* d_dst,
                                            const uint64_t* dst_offsets, const uint32_t* dst_caps,
                                            uint32_t* d_out_sizes);
/* The same with the parse the reference's settings ids stand for (lib/zstd/longtail_zstd.c:11-28, 43-60: 'ztd1' -> level 0 = the
 * default 3, 'ztd2' -> 3, 'ztd4' -> 8, 'ztd3' -> 22, 'ztd5' -> its own type id, which zstd clamps to 22; any other id -> 0 = the
 * default, it is NOT rejected there and is not here):
 *   LTHIP_ZSTD_Q_DEFAULT  'ztd1', 'ztd2', unknown ids: the lane parser's greedy parse (what lthip_zstd_compress_blocks runs); a match
 *                         of four bytes must start within 1 KiB, one of five within 4 KiB (the offset's bits are written out: a short
 *                         match from far away costs more than the literals it replaces)
 *   LTHIP_ZSTD_Q_HIGH     'ztd4': every redundant 32 KiB half of a 128 KiB piece but the piece's first is parsed with the 32 KiB in
 *                         front of it as HISTORY (matches reach 32 .. 64 KiB back wherever the half lies; by default a half only sees
 *                         what its 64 KiB group holds in front of it).  About half the match finder's throughput.
 *   LTHIP_ZSTD_Q_MAX      'ztd3', 'ztd5': the history ALSO for a piece's first half -- matches reach into the piece before -- except in
 *                         every eighth piece of a block, and the wave's table is read again after the step's inserts.  The frame's
 *                         trailer says so (directory version 4): this library's decoder runs the eight pieces in between as a CHAIN
 *                         (a piece is executed when the one before it is complete; the chains of a call side by side).  On "tokens"
 *                         2.92 / 3.33 / 3.76 (default / high / max), records 3.00 / 3.12 / 3.24 = the reference encoder's default;
 *                         restore of such frames 0.7-0.9 x the other settings' rate at 512 blocks per call, one block 2-6 ms
 *                         against 0.5-1.4 (profiles/r05_zstd_ratio_table.txt, r05_zstd_quality_rates.txt).
 * Every quality writes standard zstd frames in the sub-block layout; at the first two the pieces of a frame are independent of each
 * other; each is smaller than the one before on the synthetic kinds with structure.
 * lthip_zstd_quality_of_settings: the quality a settings id ('ztd?' as a big-endian u32, the value blocks carry) stands for. */
#define LTHIP_ZSTD_Q_DEFAULT 0
#define LTHIP_ZSTD_Q_HIGH 1
#define LTHIP_ZSTD_Q_MAX 2
LTHIP_EXPORT int lthip_zstd_quality_of_settings(uint32_t settings_id);
LTHIP_EXPORT int lthip_zstd_compress_blocks_q(lthip_ctx* ctx, const void* d_src, uint32_t block_count,
                                              const uint64_t* src_offsets, const uint32_t* src_sizes, void* d_dst,
                                              const uint64_t* dst_offsets, const uint32_t* dst_caps,
                                              uint32_t* d_out_sizes, int quality);
/* ZStdCompressionAPI_Decompress (longtail_zstd.c:144-177): every payload is one or more zstd frames (any encoder's: Huffman /
 * FSE / repeat modes / repeat offsets / skippable frames; no dictionaries; a content checksum is skipped, not verified).
 * d_out_sizes[b] = decoded size, or 0xFFFFFFFF when the payload is malformed or does not fit dst_caps[b]. */
LTHIP_EXPORT int lthip_zstd_decompress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count,
                                              const uint64_t* src_offsets, const uint32_t* src_sizes, void* d_dst,
                                              const uint64_t* dst_offsets, const uint32_t* dst_caps,
                                              uint32_t* d_out_sizes);
/* Diagnostics (tests): what the last lthip_zstd_decompress_blocks call of this process did -- out[0] payloads, out[1] blocks of
 * other encoders' frames that were listed for the block-parallel decoder, out[2] payloads a lane-parallel decoder gave back to the
 * serial one, out[3] where the first of those was sent back (a source line of k_zstd.hip).  Waits for the context's stream. */
LTHIP_EXPORT int lthip_zstd_last_decode_stats(lthip_ctx* ctx, uint32_t out[4]);
/* Tests only: the library caches its environment switches (LTHIP_ZSTD_DBG, LTHIP_LZ4_PD_WAIT, LTHIP_LZ4_SHARED, ...) per process; after
 * this call they are read again, so that one process can run a path and its ablation. */
LTHIP_EXPORT void lthip_debug_reload_env(void);
/* Tests only, ABLATION build only (the product library answers ENOTSUP / -1 and has no counter in its allocation path): make the
 * library's own device / pinned allocations number after+1 .. after+count FROM NOW fail with out-of-memory (after < 0: off);
 * lthip_debug_alloc_calls = allocations attempted by this process so far, *out_failed = how many were made to fail.  The device-side
 * counterpart of the reference's FailableStorageAPI tests (test/test.cpp:5677-5752): ENOMEM must come out of CreateVersionIndex /
 * WriteContent, nothing may leak, the same objects must work on the next call. */
LTHIP_EXPORT int lthip_debug_fail_alloc(int64_t after, int64_t count);
LTHIP_EXPORT int64_t lthip_debug_alloc_calls(int64_t* out_failed);
/* Diagnostics (parity tests): match-finder output of the last lthip_zstd_compress_blocks call (its last internal batch:
 * calls above LTHIP_BATCH_BYTES = 8 GiB of input are processed in several) on this context for the
 * 4 KiB units [first, first + count) -- 16 bytes of meta {nseq, nlit, tail, 0}, 4096 literal bytes and 1024 u64
 * records {lit | mlen << 16 | offset << 32} per unit (host buffers, any may be NULL).  A unit with nseq == 0 has no
 * literal buffer (its 4096 bytes here are unspecified): its literals are its source bytes. */
LTHIP_EXPORT int lthip_zstd_debug_units(lthip_ctx* ctx, uint64_t first, uint64_t count, void* h_meta, void* h_lits,
                                        void* h_recs);

/* ---- dedup (serial first-seen pass of Longtail_CreateVersionIndex, src/longtail.c:2951-2970) --------
 * d_first_index[i] = smallest j with d_hashes[j] == d_hashes[i]; *d_unique_count = number of i with
 * d_first_index[i] == i. */
LTHIP_EXPORT int lthip_dedup_first_seen(lthip_ctx* ctx, uint64_t count, const uint64_t* d_hashes,
                                        uint32_t* d_first_index, uint64_t* d_unique_count);
/* Multi-GPU form: all `count` (all-gathered) hashes go into the table, but only this rank's own range
 * [lookup_first, lookup_first + lookup_count) is answered (d_first_index[j] for hash lookup_first + j, global indices);
 * *d_unique_count = distinct hashes among all `count`. */
LTHIP_EXPORT int lthip_dedup_first_seen_range(lthip_ctx* ctx, uint64_t count, const uint64_t* d_hashes, uint64_t lookup_first,
                                              uint64_t lookup_count, uint32_t* d_first_index, uint64_t* d_unique_count);

/* Hash-range-sharded form of the first-seen pass (multi-GPU): this rank holds an arbitrary subset of the tree's chunk hashes, each
 * with its global chunk position; d_first_ordinal[j] = smallest position among the subset's items with the hash of item j.  The
 * ranks route every chunk to the owner of its hash (longtail_amd/dist.py: sharded_first_seen), so a rank inserts 1/N of the tree's
 * chunks instead of all of them. */
LTHIP_EXPORT int lthip_dedup_min_ordinal(lthip_ctx* ctx, uint64_t count, const uint64_t* d_hashes, const uint32_t* d_ordinals,
                                         uint32_t* d_first_ordinal, uint64_t* d_unique_count);

/* ---- bulk Longtail_CreateVersionIndex tail (src/longtail.c:2808-3017, layout :2551-2584, :2709-2806) ---
 * From the device-resident chunk lists of lthip_chunk_hash -- all assets' chunks concatenated in (asset, part, chunk) order,
 * asset a owning asset_chunk_counts[a] of them -- to the SERIALIZED VersionIndex (the bytes Longtail_WriteVersionIndexToBuffer
 * produces, :3415): first-seen unique chunk list + per-asset-chunk indexes, content hash per asset (BLAKE3 of its chunk-hash
 * array), path hashes.  The file list is a struct Longtail_FileInfos taken apart (src/longtail.h:1684-1692); directories are
 * assets with zero chunks.  Host arrays unless marked d_.  Returns ENOMEM with *out_size set when `out` is too small. */
LTHIP_EXPORT size_t lthip_version_index_size(uint32_t asset_count, uint64_t unique_chunk_count, uint64_t asset_chunk_index_count,
                                             uint32_t path_data_size);
LTHIP_EXPORT int lthip_build_version_index(lthip_ctx* ctx, uint32_t asset_count, const uint64_t* asset_sizes,
                                           const uint32_t* path_start_offsets, const uint16_t* permissions, const char* path_data,
                                           uint32_t path_data_size, const uint32_t* asset_chunk_counts, uint64_t chunk_total,
                                           const uint64_t* d_chunk_hashes, const uint32_t* d_chunk_lens,
                                           const uint32_t* asset_tags /* may be NULL */, uint32_t hash_identifier,
                                           uint32_t target_chunk_size, void* out, size_t out_capacity, size_t* out_size);

/* ---- stored blocks ------------------------------------------------------------------
 * The bytes of a stored block file (Longtail_WriteStoredBlockToBuffer, src/longtail.c:4111-4150) around a compressed
 * payload: BlockIndex data (block hash = hash of the chunk-hash array :3753-3757, hash identifier, chunk count, tag,
 * chunk hashes, chunk sizes, layout :3585-3601) then [u32 raw size][u32 compressed size] (compressblockstore.c:103-139).
 * Compress block b to  image_offsets[b] + lthip_stored_block_header_size(chunks of b)  first; this call fills in the rest,
 * so the image of block b is  d_arena[image_offsets[b] .. + header size + compressed size).  Block b holds the chunks
 * [block_first_chunk[b], block_first_chunk[b+1]) of the (unique, first-seen ordered) device arrays; image offsets must be
 * 8-byte aligned; host arrays unless marked d_. */
LTHIP_EXPORT size_t lthip_stored_block_header_size(uint32_t chunk_count);
LTHIP_EXPORT int lthip_write_stored_block_headers(lthip_ctx* ctx, uint32_t block_count, const uint64_t* block_first_chunk,
                                                  const uint64_t* d_chunk_hashes, const uint32_t* d_chunk_lens,
                                                  uint32_t hash_identifier, uint32_t tag, const uint32_t* raw_sizes,
                                                  const uint32_t* d_comp_sizes, void* d_arena, const uint64_t* image_offsets);

/* ---- bulk Longtail_CreateMissingContent (src/longtail.c:6882-6998 with DiffHashes :6620-6743 and
 * Longtail_CreateStoreIndex :6745-6880) -------------------------------------------------------------------------------
 * Which of the version's chunks (unique list, version order: device hashes + sizes, host tags or NULL) does a store holding
 * d_existing_hashes lack, and how are they packed into blocks?  Output: the serialized StoreIndex of the missing content
 * (the bytes Longtail_WriteStoreIndexToBuffer produces: blocks with their BLAKE3 block hashes, chunk lists, tags).
 * Returns ENOMEM with *out_size set when `out` is too small. */
LTHIP_EXPORT int lthip_create_missing_content(lthip_ctx* ctx, uint64_t existing_count, const uint64_t* d_existing_hashes,
                                              uint64_t chunk_count, const uint64_t* d_chunk_hashes, const uint32_t* d_chunk_lens,
                                              const uint32_t* chunk_tags, uint32_t hash_identifier, uint32_t max_block_size,
                                              uint32_t max_chunks_per_block, void* out, size_t out_capacity, size_t* out_size);

/* ---- bulk Longtail_GetExistingStoreIndex (src/longtail.c:7087-7325) -------------------------------------------
 * Which blocks of a store (its serialized StoreIndex, host memory, the bytes Longtail_WriteStoreIndexToBuffer produces) cover the
 * given chunk hashes (device)?  Blocks are kept when at least min_block_usage_percent of their bytes are wanted, walked most-used
 * first, and taken when they hold a wanted chunk no earlier block of the walk held.  Output: the serialized StoreIndex of the taken
 * blocks (== Longtail_GetExistingStoreIndex + Longtail_WriteStoreIndexToBuffer, including the reference's habit of reading a taken
 * block's tag at its first chunk's index).  Returns ENOMEM with *out_size set when `out` is too small, EBADF for a malformed index. */
LTHIP_EXPORT int lthip_get_existing_store_index(lthip_ctx* ctx, const void* store_index, size_t store_index_size, uint64_t chunk_count,
                                                const uint64_t* d_chunk_hashes, uint32_t min_block_usage_percent, void* out,
                                                size_t out_capacity, size_t* out_size);

/* ---- the ingest metric as one native session (CreateVersionIndex + CreateMissingContent + WriteContent) --------
 * For assets already resident in HBM.  The caller runs lthip_chunk_hash over its own jobs (one part per job, ascending job order),
 * then
 *   lthip_ingest_index   tail of Longtail_CreateVersionIndex (src/longtail.c:2808-3017: first-seen pass :2951-2970, content hashes
 *                        :2518-2537, path hashes :1269-1300, serialized layout :2551-2584) + Longtail_CreateMissingContent
 *                        (:6882-6998) for the chunks THIS rank writes: packing (:6801-6860), block hashes (:3753-3757)
 *   lthip_ingest_write   Longtail_WriteContent (:4760; WriteContentBlockJob :4559-4758, CompressBlock compressblockstore.c:67-141):
 *                        block assembly on the device only for blocks that are not one byte range of the data, per-block codec
 *                        straight into the stored-block image, BlockIndex + [raw][compressed] around it (:4111-4150).  The images
 *                        are produced batch after batch in the caller's arena and dropped (null sink).
 *   lthip_ingest_finish  serialized StoreIndex of what was written (:8913-8931), the session's one full synchronisation, statistics.
 * Single GPU: the "all" arrays are the local ones and tree->my_jobs is NULL.  Multi-GPU: the "all" arrays hold every rank's
 * chunks in job order (see lthip_exchange_layout) and tree->my_jobs lists this rank's jobs; a rank writes the chunks that are
 * first-seen and lie in its own jobs, i.e. Longtail_CreateMissingContent against a store that already holds the other ranks'
 * chunks.  h_version_index (may be NULL: this rank does not serialize the index) and h_store_index should be pinned memory so
 * the copies overlap the kernels.  asset_tags NULL = every asset carries cfg.compression_type (what UpSync passes).
 * Lifetimes.  The host arrays of `tree` (sizes, paths, permissions, tags, job tables) may be freed or reused as soon as
 * lthip_ingest_index has returned (the session keeps its own copy of what it reads later: round 4; round 3 read the caller's arrays from
 * a helper thread until lthip_ingest_finish).  The DEVICE arrays (d_all_hashes, d_all_lens, d_local_*) and the VersionIndex buffer must
 * stay valid, and the buffer unread, until lthip_ingest_finish has returned: the index is serialized by a helper thread next to
 * lthip_ingest_write, and the packing into blocks is finished there too (the result's block count comes from lthip_ingest_finish). */
enum lthip_codec
{
    LTHIP_CODEC_NONE = 0,
    LTHIP_CODEC_LZ4 = 1,
    LTHIP_CODEC_ZSTD = 2
};
typedef struct lthip_ingest lthip_ingest;
typedef struct lthip_ingest_config
{
    uint32_t target_chunk_size;    /* recorded in the VersionIndex */
    uint32_t hash_identifier;      /* 0x626c6b33 */
    uint32_t max_block_size;       /* cmd/main.c:3006-3009 defaults: 8 MiB */
    uint32_t max_chunks_per_block; /*                                 1024  */
    uint32_t compression_type;     /* the tag stored with chunks and blocks: 'lz42', 'ztd1'..'ztd5' */
    uint32_t codec;                /* enum lthip_codec */
    uint64_t batch_bytes;          /* raw bytes per codec batch, 0 = 8 GiB */
} lthip_ingest_config;
typedef struct lthip_ingest_tree
{
    uint32_t asset_count;               /* struct Longtail_FileInfos taken apart (src/longtail.h:1684-1692) */
    const uint64_t* asset_sizes;
    const uint32_t* path_start_offsets;
    const uint16_t* permissions;
    const char* path_data;
    uint32_t path_data_size;
    const uint32_t* asset_tags;         /* may be NULL */
    uint64_t job_count;                 /* lthip_make_jobs */
    const uint32_t* job_asset;
    const uint64_t* job_first;          /* [job_count + 1] index of each job's first chunk in the "all" arrays */
    uint64_t my_job_count;
    const uint64_t* my_jobs;            /* ascending job indices of this rank; NULL = all jobs */
} lthip_ingest_tree;
typedef struct lthip_ingest_result
{
    uint64_t struct_size;                 /* IN: sizeof(lthip_ingest_result) of the caller's header; the library fills at most that */
    uint64_t chunks_all, unique_all;      /* chunks / distinct chunks of the whole tree */
    uint64_t chunks_local, unique_local;  /* chunks of this rank's jobs / those of them this rank writes */
    uint64_t blocks, raw_bytes, compressed_bytes, gathered_blocks;
    uint64_t version_index_size, store_index_size;
    uint64_t gathered_bytes;              /* bytes of the blocks that went through the device block assembly */
} lthip_ingest_result;
LTHIP_EXPORT int lthip_ingest_create(lthip_ctx* ctx, const lthip_ingest_config* config, lthip_ingest** out);
LTHIP_EXPORT void lthip_ingest_destroy(lthip_ingest* ingest);
LTHIP_EXPORT int lthip_ingest_index(lthip_ingest* ingest, const lthip_ingest_tree* tree, const uint64_t* d_all_hashes,
                                    const uint32_t* d_all_lens, uint64_t all_chunks, const uint64_t* d_local_offsets,
                                    const uint32_t* d_local_part_first, uint64_t local_chunks, void* h_version_index,
                                    size_t version_index_capacity);
/* Multi-GPU with the hash-range-sharded first-seen table: d_first_index[i] = position of the first chunk (job order, all ranks) with
 * the hash of chunk i, for all `all_chunks` chunks of the NEXT lthip_ingest_index call, and the tree's number of distinct hashes; that
 * call then skips its own table pass (every rank inserting every rank's hashes).  The array must stay valid until the call returns. */
LTHIP_EXPORT int lthip_ingest_set_first_seen(lthip_ingest* ingest, const uint32_t* d_first_index, uint64_t unique_chunks);
LTHIP_EXPORT int lthip_ingest_write(lthip_ingest* ingest, const void* d_data, void* d_arena, uint64_t arena_bytes);
LTHIP_EXPORT int lthip_ingest_finish(lthip_ingest* ingest, void* h_store_index, size_t store_index_capacity,
                                     lthip_ingest_result* out_result);
/* The stored-block images of the LAST codec batch of lthip_ingest_write (host tables owned by the session, valid after
 * lthip_ingest_finish until the next lthip_ingest_index): blocks *out_first_block .. + *out_count of the session; image i lies at
 * d_arena + offsets[i] and is sizes[i] bytes long -- BlockIndex, [raw size][compressed size], payload: exactly what
 * Longtail_WriteStoredBlockToBuffer produces and PutStoredBlock receives (src/longtail.c:4111-4150, 4722-4757).  A session whose
 * write fit ONE batch (raw bytes <= cfg.batch_bytes, and the arena) has all of its images there: the host-fed loop of INTEGRATION.md
 * (pinned slice -> H2D -> lthip_chunk_hash -> session -> images D2H) downloads them through this table. */
LTHIP_EXPORT int lthip_ingest_images(const lthip_ingest* ingest, uint64_t* out_first_block, uint64_t* out_count,
                                     const uint64_t** out_offsets, const uint32_t** out_sizes);
/* per-block compressed sizes of the last lthip_ingest_write (host, valid after lthip_ingest_finish) */
LTHIP_EXPORT const uint32_t* lthip_ingest_compressed_sizes(const lthip_ingest* ingest);

/* ---- multi-GPU work division, host functions -----------------------------------------------------------
 * The unit of independence is the reference's own job: one (asset, target_chunk_size*1024-byte part) of ChunkAssets
 * (src/longtail.c:2396-2458).  lthip_job_count / lthip_make_jobs list the jobs of a tree exactly as :2399-2404 / :2432-2457 do
 * (asset order, 1 + size / part jobs per asset: an exact multiple ends with an empty job, a directory is one empty job).
 * lthip_partition_jobs assigns them to ranks -- deterministic, so every rank computes the same table without communication:
 *   LTHIP_PARTITION_RANGE  contiguous job ranges with equal byte shares (rank-major order == job order; an asset's parts may
 *                          straddle ranks: intra-file segment sharding, BASELINE.json configs[4])
 *   LTHIP_PARTITION_LPT    longest processing time first: jobs by size descending, each to the least loaded rank
 *   LTHIP_PARTITION_MOD    job index mod rank_count (uniform trees)
 * rank_bytes[r] (may be NULL) receives the bytes assigned to rank r. */
enum lthip_partition_policy
{
    LTHIP_PARTITION_RANGE = 0,
    LTHIP_PARTITION_LPT = 1,
    LTHIP_PARTITION_MOD = 2
};
LTHIP_EXPORT uint64_t lthip_job_count(uint32_t asset_count, const uint64_t* asset_sizes, uint32_t target_chunk_size);
LTHIP_EXPORT int lthip_make_jobs(uint32_t asset_count, const uint64_t* asset_sizes, uint32_t target_chunk_size, uint64_t capacity,
                                 uint32_t* job_asset, uint64_t* job_offset /* within the asset */, uint64_t* job_size);
LTHIP_EXPORT int lthip_partition_jobs(uint64_t job_count, const uint64_t* job_sizes, uint32_t rank_count, int policy,
                                      uint32_t* job_rank, uint64_t* rank_bytes);
/* The exchange: every rank chunks + hashes its jobs in ascending job order, then the ranks all-gather (1) their per-job chunk
 * counts, padded to count_stride entries per rank, and (2) their chunk hash / length arrays, padded to chunk_stride entries per
 * rank.  This function says where job j's run of chunks sits in the gathered arrays (job_src[j], element index) and where it
 * belongs in job order (job_dst[j]; job_dst[job_count] = total chunks) -- the order ChunkAssets concatenates in (:2499-2517),
 * which the serial first-seen pass (:2951-2970) depends on.  One lthip_gather_ranges per array applies it.  EINVAL when the
 * counts disagree with the assignment. */
LTHIP_EXPORT int lthip_exchange_layout(uint64_t job_count, const uint32_t* job_rank, uint32_t rank_count,
                                       const uint32_t* gathered_counts, uint64_t count_stride, uint64_t chunk_stride,
                                       uint64_t* job_src, uint64_t* job_dst /* job_count + 1 */, uint32_t* job_chunks /* may be NULL */);
/* The layout as the device reorder wants it: runs of jobs that are contiguous on both sides merged (range policy: one run per rank),
 * then cut into pieces of at most max_piece elements (a workgroup per piece).  Returns the number of pieces; out arrays NULL or
 * capacity too small: count only.  Host function, O(jobs). */
LTHIP_EXPORT uint64_t lthip_exchange_ranges(uint64_t job_count, const uint64_t* job_src, const uint64_t* job_dst,
                                            const uint32_t* job_chunks, uint64_t max_piece, uint64_t capacity, uint64_t* out_src,
                                            uint64_t* out_dst, uint32_t* out_cnt);
/* Applies it on the device: d_out[range_dst[i] ..) = d_gathered[range_src[i] .. + range_cnt[i]) in elements of elem_bytes (host
 * tables, staged through the context's pinned ring; asynchronous on the context's stream).  Every array of the exchange -- hashes,
 * lengths, the first-seen answers -- goes through the same tables. */
LTHIP_EXPORT int lthip_exchange_reorder(lthip_ctx* ctx, const void* d_gathered, void* d_out, uint32_t elem_bytes,
                                        uint64_t range_count, const uint64_t* range_src, const uint64_t* range_dst,
                                        const uint32_t* range_cnt);
/* Positions of this rank's chunks in job order (what the sharded first-seen table is keyed with): d_out[k] = global_first[m] +
 * (k - local_first[m]) for the own job m that holds local chunk k; local_first / global_first: host arrays of my_job_count entries,
 * index of each own job's first chunk in the rank's own lists / in the tree's job-ordered lists. */
LTHIP_EXPORT int lthip_job_ordinals(lthip_ctx* ctx, uint64_t my_job_count, const uint32_t* local_first, const uint32_t* global_first,
                                    uint64_t local_chunks, uint32_t* d_out);

/* The collective itself behind the C ABI (comm.hip): RCCL's all-gather on the context's stream, one process per GPU.  RCCL is
 * bound at run time (dlopen): ENOSYS where it is missing.  Rank 0 makes the 128-byte id, the embedder carries it to the other
 * processes (any side channel), every rank creates its communicator with it.  lthip_comm_allgather: every rank contributes
 * `count` elements of `elem_bytes` from d_send; d_recv receives rank r's contribution at element r * count (the padded
 * all-gathers lthip_exchange_layout describes).  Asynchronous on the context's stream like every other bulk call.
 *
 * lthip_comm_alltoallv: the exchange of the SHARDED first-seen table (the serial pass of src/longtail.c:2951-2970 restated as "minimum
 * position per hash", the table split by hash over the ranks): rank r sends send_counts[p] elements starting at element
 * send_displs[p] of d_send to every rank p and receives recv_counts[p] elements from it at element recv_displs[p] of d_recv
 * (host arrays of nranks entries; recv_counts[p] must equal rank p's send_counts[r] -- the ranks learn them by an all-gather of
 * their send counts).  RCCL: one group of ncclSend / ncclRecv pairs on the context's stream -- point to point, the natural shape on
 * xGMI.  lthip_comm_info: the communicator's size AS THE TRANSPORT REPORTS IT (ncclCommCount), this process's rank, the transport.
 *
 * Transports.  LTHIP_COMM_RCCL is the product path.  LTHIP_COMM_SHM is a stand-in for boxes without N GPUs: lthip_comm_unique_id
 * makes a shared-memory id when LTHIP_COMM_TRANSPORT=shm is set in the environment of the process that makes the id, and every
 * lthip_comm_create that receives such an id attaches to the segment /dev/shm/lthip_comm_<id>; the same entry points then move the
 * bytes through host memory, synchronously (with ctx == NULL the pointers are host pointers: the CPU tests).  It exists so that the
 * torch-free launch, the id hand-over and the exchange can be exercised with N processes on one GPU; it is never a measured path. */
#define LTHIP_COMM_ID_BYTES 128
#define LTHIP_COMM_RCCL 1
#define LTHIP_COMM_SHM 2
typedef struct lthip_comm lthip_comm;
LTHIP_EXPORT int lthip_comm_unique_id(void* id128);
LTHIP_EXPORT int lthip_comm_create(lthip_ctx* ctx, int nranks, int rank, const void* id128, lthip_comm** out);
LTHIP_EXPORT int lthip_comm_destroy(lthip_comm* comm);
/* The file RCCL was bound from -- $LTHIP_RCCL_PATH, else a librccl already mapped into the process (torch's), else librccl.so.1 /
 * librccl.so on the loader's path and in /opt/rocm/lib, else beside a mapped libtorch -- or, when none was found, what was tried;
 * *out_how (may be NULL) names the rule.  Loads the library on first use. */
LTHIP_EXPORT const char* lthip_comm_library(const char** out_how);
LTHIP_EXPORT int lthip_comm_info(const lthip_comm* comm, int* out_nranks, int* out_rank, int* out_transport);
LTHIP_EXPORT int lthip_comm_allgather(lthip_ctx* ctx, lthip_comm* comm, const void* d_send, void* d_recv, uint64_t count,
                                      uint32_t elem_bytes);
LTHIP_EXPORT int lthip_comm_alltoallv(lthip_ctx* ctx, lthip_comm* comm, const void* d_send, const uint64_t* send_counts /*host*/,
                                      const uint64_t* send_displs /*host*/, void* d_recv, const uint64_t* recv_counts /*host*/,
                                      const uint64_t* recv_displs /*host*/, uint32_t elem_bytes);

/* ---- synthetic assets (include/longtail_synth.h), bench/test input generator ------------------------ */
LTHIP_EXPORT int lthip_synth_fill(lthip_ctx* ctx, void* d_dst, uint32_t asset_count, const uint64_t* asset_offsets /*host*/,
                                  const uint64_t* asset_sizes /*host*/, const uint64_t* asset_seeds /*host*/, int kind);
/* the same for RANGES of assets: range i = bytes [asset_skips[i], asset_skips[i] + asset_sizes[i]) of the asset with seed
 * asset_seeds[i] (skips are multiples of 16) -- a rank generates only the parts of a large asset it owns */
LTHIP_EXPORT int lthip_synth_fill_ranges(lthip_ctx* ctx, void* d_dst, uint32_t asset_count, const uint64_t* asset_offsets /*host*/,
                                         const uint64_t* asset_sizes /*host*/, const uint64_t* asset_seeds /*host*/,
                                         const uint64_t* asset_skips /*host, may be NULL*/, int kind);

#ifdef __cplusplus
}
#endif
#endif
