/* plugin_common.h -- shared pieces of the plain-C plugin layer (host code "stays in C", BASELINE.json):
 * the longtail API structs are implemented in C99 on top of the lthip_* C ABI; no HIP headers here. */
#ifndef LTHIP_PLUGIN_COMMON_H
#define LTHIP_PLUGIN_COMMON_H

#include "../../../include/longtail_hip.h"

#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* allocator hooks (Longtail_Hip_SetAllocator) */
void* ltp_alloc(const char* context, size_t size);
void ltp_free(void* p);

/* The calling thread's GPU context (created on first use on the configured device, destroyed when the
 * thread exits).  Returns 0 when no GPU context can be created -- callers then fail with ENODEV: there
 * is deliberately no CPU fallback anywhere in this library. */
lthip_ctx* ltp_thread_ctx(void);
int ltp_device(void);

/* growable per-purpose device / pinned staging buffers owned by a thread context */
struct ltp_buf
{
    void* p;
    size_t cap;
};
int ltp_dev_reserve(lthip_ctx* ctx, struct ltp_buf* b, size_t bytes);
int ltp_pin_reserve(lthip_ctx* ctx, struct ltp_buf* b, size_t bytes);

struct ltp_thread_state
{
    lthip_ctx* ctx;
    struct ltp_buf d_in, d_out, d_aux, h_pin;
};
struct ltp_thread_state* ltp_thread_state_get(void);

/* ---- registry of windows handed out by HIP chunkers, consulted by the HIP HashAPI ---- */
struct ltp_window
{
    const uint8_t* base;     /* pinned host window */
    uint64_t size;           /* valid bytes */
    const uint64_t* offsets; /* window-relative chunk offsets (host), ascending */
    const uint32_t* lens;
    const uint64_t* hashes;
    uint32_t count;
};
/* slot = ltp_window_register(); update with ltp_window_publish(slot, &w); remove with ltp_window_unregister(slot) */
int ltp_window_register(void);
void ltp_window_publish(int slot, const struct ltp_window* w);
void ltp_window_unregister(int slot);
/* returns 1 and the digest when (data,len) is exactly a chunk of a published window */
int ltp_window_lookup(const void* data, uint32_t len, uint64_t* out_hash);
/* The window the CALLING thread's chunker handed ranges out of last (DynamicChunking calls NextChunk and HashBuffer alternately on
 * one thread, src/longtail.c:2231-2296): looked at first, without any lock.  -1 clears it. */
void ltp_window_set_current(int slot);

/* ---- bounded pool of chunker windows: pinned host window + device window + result tables + a plan of that capacity ----
 * Two size classes (LTP_WINDOW_SMALL / LTP_WINDOW_LARGE bytes); the number of windows alive per class is capped
 * (LONGTAIL_HIP_SMALL_WINDOWS, default 256; LONGTAIL_HIP_LARGE_WINDOWS, default 32: at most 0.5 + 2 GiB of pinned memory and as
 * much HBM), a thread that needs one beyond the cap waits for a release.  Windows that do not fit a class (4 * max_chunk above the
 * large size) are allocated for their chunker alone and freed with it. */
#define LTP_WINDOW_SMALL (2u << 20)
#define LTP_WINDOW_LARGE (64u << 20) /* one reference part at target_chunk_size 65536 (src/longtail.c:2396) */
struct ltp_chunk_window
{
    uint8_t* h_win; /* pinned */
    void* d_win;
    uint64_t cap;   /* bytes */
    uint64_t ccap;  /* chunk slots of the result tables */
    uint32_t min_chunk; /* the chunk parameters the plan was made for */
    uint32_t avg_chunk, max_chunk;
    uint64_t* d_off;
    uint32_t* d_len;
    uint64_t* d_hash;
    uint32_t* d_first;
    uint64_t* h_off; /* pinned */
    uint32_t* h_len;
    uint64_t* h_hash;
    lthip_plan* plan; /* one part of `cap` bytes, re-aimed per refill (lthip_plan_resize_single) */
    int cls;          /* 0 small, 1 large, 2 private */
    struct ltp_chunk_window* next;
};
/* a window of at least `bytes` for chunks of (min, avg, max); blocks while the class is at its cap; 0 + *err on failure */
struct ltp_chunk_window* ltp_window_acquire(lthip_ctx* ctx, uint64_t bytes, uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk, int* err);
void ltp_window_release(struct ltp_chunk_window* w);
void ltp_window_pool_trim(void); /* frees the idle windows (called when the last HIP ChunkerAPI is disposed) */
uint64_t ltp_window_pool_pinned_bytes(void);

/* ---- plugin_batch.c: chunk + hash of a SMALL window (have <= LTP_WINDOW_SMALL bytes in w->h_win) through the dispatcher thread,
 * together with whatever other threads have queued; fills w->h_off / h_len / h_hash like the direct path and blocks until done ---- */
int ltp_batch_chunk_hash(struct ltp_chunk_window* w, uint64_t have, uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk,
                         uint64_t* out_total);
void ltp_batch_shutdown(void);
/* content-hash memo of the batcher: 1 + the digest when `data` is byte for byte a digest array whose BLAKE3 the GPU has computed */
int ltp_memo_get(const void* data, uint32_t length, uint64_t* out_hash);

/* ---- plugin_codec_batch.c: one block (on the device, in the calling thread's buffers) through codec (0 LZ4, 1 zstd) together with the
 * blocks other threads have queued; *produced as the bulk entry points report it ---- */
int ltp_codec_batch(int codec, int decompress, int quality, const void* d_in, uint32_t n, void* d_out, uint32_t cap, uint32_t* produced);
void ltp_codec_batch_shutdown(void);

/* ---- error latch: void / value-returning entry points of the plugin structs (HashAPI.Hash, EndContext) cannot report failure;
 * the first errno of such a call on a thread is kept until read.  Exported as Longtail_Hip_GetLastError(). ---- */
void ltp_latch_error(int err);

#endif
