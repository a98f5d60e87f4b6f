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

#endif
