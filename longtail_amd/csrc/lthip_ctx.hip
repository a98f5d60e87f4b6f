// lthip_ctx.hip -- context, scratch pools, per-kernel event timing, plans and the phase-1 orchestration
// of liblongtail_hip.so.  Host code only; the kernels live in k_*.hip.
#include "lthip_internal.h"

#include <stdlib.h>

// ---------------------------------------------------------------------------------------------------
// errors / scratch
// ---------------------------------------------------------------------------------------------------
int lthip_fail(lthip_ctx* ctx, int code, const char* what, const char* detail)
{
    if (ctx)
        snprintf(ctx->err, sizeof ctx->err, "%s: %s", what ? what : "?", detail ? detail : "");
    return code;
}

int lthip_scratch(lthip_ctx* ctx, int slot, size_t bytes, void** out)
{
    if (bytes == 0)
        bytes = 256;
    if (ctx->scratch_cap[slot] < bytes)
    {
        if (ctx->scratch[slot])
        {
            // the old buffer may still be in use by work queued on the stream
            LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
            LTHIP_CHECK(ctx, hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_cap[slot] = 0;
        }
        size_t cap = bytes + bytes / 8 + 4096; // head-room so slowly growing batches do not realloc
        LTHIP_CHECK(ctx, lthip_hip_malloc(&ctx->scratch[slot], cap));
        ctx->scratch_cap[slot] = cap;
    }
    *out = ctx->scratch[slot];
    return 0;
}

volatile uint32_t g_lthip_env_gen = 1;
extern "C" void lthip_debug_reload_env(void) { g_lthip_env_gen = g_lthip_env_gen + 1u; }

// ---- allocation-failure injection (ablation build only; the product build answers ENOTSUP and has no counter in its allocation path) ----
#ifdef LTHIP_ABLATIONS
namespace
{
std::atomic<int64_t> g_alloc_calls{0};            // allocations attempted by this process (both kinds)
std::atomic<int64_t> g_fail_first{INT64_MAX};     // 1-based number of the first allocation that fails
std::atomic<int64_t> g_fail_last{INT64_MAX};      // ... and of the last one (inclusive)
std::atomic<int64_t> g_alloc_failed{0};           // how many were made to fail
bool inject_failure()
{
    const int64_t n = g_alloc_calls.fetch_add(1, std::memory_order_relaxed) + 1;
    if (n < g_fail_first.load(std::memory_order_relaxed) || n > g_fail_last.load(std::memory_order_relaxed))
        return false;
    g_alloc_failed.fetch_add(1, std::memory_order_relaxed);
    return true;
}
} // namespace
hipError_t lthip_hip_malloc(void** p, size_t bytes)
{
    if (inject_failure())
    {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(p, bytes);
}
hipError_t lthip_hip_host_malloc(void** p, size_t bytes, unsigned flags)
{
    if (inject_failure())
    {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipHostMalloc(p, bytes, flags);
}
extern "C" int lthip_debug_fail_alloc(int64_t after, int64_t count)
{
    if (after < 0 || count <= 0)
    {
        g_fail_first = INT64_MAX; // off
        g_fail_last = INT64_MAX;
        return 0;
    }
    const int64_t now = g_alloc_calls.load();
    g_fail_last = INT64_MAX;
    g_fail_first = now + after + 1;
    g_fail_last = count > INT64_MAX - (now + after) ? INT64_MAX : now + after + count;
    return 0;
}
extern "C" int64_t lthip_debug_alloc_calls(int64_t* out_failed)
{
    if (out_failed)
        *out_failed = g_alloc_failed.load();
    return g_alloc_calls.load();
}
#else
extern "C" int lthip_debug_fail_alloc(int64_t, int64_t) { return ENOTSUP; }
extern "C" int64_t lthip_debug_alloc_calls(int64_t* out_failed)
{
    if (out_failed)
        *out_failed = 0;
    return -1;
}
#endif

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
extern "C" int lthip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

extern "C" int lthip_ctx_create(int device, void* hip_stream, lthip_ctx** out_ctx)
{
    if (!out_ctx)
        return EINVAL;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return ENODEV; // fail loudly: there is no CPU fallback in this library
    if (device < 0 || device >= n)
        return EINVAL;
    if (hipSetDevice(device) != hipSuccess)
        return EIO;
    lthip_ctx* ctx = new (std::nothrow) lthip_ctx();
    if (!ctx)
        return ENOMEM;
    ctx->device = device;
    ctx->slice_ctx = nullptr;
    ctx->err[0] = 0;
    ctx->timing = false;
    memset(ctx->scratch, 0, sizeof ctx->scratch);
    memset(ctx->stage, 0, sizeof ctx->stage);
    memset(ctx->stage_small, 0, sizeof ctx->stage_small);
    ctx->stage_next = ctx->stage_small_next = 0;
    ctx->k1_lds_enabled = false;
    ctx->z_last_retry = nullptr;
    ctx->z_last_payloads = 0;
    ctx->z_last_foreign_blocks = 0;
    memset(ctx->k5_lds_enabled, 0, sizeof ctx->k5_lds_enabled);
    memset(ctx->scratch_cap, 0, sizeof ctx->scratch_cap);
    memset(ctx->total_ms, 0, sizeof ctx->total_ms);
    memset(ctx->launches, 0, sizeof ctx->launches);
    if (hip_stream != LTHIP_STREAM_PRIVATE)
    {
        ctx->stream = (hipStream_t)hip_stream; // may be the null stream
        ctx->own_stream = false;
    }
    else
    {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
        {
            delete ctx;
            return EIO;
        }
        ctx->own_stream = true;
    }
    *out_ctx = ctx;
    return 0;
}

hipError_t lthip_stream_wait(lthip_ctx* ctx) { return hipStreamSynchronize(ctx->stream); }

// How host threads wait for this device: hipDeviceScheduleBlockingSync (a waiting thread sleeps until the interrupt) instead of the
// runtime's default, which polls the completion signal.  The plugin layer's callers are the embedder's job-system workers, 32-256
// threads each blocked in a Longtail_*API call: polling was a third of the drop-in path's CPU time, and in a container with a CPU
// quota (the measured boxes grant 16 CPUs of 256: cgroup cpu.max) CPU time IS that path's throughput -- CreateVersionIndex through the
// plugins 30 -> 40 GB/s at 32 workers, 16-20 -> 39 at 64 (tools/dropin_scaling.py, profiles/r06_dropin_scaling.txt).
// It is the DEVICE's policy, process-wide, and it has to be set BEFORE the process uses the device: switched in a process that had
// already run kernels and sessions, a later event wait of the ingest session never returned (bench.py, round 6: the completion signals
// made under the polling policy do not interrupt).  So the library never sets it on its own: Longtail_Hip_SetBlockingWaits(1) is the
// embedder's FIRST call (include/longtail_hip.h).  Measured and not used instead: an event made with hipEventBlockingSync in place of
// hipStreamSynchronize (polls like the default), hipStreamQuery + nanosleep (user time 8.5 -> 2.5 s but the timer slack's 50 us per wait
// cost more than the CPU time bought: 12.4 GB/s of UpSync against 13.9 with the default and 15.5-16.5 with the policy).
extern "C" int lthip_set_blocking_waits(int device, int on)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
        return ENODEV;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess)
        return EIO;
    const hipError_t e = hipSetDeviceFlags(on ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);
    (void)hipSetDevice(prev);
    return e == hipSuccess ? 0 : EIO;
}

extern "C" void lthip_ctx_destroy(lthip_ctx* ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->slice_ctx)
        lthip_ctx_destroy(ctx->slice_ctx);
    for (auto& r : ctx->pending)
    {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : ctx->free_events)
        (void)hipEventDestroy(e);
    for (auto e : ctx->sync_events)
        (void)hipEventDestroy(e);
    if (ctx->stream2)
    {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    for (int i = 0; i < S_COUNT; ++i)
        if (ctx->scratch[i])
            (void)hipFree(ctx->scratch[i]);
    for (auto& st : ctx->stage)
    {
        if (st.done)
            (void)hipEventDestroy(st.done);
        if (st.p)
            (void)hipHostFree(st.p);
    }
    for (auto& st : ctx->stage_small)
    {
        if (st.done)
            (void)hipEventDestroy(st.done);
        if (st.p)
            (void)hipHostFree(st.p);
    }
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int lthip_ctx_sync(lthip_ctx* ctx)
{
    if (!ctx)
        return EINVAL;
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    return 0;
}

extern "C" const char* lthip_ctx_error(const lthip_ctx* ctx) { return ctx ? ctx->err : "no context"; }

// ---------------------------------------------------------------------------------------------------
// memory helpers for plain-C callers
// ---------------------------------------------------------------------------------------------------
extern "C" int lthip_malloc_device(lthip_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out)
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, lthip_hip_malloc(out, bytes ? bytes : 16));
    return 0;
}

extern "C" void lthip_free_device(lthip_ctx* ctx, void* p)
{
    if (!p)
        return;
    if (ctx)
    {
        (void)hipSetDevice(ctx->device);
        (void)lthip_stream_wait(ctx);
    }
    (void)hipFree(p);
}

extern "C" int lthip_malloc_pinned(lthip_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out)
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, lthip_hip_host_malloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}

extern "C" void lthip_free_pinned(lthip_ctx* ctx, void* p)
{
    if (!p)
        return;
    if (ctx)
    {
        (void)hipSetDevice(ctx->device);
        (void)lthip_stream_wait(ctx);
    }
    (void)hipHostFree(p);
}

extern "C" int lthip_copy_h2d(lthip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !h_src)))
        return EINVAL;
    if (bytes == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

extern "C" int lthip_copy_d2h(lthip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes)
{
    if (!ctx || (bytes && (!h_dst || !d_src)))
        return EINVAL;
    if (bytes == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------------------------------
static hipEvent_t take_event(lthip_ctx* ctx)
{
    if (!ctx->free_events.empty())
    {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

LaunchTimer::LaunchTimer(lthip_ctx* c, int kid, hipStream_t s) : ctx(c), on(c->timing), stream(s ? s : c->stream)
{
    if (on)
    {
        rec.kid = kid;
        rec.a = take_event(ctx);
        rec.b = take_event(ctx);
        (void)hipEventRecord(rec.a, stream);
    }
}

int lthip_second_stream(lthip_ctx* ctx, hipStream_t* out)
{
    if (!ctx->stream2)
        LTHIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    *out = ctx->stream2;
    return 0;
}

int lthip_stage_upload(lthip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0)
        return 0;
    const bool small = bytes <= LTHIP_STAGE_SMALL;
    LTHIP_ABLATION_ENV(env_slots, "LTHIP_STAGE_SLOTS"); // (experiment: 8 = the ring of rounds 1-2)
    const size_t nsmall = env_slots.get() > 0 && env_slots.get() < 64 ? (size_t)env_slots.get() : sizeof(ctx->stage_small) / sizeof(ctx->stage_small[0]);
    lthip_ctx::Stage& st = small ? ctx->stage_small[ctx->stage_small_next++ % nsmall]
                                 : ctx->stage[ctx->stage_next++ % (sizeof(ctx->stage) / sizeof(ctx->stage[0]))];
    if (st.used)
        LTHIP_CHECK(ctx, hipEventSynchronize(st.done)); // a ring ago: long finished in steady state
    if (st.cap < bytes)
    {
        if (st.p)
            LTHIP_CHECK(ctx, hipHostFree(st.p));
        st.p = nullptr;
        st.cap = 0;
        const size_t want = small ? LTHIP_STAGE_SMALL : bytes + bytes / 2 + 4096;
        LTHIP_CHECK(ctx, lthip_hip_host_malloc(&st.p, want, hipHostMallocDefault));
        st.cap = want;
    }
    if (!st.done)
        LTHIP_CHECK(ctx, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    memcpy(st.p, h_src, bytes);
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_dst, st.p, bytes, hipMemcpyHostToDevice, stream));
    LTHIP_CHECK(ctx, hipEventRecord(st.done, stream));
    st.used = true;
    return 0;
}

hipEvent_t lthip_sync_event(lthip_ctx* ctx)
{
    // 16 events in rotation: an event is re-recorded only long after the wait that used it has been queued behind
    // newer work on both streams
    if (ctx->sync_events.size() < 16)
    {
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        ctx->sync_events.push_back(e);
        return e;
    }
    return ctx->sync_events[ctx->sync_next++ % ctx->sync_events.size()];
}

LaunchTimer::~LaunchTimer()
{
    if (on)
    {
        (void)hipEventRecord(rec.b, stream);
        ctx->pending.push_back(rec);
    }
}

static int timing_collect(lthip_ctx* ctx)
{
    if (ctx->pending.empty())
        return 0;
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    for (auto& r : ctx->pending)
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess)
        {
            ctx->total_ms[r.kid] += (double)ms;
            ctx->launches[r.kid] += 1;
        }
        ctx->free_events.push_back(r.a);
        ctx->free_events.push_back(r.b);
    }
    ctx->pending.clear();
    return 0;
}

extern "C" int lthip_timing_enable(lthip_ctx* ctx, int on)
{
    if (!ctx)
        return EINVAL;
    int err = timing_collect(ctx);
    ctx->timing = on != 0;
    if (ctx->slice_ctx)
        (void)lthip_timing_enable(ctx->slice_ctx, on);
    return err;
}

extern "C" int lthip_timing_reset(lthip_ctx* ctx)
{
    if (!ctx)
        return EINVAL;
    int err = timing_collect(ctx);
    memset(ctx->total_ms, 0, sizeof ctx->total_ms);
    memset(ctx->launches, 0, sizeof ctx->launches);
    if (ctx->slice_ctx)
        (void)lthip_timing_reset(ctx->slice_ctx);
    return err;
}

extern "C" int lthip_timing_get(lthip_ctx* ctx, int kernel_id, double* out_total_ms, uint64_t* out_launches)
{
    if (!ctx || kernel_id < 0 || kernel_id >= LTHIP_K_COUNT)
        return EINVAL;
    int err = timing_collect(ctx);
    double ms = ctx->total_ms[kernel_id];
    uint64_t n = ctx->launches[kernel_id];
    if (ctx->slice_ctx) // (the second slice of lthip_chunk_hash runs on a context of its own: its launches count here)
    {
        double ms2 = 0;
        uint64_t n2 = 0;
        if (lthip_timing_get(ctx->slice_ctx, kernel_id, &ms2, &n2) == 0)
        {
            ms += ms2;
            n += n2;
        }
    }
    if (out_total_ms)
        *out_total_ms = ms;
    if (out_launches)
        *out_launches = n;
    return err;
}

// ---------------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------------
// hpcdcchunker.c:126-129 -- the one floating-point step of the path, evaluated on the host in double
// exactly like the reference.
static uint32_t discriminator_from_avg(uint32_t avg)
{
    double a = (double)avg;
    return (uint32_t)(a / (-1.42888852e-7 * a + 1.33237515));
}

// Division-free form of the reference's cut test `hash % d == d-1` (hpcdcchunker.c:298).
//   h % d == d-1  <=>  x = h+1 (as an integer in [1, 2^32]) is a multiple of d.
// Write d = dodd * 2^k2, inv = dodd^-1 (mod 2^32).  Multiplication by inv permutes Z/2^32 and maps the
// multiples of dodd onto [0, floor((2^32-1)/dodd)], so for x in [1, 2^32-1]:
//   d | x  <=>  M = x*inv (mod 2^32) has k2 trailing zero bits and 1 <= M >> k2 <= q,  q = floor((2^32-1)/d)
//          <=>  ror32(M - 2^k2, k2) <= q-1          (a wrong low bit, or M = 0, lands in the top bits)
// and M - 2^k2 = h*inv + (inv - 2^k2), one multiply-add.  x = 2^32 (h = 0xFFFFFFFF) gives M = 0, which is
// rejected -- correct unless d is a power of two, which gets its own mask test.
static DivTest make_div_test(uint32_t d)
{
    DivTest t;
    memset(&t, 0, sizeof t);
    t.d = d;
    if (d == 0)
        return t; // rejected by plan_create
    uint32_t k2 = 0, dodd = d;
    while ((dodd & 1u) == 0)
    {
        dodd >>= 1;
        ++k2;
    }
    uint32_t inv = dodd; // Newton iteration, 5 steps give 32 bits
    for (int i = 0; i < 5; ++i)
        inv *= 2u - dodd * inv;
    t.inv = inv;
    t.k2 = k2;
    t.addc = inv - (1u << k2);
    t.qlim = (uint32_t)(0xFFFFFFFFull / d) - 1u;
    t.pow2 = dodd == 1u;
    return t;
}

// Greedy block packing of Longtail_CreateStoreIndex (src/longtail.c:6801-6860): host control logic, serial in the
// reference as well.  block_starts receives block_count+1 chunk indices.
extern "C" int lthip_pack_blocks(uint64_t chunk_count, const uint32_t* chunk_lens, uint32_t max_block_size,
                                 uint32_t max_chunks_per_block, uint64_t* block_starts, uint64_t capacity, uint64_t* out_block_count)
{
    if (!out_block_count || (chunk_count && (!chunk_lens || !block_starts)) || max_block_size == 0 || max_chunks_per_block == 0)
        return EINVAL;
    const uint64_t limit = (uint64_t)max_block_size + max_block_size / 10; // "overshoot by 10% is ok"
    uint64_t i = 0, nb = 0;
    while (i < chunk_count)
    {
        if (nb + 1 >= capacity)
            return ENOMEM;
        block_starts[nb++] = i;
        uint64_t size = chunk_lens[i];
        uint32_t n = 1;
        while (i + 1 < chunk_count && n < max_chunks_per_block && size + chunk_lens[i + 1] <= limit)
        {
            size += chunk_lens[i + 1];
            ++n;
            ++i;
        }
        ++i;
    }
    if (capacity == 0)
        return ENOMEM;
    block_starts[nb] = chunk_count;
    *out_block_count = nb;
    return 0;
}

// The same rule, resumable: packs blocks from chunk `first_chunk` on until the batch holds max_batch_bytes of chunk data
// or its codec output bounds (size + size / bound_div + bound_add, rounded up to 64) fill arena_bytes -- always at least
// one block -- so that a caller can hand batch k to the device and pack batch k+1 while it is being compressed.
extern "C" int lthip_pack_blocks_batch(uint64_t chunk_count, const uint32_t* chunk_lens, uint64_t first_chunk,
                                       uint32_t max_block_size, uint32_t max_chunks_per_block, uint64_t max_batch_bytes,
                                       uint64_t arena_bytes, uint32_t bound_div, uint32_t bound_add, uint64_t* block_starts,
                                       uint64_t* block_sizes, uint64_t capacity, uint64_t* out_block_count,
                                       uint64_t* out_next_chunk)
{
    if (!out_block_count || !out_next_chunk || !block_starts || !block_sizes || (chunk_count && !chunk_lens) ||
        max_block_size == 0 || max_chunks_per_block == 0 || bound_div == 0 || capacity < 2 || first_chunk > chunk_count)
        return EINVAL;
    const uint64_t limit = (uint64_t)max_block_size + max_block_size / 10;
    uint64_t i = first_chunk, nb = 0, bytes = 0, arena = 0;
    while (i < chunk_count && nb + 1 < capacity)
    {
        const uint64_t start = i;
        uint64_t size = chunk_lens[i];
        uint32_t n = 1;
        while (i + 1 < chunk_count && n < max_chunks_per_block && size + chunk_lens[i + 1] <= limit)
        {
            size += chunk_lens[i + 1];
            ++n;
            ++i;
        }
        ++i;
        const uint64_t bound = (size + size / bound_div + bound_add + 63u) & ~(uint64_t)63u;
        if (nb && (bytes + size > max_batch_bytes || arena + bound > arena_bytes))
        {
            i = start; // this block opens the next batch
            break;
        }
        block_starts[nb] = start;
        block_sizes[nb] = size;
        ++nb;
        bytes += size;
        arena += bound;
    }
    block_starts[nb] = i;
    *out_block_count = nb;
    *out_next_chunk = i;
    return 0;
}

extern "C" int lthip_divtest_eval(uint32_t discriminator, uint32_t hash)
{
    if (discriminator == 0)
        return 0;
    const DivTest t = make_div_test(discriminator);
    if (t.pow2)
        return (hash & (t.d - 1u)) == t.d - 1u;
    const uint32_t m = hash * t.inv + t.addc;
    const uint32_t r = t.k2 ? ((m >> t.k2) | (m << (32u - t.k2))) : m;
    return r <= t.qlim;
}

static int plan_create_impl(lthip_ctx* ctx, uint32_t part_count, const uint64_t* part_offsets, const uint64_t* part_sizes,
                            uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk, lthip_plan** out_plan);

// where a plan's parts are cut into S runs of about equal bytes: first[0] = 0 < first[1] < ... < first[S] = n; returns S (0 = do not slice)
static uint32_t plan_slice_points(uint32_t part_count, const uint64_t* part_sizes, uint32_t want, uint32_t* first)
{
    if (part_count < 2 || want < 2)
        return 0;
    uint64_t total = 0;
    for (uint32_t p = 0; p < part_count; ++p)
        total += part_sizes[p];
    if (total < LTHIP_SLICE_MIN_BYTES)
        return 0;
    uint32_t S = want > part_count ? part_count : want;
    while (S > 2 && total / S < LTHIP_SLICE_MIN_BYTES / 2)
        --S;
    uint64_t acc = 0;
    uint32_t p = 0;
    first[0] = 0;
    for (uint32_t k = 1; k < S; ++k)
    {
        const uint64_t goal = total / S * k;
        while (p < part_count - (S - k) && acc + part_sizes[p] / 2 < goal)
            acc += part_sizes[p++];
        if (p <= first[k - 1])
        {
            acc += part_sizes[p];
            ++p;
        }
        first[k] = p;
    }
    first[S] = part_count;
    return S;
}

extern "C" int lthip_plan_create(lthip_ctx* ctx, uint32_t part_count, const uint64_t* part_offsets,
                                 const uint64_t* part_sizes, uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk,
                                 lthip_plan** out_plan)
{
    int err = plan_create_impl(ctx, part_count, part_offsets, part_sizes, min_chunk, avg_chunk, max_chunk, out_plan);
    if (err)
        return err;
    // the slices (lthip_chunk_hash): plans of their own over the same bytes.  Optional: a slice that cannot be made leaves the plan
    // unsliced, it is never an error of the call.
    lthip_plan* plan = *out_plan;
    int want = LTHIP_SLICES;
    LTHIP_ABLATION_ENV(env_slices, "LTHIP_SLICES"); // (ablation build: 1 = the single pass, for profiles of K1 / K3 alone; 2..8)
    if (env_slices.get() > 0)
        want = env_slices.get() > 8 ? 8 : env_slices.get();
    uint32_t first[9];
    const uint32_t S = plan_slice_points(part_count, part_sizes, (uint32_t)want, first);
    if (S)
    {
        bool ok = true;
        for (uint32_t k = 0; k < S && ok; ++k)
            ok = plan_create_impl(ctx, first[k + 1] - first[k], part_offsets + first[k], part_sizes + first[k], min_chunk, avg_chunk, max_chunk, &plan->slice[k]) == 0;
        if (ok)
        {
            memcpy(plan->slice_first, first, sizeof(uint32_t) * (S + 1));
            plan->nslices = S;
            plan->sliced = true;
        }
        else
        {
            for (uint32_t k = 0; k < S; ++k)
            {
                lthip_plan_destroy(ctx, plan->slice[k]);
                plan->slice[k] = nullptr;
            }
            ctx->err[0] = 0;
        }
    }
    return 0;
}

static int plan_create_impl(lthip_ctx* ctx, uint32_t part_count, const uint64_t* part_offsets, const uint64_t* part_sizes,
                            uint32_t min_chunk, uint32_t avg_chunk, uint32_t max_chunk, lthip_plan** out_plan)
{
    if (!ctx || !out_plan || (part_count && (!part_offsets || !part_sizes)))
        return EINVAL;
    *out_plan = nullptr;
    // same parameter contract as Longtail_HPCDCCreateChunker (hpcdcchunker.c:143-146)
    if (min_chunk < 48 || min_chunk > avg_chunk || avg_chunk > max_chunk)
        return lthip_fail(ctx, EINVAL, "lthip_plan_create", "need 48 <= min <= avg <= max");
    uint32_t d = discriminator_from_avg(avg_chunk);
    if (d == 0)
        return lthip_fail(ctx, EINVAL, "lthip_plan_create", "discriminator is zero");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));

    lthip_plan* plan = new (std::nothrow) lthip_plan();
    if (!plan)
        return ENOMEM;
    memset(plan, 0, sizeof *plan);
    plan->nparts = part_count;
    plan->device = ctx->device;
    plan->min_chunk = min_chunk;
    plan->avg_chunk = avg_chunk;
    plan->max_chunk = max_chunk;
    plan->div = make_div_test(d);

    std::vector<PartDev> parts(part_count ? part_count : 1);
    uint64_t tiles = 0, bm0 = 0, bm1 = 0, region = 0, bytes = 0, leaves = 0;
    for (uint32_t p = 0; p < part_count; ++p)
    {
        uint64_t sz = part_sizes[p];
        if ((part_offsets[p] & 15u) != 0 || sz > 0xFFFFFFFFull)
        {
            delete plan;
            return lthip_fail(ctx, EINVAL, "lthip_plan_create", "part offsets must be 16-byte aligned, sizes < 4 GiB");
        }
        PartDev& pd = parts[p];
        pd.off = part_offsets[p];
        pd.size = sz;
        pd.bm0_base = bm0;
        pd.bm1_base = bm1;
        pd.region_base = region;
        pd.tile_base = (uint32_t)tiles;
        uint64_t cap = sz ? sz / min_chunk + 1 : 0; // every chunk but the last is >= min bytes
        pd.region_cap = (uint32_t)cap;
        uint64_t t = div_up_u64(sz, 16384);
        tiles += t;
        bm0 += t * 256; // one 64-bit word per 64-byte run, whole tiles
        bm1 += t * 4;   // one 64-bit word per 4 KiB
        region += cap;
        bytes += sz;
        leaves += div_up_u64(sz, 1024) + cap;
    }
    if (tiles > 0xFFFFFFF0ull)
    {
        delete plan;
        return lthip_fail(ctx, EINVAL, "lthip_plan_create", "batch too large");
    }
    plan->ntiles = tiles;
    plan->bm0_words = bm0;
    plan->bm1_words = bm1;
    plan->chunk_cap = region;
    plan->total_bytes = bytes;
    plan->leaf_cap = leaves;
    plan->capacity_bytes = bytes;
    plan->cap_parts = part_count ? part_count : 1;
    plan->cap_tiles = tiles ? tiles : 1;

    hipError_t e = lthip_hip_malloc((void**)&plan->d_parts, sizeof(PartDev) * (part_count ? part_count : 1));
    if (e == hipSuccess)
        e = lthip_hip_malloc((void**)&plan->d_tile_part, sizeof(uint32_t) * (tiles ? tiles : 1));
    if (e == hipSuccess && part_count)
        e = hipMemcpyAsync(plan->d_parts, parts.data(), sizeof(PartDev) * part_count, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = lthip_stream_wait(ctx); // `parts` is a local
    if (e != hipSuccess)
    {
        lthip_plan_destroy(ctx, plan);
        return lthip_fail(ctx, e == hipErrorOutOfMemory ? ENOMEM : EIO, "lthip_plan_create", hipGetErrorString(e));
    }
    int err = lthip_launch_tile_table(ctx, plan);
    if (err)
    {
        lthip_plan_destroy(ctx, plan);
        return err;
    }
    *out_plan = plan;
    return 0;
}

// A single-part plan sized once for a CAPACITY and re-aimed at the bytes actually present: what the plugin chunker needs (one
// window of varying fill per refill) without two hipMalloc + a kernel + a synchronisation per file.  The tile -> part table of a
// single part is all zeros whatever the size, so only the part descriptor and the host-side extents change.
extern "C" int lthip_plan_resize_single(lthip_ctx* ctx, lthip_plan* plan, uint64_t size)
{
    if (!ctx || !plan || plan->nparts != 1)
        return EINVAL;
    if (size > plan->capacity_bytes)
        return lthip_fail(ctx, EINVAL, "lthip_plan_resize_single", "size above the capacity the plan was created with");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    PartDev pd;
    memset(&pd, 0, sizeof pd);
    pd.off = 0;
    pd.size = size;
    const uint64_t cap = size ? size / plan->min_chunk + 1 : 0;
    pd.region_cap = (uint32_t)cap;
    const uint64_t t = div_up_u64(size, 16384);
    plan->ntiles = t;
    plan->bm0_words = t * 256;
    plan->bm1_words = t * 4;
    plan->chunk_cap = cap;
    plan->total_bytes = size;
    plan->leaf_cap = div_up_u64(size, 1024) + cap;
    return lthip_stage_upload(ctx, plan->d_parts, &pd, sizeof pd, ctx->stream);
}

// A plan re-aimed at ANOTHER set of parts (any count up to the one it was created with, any layout whose tiles fit the tile
// table it was created with): no allocation and no synchronisation -- the part table goes through the pinned staging ring, the
// tile -> part table is rebuilt by its kernel on the stream.  What the plugin layer's batcher needs: one plan, a different set of
// windows in every submission (plugin_batch.c).
static int plan_reaim_impl(lthip_ctx* ctx, lthip_plan* plan, uint32_t part_count, const uint64_t* part_offsets, const uint64_t* part_sizes);

extern "C" int lthip_plan_reaim(lthip_ctx* ctx, lthip_plan* plan, uint32_t part_count, const uint64_t* part_offsets,
                                const uint64_t* part_sizes)
{
    int err = plan_reaim_impl(ctx, plan, part_count, part_offsets, part_sizes);
    if (err || !plan->nslices)
        return err;
    // the slices follow when the new parts fit them (they do when the layout is the one the plan was created with: bench.py's
    // steps); otherwise this aim runs as one
    plan->sliced = false;
    uint32_t first[9];
    if (plan_slice_points(part_count, part_sizes, plan->nslices, first) != plan->nslices)
        return 0;
    for (uint32_t k = 0; k < plan->nslices; ++k)
    {
        uint64_t tiles = 0;
        for (uint32_t p = first[k]; p < first[k + 1]; ++p)
            tiles += div_up_u64(part_sizes[p], 16384);
        if (first[k + 1] - first[k] > plan->slice[k]->cap_parts || tiles > plan->slice[k]->cap_tiles)
            return 0;
    }
    for (uint32_t k = 0; k < plan->nslices; ++k)
        if (plan_reaim_impl(ctx, plan->slice[k], first[k + 1] - first[k], part_offsets + first[k], part_sizes + first[k]))
        {
            ctx->err[0] = 0;
            return 0;
        }
    memcpy(plan->slice_first, first, sizeof(uint32_t) * (plan->nslices + 1));
    plan->sliced = true;
    return 0;
}

static int plan_reaim_impl(lthip_ctx* ctx, lthip_plan* plan, uint32_t part_count, const uint64_t* part_offsets, const uint64_t* part_sizes)
{
    if (!ctx || !plan || (part_count && (!part_offsets || !part_sizes)))
        return EINVAL;
    if (part_count == 0 || part_count > plan->cap_parts)
        return lthip_fail(ctx, EINVAL, "lthip_plan_reaim", "part count outside 1 .. the count the plan was created with");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<PartDev> parts(part_count);
    uint64_t tiles = 0, bm0 = 0, bm1 = 0, region = 0, bytes = 0, leaves = 0;
    for (uint32_t p = 0; p < part_count; ++p)
    {
        const uint64_t sz = part_sizes[p];
        if ((part_offsets[p] & 15u) != 0 || sz > 0xFFFFFFFFull)
            return lthip_fail(ctx, EINVAL, "lthip_plan_reaim", "part offsets must be 16-byte aligned, sizes < 4 GiB");
        PartDev& pd = parts[p];
        memset(&pd, 0, sizeof pd);
        pd.off = part_offsets[p];
        pd.size = sz;
        pd.bm0_base = bm0;
        pd.bm1_base = bm1;
        pd.region_base = region;
        pd.tile_base = (uint32_t)tiles;
        const uint64_t cap = sz ? sz / plan->min_chunk + 1 : 0;
        pd.region_cap = (uint32_t)cap;
        const uint64_t t = div_up_u64(sz, 16384);
        tiles += t;
        bm0 += t * 256;
        bm1 += t * 4;
        region += cap;
        bytes += sz;
        leaves += div_up_u64(sz, 1024) + cap;
    }
    if (tiles > plan->cap_tiles)
        return lthip_fail(ctx, EINVAL, "lthip_plan_reaim", "more 16 KiB tiles than the plan was created with");
    plan->nparts = part_count;
    plan->ntiles = tiles;
    plan->bm0_words = bm0;
    plan->bm1_words = bm1;
    plan->chunk_cap = region;
    plan->total_bytes = bytes;
    plan->leaf_cap = leaves;
    int err = lthip_stage_upload(ctx, plan->d_parts, parts.data(), sizeof(PartDev) * part_count, ctx->stream);
    if (err)
        return err;
    return lthip_launch_tile_table(ctx, plan);
}

extern "C" void lthip_plan_destroy(lthip_ctx* ctx, lthip_plan* plan)
{
    if (!plan)
        return;
    // A plan is plain device memory: it may outlive the context (and host thread) that created it.  With a
    // context only its stream is drained; without one the whole device is.
    (void)hipSetDevice(plan->device);
    if (ctx)
        (void)lthip_stream_wait(ctx);
    else
        (void)hipDeviceSynchronize();
    if (plan->d_parts)
        (void)hipFree(plan->d_parts);
    if (plan->d_tile_part)
        (void)hipFree(plan->d_tile_part);
    for (uint32_t k = 0; k < 8; ++k)
        lthip_plan_destroy(ctx, plan->slice[k]);
    delete plan;
}

extern "C" uint64_t lthip_plan_chunk_capacity(const lthip_plan* plan) { return plan ? plan->chunk_cap : 0; }
extern "C" uint32_t lthip_plan_slices(const lthip_plan* plan) { return plan && plan->sliced && plan->nslices ? plan->nslices : 1u; }

// ---------------------------------------------------------------------------------------------------
// phase 1
// ---------------------------------------------------------------------------------------------------
static int chunk_scan_one(lthip_ctx* ctx, const lthip_plan* plan, const void* d_data, uint64_t* d_chunk_offsets, uint32_t* d_chunk_lens,
                          uint32_t* d_part_first);

// a slice's lists behind the slices before it: base = part_first[0] (= the chunk count of everything before the slice, final by now),
// its lists go to [base, base + total), its part table is shifted by base
__global__ void k_slice_join(const uint64_t* __restrict__ b_off, const uint32_t* __restrict__ b_len, const uint64_t* __restrict__ b_hash,
                             const uint32_t* __restrict__ b_first, uint32_t nparts_b, uint32_t* __restrict__ part_first /* at the slice's first part */,
                             uint64_t* __restrict__ offs, uint32_t* __restrict__ lens, uint64_t* __restrict__ hashes)
{
    const uint32_t base = part_first[0], total_b = b_first[nparts_b];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total_b)
    {
        offs[base + i] = b_off[i];
        lens[base + i] = b_len[i];
        hashes[base + i] = b_hash[i];
    }
    if (i >= 1 && i <= nparts_b) // (entry 0 is the base itself: every thread reads it)
        part_first[i] = b_first[i] + base;
}

extern "C" int lthip_chunk_hash(lthip_ctx* ctx, const lthip_plan* plan, const void* d_data, uint64_t* d_chunk_offsets,
                                uint32_t* d_chunk_lens, uint64_t* d_chunk_hashes, uint32_t* d_part_first,
                                uint64_t* out_total)
{
    if (!ctx || !plan || !d_chunk_offsets || !d_chunk_lens || !d_part_first || (plan->total_bytes && !d_data))
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    int err = 0;
    bool done = false;
    // ---- slices on two streams (plans of >= 1 GiB in >= 2 parts, with hashes).  The candidate scan (K1: four waves per SIMD, its issue
    // slots 72 % used, LDS-heavy) and the leaf hashing (K3: VALU bound, no LDS) are both bound by instruction issue and leave each other
    // room: with the parts cut into S slices, the scans (+ cut selection + compaction) of the slices run one after the other on the
    // context's stream and the hashing of every slice behind its scan on a second stream -- scan i + 1 beside hashing i; the scan
    // takes its residency first (one persistent workgroup per CU), the hashing's workgroups fill what is left.  The lists are the
    // ones of the single pass, bit for bit: slice 0 writes the caller's arrays, the others go to scratch and are joined behind it in
    // order (their place depends on the counts before them).  tools/k1k3_overlap_probe.py, DESIGN.md §3.
    if (plan->sliced && plan->nslices >= 2 && d_chunk_hashes)
    {
        if (!ctx->slice_ctx && lthip_ctx_create(ctx->device, LTHIP_STREAM_PRIVATE, &ctx->slice_ctx) != 0)
            ctx->slice_ctx = nullptr;
        lthip_ctx* c2 = ctx->slice_ctx;
        const uint32_t S = plan->nslices;
        uint64_t cap_rest = 0, parts_rest = 0; // slices 1 .. S-1 in scratch, back to back
        for (uint32_t k = 1; k < S; ++k)
        {
            cap_rest += plan->slice[k]->chunk_cap + 1;
            parts_rest += (uint64_t)plan->slice[k]->nparts + 1;
        }
        void *x_off = nullptr, *x_len = nullptr, *x_hash = nullptr, *x_first = nullptr;
        if (c2 && !lthip_scratch(ctx, S_SLICE_OFFS, cap_rest * 8, &x_off) && !lthip_scratch(ctx, S_SLICE_LENS, cap_rest * 4, &x_len) &&
            !lthip_scratch(ctx, S_SLICE_HASH, cap_rest * 8, &x_hash) && !lthip_scratch(ctx, S_SLICE_FIRST, parts_rest * 4, &x_first))
        {
            c2->timing = ctx->timing;
            // all candidate scans back to back on the context's stream (scan k + 1 takes the CUs the moment scan k leaves them: its one
            // workgroup per CU needs 117 KiB of LDS and four waves per SIMD, and would wait for a whole grid of hashing workgroups to
            // drain if those got there first); cut selection, compaction and hashing of slice k on the second stream behind scan k
            void *bm0 = nullptr, *bm1 = nullptr;
            if ((err = lthip_scratch(ctx, S_BM0, plan->bm0_words * 8, &bm0)) || (err = lthip_scratch(ctx, S_BM1, plan->bm1_words * 8, &bm1)))
                return err;
            uint64_t co = 0, po = 0, b0 = 0, b1 = 0;
            uint64_t* offs_k[8];
            uint32_t *lens_k[8], *first_k[8];
            uint64_t* hash_k[8];
            for (uint32_t k = 0; k < S && !err; ++k)
            {
                const lthip_plan* pk = plan->slice[k];
                offs_k[k] = k ? (uint64_t*)x_off + co : d_chunk_offsets;
                lens_k[k] = k ? (uint32_t*)x_len + co : d_chunk_lens;
                hash_k[k] = k ? (uint64_t*)x_hash + co : d_chunk_hashes;
                first_k[k] = k ? (uint32_t*)x_first + po : d_part_first;
                if (k)
                {
                    co += pk->chunk_cap + 1;
                    po += (uint64_t)pk->nparts + 1;
                }
                uint64_t* bm0_k = (uint64_t*)bm0 + b0;
                uint64_t* bm1_k = (uint64_t*)bm1 + b1;
                b0 += pk->bm0_words;
                b1 += pk->bm1_words;
                if (pk->nparts && (err = lthip_launch_buzhash(ctx, pk, (const uint8_t*)d_data, bm0_k, bm1_k)))
                    break;
                hipEvent_t scanned = lthip_sync_event(ctx);
                LTHIP_CHECK(ctx, hipEventRecord(scanned, ctx->stream));
                LTHIP_CHECK(ctx, hipStreamWaitEvent(c2->stream, scanned, 0));
                void *region = nullptr, *pcount = nullptr;
                if ((err = lthip_scratch(c2, S_REGION, pk->chunk_cap * sizeof(uint2), &region)) ||
                    (err = lthip_scratch(c2, S_PART_COUNT, ((size_t)pk->nparts + 1) * 4, &pcount)))
                    break;
                if (pk->nparts)
                    err = lthip_launch_select(c2, pk, bm0_k, bm1_k, (uint2*)region, (uint32_t*)pcount);
                if (!err)
                    err = lthip_launch_compact(c2, pk, (const uint2*)region, (const uint32_t*)pcount, first_k[k], offs_k[k], lens_k[k]);
                if (!err)
                    err = lthip_launch_blake3(c2, (const uint8_t*)d_data, offs_k[k], lens_k[k], first_k[k] + pk->nparts, pk->chunk_cap, pk->leaf_cap,
                                              pk->max_chunk, hash_k[k]);
                if (err)
                    (void)lthip_fail(ctx, err, "lthip_chunk_hash (a slice on the second stream)", c2->err);
            }
            hipEvent_t hashed = lthip_sync_event(ctx); // (also after a failure: the first stream never runs ahead of the second)
            LTHIP_CHECK(ctx, hipEventRecord(hashed, c2->stream));
            LTHIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, hashed, 0));
            if (err)
                return err;
            LaunchTimer t(ctx, LTHIP_K_COMPACT);
            for (uint32_t k = 1; k < S; ++k) // in order: slice k's base is the total behind slice k - 1's join
            {
                const lthip_plan* pk = plan->slice[k];
                const uint64_t n = pk->chunk_cap > (uint64_t)pk->nparts + 1 ? pk->chunk_cap : (uint64_t)pk->nparts + 1;
                hipLaunchKernelGGL(k_slice_join, dim3((uint32_t)div_up_u64(n, 256)), dim3(256), 0, ctx->stream, (const uint64_t*)offs_k[k], (const uint32_t*)lens_k[k],
                                   (const uint64_t*)hash_k[k], (const uint32_t*)first_k[k], pk->nparts, d_part_first + plan->slice_first[k], d_chunk_offsets,
                                   d_chunk_lens, d_chunk_hashes);
            }
            LTHIP_LAUNCH_CHECK(ctx);
            done = true;
        }
        else
            ctx->err[0] = 0; // (no second context / scratch: the single pass)
    }
    if (!done)
    {
        err = chunk_scan_one(ctx, plan, d_data, d_chunk_offsets, d_chunk_lens, d_part_first);
        if (!err && d_chunk_hashes)
            err = lthip_launch_blake3(ctx, (const uint8_t*)d_data, d_chunk_offsets, d_chunk_lens, d_part_first + plan->nparts, plan->chunk_cap,
                                      plan->leaf_cap, plan->max_chunk, d_chunk_hashes);
    }
    if (err)
        return err;
    if (out_total)
    {
        uint32_t total = 0;
        LTHIP_CHECK(ctx, hipMemcpyAsync(&total, d_part_first + plan->nparts, 4, hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        *out_total = total;
    }
    return 0;
}

// candidate scan + cut selection + compaction of one plan on the context's stream
static int chunk_scan_one(lthip_ctx* ctx, const lthip_plan* plan, const void* d_data, uint64_t* d_chunk_offsets, uint32_t* d_chunk_lens,
                          uint32_t* d_part_first)
{
    void *bm0, *bm1, *region, *pcount;
    int err;
    if ((err = lthip_scratch(ctx, S_BM0, plan->bm0_words * 8, &bm0)))
        return err;
    if ((err = lthip_scratch(ctx, S_BM1, plan->bm1_words * 8, &bm1)))
        return err;
    if ((err = lthip_scratch(ctx, S_REGION, plan->chunk_cap * sizeof(uint2), &region)))
        return err;
    if ((err = lthip_scratch(ctx, S_PART_COUNT, ((size_t)plan->nparts + 1) * 4, &pcount)))
        return err;

    if (plan->nparts)
    {
        if ((err = lthip_launch_buzhash(ctx, plan, (const uint8_t*)d_data, (uint64_t*)bm0, (uint64_t*)bm1)))
            return err;
        if ((err = lthip_launch_select(ctx, plan, (const uint64_t*)bm0, (const uint64_t*)bm1, (uint2*)region,
                                       (uint32_t*)pcount)))
            return err;
    }
    if ((err = lthip_launch_compact(ctx, plan, (const uint2*)region, (const uint32_t*)pcount, d_part_first,
                                    d_chunk_offsets, d_chunk_lens)))
        return err;
    return 0;
}

extern "C" int lthip_chunk_from_buffer(lthip_ctx* ctx, const void* d_data, uint64_t size, uint32_t min_chunk,
                                       uint32_t avg_chunk, uint32_t max_chunk, uint64_t* out_len)
{
    if (!ctx || !d_data || !out_len || size == 0)
        return EINVAL;
    if (min_chunk < 48 || min_chunk > avg_chunk || avg_chunk > max_chunk)
        return lthip_fail(ctx, EINVAL, "lthip_chunk_from_buffer", "need 48 <= min <= avg <= max");
    if (size <= min_chunk)
    {
        *out_len = size; // hpcdcchunker.c:479-484
        return 0;
    }
    const uint32_t d = discriminator_from_avg(avg_chunk);
    if (d == 0)
        return lthip_fail(ctx, EINVAL, "lthip_chunk_from_buffer", "discriminator is zero");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    void* d_out;
    int err = lthip_scratch(ctx, S_MISC, 64, &d_out);
    if (err)
        return err;
    const uint32_t n = (uint32_t)(size > max_chunk ? max_chunk : size);
    if ((err = lthip_launch_from_buffer(ctx, (const uint8_t*)d_data, n, min_chunk, make_div_test(d), (uint64_t*)d_out)))
        return err;
    uint64_t len = 0;
    LTHIP_CHECK(ctx, hipMemcpyAsync(&len, d_out, 8, hipMemcpyDeviceToHost, ctx->stream));
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    *out_len = len;
    return 0;
}

// BLAKE3-64 of runs of 64-bit values: d_out[i] = blake3(the bytes of d_values[d_first[i] .. d_first[i + 1])).  With d_values = the
// chunk hashes of lthip_chunk_hash and d_first = its part table this is every part's CONTENT hash as ChunkAssets computes it for a
// one-part asset (src/longtail.c:2518-2537): the plugin layer's batcher keeps it so that the core's later HashBuffer over the same
// digests is answered from memory (plugin_batch.c).
__global__ void k_runs_to_ranges(const uint32_t* __restrict__ first, uint32_t n, uint64_t* __restrict__ offs, uint32_t* __restrict__ lens)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        offs[i] = 8ull * first[i];
        lens[i] = 8u * (first[i + 1] - first[i]);
    }
}

extern "C" int lthip_hash_runs_u64(lthip_ctx* ctx, const uint64_t* d_values, const uint32_t* d_first, uint32_t run_count,
                                   uint64_t* d_out)
{
    return lthip_hash_runs_u64_bounded(ctx, d_values, d_first, run_count, 0, 0, d_out);
}

extern "C" int lthip_hash_runs_u64_bounded(lthip_ctx* ctx, const uint64_t* d_values, const uint32_t* d_first, uint32_t run_count,
                                           uint64_t total_values_bound, uint64_t run_values_bound, uint64_t* d_out)
{
    if (!ctx || !d_values || !d_first || !d_out)
        return EINVAL;
    if (run_count == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    void* tab;
    int err = lthip_scratch(ctx, S_TABLES, (size_t)run_count * 16, &tab);
    if (err)
        return err;
    uint64_t* offs = (uint64_t*)tab;
    uint32_t* lens = (uint32_t*)(offs + run_count);
    hipLaunchKernelGGL(k_runs_to_ranges, dim3((run_count + 255u) / 256u), dim3(256), 0, ctx->stream, d_first, run_count, offs, lens);
    LTHIP_LAUNCH_CHECK(ctx);
    // with the caller's bounds the launch needs nothing back from the device (no read-back, no stream synchronisation): leaves <= one per
    // KiB of values + one per run
    const uint64_t leaf_bound = total_values_bound ? total_values_bound * 8u / 1024u + run_count : 0u;
    return lthip_launch_blake3(ctx, (const uint8_t*)d_values, offs, lens, nullptr, run_count, leaf_bound, run_values_bound * 8u, d_out);
}

// Streaming BLAKE3 (k_blake3.hip): a batch of LTHIP_B3_STREAM_BATCH bytes = 1024 full leaves, the `batch_index`-th of its stream, is
// reduced to its subtree's chaining value and pushed onto the stream's stack (d_stack: LTHIP_B3_STREAM_STACK_BYTES of device memory
// owned by the caller, no initialisation needed); lthip_b3_stream_final hashes the rest (tail_len <= one batch; 0 only for an empty
// stream) and folds the stack.  The caller passes how many batches came before: the stack depth and the merges follow from that
// number alone (one entry per set bit).  The low 32 bits of the BLAKE3 chunk counter are used: streams below 4 TiB.
extern "C" int lthip_b3_stream_batch(lthip_ctx* ctx, const void* d_data, uint64_t batch_index, void* d_stack)
{
    if (!ctx || !d_data || !d_stack || batch_index >= (1ull << 22))
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t depth_in = (uint32_t)__builtin_popcountll(batch_index);
    const uint32_t merges = (uint32_t)__builtin_ctzll(batch_index + 1ull);
    return lthip_launch_blake3_stream_batch(ctx, d_data, (uint32_t)(batch_index << 10), (uint32_t*)d_stack, depth_in, merges);
}

extern "C" int lthip_b3_stream_final(lthip_ctx* ctx, const void* d_tail, uint32_t tail_len, uint64_t batch_count, const void* d_stack,
                                     uint64_t* d_out)
{
    if (!ctx || !d_out || (tail_len && !d_tail) || tail_len > (1u << 20) || batch_count >= (1ull << 22) || (batch_count && (!tail_len || !d_stack)))
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    return lthip_launch_blake3_stream_final(ctx, d_tail, tail_len, (uint32_t)(batch_count << 10), (const uint32_t*)d_stack,
                                            (uint32_t)__builtin_popcountll(batch_count), d_out);
}

// One small input where it lies (see k_blake3_one): `in` and `out` must be readable / writable by the device -- pinned host memory
// (lthip_malloc_pinned) or device memory.  Asynchronous on the context's stream.
extern "C" int lthip_hash_one(lthip_ctx* ctx, const void* in, uint32_t len, uint64_t* out)
{
    if (!ctx || !out || (len && !in))
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    return lthip_launch_blake3_one(ctx, in, len, out);
}

extern "C" int lthip_hash_ranges(lthip_ctx* ctx, const void* d_data, uint64_t range_count, const uint64_t* d_offsets,
                                 const uint32_t* d_lens, uint32_t max_len, uint64_t* d_hashes)
{
    if (!ctx || (range_count && (!d_offsets || !d_lens || !d_hashes)))
        return EINVAL;
    if (range_count == 0)
        return 0;
    if (range_count > 0xFFFFFFF0ull)
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    // leaf bound unknown without reading the lengths: 0 => the launcher sizes the grid from the scanned total
    return lthip_launch_blake3(ctx, (const uint8_t*)d_data, d_offsets, d_lens, nullptr, range_count, 0, max_len, d_hashes);
}

// ... for a caller that has the lengths on the host: `leaf_total` = sum over the ranges of max(1, ceil(len / 1024)).  With ranges of
// at most 256 KiB nothing is read back and the stream is not waited for (lthip_hash_ranges stalls the caller until everything queued
// before it has run).
int lthip_hash_ranges_known(lthip_ctx* ctx, const void* d_data, uint64_t range_count, const uint64_t* d_offsets, const uint32_t* d_lens,
                            uint32_t max_len, uint64_t leaf_total, uint64_t* d_hashes)
{
    if (!ctx || (range_count && (!d_offsets || !d_lens || !d_hashes)) || range_count > 0xFFFFFFF0ull)
        return EINVAL;
    if (range_count == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    return lthip_launch_blake3(ctx, (const uint8_t*)d_data, d_offsets, d_lens, nullptr, range_count, leaf_total, max_len, d_hashes);
}
