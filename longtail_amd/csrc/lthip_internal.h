// lthip_internal.h -- shared declarations of liblongtail_hip.so (C++/HIP side, gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

#include "../../include/longtail_hip.h"

// ---------------------------------------------------------------------------------------------------
// device-side descriptors
// ---------------------------------------------------------------------------------------------------
struct PartDev
{
    uint64_t off;         // byte offset of the part in the data buffer (16-byte aligned)
    uint64_t size;        // bytes (< 2^32)
    uint64_t bm0_base;    // first 64-bit word of this part in the level-0 candidate bitmap (1 bit / byte)
    uint64_t bm1_base;    // first 64-bit word in the level-1 summary (1 bit / 64-byte run)
    uint64_t region_base; // first slot of this part in the bounded (rel_off,len) chunk region
    uint32_t tile_base;   // first buzhash tile of this part
    uint32_t region_cap;  // slots in the region
};

// exact "H % d == d-1" test without a division:  d = dodd << k2 ;  inv = dodd^-1 mod 2^32
//   t = ror32(h*inv + addc, k2) <= qlim            (lthip_plan.cpp explains the derivation)
struct DivTest
{
    uint32_t d;
    uint32_t inv;
    uint32_t addc;
    uint32_t k2;
    uint32_t qlim;
    uint32_t pow2; // d is a power of two: test (h & (d-1)) == d-1 instead
};

struct lthip_plan
{
    int device;
    uint32_t nparts;
    uint32_t min_chunk, avg_chunk, max_chunk;
    DivTest div;
    uint64_t total_bytes;
    uint64_t ntiles;
    uint64_t chunk_cap;
    uint64_t bm0_words;
    uint64_t bm1_words;
    uint64_t leaf_cap;
    uint64_t capacity_bytes; // bytes the plan was created for (lthip_plan_resize_single may aim it at fewer)
    uint32_t cap_parts;      // parts / tiles the device tables were allocated for (lthip_plan_reaim)
    uint64_t cap_tiles;
    PartDev* d_parts;
    uint32_t* d_tile_part;
    // A plan of many parts and >= LTHIP_SLICE_MIN_BYTES also exists as LTHIP_SLICES plans over consecutive runs of its parts of
    // about equal bytes (lthip_chunk_hash: the candidate scans of the slices one after the other on one stream, the leaf hashing of
    // every slice behind its scan on a second stream -- the scan of slice i + 1 runs beside the hashing of slice i)
    lthip_plan* slice[8];
    uint32_t slice_first[9]; // slice i = parts [slice_first[i], slice_first[i + 1])
    uint32_t nslices;        // 0: the plan is run as one
    bool sliced;             // the slices are aimed at the plan's current parts
};
constexpr uint64_t LTHIP_SLICE_MIN_BYTES = 1ull << 30;
#ifndef LTHIP_SLICES
#define LTHIP_SLICES 2 /* measured 2 .. 8 with the scans back to back (round 6): 2 is as good as any -- what the hashing gains is the issue slots the scans leave free while they run, whatever the granularity */
#endif

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
enum ScratchSlot
{
    S_BM0 = 0,
    S_BM1,
    S_REGION,      // uint2 {rel_off, len} per bounded chunk slot
    S_PART_COUNT,  // u32 per part
    S_LEAF_COUNT,  // u32 per dense chunk (+1)
    S_LEAF_PREFIX, // u32 per dense chunk (+1)
    S_SCAN_TMP,
    S_CV,          // 8 x u32 per leaf
    S_LZ4_SEGS,    // segment descriptors
    S_LZ4_META,    // per-segment results
    S_LZ4_STREAM,  // per-segment sequence streams
    S_LZ4_BLOCKS,  // block tables
    S_TABLES,      // small H2D tables
    S_TABLES2,
    S_MISC,
    S_Z_LITS,  // zstd: literals per 4 KiB unit
    S_Z_RECS,  // zstd: sequence records per unit
    S_Z_ENC,   // zstd: encoded 128 KiB pieces
    S_Z_WORK,  // zstd: per-encoder-wave work area
    S_Z_SUB,   // zstd: content size of every unit's sub-block (u16 per 4 KiB unit) / decoder: sequence records
    S_LZ4_WORKLIST, // groups the stitch copy has to visit
    S_LZ4_LANE_RECS, // lane parser: {start, length, offset} records, 8 per lane and unit
    S_LZ4_CLASSIFY,  // two-pass match finder: list + flags of the groups with redundancy
    S_B3_WINDOWS,   // first range of every window of leaf slots (parents kernel)
    S_Z_FR,         // zstd decoder, frames of other encoders: per block {history in, start, history out}
    S_Z_ORG,        // ... and the origins (u32 per byte of output) of the payloads in flight
    S_Z_PERM,       // zstd decoder: the items in piece-major order, per-row counters, per-item done flags
    S_XCHG,         // multi-GPU exchange: range tables of lthip_exchange_reorder
    S_XCHG2,        // ... and the job tables of lthip_job_ordinals
    S_SLICE_OFFS,   // lthip_chunk_hash in two slices: the second slice's chunk offsets / lengths / hashes / part table before the join
    S_SLICE_LENS,
    S_SLICE_HASH,
    S_SLICE_FIRST,
    S_COUNT
};

struct TimingRec
{
    int kid;
    hipEvent_t a, b;
};

struct lthip_ctx
{
    int device;
    hipStream_t stream;
    bool own_stream;
    lthip_ctx* slice_ctx;                // lazily created: context (private stream, scratch of its own) of lthip_chunk_hash's second slice
    hipStream_t stream2;                 // lazily created side stream (non-blocking), see lthip_second_stream
    std::vector<hipEvent_t> sync_events; // ordering events between the two streams, reused round-robin
    size_t sync_next;
    char err[320];
    struct Stage // pinned staging slot of lthip_stage_upload
    {
        void* p;
        size_t cap;
        hipEvent_t done;
        bool used;
    };
    // two rings: uploads of up to LTHIP_STAGE_SMALL bytes (the per-batch tables of the codec phase: with one ring of 8 slots
    // lthip_ingest_write waited for the codec batch after batch) rotate through 64 slots of that size, larger ones (plans, the
    // tree's tables) through 8 slots that grow to what they have to hold
    Stage stage[8], stage_small[64];
    size_t stage_small_next;
    size_t stage_next;
    bool k1_lds_enabled; // hipFuncAttributeMaxDynamicSharedMemorySize set for K1 on this context's device
    bool k5_lds_enabled[2][2]; // ... and for k_lz4_segments<.., FMT, CLS> with units above 4 KiB
    bool k5h_lds_enabled[2]; // ... and for k_lz4_lanes2<FMT>
    void* scratch[S_COUNT];
    size_t scratch_cap[S_COUNT];
    bool timing;
    std::vector<TimingRec> pending;
    std::vector<hipEvent_t> free_events;
    const uint32_t* z_last_retry; // what the last lthip_zstd_decompress_blocks call of THIS context did (lthip_zstd_last_decode_stats)
    uint32_t z_last_payloads, z_last_foreign_blocks;
    double total_ms[LTHIP_K_COUNT];
    uint64_t launches[LTHIP_K_COUNT];
};

int lthip_fail(lthip_ctx* ctx, int code, const char* what, const char* detail);
// Waits for everything queued on the context's stream (the one place the library waits for a stream).  HOW a thread waits is the
// device's scheduling policy: lthip_set_blocking_waits (lthip_ctx.hip).
hipError_t lthip_stream_wait(lthip_ctx* ctx);

// Environment switches (ablations, debug paths) are read once per process and cached -- not per call: the plugins call from up to 256
// threads.  lthip_debug_reload_env() (tests: one process tries a path and its ablation) bumps the generation; a cache re-reads then.
extern volatile uint32_t g_lthip_env_gen;
struct LthipEnvInt
{
    // (function-local statics shared by every host thread of the plugin layer: the value is published before the generation it
    // belongs to, so a reader that sees the generation sees the value)
    const char* name;
    std::atomic<int> value{-1};     // atoi of the variable; -1 when it is not set
    std::atomic<uint32_t> seen{0};  // generation the value was read at (0: never)
    explicit LthipEnvInt(const char* n) : name(n) {}
    int get()
    {
        const uint32_t gen = g_lthip_env_gen;
        if (seen.load(std::memory_order_acquire) != gen)
        {
            const char* e = getenv(name);
            value.store(e ? atoi(e) : -1, std::memory_order_relaxed);
            seen.store(gen, std::memory_order_release);
        }
        return value.load(std::memory_order_relaxed);
    }
};
// The PRODUCT library reads eleven environment variables, all documented in README.md ("Environment"): LONGTAIL_HIP_DEVICE / _BATCH /
// _CODEC_BATCH / _LARGE_WINDOWS / _SMALL_WINDOWS (plugin layer), LTHIP_BATCH_BYTES, LTHIP_ORIGIN_MIB, LTHIP_COMM_TRANSPORT /
// _TIMEOUT_S / _SHM_SLOT / LTHIP_RCCL_PATH.  Every other LTHIP_* switch selects an earlier formulation of a kernel, a debug path or an experiment's
// parameter; those exist in the ABLATION build only (`make ablations`: -DLTHIP_ABLATIONS, build/ablations/liblongtail_hip.so, loaded
// by the differential tests and the A/B tools through LTHIP_LIB_PATH).  In the product build such a switch is a constant "not set",
// so the branches it guards fold away, and the kernels they launch are not compiled (#ifdef LTHIP_ABLATIONS around them).
#ifdef LTHIP_ABLATIONS
#define LTHIP_ABLATION_ENV(var, name) static LthipEnvInt var{name}
#else
struct LthipEnvOff
{
    constexpr int get() const { return -1; }
};
#define LTHIP_ABLATION_ENV(var, name) constexpr LthipEnvOff var{}
#endif
// Every device / pinned allocation of the library goes through these two.  Product build: hipMalloc / hipHostMalloc, nothing else.
// Ablation build: counted, and lthip_debug_fail_alloc(after, count) makes allocations after+1 .. after+count of the process fail with
// hipErrorOutOfMemory (tests/test_gpu_alloc_failures.py: ENOMEM out of CreateVersionIndex / WriteContent, nothing leaked, the same
// objects usable again -- the device-side counterpart of the reference's FailableStorageAPI tests, test/test.cpp:5677-5752).
#ifdef LTHIP_ABLATIONS
hipError_t lthip_hip_malloc(void** p, size_t bytes);
hipError_t lthip_hip_host_malloc(void** p, size_t bytes, unsigned flags);
#else
static inline hipError_t lthip_hip_malloc(void** p, size_t bytes) { return hipMalloc(p, bytes); }
static inline hipError_t lthip_hip_host_malloc(void** p, size_t bytes, unsigned flags) { return hipHostMalloc(p, bytes, flags); }
#endif
int lthip_scratch(lthip_ctx* ctx, int slot, size_t bytes, void** out);
// A second in-order queue of the context for work that should overlap the main stream (callers order the two with
// events from lthip_sync_event and must make the main stream wait for the side stream before they return).
int lthip_second_stream(lthip_ctx* ctx, hipStream_t* out);
hipEvent_t lthip_sync_event(lthip_ctx* ctx);
// zstd frames of the "max" setting: 128 KiB pieces per chain (the match finder -- k_lz4.hip -- lets a piece see the one before it except
// in every LTHIP_ZSTD_CHAIN-th piece of a block; the decoder -- k_zstd.hip -- runs the pieces in between in order)
constexpr uint32_t LTHIP_ZSTD_CHAIN = 8;
uint64_t lthip_codec_batch_bytes(); // input bytes per internal codec batch (LTHIP_BATCH_BYTES, default 8 GiB)
uint64_t lthip_origin_budget_mib(); // (LTHIP_ORIGIN_MIB) arena of the restore paths' execution on origins (k_lz4.hip)
// Host table -> device without stalling the caller: the bytes are copied into one of a ring of pinned staging buffers and
// queued on `stream`; `h_src` may be freed on return, and the host does not wait for earlier work of the stream (a
// pageable hipMemcpyAsync + hipStreamSynchronize would wait for every kernel queued before it).
constexpr size_t LTHIP_STAGE_SMALL = 64u << 10;
int lthip_stage_upload(lthip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t stream);

#define LTHIP_CHECK(ctx, expr)                                                          \
    do                                                                                  \
    {                                                                                   \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess)                                                          \
            return lthip_fail((ctx), e__ == hipErrorOutOfMemory ? ENOMEM : EIO, #expr, hipGetErrorString(e__)); \
    } while (0)

// timing brackets around a launch on ctx->stream
struct LaunchTimer
{
    lthip_ctx* ctx;
    TimingRec rec;
    bool on;
    hipStream_t stream;
    LaunchTimer(lthip_ctx* c, int kid, hipStream_t s = nullptr); // s == nullptr: the context's main stream
    ~LaunchTimer();
};

#define LTHIP_LAUNCH_CHECK(ctx)                                                          \
    do                                                                                   \
    {                                                                                    \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess)                                                           \
            return lthip_fail((ctx), EIO, "kernel launch", hipGetErrorString(e__));      \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// kernel launchers implemented in the k_*.hip files
// ---------------------------------------------------------------------------------------------------
int lthip_launch_tile_table(lthip_ctx* ctx, lthip_plan* plan);
int lthip_launch_buzhash(lthip_ctx* ctx, const lthip_plan* plan, const uint8_t* d_data, uint64_t* bm0, uint64_t* bm1);
int lthip_launch_select(lthip_ctx* ctx, const lthip_plan* plan, const uint64_t* bm0, const uint64_t* bm1, uint2* region,
                        uint32_t* part_count);
int lthip_launch_compact(lthip_ctx* ctx, const lthip_plan* plan, const uint2* region, const uint32_t* part_count,
                         uint32_t* d_part_first, uint64_t* d_chunk_offsets, uint32_t* d_chunk_lens);
int lthip_exclusive_scan_u32(lthip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n_bound, const uint32_t* d_n,
                             int kid);
// BLAKE3-64 of count ranges (count = min(count_bound, *d_count) when d_count != null).  leaf_bound = upper bound on
// the total number of 1 KiB leaves (0 = unknown: read it back from the device), max_len = upper bound on a single
// range's length (0 = unknown).
int lthip_launch_blake3(lthip_ctx* ctx, const uint8_t* d_data, const uint64_t* d_offsets, const uint32_t* d_lens,
                        const uint32_t* d_count, uint64_t count_bound, uint64_t leaf_bound, uint64_t max_len,
                        uint64_t* d_hashes);

int lthip_hash_ranges_known(lthip_ctx* ctx, const void* d_data, uint64_t range_count, const uint64_t* d_offsets, const uint32_t* d_lens,
                            uint32_t max_len, uint64_t leaf_total, uint64_t* d_hashes); // lthip_hash_ranges without its read-back

int lthip_launch_blake3_stream_batch(lthip_ctx* ctx, const void* d_data, uint32_t leaf0, uint32_t* d_stack, uint32_t depth_in, uint32_t merges);
int lthip_launch_blake3_stream_final(lthip_ctx* ctx, const void* d_tail, uint32_t tail_len, uint32_t leaf0, const uint32_t* d_stack,
                                     uint32_t depth, uint64_t* d_out);
int lthip_launch_blake3_one(lthip_ctx* ctx, const void* in, uint32_t len, uint64_t* out);

int lthip_launch_from_buffer(lthip_ctx* ctx, const uint8_t* d_data, uint32_t n, uint32_t min_chunk, const DivTest& dv,
                             uint64_t* d_out);

// zstd front end (k_lz4.hip): the LZ4 match finder run with sequence output.  Per 4 KiB unit u of the batch (units
// are numbered block after block, `unit_base[b]` first): literals at d_lits + u*4096, records at d_recs + u*1024,
// ZbUnitMeta at d_meta + u (zstd_block_core.h).
// quality: LTHIP_ZSTD_Q_* (the parse the zstd settings select)
int lthip_launch_lz_sequences(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                              const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                              uint8_t** d_lits, uint64_t** d_recs, void** d_meta, uint32_t* unit_base, uint64_t* total_units, int quality);

static inline uint64_t div_up_u64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
