// k_blake3.hip -- BLAKE3-64 chunk hashing on gfx950.
//
// Reference behaviour: Blake3Hash_HashBuffer (lib/blake3/longtail_blake3.c:81-102) = unkeyed BLAKE3, first 8
// output bytes as a little-endian u64; arithmetic per lib/blake3/ext/blake3_portable.c:8-98 (compression
// function), blake3_impl.h:85-97 (IV, message schedule), blake3.c:118-166, 216-249, 576-618 (leaf/"chunk"
// state machine, left-heavy tree, ROOT finalisation).
//
// MI355X formulation: pure 32-bit integer ALU work (no MFMA):
//   leaf kernel   one thread per 1 KiB leaf of every range: 16 state + 16 message words live in VGPRs, the 7
//                 rounds are fully unrolled with the message schedule folded into register names; ranges start
//                 at arbitrary byte offsets, so message words are rebuilt from 4-byte aligned dwords with one
//                 v_alignbit each.  Leaf -> range mapping is a binary search over the exclusive scan of leaf
//                 counts (no per-leaf descriptor traffic).
//   parent kernel in-place left-heavy reduction of each range's chaining values (level by level, stride
//                 doubling == blake3's compress_parents_parallel with the odd node carried up), ROOT on the
//                 last merge; writes output words 0,1 as the u64 digest.
#include "lthip_internal.h"

#include <stdlib.h>

namespace
{

enum
{
    F_CHUNK_START = 1, // blake3_impl.h:14-22
    F_CHUNK_END = 2,
    F_PARENT = 4,
    F_ROOT = 8
};

#define B3_IV0 0x6A09E667u
#define B3_IV1 0xBB67AE85u
#define B3_IV2 0x3C6EF372u
#define B3_IV3 0xA54FF53Au
#define B3_IV4 0x510E527Fu
#define B3_IV5 0x9B05688Cu
#define B3_IV6 0x1F83D9ABu
#define B3_IV7 0x5BE0CD19u

__device__ __forceinline__ uint32_t rotr32(uint32_t x, uint32_t r) { return __builtin_amdgcn_alignbit(x, x, r); }

__device__ __forceinline__ void b3_g(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t mx, uint32_t my)
{
    a = a + b + mx;
    d = rotr32(d ^ a, 16);
    c = c + d;
    b = rotr32(b ^ c, 12);
    a = a + b + my;
    d = rotr32(d ^ a, 8);
    c = c + d;
    b = rotr32(b ^ c, 7);
}

// one round with message words named explicitly (the schedule is applied by the caller's argument order)
#define B3_ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
    b3_g(s0, s4, s8, s12, m0, m1);                                                      \
    b3_g(s1, s5, s9, s13, m2, m3);                                                      \
    b3_g(s2, s6, s10, s14, m4, m5);                                                     \
    b3_g(s3, s7, s11, s15, m6, m7);                                                     \
    b3_g(s0, s5, s10, s15, m8, m9);                                                     \
    b3_g(s1, s6, s11, s12, m10, m11);                                                   \
    b3_g(s2, s7, s8, s13, m12, m13);                                                    \
    b3_g(s3, s4, s9, s14, m14, m15);

// cv <- first 8 output words of compress(cv, m, counter, block_len, flags)
__device__ __forceinline__ void b3_compress(uint32_t (&cv)[8], const uint32_t (&m)[16], uint32_t counter_lo,
                                            uint32_t block_len, uint32_t flags)
{
    uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
    uint32_t s8 = B3_IV0, s9 = B3_IV1, s10 = B3_IV2, s11 = B3_IV3;
    uint32_t s12 = counter_lo, s13 = 0u, s14 = block_len, s15 = flags;
    // rows of MSG_SCHEDULE (blake3_impl.h:89-97): row r+1 = row r permuted by {2,6,3,10,7,0,4,13,1,11,12,5,9,14,15,8}
    B3_ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15])
    B3_ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8])
    B3_ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1])
    B3_ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6])
    B3_ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4])
    B3_ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7])
    B3_ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13])
    cv[0] = s0 ^ s8;
    cv[1] = s1 ^ s9;
    cv[2] = s2 ^ s10;
    cv[3] = s3 ^ s11;
    cv[4] = s4 ^ s12;
    cv[5] = s5 ^ s13;
    cv[6] = s6 ^ s14;
    cv[7] = s7 ^ s15;
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// ---------------------------------------------------------------------------------------------------
// leaf counts
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t leaves_of(uint32_t len) { return len ? (len + 1023u) >> 10 : 1u; }

__global__ void k_leaf_counts(const uint32_t* __restrict__ lens, uint64_t bound, const uint32_t* __restrict__ n_dev,
                              uint32_t* __restrict__ counts)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t n = n_dev ? (*n_dev < bound ? *n_dev : bound) : bound;
    if (i < n)
        counts[i] = leaves_of(lens[i]);
}

// ---------------------------------------------------------------------------------------------------
// leaves
// ---------------------------------------------------------------------------------------------------
#ifndef LTHIP_B3_PW
#define LTHIP_B3_PW 1024
#endif
constexpr int PW = LTHIP_B3_PW; // parents kernel: window of leaf slots per workgroup (a power of two)
constexpr int PW_MAXN = 256; // leaves of the largest range (small trees: <= 256 KiB)
constexpr int PW_SLOTS = PW + PW_MAXN;
#ifndef LTHIP_B3_PW_THREADS
#define LTHIP_B3_PW_THREADS 512
#endif
constexpr int PW_THREADS = LTHIP_B3_PW_THREADS; // threads of a parents workgroup (a power of two).  Measured on the 64 GiB tree (same box,
// builds side by side through LTHIP_LIB_PATH): windows of 256 / 512 / 1024 / 2048 slots with 256 threads 4.19 / 3.54 / 3.12 / 5.55 ms
// (2048: two workgroups per CU); 1024 slots with 256 / 512 / 1024 threads 3.12 / 2.80 / 3.19 ms -- 512 threads have one merge each
// on the first level and three workgroups of them are 24 waves per CU.

__global__ __launch_bounds__(256) void k_blake3_leaves(const uint8_t* __restrict__ data,
                                                       const uint64_t* __restrict__ offsets,
                                                       const uint32_t* __restrict__ lens,
                                                       const uint32_t* __restrict__ leaf_prefix, // [count+1]
                                                       uint64_t count_bound, const uint32_t* __restrict__ n_dev,
                                                       uint32_t* __restrict__ cvs, uint32_t* __restrict__ win_first)
{
    const uint32_t count = (uint32_t)(n_dev ? (*n_dev < count_bound ? *n_dev : count_bound) : count_bound);
    const uint32_t total = leaf_prefix[count];
    const uint64_t g64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g64 >= total)
        return;
    const uint32_t g = (uint32_t)g64;
    // range c with leaf_prefix[c] <= g < leaf_prefix[c+1]
    uint32_t lo = 0, hi = count;
    while (hi - lo > 1)
    {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (leaf_prefix[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t c = lo;
    const uint32_t first = leaf_prefix[c];
    const uint32_t k = g - first; // leaf index inside the range == BLAKE3 chunk counter
    if (win_first && (g & (uint32_t)(PW - 1)) == 0u)
        win_first[g / (uint32_t)PW] = k == 0u ? c : c + 1u; // first range that starts at or after this window (parents kernel)
    const uint32_t rlen = lens[c];
    const uint32_t nleaf = leaves_of(rlen);
    const uint32_t llen = rlen - (k << 10) < 1024u ? rlen - (k << 10) : 1024u; // 0 only for the empty range
    const uint8_t* p = data + offsets[c] + ((uint64_t)k << 10);
    const uint32_t root = nleaf == 1u ? (uint32_t)F_ROOT : 0u;

    const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
    const uint32_t sh = mis * 8u;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p - mis); // 4-byte aligned

    uint32_t cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
    const uint32_t nblocks = llen ? (llen + 63u) >> 6 : 1u;

    // interior blocks: the 64 bytes (and the spill-over dword when misaligned) are inside the leaf
    for (uint32_t b = 0; b + 1 < nblocks; ++b)
    {
        const uint32_t* src = q + b * 16u;
        uint32_t w[17];
        const u32x4_a4 v0 = *reinterpret_cast<const u32x4_a4*>(src);
        const u32x4_a4 v1 = *reinterpret_cast<const u32x4_a4*>(src + 4);
        const u32x4_a4 v2 = *reinterpret_cast<const u32x4_a4*>(src + 8);
        const u32x4_a4 v3 = *reinterpret_cast<const u32x4_a4*>(src + 12);
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
        w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        w[8] = v2.x; w[9] = v2.y; w[10] = v2.z; w[11] = v2.w;
        w[12] = v3.x; w[13] = v3.y; w[14] = v3.z; w[15] = v3.w;
        w[16] = mis ? src[16] : 0u;
        uint32_t m[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            m[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], sh);
        b3_compress(cv, m, k, 64u, b == 0 ? (uint32_t)F_CHUNK_START : 0u);
    }
    // last block: 0..64 bytes, touch only dwords that hold at least one byte of the leaf
    {
        const uint32_t b = nblocks - 1u;
        const uint32_t bl = llen - b * 64u;
        const uint32_t* src = q + b * 16u;
        const uint32_t nd = bl ? (mis + bl + 3u) >> 2 : 0u; // dwords that intersect [p, p+bl)
        uint32_t w[17];
#pragma unroll
        for (int i = 0; i < 17; ++i)
            w[i] = (uint32_t)i < nd ? src[i] : 0u;
        uint32_t m[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            uint32_t v = __builtin_amdgcn_alignbit(w[i + 1], w[i], sh);
            const int rem = (int)bl - 4 * i; // valid bytes in this word
            if (rem <= 0)
                v = 0u;
            else if (rem < 4)
                v &= (1u << (8 * rem)) - 1u;
            m[i] = v;
        }
        uint32_t fl = (uint32_t)F_CHUNK_END | root;
        if (b == 0)
            fl |= (uint32_t)F_CHUNK_START;
        b3_compress(cv, m, k, bl, fl);
    }
    uint4* out = reinterpret_cast<uint4*>(cvs + (uint64_t)g * 8u);
    out[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    out[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// ---------------------------------------------------------------------------------------------------
// parents
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void b3_parent(uint32_t* __restrict__ left, const uint32_t* __restrict__ right, uint32_t flags)
{
    uint32_t m[16];
    const uint4 a0 = reinterpret_cast<const uint4*>(left)[0], a1 = reinterpret_cast<const uint4*>(left)[1];
    const uint4 b0 = reinterpret_cast<const uint4*>(right)[0], b1 = reinterpret_cast<const uint4*>(right)[1];
    m[0] = a0.x; m[1] = a0.y; m[2] = a0.z; m[3] = a0.w; m[4] = a1.x; m[5] = a1.y; m[6] = a1.z; m[7] = a1.w;
    m[8] = b0.x; m[9] = b0.y; m[10] = b0.z; m[11] = b0.w; m[12] = b1.x; m[13] = b1.y; m[14] = b1.z; m[15] = b1.w;
    uint32_t cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
    b3_compress(cv, m, 0u, 64u, flags);
    reinterpret_cast<uint4*>(left)[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    reinterpret_cast<uint4*>(left)[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

#ifdef LTHIP_ABLATIONS
#include "ablations/k_blake3_parents_small.inc"
#endif

// ONE small input (<= 64 KiB) in ONE launch, read where it lies -- pinned host memory, over the bus -- and answered into pinned host
// memory: the plugin layer's HashBuffer of a path string, a chunk-hash array or a block's hash array (src/longtail.c:1272, 2522,
// 3757), which used to cost two uploads, seven launches (counts, scan, leaves, parents) and a download per call.  Lane l hashes leaf
// l byte by byte (speed is irrelevant here: the call is a bus round trip), lane 0 reduces the tree.
constexpr uint32_t B3_ONE_PITCH = 1024u + 4u; // LDS bytes per staged leaf: lanes read their leaves a dword at a time, 257 dwords apart
__global__ __launch_bounds__(64) void k_blake3_one(const uint8_t* __restrict__ in, uint32_t len, uint64_t* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_one[]; // [8 x 64 chaining values][leaves x B3_ONE_PITCH bytes of input]
    uint32_t* s_cv = s_one;
    uint32_t* s_in = s_one + 64 * 8;
    const uint32_t lane = threadIdx.x;
    const uint32_t nleaf = leaves_of(len);
    // The input is the caller's PINNED HOST block as a rule (plugin_hash.c): a lane that reads its 1 KiB leaf where it lies pays a
    // trip over the link per load, sixteen dependent rounds of them (byte loads: 115 us per 1 KiB call, 16-byte loads: 47).  So the wave
    // first brings the whole input into LDS with coalesced 16-byte loads that are all in flight together, then every lane hashes its
    // leaf from there (tools/hash_latency.py: 8 KiB 130 -> 31 us per HashBuffer call; 17 of them are the launch and the wait).
    {
        const uint32_t nvec = (((uintptr_t)in & 15u) == 0u) ? len >> 4 : 0u;
        const uint4* in4 = reinterpret_cast<const uint4*>(in);
        for (uint32_t v0 = 0; v0 < nvec; v0 += 64u * 8u)
        {
            uint4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
                const uint32_t v = v0 + (uint32_t)u * 64u + lane;
                q[u] = v < nvec ? in4[v] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
            {
                const uint32_t v = v0 + (uint32_t)u * 64u + lane;
                if (v < nvec)
                {
                    uint32_t* d = s_in + (v >> 6) * (B3_ONE_PITCH / 4u) + (v & 63u) * 4u; // (64 vectors per leaf)
                    d[0] = q[u].x;
                    d[1] = q[u].y;
                    d[2] = q[u].z;
                    d[3] = q[u].w;
                }
            }
        }
        for (uint32_t j = (nvec << 4) + lane; j < len; j += 64u) // the tail (or everything, from an unaligned address): bytes
            reinterpret_cast<uint8_t*>(s_in)[(j >> 10) * B3_ONE_PITCH + (j & 1023u)] = in[j];
    }
    __syncthreads();
    if (lane < nleaf)
    {
        const uint32_t llen = len - (lane << 10) < 1024u ? len - (lane << 10) : 1024u;
        const uint32_t* pw = s_in + lane * (B3_ONE_PITCH / 4u);
        const uint8_t* p = reinterpret_cast<const uint8_t*>(pw);
        const uint32_t nblocks = llen ? (llen + 63u) >> 6 : 1u;
        uint32_t cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
        for (uint32_t b = 0; b < nblocks; ++b)
        {
            const uint32_t bl = llen - b * 64u < 64u ? llen - b * 64u : 64u;
            uint32_t m[16];
            if (bl == 64u)
            {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    m[i] = pw[b * 16u + (uint32_t)i];
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    m[i] = 0u;
                for (uint32_t j = 0; j < bl; ++j)
                    m[j >> 2] |= (uint32_t)p[b * 64u + j] << (8u * (j & 3u));
            }
            uint32_t fl = b == 0 ? (uint32_t)F_CHUNK_START : 0u;
            if (b + 1 == nblocks)
                fl |= (uint32_t)F_CHUNK_END | (nleaf == 1u ? (uint32_t)F_ROOT : 0u);
            b3_compress(cv, m, lane, bl, fl);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            s_cv[lane * 8 + i] = cv[i];
    }
    __syncthreads();
    // the tree, level by level: the merges of a level are independent (left-heavy tree == in-place stride doubling with the odd node
    // carried), lane j takes the j-th of them (64 leaves: six rounds instead of 63 merges on one lane)
    for (uint32_t stride = 1; stride < nleaf; stride <<= 1)
    {
        const uint32_t last = (stride << 1) >= nleaf;
        const uint32_t k = lane * (stride << 1);
        if (k + stride < nleaf)
            b3_parent(s_cv + k * 8u, s_cv + (k + stride) * 8u, (uint32_t)F_PARENT | (last ? (uint32_t)F_ROOT : 0u));
        __syncthreads();
    }
    if (lane == 0)
        *out = (uint64_t)s_cv[0] | ((uint64_t)s_cv[1] << 32);
}

// ---------------------------------------------------------------------------------------------------
// Streaming (Blake3Hash_BeginContext / _Hash / _EndContext, lib/blake3/longtail_blake3.c:24-79; ext/blake3.c:462-618).
// The reference keeps O(1) state: a partial 1 KiB chunk and a stack of subtree chaining values (one per set bit of the chunk
// count).  Same here, with the stack in device memory and the unit of work a BATCH of 1024 full leaves (1 MiB):
//   k_blake3_stream_batch  leaves leaf0 .. leaf0 + 1023 of the stream -> their complete subtree's chaining value (1024 threads, ten
//                          merge levels in LDS) -> pushed onto the stack, followed by `merges` = trailing-zeros(batch count) parent
//                          compressions (equal-sized neighbours merge: blake3.c:add_chunk_chaining_value restated per batch)
//   k_blake3_stream_final  the rest of the stream (0 .. 1 MiB, never empty unless the whole stream is): its leaves in parallel, then
//                          one thread pushes them the same way and folds the stack from the top, ROOT on the last compression
//                          (blake3.c:576-618); the empty stream is the one empty ROOT leaf.
// A batch is only hashed once a byte beyond it has arrived (the host holds it back), so the final kernel always has the last leaf.
// ---------------------------------------------------------------------------------------------------
constexpr int B3S_LEAVES = 1024;

__global__ __launch_bounds__(B3S_LEAVES) void k_blake3_stream_batch(const uint8_t* __restrict__ data, uint32_t leaf0,
                                                                      uint32_t* __restrict__ stack, uint32_t depth_in, uint32_t merges)
{
    __shared__ uint32_t s_cv[B3S_LEAVES * 8];
    const uint32_t t = threadIdx.x;
    {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(data) + (size_t)t * 256u; // 16-byte aligned device buffer
        uint32_t cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
        for (uint32_t b = 0; b < 16u; ++b)
        {
            uint32_t m[16];
            const uint4* q = reinterpret_cast<const uint4*>(p + b * 16u);
            const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
            m[0] = v0.x; m[1] = v0.y; m[2] = v0.z; m[3] = v0.w; m[4] = v1.x; m[5] = v1.y; m[6] = v1.z; m[7] = v1.w;
            m[8] = v2.x; m[9] = v2.y; m[10] = v2.z; m[11] = v2.w; m[12] = v3.x; m[13] = v3.y; m[14] = v3.z; m[15] = v3.w;
            b3_compress(cv, m, leaf0 + t, 64u, (b == 0 ? (uint32_t)F_CHUNK_START : 0u) | (b == 15u ? (uint32_t)F_CHUNK_END : 0u));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            s_cv[t * 8 + i] = cv[i];
    }
    __syncthreads();
    for (uint32_t stride = 1; stride < (uint32_t)B3S_LEAVES; stride <<= 1)
    {
        if ((t & (2u * stride - 1u)) == 0u)
            b3_parent(s_cv + t * 8u, s_cv + (t + stride) * 8u, (uint32_t)F_PARENT);
        __syncthreads();
    }
    if (t == 0)
    {
        uint32_t depth = depth_in;
        for (int i = 0; i < 8; ++i)
            stack[depth * 8u + i] = s_cv[i];
        ++depth;
        for (uint32_t m = 0; m < merges; ++m)
        {
            b3_parent(stack + (depth - 2u) * 8u, stack + (depth - 1u) * 8u, (uint32_t)F_PARENT);
            --depth;
        }
    }
}

__global__ __launch_bounds__(B3S_LEAVES) void k_blake3_stream_final(const uint8_t* __restrict__ tail, uint32_t tail_len, uint32_t leaf0,
                                                                      const uint32_t* __restrict__ stack, uint32_t depth_in,
                                                                      uint64_t* __restrict__ out)
{
    __shared__ uint32_t s_cv[B3S_LEAVES * 8];
    __shared__ uint32_t s_stack[64 * 8];
    const uint32_t t = threadIdx.x;
    const uint32_t nleaf = tail_len ? (tail_len + 1023u) >> 10 : (leaf0 == 0u ? 1u : 0u);
    const bool single = leaf0 == 0u && nleaf == 1u; // the whole stream is one leaf: it carries ROOT itself
    if (t < nleaf)
    {
        const uint32_t llen = tail_len - (t << 10) < 1024u ? tail_len - (t << 10) : 1024u;
        const uint8_t* p = tail + ((size_t)t << 10);
        const uint32_t nblocks = llen ? (llen + 63u) >> 6 : 1u;
        uint32_t cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
        for (uint32_t b = 0; b < nblocks; ++b)
        {
            const uint32_t bl = llen - b * 64u < 64u ? llen - b * 64u : 64u;
            uint32_t m[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                m[i] = 0u;
            for (uint32_t j = 0; j < bl; ++j)
                m[j >> 2] |= (uint32_t)p[b * 64u + j] << (8u * (j & 3u));
            uint32_t fl = b == 0 ? (uint32_t)F_CHUNK_START : 0u;
            if (b + 1 == nblocks)
                fl |= (uint32_t)F_CHUNK_END | (single ? (uint32_t)F_ROOT : 0u);
            b3_compress(cv, m, leaf0 + t, bl, fl);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            s_cv[t * 8 + i] = cv[i];
    }
    for (uint32_t i = t; i < depth_in * 8u; i += (uint32_t)B3S_LEAVES)
        s_stack[i] = stack[i];
    __syncthreads();
    if (t != 0)
        return;
    if (single)
    {
        *out = (uint64_t)s_cv[0] | ((uint64_t)s_cv[1] << 32);
        return;
    }
    // every leaf but the last is pushed and merged eagerly: after leaf number g (0-based) the stack holds one subtree per set bit of
    // g + 1.  The stack counts BATCHES below the tail (leaf0 is a multiple of 1024, so its entries are the set bits of leaf0 >> 10
    // and sit below anything the tail pushes).
    uint32_t depth = depth_in;
    for (uint32_t j = 0; j + 1 < nleaf; ++j)
    {
        for (int i = 0; i < 8; ++i)
            s_stack[depth * 8u + i] = s_cv[j * 8u + i];
        ++depth;
        uint32_t cnt = leaf0 + j + 1u; // chunks so far
        while ((cnt & 1u) == 0u)      // equal-sized neighbours merge
        {
            b3_parent(s_stack + (depth - 2u) * 8u, s_stack + (depth - 1u) * 8u, (uint32_t)F_PARENT);
            --depth;
            cnt >>= 1;
        }
    }
    // the last leaf's value, folded with the stack from the top; ROOT on the final compression
    uint32_t* cur = s_cv + (nleaf - 1u) * 8u;
    while (depth > 0u)
    {
        uint32_t* left = s_stack + (depth - 1u) * 8u;
        b3_parent(left, cur, (uint32_t)F_PARENT | (depth == 1u ? (uint32_t)F_ROOT : 0u));
        cur = left;
        --depth;
    }
    *out = (uint64_t)cur[0] | ((uint64_t)cur[1] << 32);
}

// small trees, lane-dense: a workgroup owns the ranges whose first leaf slot lies in its window of PW slots, keeps their
// chaining values in LDS and reduces ALL of them level by level -- the merges of one level (any range, any position) are
// compacted into a list so that every lane of every wave has one, instead of one thread walking one tree serially with
// its 2 x 32-byte loads uncoalesced (k_blake3_parents_small).  Left-heavy tree as above: the node of leaves [k, k + 2 st)
// lives at slot k; a level's merges read slots (k, k + st) and write slot k, so they are independent.
__global__ __launch_bounds__(PW_THREADS) void k_blake3_parents_window(const uint32_t* __restrict__ leaf_prefix, uint64_t count_bound,
                                                               const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ cvs,
                                                               const uint32_t* __restrict__ win_first, uint64_t* __restrict__ hashes)
{
    __shared__ uint4 s_cv[PW_SLOTS * 2];
    __shared__ uint32_t s_info[PW_SLOTS]; // leaf index inside its range | leaves of the range << 16
    __shared__ uint16_t s_list[PW_SLOTS / 2 + 64];
    __shared__ uint32_t s_cnt, s_maxn;
    const uint32_t count = (uint32_t)(n_dev ? (*n_dev < count_bound ? *n_dev : count_bound) : count_bound);
    const uint32_t total = leaf_prefix[count];
    const uint64_t w0 = (uint64_t)blockIdx.x * PW;
    if (w0 >= total)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // first range that starts at or after the window / the next window: noted by the leaf kernel's thread of that slot (a
    // binary search here would be ~40 dependent loads before the workgroup can start)
    const uint32_t c_lo = win_first[blockIdx.x];
    const uint32_t c_hi = w0 + PW >= total ? count : win_first[blockIdx.x + 1u];
    const uint32_t nch = c_hi - c_lo;
    if (nch == 0)
        return;
    const uint32_t wbase = leaf_prefix[c_lo];
    const uint32_t nslots = leaf_prefix[c_hi] - wbase;
    if (tid == 0)
    {
        s_cnt = 0;
        s_maxn = 0;
    }
    {
        const uint4* g = reinterpret_cast<const uint4*>(cvs) + (uint64_t)wbase * 2u;
        for (uint32_t v = tid; v < nslots * 2u; v += PW_THREADS)
            s_cv[v] = g[v];
    }
    __syncthreads();
    if (nch * 8u < nslots)
    {
        // few large ranges: a wave per range
        for (uint32_t j = tid >> 6; j < nch; j += PW_THREADS / 64)
        {
            const uint32_t p0 = leaf_prefix[c_lo + j], n = leaf_prefix[c_lo + j + 1] - p0;
            for (uint32_t k = lane; k < n; k += 64)
                s_info[p0 - wbase + k] = k | (n << 16);
            if (lane == 0)
                atomicMax(&s_maxn, n);
        }
    }
    else
    {
        for (uint32_t j = tid; j < nch; j += PW_THREADS)
        {
            const uint32_t p0 = leaf_prefix[c_lo + j], n = leaf_prefix[c_lo + j + 1] - p0;
            for (uint32_t k = 0; k < n; ++k)
                s_info[p0 - wbase + k] = k | (n << 16);
            atomicMax(&s_maxn, n);
        }
    }
    __syncthreads();
    const uint32_t maxn = s_maxn;
    uint32_t rot = blockIdx.x; // which wave takes the first 64 merges of a level: rotates, or the SIMD of wave 0 does all the thin levels
    for (uint32_t st = 1; st < maxn; st <<= 1, ++rot)
    {
        for (uint32_t s0 = 0; s0 < nslots; s0 += PW_THREADS)
        {
            const uint32_t sl = s0 + (uint32_t)tid;
            const uint32_t info = sl < nslots ? s_info[sl] : 0u;
            const uint32_t k = info & 0xFFFFu, n = info >> 16;
            const bool is = (k & ((st << 1) - 1u)) == 0u && k + st < n;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(is);
            uint32_t base = 0;
            if (lane == 0 && mask)
                base = atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            if (is)
                s_list[base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint16_t)sl;
        }
        __syncthreads();
        const uint32_t nm = s_cnt;
        for (uint32_t mi = ((uint32_t)tid + 64u * rot) & (uint32_t)(PW_THREADS - 1); mi < nm; mi += PW_THREADS)
        {
            const uint32_t sl = s_list[mi];
            const uint32_t n = s_info[sl] >> 16;
            const uint4 a0 = s_cv[2u * sl], a1 = s_cv[2u * sl + 1u];
            const uint4 b0 = s_cv[2u * (sl + st)], b1 = s_cv[2u * (sl + st) + 1u];
            uint32_t m[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            uint32_t cv[8] = {B3_IV0, B3_IV1, B3_IV2, B3_IV3, B3_IV4, B3_IV5, B3_IV6, B3_IV7};
            b3_compress(cv, m, 0u, 64u, (uint32_t)F_PARENT | ((st << 1) >= n ? (uint32_t)F_ROOT : 0u));
            s_cv[2u * sl] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
            s_cv[2u * sl + 1u] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
        }
        __syncthreads();
        if (tid == 0)
            s_cnt = 0;
        __syncthreads();
    }
    for (uint32_t j = tid; j < nch; j += PW_THREADS)
    {
        const uint4 r = s_cv[2u * (leaf_prefix[c_lo + j] - wbase)];
        hashes[c_lo + j] = (uint64_t)r.x | ((uint64_t)r.y << 32);
    }
}

// big trees: one launch per level, one thread per leaf slot
__global__ __launch_bounds__(256) void k_blake3_parents_level(const uint32_t* __restrict__ leaf_prefix, uint32_t count,
                                                              uint32_t stride, uint32_t* __restrict__ cvs)
{
    const uint32_t total = leaf_prefix[count];
    const uint64_t g64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g64 >= total)
        return;
    const uint32_t g = (uint32_t)g64;
    uint32_t lo = 0, hi = count;
    while (hi - lo > 1)
    {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (leaf_prefix[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t slot0 = leaf_prefix[lo];
    const uint32_t n = leaf_prefix[lo + 1] - slot0;
    const uint32_t k = g - slot0;
    if ((k & ((stride << 1) - 1u)) != 0u || k + stride >= n)
        return;
    const uint32_t last = (stride << 1) >= n;
    b3_parent(cvs + (uint64_t)g * 8u, cvs + (uint64_t)(g + stride) * 8u, (uint32_t)F_PARENT | (last ? (uint32_t)F_ROOT : 0u));
}

__global__ void k_blake3_emit(const uint32_t* __restrict__ leaf_prefix, uint32_t count, const uint32_t* __restrict__ cvs,
                              uint64_t* __restrict__ hashes)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count)
        return;
    const uint32_t* base = cvs + (uint64_t)leaf_prefix[c] * 8u;
    hashes[c] = (uint64_t)base[0] | ((uint64_t)base[1] << 32);
}

} // namespace

int lthip_launch_blake3(lthip_ctx* ctx, const uint8_t* d_data, const uint64_t* d_offsets, const uint32_t* d_lens,
                        const uint32_t* d_count, uint64_t count_bound, uint64_t leaf_bound, uint64_t max_len,
                        uint64_t* d_hashes)
{
    if (count_bound == 0)
        return 0;
    void *lc, *lp, *cv;
    int err;
    if ((err = lthip_scratch(ctx, S_LEAF_COUNT, (count_bound + 1) * 4, &lc)))
        return err;
    if ((err = lthip_scratch(ctx, S_LEAF_PREFIX, (count_bound + 1) * 4, &lp)))
        return err;
    {
        LaunchTimer t(ctx, LTHIP_K_COMPACT);
        hipLaunchKernelGGL(k_leaf_counts, dim3((uint32_t)div_up_u64(count_bound, 256)), dim3(256), 0, ctx->stream, d_lens,
                           count_bound, d_count, (uint32_t*)lc);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    if ((err = lthip_exclusive_scan_u32(ctx, (const uint32_t*)lc, (uint32_t*)lp, count_bound, d_count, LTHIP_K_COMPACT)))
        return err;

    const bool small_trees = max_len != 0 && max_len <= 256u * 1024u;
    uint32_t host_count = 0;
    if (leaf_bound == 0 || !small_trees)
    {
        // need exact numbers on the host: the count (when it lives on the device) and the leaf total
        if (d_count)
        {
            LTHIP_CHECK(ctx, hipMemcpyAsync(&host_count, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
            LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
            if (host_count > count_bound)
                host_count = (uint32_t)count_bound;
        }
        else
            host_count = (uint32_t)count_bound;
        uint32_t total = 0;
        LTHIP_CHECK(ctx, hipMemcpyAsync(&total, (const uint32_t*)lp + host_count, 4, hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        leaf_bound = total;
        if (host_count == 0)
            return 0;
    }
    if (leaf_bound == 0)
        return 0;
    if (leaf_bound > 0xFFFFFFF0ull)
        return lthip_fail(ctx, EINVAL, "blake3", "too many leaves in one batch");
    if ((err = lthip_scratch(ctx, S_CV, leaf_bound * 32, &cv)))
        return err;
    void* wf = nullptr;
    if (small_trees && (err = lthip_scratch(ctx, S_B3_WINDOWS, (leaf_bound / PW + 2) * 4, &wf)))
        return err;
    {
        LaunchTimer t(ctx, LTHIP_K_B3_LEAF);
        LTHIP_ABLATION_ENV(env_pad, "LTHIP_B3_PAD_LDS"); // experiment: unused LDS limits the waves per CU
        const int pad_lds = env_pad.get() > 0 ? env_pad.get() : 0;
        hipLaunchKernelGGL(k_blake3_leaves, dim3((uint32_t)div_up_u64(leaf_bound, 256)), dim3(256), (size_t)pad_lds, ctx->stream, d_data,
                           d_offsets, d_lens, (const uint32_t*)lp, count_bound, d_count, (uint32_t*)cv, (uint32_t*)wf);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    if (small_trees)
    {
        LaunchTimer t(ctx, LTHIP_K_B3_PARENT);
#ifdef LTHIP_ABLATIONS
        LTHIP_ABLATION_ENV(env_serial, "LTHIP_B3_SERIAL_PARENTS"); // one thread per tree
        if (env_serial.get() >= 0)
            hipLaunchKernelGGL(k_blake3_parents_small, dim3((uint32_t)div_up_u64(count_bound, 256)), dim3(256), 0, ctx->stream,
                               d_lens, (const uint32_t*)lp, count_bound, d_count, (uint32_t*)cv, d_hashes);
        else
#endif
            hipLaunchKernelGGL(k_blake3_parents_window, dim3((uint32_t)div_up_u64(leaf_bound, PW)), dim3(PW_THREADS), 0, ctx->stream,
                               (const uint32_t*)lp, count_bound, d_count, (const uint32_t*)cv, (const uint32_t*)wf, d_hashes);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    else
    {
        LaunchTimer t(ctx, LTHIP_K_B3_PARENT);
        // deepest possible tree: a u32 length has at most 2^22 leaves
        uint64_t max_leaves = max_len ? div_up_u64(max_len, 1024) : (1ull << 22);
        for (uint64_t stride = 1; stride < max_leaves; stride <<= 1)
            hipLaunchKernelGGL(k_blake3_parents_level, dim3((uint32_t)div_up_u64(leaf_bound, 256)), dim3(256), 0,
                               ctx->stream, (const uint32_t*)lp, host_count, (uint32_t)stride, (uint32_t*)cv);
        hipLaunchKernelGGL(k_blake3_emit, dim3((uint32_t)div_up_u64(host_count, 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)lp, host_count, (const uint32_t*)cv, d_hashes);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    return 0;
}

// BLAKE3-64 of one input of at most 64 KiB that the device can read where it is (pinned host memory or device memory), result
// written to `out` (pinned host or device): one launch on the context's stream
int lthip_launch_blake3_one(lthip_ctx* ctx, const void* in, uint32_t len, uint64_t* out)
{
    if (len > 65536u)
        return lthip_fail(ctx, EINVAL, "blake3_one", "input above 64 KiB");
    LaunchTimer t(ctx, LTHIP_K_B3_LEAF);
    const size_t lds = 64 * 8 * 4 + (size_t)(len ? (len + 1023u) >> 10 : 1u) * B3_ONE_PITCH;
    if (lds > 64u * 1024u)
    {
        static bool granted[64] = {}; // per device: more than 64 KiB of dynamic LDS has to be granted explicitly
        if (ctx->device < 0 || ctx->device >= 64 || !granted[ctx->device])
        {
            LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_blake3_one), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            if (ctx->device >= 0 && ctx->device < 64)
                granted[ctx->device] = true;
        }
    }
    hipLaunchKernelGGL(k_blake3_one, dim3(1), dim3(64), lds, ctx->stream, (const uint8_t*)in, len, out);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

int lthip_launch_blake3_stream_batch(lthip_ctx* ctx, const void* d_data, uint32_t leaf0, uint32_t* d_stack, uint32_t depth_in, uint32_t merges)
{
    LaunchTimer t(ctx, LTHIP_K_B3_LEAF);
    hipLaunchKernelGGL(k_blake3_stream_batch, dim3(1), dim3(B3S_LEAVES), 0, ctx->stream, (const uint8_t*)d_data, leaf0, d_stack, depth_in, merges);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

int lthip_launch_blake3_stream_final(lthip_ctx* ctx, const void* d_tail, uint32_t tail_len, uint32_t leaf0, const uint32_t* d_stack,
                                     uint32_t depth, uint64_t* d_out)
{
    LaunchTimer t(ctx, LTHIP_K_B3_PARENT);
    hipLaunchKernelGGL(k_blake3_stream_final, dim3(1), dim3(B3S_LEAVES), 0, ctx->stream, (const uint8_t*)d_tail, tail_len, leaf0, d_stack, depth,
                       d_out);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
