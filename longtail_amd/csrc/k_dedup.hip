// k_dedup.hip -- first-seen dedup of chunk hashes on gfx950.
//
// Reference behaviour: the serial pass of Longtail_CreateVersionIndex (src/longtail.c:2951-2970):
// LookupTable_PutUnique over all chunk hashes in (asset, part, chunk) order; a chunk is "unique" the first time
// its hash appears and every later occurrence maps to that first index.  On the GPU the same mapping is the
// minimum index per key in an open-addressing table (atomicCAS on the key, atomicMin on the index), which is
// order independent, so 8 ranks that all-gather their hash arrays (RCCL) derive identical results.
#include "lthip_internal.h"

namespace
{

constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdull;
    z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ull;
    return z ^ (z >> 33);
}

__global__ void k_dedup_clear(uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, uint64_t slots, uint32_t* special,
                              unsigned long long* unique)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x)
    {
        keys[i] = EMPTY_KEY;
        idx[i] = 0xFFFFFFFFu;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        *special = 0xFFFFFFFFu;
        *unique = 0ull;
    }
}

// `ordinals` (may be null): the value kept per hash is the MINIMUM of ordinals[i] instead of the minimum position i -- the
// hash-range-sharded first-seen table of the multi-GPU path, where a rank holds an arbitrary subset of the chunks and each comes
// with its global position (lthip_dedup_min_ordinal)
__global__ void k_dedup_insert(const uint64_t* __restrict__ hashes, uint64_t n, uint64_t* __restrict__ keys,
                               uint32_t* __restrict__ idx, uint64_t mask, uint32_t* special, unsigned long long* distinct,
                               const uint32_t* __restrict__ ordinals)
{
    const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool claimed = false; // this insert claimed a slot: one more distinct hash
    if (pos < n)
    {
        const uint64_t h = hashes[pos];
        const uint32_t i = ordinals ? ordinals[pos] : (uint32_t)pos;
        if (h == EMPTY_KEY)
            claimed = atomicMin(special, i) == 0xFFFFFFFFu;
        else
        {
            uint64_t slot = mix64(h) & mask;
            for (;;)
            {
                const unsigned long long prev =
                    atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]), (unsigned long long)EMPTY_KEY, (unsigned long long)h);
                if (prev == EMPTY_KEY || prev == h)
                {
                    atomicMin(&idx[slot], i);
                    claimed = prev == EMPTY_KEY;
                    break;
                }
                slot = (slot + 1) & mask;
            }
        }
    }
    // one add per wave (measured: not what the kernel's time is -- 0.45 ms for 2.15 M hashes either way: the CAS and the minimum are
    // 4.3 M device-scope atomics on random slots)
    const uint64_t b = __builtin_amdgcn_ballot_w64(claimed);
    if (distinct && b && (threadIdx.x & 63) == 0)
        atomicAdd(distinct, (unsigned long long)__builtin_popcountll(b));
}

__global__ void k_dedup_lookup(const uint64_t* __restrict__ hashes, uint64_t i0, uint64_t n, const uint64_t* __restrict__ keys,
                               const uint32_t* __restrict__ idx, uint64_t mask, const uint32_t* special,
                               uint32_t* __restrict__ first_index, unsigned long long* unique)
{
    const uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mine = 0;
    if (i < i0 + n)
    {
        const uint64_t h = hashes[i];
        uint32_t f;
        if (h == EMPTY_KEY)
            f = *special;
        else
        {
            uint64_t slot = mix64(h) & mask;
            while (keys[slot] != h)
                slot = (slot + 1) & mask;
            f = idx[slot];
        }
        first_index[i - i0] = f;
        mine = f == (uint32_t)i;
    }
    const uint64_t b = __builtin_amdgcn_ballot_w64(mine != 0);
    if (unique && (threadIdx.x & 63) == 0 && b)
        atomicAdd(unique, (unsigned long long)__builtin_popcountll(b));
}

} // namespace

// Inserts all `count` hashes, answers for [lookup_first, lookup_first + lookup_count): d_first_index[j] = global index of the
// first occurrence of hash lookup_first + j.  *d_unique_count = number of DISTINCT hashes among all `count` (counted at
// insertion, so it does not need the lookups of the other ranks' ranges).
extern "C" int lthip_dedup_first_seen_range(lthip_ctx* ctx, uint64_t count, const uint64_t* d_hashes, uint64_t lookup_first,
                                            uint64_t lookup_count, uint32_t* d_first_index, uint64_t* d_unique_count)
{
    if (!ctx || !d_unique_count || (count && !d_hashes) || (lookup_count && !d_first_index) || lookup_first + lookup_count > count)
        return EINVAL;
    if (count > 0x7FFFFFFFull)
        return lthip_fail(ctx, EINVAL, "dedup", "too many hashes");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint64_t slots = 1024;
    while (slots < count * 2)
        slots <<= 1;
    void *keys, *idx, *misc;
    int err;
    if ((err = lthip_scratch(ctx, S_TABLES, slots * 8, &keys)))
        return err;
    if ((err = lthip_scratch(ctx, S_TABLES2, slots * 4, &idx)))
        return err;
    if ((err = lthip_scratch(ctx, S_MISC, 64, &misc)))
        return err;
    uint32_t* special = (uint32_t*)misc;
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_dedup_clear, dim3(2048), dim3(256), 0, ctx->stream, (uint64_t*)keys, (uint32_t*)idx, slots, special,
                       (unsigned long long*)d_unique_count);
    if (count)
        hipLaunchKernelGGL(k_dedup_insert, dim3((uint32_t)div_up_u64(count, 256)), dim3(256), 0, ctx->stream, d_hashes, count,
                           (uint64_t*)keys, (uint32_t*)idx, slots - 1, special, (unsigned long long*)d_unique_count, (const uint32_t*)nullptr);
    if (lookup_count)
        hipLaunchKernelGGL(k_dedup_lookup, dim3((uint32_t)div_up_u64(lookup_count, 256)), dim3(256), 0, ctx->stream, d_hashes,
                           lookup_first, lookup_count, (const uint64_t*)keys, (const uint32_t*)idx, slots - 1, (const uint32_t*)special,
                           d_first_index, (unsigned long long*)nullptr);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

extern "C" int lthip_dedup_first_seen(lthip_ctx* ctx, uint64_t count, const uint64_t* d_hashes, uint32_t* d_first_index,
                                      uint64_t* d_unique_count)
{
    if (count && !d_first_index)
        return EINVAL;
    return lthip_dedup_first_seen_range(ctx, count, d_hashes, 0, count, d_first_index, d_unique_count);
}

// The sharded form: THIS rank owns an arbitrary subset of all chunks' hashes (those of its hash range), each with its global
// position; d_first_ordinal[j] = the smallest position among the items with the hash of item j -- what the owner answers to the
// rank that sent item j.  *d_unique_count = distinct hashes of the subset (summed over the owners: the tree's unique chunks).
extern "C" int lthip_dedup_min_ordinal(lthip_ctx* ctx, uint64_t count, const uint64_t* d_hashes, const uint32_t* d_ordinals,
                                       uint32_t* d_first_ordinal, uint64_t* d_unique_count)
{
    if (!ctx || !d_unique_count || (count && (!d_hashes || !d_ordinals || !d_first_ordinal)))
        return EINVAL;
    if (count > 0x7FFFFFFFull)
        return lthip_fail(ctx, EINVAL, "dedup", "too many hashes");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint64_t slots = 1024;
    while (slots < count * 2)
        slots <<= 1;
    void *keys, *idx, *misc;
    int err;
    if ((err = lthip_scratch(ctx, S_TABLES, slots * 8, &keys)))
        return err;
    if ((err = lthip_scratch(ctx, S_TABLES2, slots * 4, &idx)))
        return err;
    if ((err = lthip_scratch(ctx, S_MISC, 64, &misc)))
        return err;
    uint32_t* special = (uint32_t*)misc;
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_dedup_clear, dim3(2048), dim3(256), 0, ctx->stream, (uint64_t*)keys, (uint32_t*)idx, slots, special,
                       (unsigned long long*)d_unique_count);
    if (count)
    {
        hipLaunchKernelGGL(k_dedup_insert, dim3((uint32_t)div_up_u64(count, 256)), dim3(256), 0, ctx->stream, d_hashes, count,
                           (uint64_t*)keys, (uint32_t*)idx, slots - 1, special, (unsigned long long*)d_unique_count, d_ordinals);
        hipLaunchKernelGGL(k_dedup_lookup, dim3((uint32_t)div_up_u64(count, 256)), dim3(256), 0, ctx->stream, d_hashes, (uint64_t)0, count,
                           (const uint64_t*)keys, (const uint32_t*)idx, slots - 1, (const uint32_t*)special, d_first_ordinal,
                           (unsigned long long*)nullptr);
    }
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
