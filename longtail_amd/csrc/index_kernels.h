// index_kernels.h -- small kernels shared by version_index.hip (the stand-alone f1/f2/f4 entry points) and ingest.hip (the
// fused ingest session): unique-index compaction of the first-seen pass (src/longtail.c:2951-2970) and the bytes around a
// stored block's payload (:4111-4150, :3585-3601; compressblockstore.c:103-139).  Included into both translation units.
#pragma once
#include "lthip_internal.h"

namespace
{
// is_first[i] = first_index[i] == i
__global__ void k_vi_mark(const uint32_t* __restrict__ first_index, uint64_t n, uint32_t* __restrict__ is_first)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        is_first[i] = first_index[i] == (uint32_t)i ? 1u : 0u;
}

// asset_chunk_indexes[i] = rank[first_index[i]]; first occurrences also fill the compact arrays
__global__ void k_vi_compact(const uint32_t* __restrict__ first_index, const uint32_t* __restrict__ rank, uint64_t n,
                             const uint64_t* __restrict__ hashes, const uint32_t* __restrict__ lens,
                             const uint32_t* __restrict__ asset_first_chunk /* [assets + 1] */, uint32_t asset_count,
                             const uint32_t* __restrict__ asset_tags /* may be null */, uint32_t* __restrict__ indexes,
                             uint64_t* __restrict__ uniq_hashes, uint32_t* __restrict__ uniq_sizes, uint32_t* __restrict__ uniq_tags)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t f = first_index[i];
    const uint32_t r = rank[f];
    indexes[i] = r;
    if (f == (uint32_t)i)
    {
        uniq_hashes[r] = hashes[i];
        uniq_sizes[r] = lens[i];
        uint32_t tag = 0;
        if (asset_tags)
        {
            uint32_t lo = 0, hi = asset_count; // asset a with asset_first_chunk[a] <= i < asset_first_chunk[a + 1]
            while (hi - lo > 1)
            {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (asset_first_chunk[mid] <= (uint32_t)i)
                    lo = mid;
                else
                    hi = mid;
            }
            tag = asset_tags[lo];
        }
        uniq_tags[r] = tag;
    }
}

} // namespace

namespace
{
__global__ __launch_bounds__(64) void k_stored_block_headers(const uint32_t* __restrict__ block_first_chunk /* [nblocks + 1] */,
                                                             uint32_t nblocks, const uint64_t* __restrict__ chunk_hashes,
                                                             const uint32_t* __restrict__ chunk_lens,
                                                             const uint64_t* __restrict__ block_hashes, uint32_t hash_identifier,
                                                             uint32_t tag, const uint32_t* __restrict__ block_tags /* null: `tag` */,
                                                             const uint32_t* __restrict__ raw_sizes,
                                                             const uint32_t* __restrict__ comp_sizes,
                                                             const uint64_t* __restrict__ image_offsets, uint8_t* __restrict__ arena)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks)
        return;
    const uint32_t c0 = block_first_chunk[b], n = block_first_chunk[b + 1] - c0;
    uint8_t* w = arena + image_offsets[b]; // 8-byte aligned by contract
    const int lane = threadIdx.x;
    if (lane == 0)
    {
        *reinterpret_cast<uint64_t*>(w) = block_hashes[b];
        uint32_t* h = reinterpret_cast<uint32_t*>(w + 8);
        h[0] = hash_identifier;
        h[1] = n;
        h[2] = block_tags ? block_tags[b] : tag;
    }
    uint8_t* hashes = w + 20; // only 4-byte aligned
    for (uint32_t i = lane; i < n; i += 64)
    {
        const uint64_t v = chunk_hashes[c0 + i];
        uint32_t* p = reinterpret_cast<uint32_t*>(hashes + (size_t)i * 8);
        p[0] = (uint32_t)v;
        p[1] = (uint32_t)(v >> 32);
    }
    uint32_t* sizes = reinterpret_cast<uint32_t*>(hashes + (size_t)n * 8);
    for (uint32_t i = lane; i < n; i += 64)
        sizes[i] = chunk_lens[c0 + i];
    if (lane == 0)
    {
        sizes[n] = raw_sizes[b];
        sizes[n + 1] = comp_sizes[b];
    }
}
} // namespace
