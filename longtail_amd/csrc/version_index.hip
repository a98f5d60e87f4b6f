// version_index.hip -- SURVEY.md §8 f1: the tail of Longtail_CreateVersionIndex as one bulk call.
//
// The reference (src/longtail.c:2808-3017) takes the per-asset chunk lists that ChunkAssets concatenated in (asset,
// part, chunk) order (:2499-2517), hashes every asset's array of chunk hashes into its content hash (:2518-2537), hashes
// every path (:2008, 1269-1300), walks all chunks once to keep the FIRST occurrence of every chunk hash -- the unique
// chunk list and the per-asset-chunk index into it (:2951-2970) -- and lays the result out with
// Longtail_BuildVersionIndex (:2709-2806, layout :2551-2584).  Here the chunk lists are already on the device
// (lthip_chunk_hash), so the same steps run there: first-seen table (k_dedup.hip), a scan that turns "is first" into the
// unique index, BLAKE3 of the hash arrays and of the path strings (k_blake3.hip), and a final copy into the serialized
// layout, byte for byte what Longtail_WriteVersionIndexToBuffer (:3415) would write.
#include "lthip_internal.h"
#include "index_kernels.h"

#include <algorithm>
#include <unordered_set>

namespace
{

struct DevBuf
{
    void* p = nullptr;
    ~DevBuf()
    {
        if (p)
            (void)hipFree(p);
    }
    int alloc(lthip_ctx* ctx, size_t bytes)
    {
        LTHIP_CHECK(ctx, lthip_hip_malloc(&p, bytes ? bytes : 16));
        return 0;
    }
};

} // namespace

extern "C" size_t lthip_version_index_size(uint32_t asset_count, uint64_t unique_chunk_count, uint64_t asset_chunk_index_count,
                                           uint32_t path_data_size)
{
    // Longtail_GetVersionIndexDataSize, src/longtail.c:2551-2584
    return 6 * sizeof(uint32_t) + (size_t)asset_count * (8 + 8 + 8 + 4 + 4) + (size_t)asset_chunk_index_count * 4 +
           (size_t)unique_chunk_count * (8 + 4 + 4) + (size_t)asset_count * (4 + 2) + path_data_size;
}

extern "C" int lthip_build_version_index(lthip_ctx* ctx, uint32_t asset_count, const uint64_t* asset_sizes,
                                         const uint32_t* path_start_offsets, const uint16_t* permissions, const char* path_data,
                                         uint32_t path_data_size, const uint32_t* asset_chunk_counts, uint64_t chunk_total,
                                         const uint64_t* d_chunk_hashes, const uint32_t* d_chunk_lens, const uint32_t* asset_tags,
                                         uint32_t hash_identifier, uint32_t target_chunk_size, void* out, size_t out_capacity,
                                         size_t* out_size)
{
    if (!ctx || !out_size || (asset_count && (!asset_sizes || !path_start_offsets || !permissions || !path_data || !asset_chunk_counts)) ||
        (chunk_total && (!d_chunk_hashes || !d_chunk_lens)))
        return EINVAL;
    if (chunk_total > 0x7FFFFFF0ull)
        return lthip_fail(ctx, EINVAL, "version index", "more than 2^31 chunks");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n = (uint32_t)chunk_total;
    std::vector<uint32_t> starts((size_t)asset_count + 1, 0);
    for (uint32_t a = 0; a < asset_count; ++a)
        starts[a + 1] = starts[a] + asset_chunk_counts[a];
    if (starts[asset_count] != n)
        return lthip_fail(ctx, EINVAL, "version index", "asset chunk counts do not add up to the chunk total");

    int err;
    DevBuf d_first, d_isfirst, d_rank, d_idx, d_uh, d_us, d_ut, d_starts, d_tags, d_uniq, d_paths, d_off, d_len, d_ph, d_ch;
    if ((err = d_first.alloc(ctx, (size_t)n * 4)) || (err = d_isfirst.alloc(ctx, (size_t)n * 4)) ||
        (err = d_rank.alloc(ctx, ((size_t)n + 1) * 4)) || (err = d_idx.alloc(ctx, (size_t)n * 4)) ||
        (err = d_uh.alloc(ctx, (size_t)n * 8)) || (err = d_us.alloc(ctx, (size_t)n * 4)) || (err = d_ut.alloc(ctx, (size_t)n * 4)) ||
        (err = d_starts.alloc(ctx, ((size_t)asset_count + 1) * 4)) || (err = d_tags.alloc(ctx, (size_t)asset_count * 4)) ||
        (err = d_uniq.alloc(ctx, 8)) || (err = d_paths.alloc(ctx, (size_t)path_data_size + 16)) ||
        (err = d_off.alloc(ctx, (size_t)asset_count * 8)) || (err = d_len.alloc(ctx, (size_t)asset_count * 4)) ||
        (err = d_ph.alloc(ctx, (size_t)asset_count * 8)) || (err = d_ch.alloc(ctx, (size_t)asset_count * 8)))
        return err;

    // ---- first-seen dedup -> unique index of every asset chunk, compact unique arrays (:2951-2970) ----
    uint64_t unique = 0;
    if ((err = lthip_dedup_first_seen(ctx, n, d_chunk_hashes, (uint32_t*)d_first.p, (uint64_t*)d_uniq.p)))
        return err;
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_starts.p, starts.data(), ((size_t)asset_count + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (asset_tags && asset_count)
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_tags.p, asset_tags, (size_t)asset_count * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n)
    {
        const uint32_t blocks = (uint32_t)div_up_u64(n, 256);
        hipLaunchKernelGGL(k_vi_mark, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)d_first.p, (uint64_t)n,
                           (uint32_t*)d_isfirst.p);
        if ((err = lthip_exclusive_scan_u32(ctx, (const uint32_t*)d_isfirst.p, (uint32_t*)d_rank.p, n, nullptr, LTHIP_K_OTHER)))
            return err;
        hipLaunchKernelGGL(k_vi_compact, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)d_first.p, (const uint32_t*)d_rank.p,
                           (uint64_t)n, d_chunk_hashes, d_chunk_lens, (const uint32_t*)d_starts.p, asset_count,
                           asset_tags ? (const uint32_t*)d_tags.p : (const uint32_t*)nullptr, (uint32_t*)d_idx.p, (uint64_t*)d_uh.p,
                           (uint32_t*)d_us.p, (uint32_t*)d_ut.p);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    LTHIP_CHECK(ctx, hipMemcpyAsync(&unique, d_uniq.p, 8, hipMemcpyDeviceToHost, ctx->stream));

    // ---- content hash of every asset = BLAKE3 of its chunk-hash array (:2518-2537); path hashes (:1269-1300) ----
    std::vector<uint64_t> h_off(asset_count);
    std::vector<uint32_t> h_len(asset_count);
    uint32_t max_len = 0;
    for (uint32_t a = 0; a < asset_count; ++a)
    {
        h_off[a] = (uint64_t)starts[a] * 8u;
        if ((uint64_t)asset_chunk_counts[a] * 8u > 0xFFFFFFFFull)
            return lthip_fail(ctx, EINVAL, "version index", "asset with more than 2^29 chunks");
        h_len[a] = asset_chunk_counts[a] * 8u; // the reference's hash_size is a uint32_t as well (:2522)
        max_len = h_len[a] > max_len ? h_len[a] : max_len;
    }
    if (asset_count)
    {
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_off.p, h_off.data(), (size_t)asset_count * 8, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_len.p, h_len.data(), (size_t)asset_count * 4, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx)); // h_off / h_len are reused below
        if ((err = lthip_hash_ranges(ctx, d_chunk_hashes ? (const void*)d_chunk_hashes : d_paths.p, asset_count, (const uint64_t*)d_off.p,
                                     (const uint32_t*)d_len.p, max_len, (uint64_t*)d_ch.p)))
            return err;
        max_len = 0;
        for (uint32_t a = 0; a < asset_count; ++a)
        {
            if (path_start_offsets[a] >= path_data_size)
                return lthip_fail(ctx, EINVAL, "version index", "path offset outside the path data");
            h_off[a] = path_start_offsets[a];
            h_len[a] = (uint32_t)strnlen(path_data + path_start_offsets[a], path_data_size - path_start_offsets[a]);
            max_len = h_len[a] > max_len ? h_len[a] : max_len;
        }
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_paths.p, path_data, path_data_size, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_off.p, h_off.data(), (size_t)asset_count * 8, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_len.p, h_len.data(), (size_t)asset_count * 4, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        if ((err = lthip_hash_ranges(ctx, d_paths.p, asset_count, (const uint64_t*)d_off.p, (const uint32_t*)d_len.p, max_len,
                                     (uint64_t*)d_ph.p)))
            return err;
    }
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));

    // ---- serialized layout (Longtail_BuildVersionIndex :2757-2806 over InitVersionIndexFromData's section order) ----
    const size_t size = lthip_version_index_size(asset_count, unique, n, path_data_size);
    *out_size = size;
    if (!out || out_capacity < size)
        return ENOMEM;
    uint8_t* w = (uint8_t*)out;
    const uint32_t head[6] = {(0u << 24) | (0u << 16) | 2u /* LONGTAIL_VERSION_INDEX_VERSION_0_0_2, :16-22 */, hash_identifier,
                              target_chunk_size, asset_count, (uint32_t)unique, n};
    memcpy(w, head, sizeof head);
    w += sizeof head;
#define LT_D2H(SRC, BYTES)                                                                      \
    do                                                                                          \
    {                                                                                           \
        if (BYTES)                                                                              \
            LTHIP_CHECK(ctx, hipMemcpy(w, (SRC), (BYTES), hipMemcpyDeviceToHost));              \
        w += (BYTES);                                                                           \
    } while (0)
    LT_D2H(d_ph.p, (size_t)asset_count * 8);                  // m_PathHashes
    LT_D2H(d_ch.p, (size_t)asset_count * 8);                  // m_ContentHashes
    memcpy(w, asset_sizes, (size_t)asset_count * 8);          // m_AssetSizes
    w += (size_t)asset_count * 8;
    memcpy(w, asset_chunk_counts, (size_t)asset_count * 4);   // m_AssetChunkCounts
    w += (size_t)asset_count * 4;
    memcpy(w, starts.data(), (size_t)asset_count * 4);        // m_AssetChunkIndexStarts
    w += (size_t)asset_count * 4;
    LT_D2H(d_idx.p, (size_t)n * 4);                           // m_AssetChunkIndexes
    LT_D2H(d_uh.p, (size_t)unique * 8);                       // m_ChunkHashes
    LT_D2H(d_us.p, (size_t)unique * 4);                       // m_ChunkSizes
    LT_D2H(d_ut.p, (size_t)unique * 4);                       // m_ChunkTags
#undef LT_D2H
    memcpy(w, path_start_offsets, (size_t)asset_count * 4);   // m_NameOffsets
    w += (size_t)asset_count * 4;
    memcpy(w, permissions, (size_t)asset_count * 2);          // m_Permissions
    w += (size_t)asset_count * 2;
    memcpy(w, path_data, path_data_size);                     // m_NameData
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// stored blocks.  What fsblockstore puts in a .lsb file (Longtail_WriteStoredBlockToBuffer,
// src/longtail.c:4111-4150) is the BlockIndex data (layout :3585-3601: block hash, hash identifier, chunk count, tag,
// chunk hashes, chunk sizes; the block hash is the hash of the block's chunk-hash array, :3753-3757) followed by the
// block data, which for a compressed store is [u32 raw size][u32 compressed size][payload] (CompressBlock,
// lib/compressblockstore/longtail_compressblockstore.c:103-139).  The bulk path compresses straight to
// image_offset + lthip_stored_block_header_size(chunk count), so the payload is never copied; the kernel below writes
// the bytes around it.
// ---------------------------------------------------------------------------------------------------

extern "C" size_t lthip_stored_block_header_size(uint32_t chunk_count)
{
    return 8 + 4 + 4 + 4 + (size_t)chunk_count * 12 /* Longtail_GetBlockIndexDataSize */ + 8 /* raw + compressed size */;
}

extern "C" int lthip_write_stored_block_headers(lthip_ctx* ctx, uint32_t block_count, const uint64_t* block_first_chunk,
                                                const uint64_t* d_chunk_hashes, const uint32_t* d_chunk_lens,
                                                uint32_t hash_identifier, uint32_t tag, const uint32_t* raw_sizes,
                                                const uint32_t* d_comp_sizes, void* d_arena, const uint64_t* image_offsets)
{
    if (!ctx || (block_count && (!block_first_chunk || !d_chunk_hashes || !d_chunk_lens || !raw_sizes || !d_comp_sizes || !d_arena ||
                                 !image_offsets)))
        return EINVAL;
    if (block_count == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> first((size_t)block_count + 1), lens(block_count);
    std::vector<uint64_t> offs(block_count);
    uint32_t max_len = 0;
    for (uint32_t b = 0; b <= block_count; ++b)
    {
        if (block_first_chunk[b] > 0x7FFFFFF0ull || (b && block_first_chunk[b] < block_first_chunk[b - 1]))
            return lthip_fail(ctx, EINVAL, "stored blocks", "block_first_chunk must be non-decreasing and below 2^31");
        first[b] = (uint32_t)block_first_chunk[b];
    }
    for (uint32_t b = 0; b < block_count; ++b)
    {
        if (image_offsets[b] & 7u)
            return lthip_fail(ctx, EINVAL, "stored blocks", "image offsets must be 8-byte aligned");
        offs[b] = (uint64_t)first[b] * 8u;
        lens[b] = (first[b + 1] - first[b]) * 8u;
        max_len = lens[b] > max_len ? lens[b] : max_len;
    }
    DevBuf d_first, d_off, d_len, d_bh, d_raw, d_img;
    int err;
    if ((err = d_first.alloc(ctx, ((size_t)block_count + 1) * 4)) || (err = d_off.alloc(ctx, (size_t)block_count * 8)) ||
        (err = d_len.alloc(ctx, (size_t)block_count * 4)) || (err = d_bh.alloc(ctx, (size_t)block_count * 8)) ||
        (err = d_raw.alloc(ctx, (size_t)block_count * 4)) || (err = d_img.alloc(ctx, (size_t)block_count * 8)))
        return err;
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_first.p, first.data(), ((size_t)block_count + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_off.p, offs.data(), (size_t)block_count * 8, hipMemcpyHostToDevice, ctx->stream));
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_len.p, lens.data(), (size_t)block_count * 4, hipMemcpyHostToDevice, ctx->stream));
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_raw.p, raw_sizes, (size_t)block_count * 4, hipMemcpyHostToDevice, ctx->stream));
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_img.p, image_offsets, (size_t)block_count * 8, hipMemcpyHostToDevice, ctx->stream));
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx)); // host vectors go out of scope
    if ((err = lthip_hash_ranges(ctx, d_chunk_hashes, block_count, (const uint64_t*)d_off.p, (const uint32_t*)d_len.p, max_len,
                                 (uint64_t*)d_bh.p)))
        return err;
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_stored_block_headers, dim3(block_count), dim3(64), 0, ctx->stream, (const uint32_t*)d_first.p, block_count,
                       d_chunk_hashes, d_chunk_lens, (const uint64_t*)d_bh.p, hash_identifier, tag, (const uint32_t*)nullptr, (const uint32_t*)d_raw.p, d_comp_sizes,
                       (const uint64_t*)d_img.p, (uint8_t*)d_arena);
    LTHIP_LAUNCH_CHECK(ctx);
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx)); // the DevBufs are freed on return
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Longtail_CreateMissingContent (src/longtail.c:6882-6998) as a bulk call.
// DiffHashes (:6620-6743) keeps the version's chunks that the store does not have, in the order of their first occurrence
// in the version; Longtail_CreateStoreIndex (:6745-6880) packs them into blocks (same tag, <= max_chunks_per_block chunks,
// size <= max_block_size * 1.1) and Longtail_CreateStoreIndexFromBlocks (:9060-9125) lays the StoreIndex out (layout
// :8913-8931).  The set difference runs on the first-seen table: store hashes and version hashes are concatenated, a
// version chunk is missing iff its first occurrence in the concatenation is itself.  Block hashes come from the BLAKE3
// kernels.  Output: the bytes Longtail_WriteStoreIndexToBuffer would produce.
// ---------------------------------------------------------------------------------------------------
extern "C" int lthip_create_missing_content(lthip_ctx* ctx, uint64_t existing_count, const uint64_t* d_existing_hashes,
                                            uint64_t chunk_count, const uint64_t* d_chunk_hashes, const uint32_t* d_chunk_lens,
                                            const uint32_t* chunk_tags, uint32_t hash_identifier, uint32_t max_block_size,
                                            uint32_t max_chunks_per_block, void* out, size_t out_capacity, size_t* out_size)
{
    if (!ctx || !out_size || (existing_count && !d_existing_hashes) || (chunk_count && (!d_chunk_hashes || !d_chunk_lens)) ||
        max_chunks_per_block == 0)
        return EINVAL;
    if (existing_count + chunk_count > 0x7FFFFFF0ull)
        return lthip_fail(ctx, EINVAL, "missing content", "more than 2^31 hashes");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t ne = (size_t)existing_count, n = (size_t)chunk_count;
    int err;
    DevBuf d_all, d_first, d_uniq;
    if ((err = d_all.alloc(ctx, (ne + n) * 8)) || (err = d_first.alloc(ctx, (ne + n) * 4)) || (err = d_uniq.alloc(ctx, 8)))
        return err;
    if (ne)
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_all.p, d_existing_hashes, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (n)
        LTHIP_CHECK(ctx, hipMemcpyAsync((uint8_t*)d_all.p + ne * 8, d_chunk_hashes, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if ((err = lthip_dedup_first_seen(ctx, ne + n, (const uint64_t*)d_all.p, (uint32_t*)d_first.p, (uint64_t*)d_uniq.p)))
        return err;
    std::vector<uint32_t> first(n), lens(n);
    std::vector<uint64_t> hashes(n);
    if (n)
    {
        LTHIP_CHECK(ctx, hipMemcpyAsync(first.data(), (const uint32_t*)d_first.p + ne, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, hipMemcpyAsync(lens.data(), d_chunk_lens, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, hipMemcpyAsync(hashes.data(), d_chunk_hashes, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    // the missing chunks, version order
    std::vector<uint64_t> m_hash;
    std::vector<uint32_t> m_size, m_tag;
    for (size_t i = 0; i < n; ++i)
        if (first[i] == (uint32_t)(ne + i))
        {
            m_hash.push_back(hashes[i]);
            m_size.push_back(lens[i]);
            m_tag.push_back(chunk_tags ? chunk_tags[i] : 0u);
        }
    const size_t m = m_hash.size();
    // greedy packing, :6801-6860
    std::vector<uint32_t> b_off, b_cnt, b_tag;
    const uint64_t limit = (uint64_t)max_block_size + max_block_size / 10;
    for (size_t i = 0; i < m;)
    {
        uint64_t size = m_size[i];
        size_t j = i + 1;
        while (j < m && m_tag[j] == m_tag[i] && j - i < max_chunks_per_block && size + m_size[j] <= limit)
            size += m_size[j++];
        b_off.push_back((uint32_t)i);
        b_cnt.push_back((uint32_t)(j - i));
        b_tag.push_back(m_tag[i]);
        i = j;
    }
    const size_t nb = b_off.size();
    // block hashes = BLAKE3 of each block's chunk-hash array (:3753-3757)
    std::vector<uint64_t> b_hash(nb);
    if (nb)
    {
        DevBuf d_mh, d_o, d_l, d_bh;
        std::vector<uint64_t> o(nb);
        std::vector<uint32_t> l(nb);
        uint32_t max_len = 0;
        for (size_t b = 0; b < nb; ++b)
        {
            o[b] = (uint64_t)b_off[b] * 8u;
            l[b] = b_cnt[b] * 8u;
            max_len = l[b] > max_len ? l[b] : max_len;
        }
        if ((err = d_mh.alloc(ctx, m * 8)) || (err = d_o.alloc(ctx, nb * 8)) || (err = d_l.alloc(ctx, nb * 4)) ||
            (err = d_bh.alloc(ctx, nb * 8)))
            return err;
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_mh.p, m_hash.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_o.p, o.data(), nb * 8, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_l.p, l.data(), nb * 4, hipMemcpyHostToDevice, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        if ((err = lthip_hash_ranges(ctx, d_mh.p, nb, (const uint64_t*)d_o.p, (const uint32_t*)d_l.p, max_len, (uint64_t*)d_bh.p)))
            return err;
        LTHIP_CHECK(ctx, hipMemcpyAsync(b_hash.data(), d_bh.p, nb * 8, hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    }
    // Longtail_GetStoreIndexDataSize, :8913-8931
    const size_t size = 16 + nb * 8 + m * 8 + nb * 12 + m * 4;
    *out_size = size;
    if (!out || out_capacity < size)
        return ENOMEM;
    uint8_t* w = (uint8_t*)out;
    const uint32_t head[4] = {(1u << 24) /* LONGTAIL_STORE_INDEX_VERSION_1_0_0, :19-23 */, hash_identifier, (uint32_t)nb, (uint32_t)m};
    memcpy(w, head, 16);
    w += 16;
    memcpy(w, b_hash.data(), nb * 8);  // m_BlockHashes
    w += nb * 8;
    memcpy(w, m_hash.data(), m * 8);   // m_ChunkHashes
    w += m * 8;
    memcpy(w, b_off.data(), nb * 4);   // m_BlockChunksOffsets
    w += nb * 4;
    memcpy(w, b_cnt.data(), nb * 4);   // m_BlockChunkCounts
    w += nb * 4;
    memcpy(w, b_tag.data(), nb * 4);   // m_BlockTags
    w += nb * 4;
    memcpy(w, m_size.data(), m * 4);   // m_ChunkSizes
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// SURVEY.md §8 f4 (second half): Longtail_GetExistingStoreIndex (src/longtail.c:7087-7325) as a bulk call.
// Which blocks of a store cover the given chunk hashes?  The reference (1) counts the distinct wanted hashes, (2) measures every
// store block's usage = bytes of wanted chunks / block bytes and keeps the blocks at or above min_block_usage_percent, (3) sorts
// them by usage, most used first (qsort_r with a usage-only comparator: glibc's merge sort keeps equal usages in store order, and
// that is the order reproduced here), (4) walks them in that order: a block is taken when it holds a wanted chunk that no earlier
// block of the walk held, (5) lays the taken blocks out with Longtail_CreateStoreIndexFromBlocks (:9063-9125).
// On the device the serial walk (4) becomes order independent: every wanted chunk is claimed by the block of smallest RANK that
// holds it (atomicMin), and a block is taken iff it claims something.  The hash-set membership tests of (2) and the claims of (4)
// run over the store's chunk list in parallel; the sort of a few thousand blocks stays on the host.
// One quirk is mirrored, not fixed: the reference takes a taken block's tag from m_BlockTags[first chunk index] instead of
// m_BlockTags[block index] (:7307) -- an index that can run past the tag array into the chunk sizes that follow it.
// ---------------------------------------------------------------------------------------------------
namespace
{
constexpr uint64_t GE_EMPTY = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ uint64_t ge_mix(uint64_t z)
{
    z = (z ^ (z >> 33)) * 0xff51afd7ed558ccdull;
    z = (z ^ (z >> 33)) * 0xc4ceb9fe1a85ec53ull;
    return z ^ (z >> 33);
}

__global__ void k_ge_clear(uint64_t* __restrict__ keys, uint32_t* __restrict__ val, uint64_t slots, uint32_t* __restrict__ misc)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x)
    {
        keys[i] = GE_EMPTY;
        val[i] = 0xFFFFFFFFu;
    }
    if (blockIdx.x == 0 && threadIdx.x < 4)
        misc[threadIdx.x] = threadIdx.x == 1 ? 0xFFFFFFFFu : 0u; // [0] distinct wanted, [1] rank of the all-ones hash, [2] all-ones wanted
}

__global__ void k_ge_insert(const uint64_t* __restrict__ wanted, uint64_t n, uint64_t* __restrict__ keys, uint64_t mask,
                            uint32_t* __restrict__ misc)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t h = wanted[i];
    if (h == GE_EMPTY)
    {
        if (atomicExch(&misc[2], 1u) == 0u)
            atomicAdd(&misc[0], 1u);
        return;
    }
    uint64_t slot = ge_mix(h) & mask;
    for (;;)
    {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]), (unsigned long long)GE_EMPTY, (unsigned long long)h);
        if (prev == GE_EMPTY)
        {
            atomicAdd(&misc[0], 1u);
            return;
        }
        if (prev == h)
            return;
        slot = (slot + 1) & mask;
    }
}

// slot of a store chunk hash in the wanted set, or ~0 (the all-ones hash lives in misc)
__device__ __forceinline__ uint64_t ge_find(uint64_t h, const uint64_t* __restrict__ keys, uint64_t mask)
{
    uint64_t slot = ge_mix(h) & mask;
    for (;;)
    {
        const uint64_t k = keys[slot];
        if (k == h)
            return slot;
        if (k == GE_EMPTY)
            return ~0ull;
        slot = (slot + 1) & mask;
    }
}

// one wave per store block: bytes of the block, bytes of its wanted chunks
__global__ __launch_bounds__(64) void k_ge_usage(const uint32_t* __restrict__ b_off, const uint32_t* __restrict__ b_cnt, uint32_t nblocks,
                                                 const uint64_t* __restrict__ c_hash, const uint32_t* __restrict__ c_size,
                                                 const uint64_t* __restrict__ keys, uint64_t mask, const uint32_t* __restrict__ misc,
                                                 uint32_t* __restrict__ use, uint32_t* __restrict__ size)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks)
        return;
    const uint32_t o = b_off[b], n = b_cnt[b];
    uint32_t u = 0, s = 0; // uint32_t like the reference's block_use / block_size
    for (uint32_t i = threadIdx.x; i < n; i += 64)
    {
        const uint64_t h = c_hash[o + i];
        const uint32_t sz = c_size[o + i];
        s += sz;
        const bool in = h == GE_EMPTY ? misc[2] != 0u : ge_find(h, keys, mask) != ~0ull;
        u += in ? sz : 0u;
    }
    for (int d = 32; d > 0; d >>= 1)
    {
        u += __shfl_down(u, d, 64);
        s += __shfl_down(s, d, 64);
    }
    if (threadIdx.x == 0)
    {
        use[b] = u;
        size[b] = s;
    }
}

// every wanted chunk of a ranked block: claimed by the smallest rank
__global__ __launch_bounds__(64) void k_ge_claim(const uint32_t* __restrict__ b_off, const uint32_t* __restrict__ b_cnt, uint32_t nblocks,
                                                 const uint32_t* __restrict__ rank, const uint64_t* __restrict__ c_hash,
                                                 const uint64_t* __restrict__ keys, uint32_t* __restrict__ val, uint64_t mask,
                                                 uint32_t* __restrict__ misc)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks || rank[b] == 0xFFFFFFFFu)
        return;
    const uint32_t o = b_off[b], n = b_cnt[b], r = rank[b];
    for (uint32_t i = threadIdx.x; i < n; i += 64)
    {
        const uint64_t h = c_hash[o + i];
        if (h == GE_EMPTY)
        {
            if (misc[2] != 0u)
                atomicMin(&misc[1], r);
            continue;
        }
        const uint64_t slot = ge_find(h, keys, mask);
        if (slot != ~0ull)
            atomicMin(&val[slot], r);
    }
}

__global__ __launch_bounds__(64) void k_ge_taken(const uint32_t* __restrict__ b_off, const uint32_t* __restrict__ b_cnt, uint32_t nblocks,
                                                 const uint32_t* __restrict__ rank, const uint64_t* __restrict__ c_hash,
                                                 const uint64_t* __restrict__ keys, const uint32_t* __restrict__ val, uint64_t mask,
                                                 const uint32_t* __restrict__ misc, uint32_t* __restrict__ taken)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks)
        return;
    const uint32_t o = b_off[b], n = b_cnt[b], r = rank[b];
    bool mine = false;
    if (r != 0xFFFFFFFFu)
        for (uint32_t i = threadIdx.x; i < n; i += 64)
        {
            const uint64_t h = c_hash[o + i];
            if (h == GE_EMPTY)
                mine |= misc[2] != 0u && misc[1] == r;
            else
            {
                const uint64_t slot = ge_find(h, keys, mask);
                mine |= slot != ~0ull && val[slot] == r;
            }
        }
    const uint64_t any = __builtin_amdgcn_ballot_w64(mine);
    if (threadIdx.x == 0)
        taken[b] = any != 0ull ? 1u : 0u;
}
} // namespace

extern "C" int lthip_get_existing_store_index(lthip_ctx* ctx, const void* store_index, size_t store_index_size, uint64_t chunk_count,
                                              const uint64_t* d_chunk_hashes, uint32_t min_block_usage_percent, void* out, size_t out_capacity,
                                              size_t* out_size)
{
    if (!ctx || !store_index || !out_size || (chunk_count && !d_chunk_hashes))
        return EINVAL;
    if (store_index_size < 16)
        return lthip_fail(ctx, EBADF, "existing store index", "store index shorter than its header");
    const uint8_t* r = (const uint8_t*)store_index;
    uint32_t head[4];
    memcpy(head, r, 16);
    const uint32_t nb = head[2], m = head[3];
    if (head[0] != (1u << 24)) // LONGTAIL_STORE_INDEX_VERSION_1_0_0 (src/longtail.c:19-23; InitStoreIndexFromData rejects others)
        return lthip_fail(ctx, EBADF, "existing store index", "unsupported store index version");
    const size_t need = 16 + (size_t)nb * 8 + (size_t)m * 8 + (size_t)nb * 12 + (size_t)m * 4;
    if (store_index_size < need || chunk_count > 0x7FFFFFF0ull)
        return lthip_fail(ctx, EBADF, "existing store index", "store index truncated");
    // unaligned-safe views (the serialized index is only 4-byte aligned after its 16-byte header)
    const uint8_t* p_bhash = r + 16;
    const uint8_t* p_chash = p_bhash + (size_t)nb * 8;
    const uint8_t* p_boff = p_chash + (size_t)m * 8;
    const uint8_t* p_bcnt = p_boff + (size_t)nb * 4;
    const uint8_t* p_btag = p_bcnt + (size_t)nb * 4;
    const uint8_t* p_csize = p_btag + (size_t)nb * 4;
    std::vector<uint32_t> b_off(nb), b_cnt(nb);
    if (nb)
    {
        memcpy(b_off.data(), p_boff, (size_t)nb * 4);
        memcpy(b_cnt.data(), p_bcnt, (size_t)nb * 4);
    }
    for (uint32_t b = 0; b < nb; ++b)
        if ((uint64_t)b_off[b] + b_cnt[b] > m)
            return lthip_fail(ctx, EBADF, "existing store index", "block chunk range outside the chunk list");
    auto write_empty = [&]() -> int {
        // Longtail_CreateStoreIndexFromBlocks(0, 0): hash identifier 0, no blocks, no chunks
        *out_size = 16;
        if (!out || out_capacity < 16)
            return ENOMEM;
        const uint32_t e[4] = {1u << 24, 0u, 0u, 0u};
        memcpy(out, e, 16);
        return 0;
    };
    if (nb == 0 || min_block_usage_percent > 100)
        return write_empty();

    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    uint64_t slots = 1024;
    while (slots < chunk_count * 2)
        slots <<= 1;
    DevBuf d_keys, d_val, d_misc, d_chash, d_csize, d_boff, d_bcnt, d_use, d_size, d_rank, d_taken;
    int err;
    if ((err = d_keys.alloc(ctx, slots * 8)) || (err = d_val.alloc(ctx, slots * 4)) || (err = d_misc.alloc(ctx, 16)) ||
        (err = d_chash.alloc(ctx, (size_t)m * 8)) || (err = d_csize.alloc(ctx, (size_t)m * 4)) || (err = d_boff.alloc(ctx, (size_t)nb * 4)) ||
        (err = d_bcnt.alloc(ctx, (size_t)nb * 4)) || (err = d_use.alloc(ctx, (size_t)nb * 4)) || (err = d_size.alloc(ctx, (size_t)nb * 4)) ||
        (err = d_rank.alloc(ctx, (size_t)nb * 4)) || (err = d_taken.alloc(ctx, (size_t)nb * 4)))
        return err;
    if (m)
    {
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_chash.p, p_chash, (size_t)m * 8, hipMemcpyHostToDevice, s));
        LTHIP_CHECK(ctx, hipMemcpyAsync(d_csize.p, p_csize, (size_t)m * 4, hipMemcpyHostToDevice, s));
    }
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_boff.p, b_off.data(), (size_t)nb * 4, hipMemcpyHostToDevice, s));
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_bcnt.p, b_cnt.data(), (size_t)nb * 4, hipMemcpyHostToDevice, s));
    const uint64_t mask = slots - 1;
    hipLaunchKernelGGL(k_ge_clear, dim3(1024), dim3(256), 0, s, (uint64_t*)d_keys.p, (uint32_t*)d_val.p, slots, (uint32_t*)d_misc.p);
    if (chunk_count)
        hipLaunchKernelGGL(k_ge_insert, dim3((uint32_t)div_up_u64(chunk_count, 256)), dim3(256), 0, s, d_chunk_hashes, chunk_count,
                           (uint64_t*)d_keys.p, mask, (uint32_t*)d_misc.p);
    hipLaunchKernelGGL(k_ge_usage, dim3(nb), dim3(64), 0, s, (const uint32_t*)d_boff.p, (const uint32_t*)d_bcnt.p, nb, (const uint64_t*)d_chash.p,
                       (const uint32_t*)d_csize.p, (const uint64_t*)d_keys.p, mask, (const uint32_t*)d_misc.p, (uint32_t*)d_use.p,
                       (uint32_t*)d_size.p);
    LTHIP_LAUNCH_CHECK(ctx);
    std::vector<uint32_t> use(nb), size(nb);
    LTHIP_CHECK(ctx, hipMemcpyAsync(use.data(), d_use.p, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
    LTHIP_CHECK(ctx, hipMemcpyAsync(size.data(), d_size.p, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
    LTHIP_CHECK(ctx, hipStreamSynchronize(s));
    // (2)+(3): the potential blocks, most used first, equal usages in store order
    std::vector<uint32_t> order, pct(nb, 0);
    for (uint32_t b = 0; b < nb; ++b)
        if (use[b] > 0)
        {
            pct[b] = (uint32_t)(((uint64_t)use[b] * 100) / size[b]);
            if (min_block_usage_percent > 0 && pct[b] < min_block_usage_percent)
                continue;
            order.push_back(b);
        }
    if (order.empty())
        return write_empty();
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pct[a] > pct[b]; });
    std::vector<uint32_t> rank(nb, 0xFFFFFFFFu);
    for (size_t i = 0; i < order.size(); ++i)
        rank[order[i]] = (uint32_t)i;
    LTHIP_CHECK(ctx, hipMemcpyAsync(d_rank.p, rank.data(), (size_t)nb * 4, hipMemcpyHostToDevice, s));
    // (4): claims
    hipLaunchKernelGGL(k_ge_claim, dim3(nb), dim3(64), 0, s, (const uint32_t*)d_boff.p, (const uint32_t*)d_bcnt.p, nb, (const uint32_t*)d_rank.p,
                       (const uint64_t*)d_chash.p, (const uint64_t*)d_keys.p, (uint32_t*)d_val.p, mask, (uint32_t*)d_misc.p);
    hipLaunchKernelGGL(k_ge_taken, dim3(nb), dim3(64), 0, s, (const uint32_t*)d_boff.p, (const uint32_t*)d_bcnt.p, nb, (const uint32_t*)d_rank.p,
                       (const uint64_t*)d_chash.p, (const uint64_t*)d_keys.p, (const uint32_t*)d_val.p, mask, (const uint32_t*)d_misc.p,
                       (uint32_t*)d_taken.p);
    LTHIP_LAUNCH_CHECK(ctx);
    std::vector<uint32_t> taken(nb);
    LTHIP_CHECK(ctx, hipMemcpyAsync(taken.data(), d_taken.p, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
    LTHIP_CHECK(ctx, hipStreamSynchronize(s));
    // (5): the taken blocks in walk order; a block hash that occurs twice in the store is taken once (block_to_index_lookup, :7246)
    std::vector<uint32_t> found;
    {
        std::unordered_set<uint64_t> seen; // one hash look-up per taken block, like the reference's table (:7246)
        seen.reserve(nb);
        for (uint32_t b : order)
            if (taken[b])
            {
                uint64_t bh;
                memcpy(&bh, p_bhash + (size_t)b * 8, 8);
                if (!seen.insert(bh).second)
                    continue;
                found.push_back(b);
            }
    }
    if (found.empty())
        return write_empty();
    size_t fm = 0;
    for (uint32_t b : found)
        fm += b_cnt[b];
    const size_t fb = found.size();
    const size_t osize = 16 + fb * 8 + fm * 8 + fb * 12 + fm * 4;
    *out_size = osize;
    if (!out || out_capacity < osize)
        return ENOMEM;
    uint8_t* w = (uint8_t*)out;
    const uint32_t ohead[4] = {1u << 24, head[1], (uint32_t)fb, (uint32_t)fm};
    memcpy(w, ohead, 16);
    uint8_t* o_bhash = w + 16;
    uint8_t* o_chash = o_bhash + fb * 8;
    uint8_t* o_boff = o_chash + fm * 8;
    uint8_t* o_bcnt = o_boff + fb * 4;
    uint8_t* o_btag = o_bcnt + fb * 4;
    uint8_t* o_csize = o_btag + fb * 4;
    uint32_t c = 0;
    for (size_t i = 0; i < fb; ++i)
    {
        const uint32_t b = found[i], n = b_cnt[b], off = b_off[b];
        memcpy(o_bhash + i * 8, p_bhash + (size_t)b * 8, 8);
        // the reference's tag quirk: m_BlockTags[first chunk index] (:7307), read where that index lands in the serialized arrays
        uint32_t tag = 0;
        if ((size_t)(p_btag - r) + (size_t)off * 4 + 4 <= store_index_size)
            memcpy(&tag, p_btag + (size_t)off * 4, 4);
        memcpy(o_btag + i * 4, &tag, 4);
        memcpy(o_bcnt + i * 4, &n, 4);
        memcpy(o_boff + i * 4, &c, 4);
        memcpy(o_chash + (size_t)c * 8, p_chash + (size_t)off * 8, (size_t)n * 8);
        memcpy(o_csize + (size_t)c * 4, p_csize + (size_t)off * 4, (size_t)n * 4);
        c += n;
    }
    return 0;
}
