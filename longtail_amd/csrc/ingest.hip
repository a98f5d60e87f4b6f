// ingest.hip -- the ingest metric of SURVEY.md §8(d) as ONE native session over device-resident assets:
//
//     Longtail_CreateVersionIndex   (src/longtail.c:2808)   chunk + hash (lthip_chunk_hash) -> [multi-GPU: exchange] ->
//                                                           first-seen dedup, content / path hashes, serialized VersionIndex
//     Longtail_CreateMissingContent (src/longtail.c:6882)   the chunks this rank saw first, packed into blocks
//                                                           (Longtail_CreateStoreIndex :6745-6880), block hashes, serialized StoreIndex
//     Longtail_WriteContent         (src/longtail.c:4760)   block assembly (device gather only where a block is not one byte
//                                                           range), per-block LZ4 / ZStd straight into the stored-block image,
//                                                           BlockIndex + [raw][compressed] around it (:4111-4150) -- a null sink:
//                                                           images are produced in a bounded device arena and dropped
//
// Everything between the phases stays on the device; the host sees the unique chunks' lengths and offsets once (the greedy
// packing is serial in the reference too) and does its serial work while the GPU is busy with work that does not depend on
// it: the VersionIndex sections are hashed and copied out while the host packs blocks, the StoreIndex is laid out while the
// codec runs.  No allocation in steady state: all workspaces are grown once and kept.
//
// Multi-GPU: the session is given ALL ranks' chunk hashes / lengths in job order (dist.exchange_chunks) plus
// the list of its own jobs.  Every rank derives the same first-seen table; a rank writes the chunks that are first-seen AND
// lie in its own jobs -- exactly Longtail_CreateMissingContent against a store that already holds the other ranks' chunks.
#include "lthip_internal.h"
#include "index_kernels.h"

#include <algorithm>
#include <chrono>
#include <thread>

namespace
{

struct DBuf
{
    void* p = nullptr;
    size_t cap = 0;
};
struct HBuf
{
    void* p = nullptr;
    size_t cap = 0;
};

int reserve_dev(lthip_ctx* ctx, DBuf& b, size_t bytes)
{
    if (bytes == 0)
        bytes = 256;
    if (b.cap >= bytes)
        return 0;
    if (b.p)
    {
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        LTHIP_CHECK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t cap = bytes + bytes / 8 + 4096;
    LTHIP_CHECK(ctx, lthip_hip_malloc(&b.p, cap));
    b.cap = cap;
    return 0;
}

int reserve_pinned(lthip_ctx* ctx, HBuf& b, size_t bytes)
{
    if (bytes == 0)
        bytes = 256;
    if (b.cap >= bytes)
        return 0;
    if (b.p)
    {
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        LTHIP_CHECK(ctx, hipHostFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t cap = bytes + bytes / 8 + 4096;
    LTHIP_CHECK(ctx, lthip_hip_host_malloc(&b.p, cap, hipHostMallocDefault));
    b.cap = cap;
    return 0;
}

// local chunk k of this rank -> its index in the job-ordered arrays of all ranks, and "this rank writes it":
// owned[k] = first_index[g(k)] == g(k).  part_first = the rank's own chunk-list starts (lthip_chunk_hash), one part per own job,
// job_gfirst[m] = index of own job m's first chunk in the global arrays.  With job_gfirst == null the arrays are the same.
__global__ void k_ing_owned(const uint32_t* __restrict__ first_index, const uint32_t* __restrict__ part_first, uint32_t nparts,
                            const uint32_t* __restrict__ job_gfirst, uint32_t nlocal, uint32_t* __restrict__ owned,
                            uint32_t* __restrict__ l2g)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nlocal)
        return;
    uint32_t g = k;
    if (job_gfirst)
    {
        uint32_t lo = 0, hi = nparts; // part m with part_first[m] <= k < part_first[m + 1] (empty parts: take the last such)
        while (hi - lo > 1)
        {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (part_first[mid] <= k)
                lo = mid;
            else
                hi = mid;
        }
        g = job_gfirst[lo] + (k - part_first[lo]);
    }
    l2g[k] = g;
    owned[k] = first_index[g] == g ? 1u : 0u;
}

// compaction of the owned chunks: hash, length, byte offset in the rank's data and the tag of the chunk's asset
__global__ void k_ing_compact(const uint32_t* __restrict__ owned, const uint32_t* __restrict__ orank, const uint32_t* __restrict__ l2g,
                              uint32_t nlocal, const uint64_t* __restrict__ all_hashes, const uint32_t* __restrict__ all_lens,
                              const uint64_t* __restrict__ local_offsets, const uint32_t* __restrict__ asset_first_chunk,
                              uint32_t asset_count, const uint32_t* __restrict__ asset_tags, uint64_t* __restrict__ u_hash,
                              uint32_t* __restrict__ u_len, uint64_t* __restrict__ u_off, uint32_t* __restrict__ u_tag)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nlocal || !owned[k])
        return;
    const uint32_t r = orank[k], g = l2g[k];
    u_hash[r] = all_hashes[g];
    u_len[r] = all_lens[g];
    u_off[r] = local_offsets[k];
    if (asset_tags)
    {
        uint32_t lo = 0, hi = asset_count;
        while (hi - lo > 1)
        {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (asset_first_chunk[mid] <= g)
                lo = mid;
            else
                hi = mid;
        }
        u_tag[r] = asset_tags[lo];
    }
}

// brk[i] = chunk i does not continue the byte range of chunk i - 1 (what the packing loop needs of the offsets: a byte instead of 8)
__global__ void k_ing_breaks(const uint64_t* __restrict__ off, const uint32_t* __restrict__ len, uint32_t n, uint8_t* __restrict__ brk)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        brk[i] = i != 0u && off[i] != off[i - 1u] + len[i - 1u] ? 1 : 0;
}

__global__ void k_ing_sum_u32(const uint32_t* __restrict__ v, uint32_t n, unsigned long long* __restrict__ out)
{
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        acc += v[i];
    for (int o = 32; o > 0; o >>= 1)
        acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc)
        atomicAdd(out, acc);
}

} // namespace

struct lthip_ingest
{
    lthip_ctx* ctx;
    lthip_ingest_config cfg;
    // ---- index phase ----
    DBuf d_first, d_isfirst, d_rank, d_idx, d_uh, d_us, d_ut, d_starts, d_tags, d_counts, d_paths, d_aoff, d_alen, d_ph, d_ch;
    DBuf d_gfirst, d_owned, d_orank, d_l2g, d_mu_hash, d_mu_len, d_mu_off, d_mu_tag;
    HBuf h_counts, h_mu_len, h_mu_off, h_mu_hash, h_mu_tag, h_bhash, h_comp, h_brk;
    DBuf d_brk;
    DBuf d_bhash, d_boff, d_blen, d_comp, d_sum;
    DBuf d_gather, d_gsrc, d_glen, d_gdst, d_bfirst, d_braw, d_bimg, d_btag, d_tmpsz;
    hipEvent_t ev_counts, ev_lens, ev_offs, ev_index;
    // the first-seen index of every chunk computed elsewhere (the sharded table of the multi-GPU path): consumed by the next
    // lthip_ingest_index instead of its own table pass
    const uint32_t* ext_first;
    uint64_t ext_unique;
    // host state between the phases
    uint64_t n_all, n_local, unique_all, n_mine;
    std::vector<uint64_t> b_first;   // nb + 1 chunk indices into the owned-unique list
    std::vector<uint64_t> b_size;    // raw bytes
    std::vector<uint8_t> b_is_range; // the block's chunks are one byte range of the rank's data
    std::vector<uint32_t> b_tag;
    bool has_tags;
    size_t vi_size;
    bool indexed, written;
    // the packing runs in slices (ingest_pack): lthip_ingest_index packs what the first codec batch takes, lthip_ingest_write the rest
    // once that batch is queued
    uint32_t pack_next;   // first owned chunk that is in no block yet
    uint32_t pack_avail;  // owned chunks whose lengths / break flags / tags / offsets are on the host
    uint64_t pack_raw;    // raw bytes of the blocks so far
    bool blocks_done;     // all blocks packed and hashed, their hashes on the way to the host (ev_index)
    hipEvent_t ev_hashes; // the owned chunks' hashes are on the host (side stream)
    lthip_ingest_result res;
    // the VersionIndex sections are put together by a helper thread on a context of its own (stream, staging ring, BLAKE3 scratch), so
    // that the calling thread keeps the session's context to itself and does not wait for it before lthip_ingest_finish
    lthip_ctx* vi_ctx;
    std::thread vi_thread;
    int vi_err;
    bool vi_pending; // prepared by lthip_ingest_index, not started yet
    // what the VersionIndex helper reads of the caller's tree AFTER lthip_ingest_index has returned: a deep copy (O(assets): sizes, path
    // offsets, permissions, path data), so that a caller may free or reuse its lthip_ingest_tree arrays as soon as the call returns --
    // the contract of round 2.  Only the device arrays and the output buffer live until lthip_ingest_finish (include/longtail_hip.h).
    lthip_ingest_tree vi_tree;
    std::vector<uint64_t> vi_asset_sizes;
    std::vector<uint32_t> vi_path_offsets;
    std::vector<uint16_t> vi_permissions;
    std::vector<char> vi_path_data;
    std::vector<uint32_t> vi_starts, vi_counts;
    const uint64_t* vi_hashes;
    void* vi_out;
    // the stored-block images of the last codec batch (lthip_ingest_images): first block, offsets in the arena, header sizes; the image
    // sizes are completed by lthip_ingest_finish (they need the compressed sizes)
    uint64_t img_first;
    std::vector<uint64_t> img_offsets;
    std::vector<uint32_t> img_sizes; // header + payload per image: computed by every lthip_ingest_finish from img_hdr (calling it twice adds nothing twice)
    std::vector<uint32_t> img_hdr;   // BlockIndex + [raw][compressed] of the last batch's images
};

// LTHIP_INGEST_TRACE=1: host time between the marks of lthip_ingest_index / _write, to stderr
struct IngTrace
{
    bool on;
    const char* what;
    std::chrono::steady_clock::time_point t0, last;
    char line[512];
    size_t len;
    explicit IngTrace(const char* w) : what(w), len(0)
    {
        LTHIP_ABLATION_ENV(env, "LTHIP_INGEST_TRACE");
        on = env.get() > 0;
        if (on)
            t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char* name)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        if (len < sizeof line - 48)
            len += (size_t)snprintf(line + len, sizeof line - len, " %s %.0f", name, std::chrono::duration<double, std::micro>(now - last).count());
        last = now;
    }
    ~IngTrace()
    {
        if (on)
            fprintf(stderr, "%s (us):%s | total %.0f\n", what, line, std::chrono::duration<double, std::micro>(last - t0).count());
    }
};

static size_t codec_bound(const lthip_ingest* g, size_t n)
{
    if (g->cfg.codec == LTHIP_CODEC_LZ4)
        return lthip_lz4_bound(n);
    if (g->cfg.codec == LTHIP_CODEC_ZSTD)
        return lthip_zstd_bound(n);
    return n;
}

extern "C" int lthip_ingest_create(lthip_ctx* ctx, const lthip_ingest_config* cfg, lthip_ingest** out)
{
    if (!ctx || !cfg || !out)
        return EINVAL;
    *out = nullptr;
    if (cfg->max_block_size == 0 || cfg->max_chunks_per_block == 0 || cfg->codec > LTHIP_CODEC_ZSTD)
        return lthip_fail(ctx, EINVAL, "lthip_ingest_create", "bad block / codec parameters");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    lthip_ingest* g = new (std::nothrow) lthip_ingest();
    if (!g)
        return ENOMEM;
    g->ctx = ctx;
    g->cfg = *cfg;
    if (g->cfg.batch_bytes == 0)
        g->cfg.batch_bytes = 8ull << 30;
    g->indexed = g->written = false;
    g->vi_ctx = nullptr;
    g->vi_err = 0;
    g->vi_pending = false;
    g->ev_counts = g->ev_lens = g->ev_index = g->ev_offs = g->ev_hashes = nullptr;
    g->blocks_done = false;
    if (hipEventCreateWithFlags(&g->ev_counts, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_lens, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_offs, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_index, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_hashes, hipEventDisableTiming) != hipSuccess)
    {
        lthip_ingest_destroy(g);
        return lthip_fail(ctx, EIO, "lthip_ingest_create", "hipEventCreate");
    }
    *out = g;
    return 0;
}

static int ingest_vi_join(lthip_ingest* g);

extern "C" void lthip_ingest_destroy(lthip_ingest* g)
{
    if (!g)
        return;
    (void)hipSetDevice(g->ctx->device);
    (void)ingest_vi_join(g);
    if (g->vi_ctx)
        lthip_ctx_destroy(g->vi_ctx);
    (void)hipStreamSynchronize(g->ctx->stream);
    DBuf* dev[] = {&g->d_first, &g->d_isfirst, &g->d_rank, &g->d_idx, &g->d_uh, &g->d_us, &g->d_ut, &g->d_starts, &g->d_tags, &g->d_counts,
                   &g->d_paths, &g->d_aoff, &g->d_alen, &g->d_ph, &g->d_ch, &g->d_gfirst, &g->d_owned, &g->d_orank, &g->d_l2g, &g->d_mu_hash,
                   &g->d_mu_len, &g->d_mu_off, &g->d_mu_tag, &g->d_bhash, &g->d_boff, &g->d_blen, &g->d_comp, &g->d_sum, &g->d_gather,
                   &g->d_gsrc, &g->d_glen, &g->d_gdst, &g->d_bfirst, &g->d_braw, &g->d_bimg, &g->d_btag, &g->d_tmpsz, &g->d_brk};
    for (DBuf* b : dev)
        if (b->p)
            (void)hipFree(b->p);
    HBuf* pin[] = {&g->h_counts, &g->h_mu_len, &g->h_mu_off, &g->h_mu_hash, &g->h_mu_tag, &g->h_bhash, &g->h_comp, &g->h_brk};
    for (HBuf* b : pin)
        if (b->p)
            (void)hipHostFree(b->p);
    if (g->ev_counts)
        (void)hipEventDestroy(g->ev_counts);
    if (g->ev_lens)
        (void)hipEventDestroy(g->ev_lens);
    if (g->ev_offs)
        (void)hipEventDestroy(g->ev_offs);
    if (g->ev_index)
        (void)hipEventDestroy(g->ev_index);
    if (g->ev_hashes)
    {
        (void)hipEventSynchronize(g->ev_hashes);
        (void)hipEventDestroy(g->ev_hashes);
    }
    delete g;
}

// ---------------------------------------------------------------------------------------------------------------------
// phase 2: the tail of CreateVersionIndex + CreateMissingContent
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int lthip_ingest_set_first_seen(lthip_ingest* g, const uint32_t* d_first_index, uint64_t unique_chunks)
{
    if (!g)
        return EINVAL;
    g->ext_first = d_first_index;
    g->ext_unique = unique_chunks;
    return 0;
}

// Greedy packing of the owned chunks into blocks (Longtail_CreateStoreIndex :6801-6860), serial like the reference's, continued from
// where it stopped: until the new blocks hold `raw_budget` bytes or the chunks on the host (pack_avail) run out.  A block is only
// closed when the chunk that does not fit any more has been seen (or there is none).
static void ingest_pack(lthip_ingest* g, uint64_t raw_budget)
{
    const uint32_t nm = (uint32_t)g->n_mine, avail = g->pack_avail;
    const uint32_t* lens = (const uint32_t*)g->h_mu_len.p;
    const uint8_t* brk = (const uint8_t*)g->h_brk.p;
    const uint32_t* tags = g->has_tags ? (const uint32_t*)g->h_mu_tag.p : nullptr;
    const uint64_t limit = (uint64_t)g->cfg.max_block_size + g->cfg.max_block_size / 10;
    const uint32_t max_chunks = g->cfg.max_chunks_per_block;
    uint64_t added = 0;
    uint32_t i = g->pack_next;
    while (i < avail && added < raw_budget)
    {
        uint64_t size = lens[i];
        uint32_t j = i + 1;
        bool range = true;
        const uint32_t tag = tags ? tags[i] : g->cfg.compression_type;
        while (j < avail && j - i < max_chunks && (!tags || tags[j] == tag) && size + lens[j] <= limit)
        {
            range &= brk[j] == 0;
            size += lens[j];
            ++j;
        }
        if (j == avail && avail < nm && j - i < max_chunks)
            break; // the next chunk may still belong to this block
        g->b_size.push_back(size);
        g->b_first.push_back(j); // (b_first[b + 1]: where block b ends)
        g->b_is_range.push_back(range ? 1 : 0);
        g->b_tag.push_back(tag);
        added += size;
        i = j;
    }
    g->pack_next = i;
    g->pack_raw += added;
}

// The rest of the packing, then the block hashes = BLAKE3 of each block's chunk-hash array (:3753-3757) on their way to the host.
// lthip_ingest_write calls this behind the launches of its first batch (the codec runs while the host packs), lthip_ingest_finish when
// nothing was written.
static int ingest_blocks_done(lthip_ingest* g)
{
    if (g->blocks_done)
        return 0;
    lthip_ctx* ctx = g->ctx;
    hipStream_t s = ctx->stream;
    int err;
    if (g->pack_avail < g->n_mine)
    {
        LTHIP_CHECK(ctx, hipEventSynchronize(g->ev_offs));
        g->pack_avail = (uint32_t)g->n_mine;
    }
    ingest_pack(g, ~0ull);
    const size_t nb = g->b_size.size();
    g->res.blocks = nb;
    g->res.raw_bytes = g->pack_raw;
    if ((err = reserve_pinned(ctx, g->h_bhash, nb * 8)) || (err = reserve_pinned(ctx, g->h_comp, nb * 4 + 8)))
        return err;
    if (nb)
    {
        std::vector<uint64_t> o(nb);
        std::vector<uint32_t> l(nb);
        uint32_t max_len = 0;
        uint64_t leaves = 0; // (1 KiB leaves of the hash arrays: known here, so the hash launcher reads nothing back -- the stream holds
                             // the first codec batch by now)
        for (size_t b = 0; b < nb; ++b)
        {
            o[b] = g->b_first[b] * 8u;
            l[b] = (uint32_t)(g->b_first[b + 1] - g->b_first[b]) * 8u;
            max_len = std::max(max_len, l[b]);
            leaves += l[b] ? (l[b] + 1023u) >> 10 : 1u;
        }
        if ((err = lthip_stage_upload(ctx, g->d_boff.p, o.data(), nb * 8, s)) || (err = lthip_stage_upload(ctx, g->d_blen.p, l.data(), nb * 4, s)))
            return err;
        if ((err = lthip_hash_ranges_known(ctx, g->d_mu_hash.p, nb, (const uint64_t*)g->d_boff.p, (const uint32_t*)g->d_blen.p, max_len, leaves,
                                           (uint64_t*)g->d_bhash.p)))
            return err;
        LTHIP_CHECK(ctx, hipMemcpyAsync(g->h_bhash.p, g->d_bhash.p, nb * 8, hipMemcpyDeviceToHost, s));
    }
    LTHIP_CHECK(ctx, hipEventRecord(g->ev_index, s));
    g->blocks_done = true;
    return 0;
}

// The serialized VersionIndex (Longtail_BuildVersionIndex :2757-2806 over InitVersionIndexFromData's section order) into the caller's
// buffer: content hash of every asset = BLAKE3 of its chunk-hash array (:2518-2537), path hashes (:1269-1300), the sections copied or
// written.  `ctx`: the context whose stream, staging ring and scratch it may use -- the helper thread's own, or the session's.
static int ingest_vi_work(lthip_ingest* g, lthip_ctx* ctx)
{
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const lthip_ingest_tree* t = &g->vi_tree;
    const std::vector<uint32_t>&starts = g->vi_starts, &counts = g->vi_counts;
    const uint32_t n = (uint32_t)g->n_all, na = t->asset_count;
    const uint64_t unique = g->unique_all;
    const uint64_t* d_all_hashes = g->vi_hashes;
    void* h_version_index = g->vi_out;
    int err = 0;
    if (ctx != g->ctx)
        LTHIP_CHECK(ctx, hipStreamWaitEvent(s, g->ev_lens, 0)); // (recorded behind the first-seen pass on the session's stream)
    // content hash of every asset = BLAKE3 of its chunk-hash array (:2518-2537); path hashes (:1269-1300)
    std::vector<uint64_t> h_off(na);
    std::vector<uint32_t> h_len(na);
    uint32_t max_len = 0;
    for (uint32_t a = 0; a < na; ++a)
    {
        if ((uint64_t)counts[a] * 8u > 0xFFFFFFFFull)
            return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "asset with more than 2^29 chunks");
        h_off[a] = (uint64_t)starts[a] * 8u;
        h_len[a] = counts[a] * 8u;
        max_len = std::max(max_len, h_len[a]);
    }
    if (na)
    {
        if ((err = lthip_stage_upload(ctx, g->d_aoff.p, h_off.data(), (size_t)na * 8, s)) ||
            (err = lthip_stage_upload(ctx, g->d_alen.p, h_len.data(), (size_t)na * 4, s)))
            return err;
        if ((err = lthip_hash_ranges(ctx, n ? (const void*)d_all_hashes : g->d_paths.p, na, (const uint64_t*)g->d_aoff.p,
                                     (const uint32_t*)g->d_alen.p, max_len, (uint64_t*)g->d_ch.p)))
            return err;
        max_len = 0;
        for (uint32_t a = 0; a < na; ++a)
        {
            if (t->path_start_offsets[a] >= t->path_data_size)
                return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "path offset outside the path data");
            h_off[a] = t->path_start_offsets[a];
            h_len[a] = (uint32_t)strnlen(t->path_data + t->path_start_offsets[a], t->path_data_size - t->path_start_offsets[a]);
            max_len = std::max(max_len, h_len[a]);
        }
        // the staging ring holds 8 uploads: the offset / length tables of the content hashes were consumed by a kernel
        // queued before these, and the uploads are ordered on the stream, so reusing d_aoff / d_alen is safe
        if ((err = lthip_stage_upload(ctx, g->d_paths.p, t->path_data, t->path_data_size, s)) ||
            (err = lthip_stage_upload(ctx, g->d_aoff.p, h_off.data(), (size_t)na * 8, s)) ||
            (err = lthip_stage_upload(ctx, g->d_alen.p, h_len.data(), (size_t)na * 4, s)))
            return err;
        if ((err = lthip_hash_ranges(ctx, g->d_paths.p, na, (const uint64_t*)g->d_aoff.p, (const uint32_t*)g->d_alen.p, max_len,
                                     (uint64_t*)g->d_ph.p)))
            return err;
    }
    // serialized layout (Longtail_BuildVersionIndex :2757-2806 over InitVersionIndexFromData's section order)
    uint8_t* w = (uint8_t*)h_version_index;
    const uint32_t head[6] = {2u /* LONGTAIL_VERSION_INDEX_VERSION_0_0_2, :16-22 */, g->cfg.hash_identifier, g->cfg.target_chunk_size, na,
                              (uint32_t)unique, n};
    memcpy(w, head, sizeof head);
    w += sizeof head;
    #define LT_D2H(SRC, BYTES)                                                                    \
do                                                                                        \
{                                                                                         \
    if (BYTES)                                                                            \
        LTHIP_CHECK(ctx, hipMemcpyAsync(w, (SRC), (BYTES), hipMemcpyDeviceToHost, s));    \
    w += (BYTES);                                                                         \
} while (0)
    LT_D2H(g->d_ph.p, (size_t)na * 8);             // m_PathHashes
    LT_D2H(g->d_ch.p, (size_t)na * 8);             // m_ContentHashes
    memcpy(w, t->asset_sizes, (size_t)na * 8);     // m_AssetSizes
    w += (size_t)na * 8;
    memcpy(w, counts.data(), (size_t)na * 4);      // m_AssetChunkCounts
    w += (size_t)na * 4;
    memcpy(w, starts.data(), (size_t)na * 4);      // m_AssetChunkIndexStarts
    w += (size_t)na * 4;
    LT_D2H(g->d_idx.p, (size_t)n * 4);             // m_AssetChunkIndexes
    LT_D2H(g->d_uh.p, (size_t)unique * 8);         // m_ChunkHashes
    LT_D2H(g->d_us.p, (size_t)unique * 4);         // m_ChunkSizes
    if (g->has_tags)
        LT_D2H(g->d_ut.p, (size_t)unique * 4);     // m_ChunkTags
    else
    {
        uint32_t* tg = (uint32_t*)w;               // one tag for the whole tree (what UpSync passes, cmd/main.c:1038-1046)
        for (uint64_t i = 0; i < unique; ++i)
            tg[i] = g->cfg.compression_type;
        w += (size_t)unique * 4;
    }
    #undef LT_D2H
    memcpy(w, t->path_start_offsets, (size_t)na * 4); // m_NameOffsets
    w += (size_t)na * 4;
    memcpy(w, t->permissions, (size_t)na * 2);        // m_Permissions
    w += (size_t)na * 2;
    memcpy(w, t->path_data, t->path_data_size);       // m_NameData

    return 0;
}

// collects the helper (and what it queued): its error, if any
static int ingest_vi_join(lthip_ingest* g)
{
    if (g->vi_thread.joinable())
    {
        g->vi_thread.join();
        if (g->vi_ctx)
            (void)hipStreamSynchronize(g->vi_ctx->stream);
    }
    const int e = g->vi_err;
    g->vi_err = 0;
    return e;
}

// starts what lthip_ingest_index prepared (once): the helper thread, or the work itself when the thread is switched off
static int ingest_vi_start(lthip_ingest* g)
{
    if (!g->vi_pending)
        return 0;
    g->vi_pending = false;
    lthip_ctx* ctx = g->ctx;
    LTHIP_ABLATION_ENV(env_vit, "LTHIP_INGEST_VI_THREAD");
    if (env_vit.get() == 0)
        return ingest_vi_work(g, ctx);
    if (!g->vi_ctx && lthip_ctx_create(ctx->device, LTHIP_STREAM_PRIVATE, &g->vi_ctx) != 0)
        return lthip_fail(ctx, ENOMEM, "lthip_ingest", "no context for the VersionIndex helper");
    g->vi_thread = std::thread([g] { g->vi_err = ingest_vi_work(g, g->vi_ctx); });
    return 0;
}

extern "C" int lthip_ingest_index(lthip_ingest* g, const lthip_ingest_tree* t, const uint64_t* d_all_hashes, const uint32_t* d_all_lens,
                                  uint64_t all_chunks, const uint64_t* d_local_offsets, const uint32_t* d_local_part_first,
                                  uint64_t local_chunks, void* h_version_index, size_t version_index_capacity)
{
    if (!g || !t || (all_chunks && (!d_all_hashes || !d_all_lens)) || (local_chunks && (!d_local_offsets || !d_local_part_first)) ||
        (t->job_count && (!t->job_asset || !t->job_first)) ||
        (t->asset_count && (!t->asset_sizes || !t->path_start_offsets || !t->permissions || !t->path_data)))
        return EINVAL;
    lthip_ctx* ctx = g->ctx;
    if (all_chunks > 0x7FFFFFF0ull || local_chunks > all_chunks)
        return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "chunk counts out of range");
    if (t->job_count && t->job_first[t->job_count] != all_chunks)
        return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "job_first[job_count] must be the number of chunks");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    g->vi_pending = false;
    (void)ingest_vi_join(g); // (an index that was never finished: its helper reads what this call is about to replace ...
    (void)hipEventSynchronize(g->ev_hashes); // ... and so does the side stream)
    IngTrace tr("lthip_ingest_index");
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t)all_chunks, nl = (uint32_t)local_chunks, na = t->asset_count;
    const bool all_mine = t->my_jobs == nullptr;
    const uint64_t my_jobs = all_mine ? t->job_count : t->my_job_count;
    g->indexed = g->written = false;
    g->n_all = n;
    g->n_local = nl;
    g->has_tags = t->asset_tags != nullptr;
    memset(&g->res, 0, sizeof g->res);

    tr.mark("checks");
    int err;
    if ((err = reserve_dev(ctx, g->d_first, (size_t)n * 4)) || (err = reserve_dev(ctx, g->d_isfirst, (size_t)n * 4)) ||
        (err = reserve_dev(ctx, g->d_rank, ((size_t)n + 1) * 4)) || (err = reserve_dev(ctx, g->d_idx, (size_t)n * 4)) ||
        (err = reserve_dev(ctx, g->d_uh, (size_t)n * 8)) || (err = reserve_dev(ctx, g->d_us, (size_t)n * 4)) ||
        (err = reserve_dev(ctx, g->d_ut, (size_t)n * 4)) || (err = reserve_dev(ctx, g->d_starts, ((size_t)na + 1) * 4)) ||
        (err = reserve_dev(ctx, g->d_tags, (size_t)na * 4)) || (err = reserve_dev(ctx, g->d_counts, 64)) ||
        (err = reserve_dev(ctx, g->d_paths, (size_t)t->path_data_size + 16)) || (err = reserve_dev(ctx, g->d_aoff, (size_t)na * 8)) ||
        (err = reserve_dev(ctx, g->d_alen, (size_t)na * 4)) || (err = reserve_dev(ctx, g->d_ph, (size_t)na * 8)) ||
        (err = reserve_dev(ctx, g->d_ch, (size_t)na * 8)) || (err = reserve_dev(ctx, g->d_gfirst, ((size_t)my_jobs + 1) * 4)) ||
        (err = reserve_dev(ctx, g->d_owned, (size_t)nl * 4)) || (err = reserve_dev(ctx, g->d_orank, ((size_t)nl + 1) * 4)) ||
        (err = reserve_dev(ctx, g->d_l2g, (size_t)nl * 4)) || (err = reserve_dev(ctx, g->d_mu_hash, (size_t)nl * 8)) ||
        (err = reserve_dev(ctx, g->d_mu_len, (size_t)nl * 4)) || (err = reserve_dev(ctx, g->d_mu_off, (size_t)nl * 8)) ||
        (err = reserve_dev(ctx, g->d_mu_tag, (size_t)nl * 4)) || (err = reserve_pinned(ctx, g->h_counts, 64)) ||
        (err = reserve_pinned(ctx, g->h_mu_len, (size_t)nl * 4)) || (err = reserve_pinned(ctx, g->h_mu_off, (size_t)nl * 8)) ||
        (err = reserve_pinned(ctx, g->h_brk, (size_t)nl + 16)) || (err = reserve_dev(ctx, g->d_brk, (size_t)nl + 16)) ||
        (err = reserve_pinned(ctx, g->h_mu_hash, (size_t)nl * 8)) || (err = reserve_pinned(ctx, g->h_mu_tag, (size_t)nl * 4)) ||
        // (per block; the number of blocks is known when the packing ends, which is after the first codec batch was queued: a block
        // holds at least one chunk)
        (err = reserve_dev(ctx, g->d_bhash, (size_t)nl * 8)) || (err = reserve_dev(ctx, g->d_boff, (size_t)nl * 8)) ||
        (err = reserve_dev(ctx, g->d_blen, (size_t)nl * 4)) || (err = reserve_dev(ctx, g->d_comp, (size_t)nl * 4)) ||
        (err = reserve_dev(ctx, g->d_sum, 8)))
        return err;
    tr.mark("reserve");
    uint64_t* d_counts = (uint64_t*)g->d_counts.p; // [0] distinct hashes of all ranks, [1] chunks this rank writes (u32 in the low half)
    volatile uint64_t* h_counts = (volatile uint64_t*)g->h_counts.p;

    // ---- first-seen pass over ALL chunks (:2951-2970), unique index of every asset chunk ----
    if (g->ext_first)
    {
        // computed by the ranks together (lthip_dedup_min_ordinal on every rank's share of the hash space): nothing to insert here
        const uint64_t u = g->ext_unique;
        LTHIP_CHECK(ctx, hipMemcpyAsync(g->d_first.p, g->ext_first, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if ((err = lthip_stage_upload(ctx, d_counts, &u, 8, s)))
            return err;
        g->ext_first = nullptr;
    }
    else if ((err = lthip_dedup_first_seen(ctx, n, d_all_hashes, (uint32_t*)g->d_first.p, d_counts)))
        return err;
    tr.mark("first-seen");

    // ---- host tables that only depend on the job layout (and its validation): 0.05-0.1 ms on the 64 GiB tree, while the first-seen pass
    // (0.55 ms) is running ----
    std::vector<uint32_t> starts((size_t)na + 1, 0), counts(na, 0);
    for (uint64_t j = 0; j < t->job_count; ++j)
    {
        if (t->job_asset[j] >= na || t->job_first[j + 1] < t->job_first[j])
            return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "bad job table");
        counts[t->job_asset[j]] += (uint32_t)(t->job_first[j + 1] - t->job_first[j]);
    }
    for (uint32_t a = 0; a < na; ++a)
        starts[a + 1] = starts[a] + counts[a];
    if (starts[na] != n)
        return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "jobs do not cover the chunk arrays");
    std::vector<uint32_t> gfirst;
    if (!all_mine)
    {
        gfirst.resize((size_t)my_jobs + 1);
        uint64_t mine = 0;
        for (uint64_t m = 0; m < my_jobs; ++m)
        {
            const uint64_t j = t->my_jobs[m];
            if (j >= t->job_count || (m && j <= t->my_jobs[m - 1]))
                return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "my_jobs must be ascending job indices");
            gfirst[m] = (uint32_t)t->job_first[j];
            mine += t->job_first[j + 1] - t->job_first[j];
        }
        gfirst[my_jobs] = 0;
        if (mine != nl)
            return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "own jobs do not add up to the local chunk count");
    }
    else if (nl != n)
        return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "without my_jobs the local arrays are the global ones");

    tr.mark("tables");
    if ((err = lthip_stage_upload(ctx, g->d_starts.p, starts.data(), ((size_t)na + 1) * 4, s)))
        return err;
    if (g->has_tags && na && (err = lthip_stage_upload(ctx, g->d_tags.p, t->asset_tags, (size_t)na * 4, s)))
        return err;
    if (!all_mine && (err = lthip_stage_upload(ctx, g->d_gfirst.p, gfirst.data(), ((size_t)my_jobs + 1) * 4, s)))
        return err;
    const bool want_vi = h_version_index != nullptr;
    if (n)
    {
        const uint32_t blocks = (uint32_t)div_up_u64(n, 256);
        LaunchTimer tm(ctx, LTHIP_K_OTHER);
        hipLaunchKernelGGL(k_vi_mark, dim3(blocks), dim3(256), 0, s, (const uint32_t*)g->d_first.p, (uint64_t)n, (uint32_t*)g->d_isfirst.p);
        if ((err = lthip_exclusive_scan_u32(ctx, (const uint32_t*)g->d_isfirst.p, (uint32_t*)g->d_rank.p, n, nullptr, LTHIP_K_OTHER)))
            return err;
        hipLaunchKernelGGL(k_vi_compact, dim3(blocks), dim3(256), 0, s, (const uint32_t*)g->d_first.p, (const uint32_t*)g->d_rank.p,
                           (uint64_t)n, d_all_hashes, d_all_lens, (const uint32_t*)g->d_starts.p, na,
                           g->has_tags ? (const uint32_t*)g->d_tags.p : (const uint32_t*)nullptr, (uint32_t*)g->d_idx.p,
                           (uint64_t*)g->d_uh.p, (uint32_t*)g->d_us.p, (uint32_t*)g->d_ut.p);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    // ---- the chunks this rank writes: first-seen and in one of its own jobs, in version order ----
    h_counts[2] = h_counts[3] = 0;
    if (nl)
    {
        const uint32_t blocks = (uint32_t)div_up_u64(nl, 256);
        LaunchTimer tm(ctx, LTHIP_K_OTHER);
        hipLaunchKernelGGL(k_ing_owned, dim3(blocks), dim3(256), 0, s, (const uint32_t*)g->d_first.p, d_local_part_first, (uint32_t)my_jobs,
                           all_mine ? (const uint32_t*)nullptr : (const uint32_t*)g->d_gfirst.p, nl, (uint32_t*)g->d_owned.p,
                           (uint32_t*)g->d_l2g.p);
        if ((err = lthip_exclusive_scan_u32(ctx, (const uint32_t*)g->d_owned.p, (uint32_t*)g->d_orank.p, nl, nullptr, LTHIP_K_OTHER)))
            return err;
        hipLaunchKernelGGL(k_ing_compact, dim3(blocks), dim3(256), 0, s, (const uint32_t*)g->d_owned.p, (const uint32_t*)g->d_orank.p,
                           (const uint32_t*)g->d_l2g.p, nl, d_all_hashes, d_all_lens, d_local_offsets, (const uint32_t*)g->d_starts.p, na,
                           g->has_tags ? (const uint32_t*)g->d_tags.p : (const uint32_t*)nullptr, (uint64_t*)g->d_mu_hash.p,
                           (uint32_t*)g->d_mu_len.p, (uint64_t*)g->d_mu_off.p, (uint32_t*)g->d_mu_tag.p);
        LTHIP_LAUNCH_CHECK(ctx);
        // number of owned chunks = orank[nl - 1] + owned[nl - 1]; the scan wrote nl entries, so read both
        LTHIP_CHECK(ctx, hipMemcpyAsync((void*)(h_counts + 2), (const uint32_t*)g->d_orank.p + (nl - 1), 4, hipMemcpyDeviceToHost, s));
        LTHIP_CHECK(ctx, hipMemcpyAsync((void*)(h_counts + 3), (const uint32_t*)g->d_owned.p + (nl - 1), 4, hipMemcpyDeviceToHost, s));
    }
    LTHIP_CHECK(ctx, hipMemcpyAsync((void*)h_counts, d_counts, 8, hipMemcpyDeviceToHost, s));
    tr.mark("queued");
    LTHIP_CHECK(ctx, hipEventRecord(g->ev_counts, s));
    LTHIP_CHECK(ctx, hipEventSynchronize(g->ev_counts));
    tr.mark("counts");
    const uint64_t unique = h_counts[0];
    const uint32_t nm = nl ? (uint32_t)(h_counts[2] & 0xFFFFFFFFu) + (uint32_t)(h_counts[3] & 0xFFFFFFFFu) : 0u;
    g->unique_all = unique;
    g->n_mine = nm;

    // ---- the host needs the owned chunks' lengths, offsets (and tags) for the packing and the codec calls.  Of the offsets the packing
    // loop reads only whether a chunk continues the range of the one before: a byte per chunk from k_ing_breaks.  The copies are 44 MB on
    // the 64 GiB tree (0.8 ms of the link), and the first codec batch needs the head of the lists only: the chunks of about that batch
    // are copied on the session's stream (ev_lens), the rest -- and the chunk hashes, which lthip_ingest_finish reads -- by the side
    // stream next to the codec kernels (ev_offs, ev_hashes)
    uint32_t head = nm;
    {
        LTHIP_ABLATION_ENV(env_slices, "LTHIP_INGEST_PACK_SLICES");
        uint64_t tree_bytes = 0;
        for (uint32_t a = 0; a < na; ++a)
            tree_bytes += t->asset_sizes[a];
        if (env_slices.get() != 0 && n && tree_bytes > 2 * g->cfg.batch_bytes)
        {
            // chunks of one batch at the tree's mean chunk size, a quarter more, and the chunks of one more block
            const double per_byte = (double)n / (double)tree_bytes;
            const double want = 1.25 * per_byte * (double)g->cfg.batch_bytes + 2.0 * g->cfg.max_chunks_per_block + 4096.0;
            if (want < (double)nm)
                head = (uint32_t)want;
        }
    }
    g->pack_avail = head;
    hipStream_t s2 = s;
    if (head < nm && (err = lthip_second_stream(ctx, &s2)))
        return err;
    auto d2h_lists = [&](uint32_t c0, uint32_t c1, hipStream_t st) -> int {
        const size_t k = (size_t)c1 - c0;
        if (!k)
            return 0;
        LTHIP_CHECK(ctx, hipMemcpyAsync((uint32_t*)g->h_mu_len.p + c0, (const uint32_t*)g->d_mu_len.p + c0, k * 4, hipMemcpyDeviceToHost, st));
        LTHIP_CHECK(ctx, hipMemcpyAsync((uint8_t*)g->h_brk.p + c0, (const uint8_t*)g->d_brk.p + c0, k, hipMemcpyDeviceToHost, st));
        if (g->has_tags)
            LTHIP_CHECK(ctx, hipMemcpyAsync((uint32_t*)g->h_mu_tag.p + c0, (const uint32_t*)g->d_mu_tag.p + c0, k * 4, hipMemcpyDeviceToHost, st));
        LTHIP_CHECK(ctx, hipMemcpyAsync((uint64_t*)g->h_mu_off.p + c0, (const uint64_t*)g->d_mu_off.p + c0, k * 8, hipMemcpyDeviceToHost, st));
        return 0;
    };
    if (nm)
    {
        hipLaunchKernelGGL(k_ing_breaks, dim3((uint32_t)div_up_u64(nm, 256)), dim3(256), 0, s, (const uint64_t*)g->d_mu_off.p,
                           (const uint32_t*)g->d_mu_len.p, nm, (uint8_t*)g->d_brk.p);
        LTHIP_LAUNCH_CHECK(ctx);
        if ((err = d2h_lists(0, head, s)))
            return err;
    }
    LTHIP_CHECK(ctx, hipEventRecord(g->ev_lens, s));
    if (s2 != s)
        LTHIP_CHECK(ctx, hipStreamWaitEvent(s2, g->ev_lens, 0));
    if ((err = d2h_lists(head, nm, s2)))
        return err;
    LTHIP_CHECK(ctx, hipEventRecord(g->ev_offs, s2));
    if (nm)
        LTHIP_CHECK(ctx, hipMemcpyAsync(g->h_mu_hash.p, g->d_mu_hash.p, (size_t)nm * 8, hipMemcpyDeviceToHost, s2)); // StoreIndex, read in finish
    LTHIP_CHECK(ctx, hipEventRecord(g->ev_hashes, s2));

    tr.mark("lists");
    // ---- greedy packing of the owned chunks (Longtail_CreateStoreIndex :6801-6860): here the blocks of the first codec batch ----
    LTHIP_CHECK(ctx, hipEventSynchronize(g->ev_lens));
    g->b_first.clear();
    g->b_size.clear();
    g->b_is_range.clear();
    g->b_tag.clear();
    g->b_first.push_back(0);
    g->pack_next = 0;
    g->pack_raw = 0;
    g->blocks_done = false;
    g->written = false;
    ingest_pack(g, g->cfg.batch_bytes + 2ull * g->cfg.max_block_size);
    tr.mark("pack");

    // ---- the VersionIndex sections: 1.2-2.4 ms of host work on the 64 GiB tree (the tables of 65 536 assets, the tag column
    // of 2.1 M chunks) plus copies and two small hash launches, none of which the rest of the session waits for: a helper thread with a
    // context of its own does them (LTHIP_INGEST_VI_THREAD=0: the calling thread, on the session's context), started by ingest_vi_start once
    // the first codec batch is queued -- a thread's start is 0.1 ms -- and collected by lthip_ingest_finish.
    g->vi_size = 0;
    if (want_vi)
    {
        const size_t size = lthip_version_index_size(na, unique, n, t->path_data_size);
        g->vi_size = size;
        if (version_index_capacity < size)
            return lthip_fail(ctx, ENOMEM, "lthip_ingest_index", "version index buffer too small");
        for (uint32_t a = 0; a < na; ++a)
        {
            if ((uint64_t)counts[a] * 8u > 0xFFFFFFFFull)
                return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "asset with more than 2^29 chunks");
            if (t->path_start_offsets[a] >= t->path_data_size)
                return lthip_fail(ctx, EINVAL, "lthip_ingest_index", "path offset outside the path data");
        }
        g->vi_tree = *t;
        g->vi_asset_sizes.assign(t->asset_sizes, t->asset_sizes + na);
        g->vi_path_offsets.assign(t->path_start_offsets, t->path_start_offsets + na);
        g->vi_permissions.assign(t->permissions, t->permissions + na);
        g->vi_path_data.assign(t->path_data, t->path_data + t->path_data_size);
        g->vi_tree.asset_sizes = g->vi_asset_sizes.data();
        g->vi_tree.path_start_offsets = g->vi_path_offsets.data();
        g->vi_tree.permissions = g->vi_permissions.data();
        g->vi_tree.path_data = g->vi_path_data.data();
        g->vi_tree.asset_tags = nullptr; // (not read after the call)
        g->vi_tree.job_asset = nullptr;
        g->vi_tree.job_first = nullptr;
        g->vi_tree.my_jobs = nullptr;
        g->vi_starts.swap(starts);
        g->vi_counts.swap(counts);
        g->vi_hashes = d_all_hashes;
        g->vi_out = h_version_index;
        g->vi_pending = true; // (ingest_vi_start: behind the first codec batch's launches, or in lthip_ingest_finish)
    }

    tr.mark("helper");
    g->res.chunks_all = n;
    g->res.unique_all = unique;
    g->res.chunks_local = nl;
    g->res.unique_local = nm;
    g->res.blocks = 0; // (ingest_blocks_done)
    g->res.raw_bytes = 0;
    g->res.version_index_size = g->vi_size;
    g->indexed = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// phase 3: WriteContent into a bounded device arena (null sink)
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int lthip_ingest_write(lthip_ingest* g, const void* d_data, void* d_arena, uint64_t arena_bytes)
{
    if (!g || !g->indexed || (g->n_mine && (!d_data || !d_arena)))
        return EINVAL;
    IngTrace tr("lthip_ingest_write");
    lthip_ctx* ctx = g->ctx;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t* lens = (const uint32_t*)g->h_mu_len.p;
    const uint64_t* offs = (const uint64_t*)g->h_mu_off.p;
    int err;
    uint64_t gathered_blocks = 0, gathered_bytes = 0;
    g->img_first = 0;
    g->img_offsets.clear();
    g->img_sizes.clear();
    g->img_hdr.clear();
    std::vector<uint64_t> src_off, dst_off, img_off, g_src, g_dst, bfirst;
    std::vector<uint32_t> src_size, dst_cap, g_len, braw;
    // more blocks, when the batch being put together has taken all there are and chunks are left
    auto more_blocks = [&](size_t have) -> int {
        while (g->b_size.size() == have && g->pack_next < g->n_mine)
        {
            if (g->pack_next + 2ull * g->cfg.max_chunks_per_block >= g->pack_avail && g->pack_avail < g->n_mine)
            {
                LTHIP_CHECK(ctx, hipEventSynchronize(g->ev_offs));
                g->pack_avail = (uint32_t)g->n_mine;
            }
            ingest_pack(g, 1ull << 30);
        }
        return 0;
    };
    for (size_t b0 = 0;;)
    {
        if ((err = more_blocks(b0)))
            return err;
        if (b0 == g->b_size.size())
            break;
        // ---- one batch: as many blocks as the arena and the codec's batch size hold (always at least one) ----
        size_t b1 = b0;
        uint64_t arena = 0, bytes = 0, gather_chunks = 0;
        img_off.clear();
        while ((err = more_blocks(b1)) == 0 && b1 < g->b_size.size())
        {
            const uint32_t nchunks = (uint32_t)(g->b_first[b1 + 1] - g->b_first[b1]);
            const uint64_t need = ((uint64_t)lthip_stored_block_header_size(nchunks) + codec_bound(g, g->b_size[b1]) + 63u) & ~(uint64_t)63u;
            if (b1 > b0 && (arena + need > arena_bytes || bytes + g->b_size[b1] > g->cfg.batch_bytes))
                break;
            if (arena + need > arena_bytes)
                return lthip_fail(ctx, ENOMEM, "lthip_ingest_write", "the arena does not hold a single stored block");
            img_off.push_back(arena);
            arena += need;
            bytes += g->b_size[b1];
            if (!g->b_is_range[b1])
                gather_chunks += nchunks;
            ++b1;
        }
        if (err)
            return err;
        tr.mark("batch");
        const size_t cnt = b1 - b0;
        // ---- block assembly (WriteContentBlockJob, :4640-4721) only for blocks that are not one byte range of the data ----
        if (gather_chunks)
        {
            if ((err = reserve_dev(ctx, g->d_gsrc, gather_chunks * 8)) ||
                (err = reserve_dev(ctx, g->d_glen, gather_chunks * 4)) || (err = reserve_dev(ctx, g->d_gdst, gather_chunks * 8)))
                return err;
            g_src.clear();
            g_len.clear();
            g_dst.clear();
            uint64_t pos = 0;
            for (size_t b = b0; b < b1; ++b)
                if (!g->b_is_range[b])
                {
                    pos = (pos + 15u) & ~(uint64_t)15u;
                    for (uint64_t c = g->b_first[b]; c < g->b_first[b + 1]; ++c)
                    {
                        g_src.push_back(offs[c]);
                        g_len.push_back(lens[c]);
                        g_dst.push_back(pos);
                        pos += lens[c];
                    }
                }
            if ((err = reserve_dev(ctx, g->d_gather, pos + 256)))
                return err;
            if ((err = lthip_stage_upload(ctx, g->d_gsrc.p, g_src.data(), g_src.size() * 8, s)) ||
                (err = lthip_stage_upload(ctx, g->d_glen.p, g_len.data(), g_len.size() * 4, s)) ||
                (err = lthip_stage_upload(ctx, g->d_gdst.p, g_dst.data(), g_dst.size() * 8, s)))
                return err;
            if ((err = lthip_gather_ranges(ctx, d_data, g_src.size(), (const uint64_t*)g->d_gsrc.p, (const uint32_t*)g->d_glen.p, g->d_gather.p,
                                           (const uint64_t*)g->d_gdst.p)))
                return err;
        }
        // ---- compress straight to image + header size; the blocks in place first, then the assembled ones ----
        for (int pass = 0; pass < 2; ++pass)
        {
            src_off.clear();
            src_size.clear();
            dst_off.clear();
            dst_cap.clear();
            std::vector<uint32_t> which;
            uint64_t pos = 0;
            for (size_t b = b0; b < b1; ++b)
            {
                const bool range = g->b_is_range[b] != 0;
                if (!range)
                    pos = (pos + 15u) & ~(uint64_t)15u;
                if (range == (pass == 0))
                {
                    const uint32_t nchunks = (uint32_t)(g->b_first[b + 1] - g->b_first[b]);
                    src_off.push_back(range ? offs[g->b_first[b]] : pos);
                    src_size.push_back((uint32_t)g->b_size[b]);
                    dst_off.push_back(img_off[b - b0] + lthip_stored_block_header_size(nchunks));
                    dst_cap.push_back((uint32_t)codec_bound(g, g->b_size[b]));
                    which.push_back((uint32_t)b);
                }
                if (!range)
                    pos += g->b_size[b];
            }
            if (src_off.empty())
                continue;
            // the codec entry points write one size per block of the call: the two passes are contiguous runs only when the
            // batch is all-range or all-gathered, so sizes go through a per-call list and are scattered by `which`
            const void* src = pass == 0 ? d_data : g->d_gather.p;
            const uint32_t k = (uint32_t)src_off.size();
            bool contiguous = true;
            for (uint32_t i = 1; i < k; ++i)
                contiguous &= which[i] == which[i - 1] + 1;
            uint32_t* d_sizes = (uint32_t*)g->d_comp.p + which[0];
            if (!contiguous)
            {
                if ((err = reserve_dev(ctx, g->d_tmpsz, (size_t)k * 4)))
                    return err;
                d_sizes = (uint32_t*)g->d_tmpsz.p;
            }
            if (g->cfg.codec == LTHIP_CODEC_LZ4)
                err = lthip_lz4_compress_blocks(ctx, src, k, src_off.data(), src_size.data(), d_arena, dst_off.data(), dst_cap.data(), d_sizes, 0);
            else if (g->cfg.codec == LTHIP_CODEC_ZSTD)
                err = lthip_zstd_compress_blocks_q(ctx, src, k, src_off.data(), src_size.data(), d_arena, dst_off.data(), dst_cap.data(), d_sizes,
                                                   lthip_zstd_quality_of_settings(g->cfg.compression_type)); // ('ztd4': high, 'ztd3' / 'ztd5': max)
            else
                err = lthip_fail(ctx, EINVAL, "lthip_ingest_write", "codec 0 (store raw) is not implemented");
            if (err)
                return err;
            if (!contiguous)
            {
                // scatter: comp[which[i]] = sizes[i]  (gather kernel on 4-byte ranges)
                std::vector<uint64_t> so(k), dof(k);
                std::vector<uint32_t> four(k, 4u);
                for (uint32_t i = 0; i < k; ++i)
                {
                    so[i] = (uint64_t)i * 4u;
                    dof[i] = (uint64_t)which[i] * 4u;
                }
                if ((err = reserve_dev(ctx, g->d_gsrc, (size_t)k * 8)) || (err = reserve_dev(ctx, g->d_glen, (size_t)k * 4)) ||
                    (err = reserve_dev(ctx, g->d_gdst, (size_t)k * 8)))
                    return err;
                if ((err = lthip_stage_upload(ctx, g->d_gsrc.p, so.data(), (size_t)k * 8, s)) ||
                    (err = lthip_stage_upload(ctx, g->d_glen.p, four.data(), (size_t)k * 4, s)) ||
                    (err = lthip_stage_upload(ctx, g->d_gdst.p, dof.data(), (size_t)k * 8, s)))
                    return err;
                if ((err = lthip_gather_ranges(ctx, g->d_tmpsz.p, k, (const uint64_t*)g->d_gsrc.p, (const uint32_t*)g->d_glen.p, g->d_comp.p,
                                               (const uint64_t*)g->d_gdst.p)))
                    return err;
            }
            if (pass == 1)
            {
                gathered_blocks += k;
                for (uint32_t i = 0; i < k; ++i)
                    gathered_bytes += src_size[i];
            }
        }
        tr.mark("codec");
        // ---- (first batch: the codec has work now; the rest of the packing and all block hashes) ----
        if ((err = ingest_vi_start(g)) || (err = ingest_blocks_done(g)))
            return err;
        tr.mark("blocks");
        // ---- BlockIndex + [raw][compressed] around the payloads (:4111-4150; compressblockstore.c:103-139) ----
        if ((err = reserve_dev(ctx, g->d_bfirst, (cnt + 1) * 4)) || (err = reserve_dev(ctx, g->d_braw, cnt * 4)) ||
            (err = reserve_dev(ctx, g->d_bimg, cnt * 8)) || (err = reserve_dev(ctx, g->d_btag, cnt * 4)))
            return err;
        std::vector<uint32_t> first32(cnt + 1);
        braw.resize(cnt);
        for (size_t b = b0; b <= b1; ++b)
            first32[b - b0] = (uint32_t)g->b_first[b];
        for (size_t b = b0; b < b1; ++b)
            braw[b - b0] = (uint32_t)g->b_size[b];
        if ((err = lthip_stage_upload(ctx, g->d_bfirst.p, first32.data(), (cnt + 1) * 4, s)) ||
            (err = lthip_stage_upload(ctx, g->d_braw.p, braw.data(), cnt * 4, s)) ||
            (err = lthip_stage_upload(ctx, g->d_bimg.p, img_off.data(), cnt * 8, s)))
            return err;
        const uint32_t* d_tags = nullptr;
        if (g->has_tags)
        {
            if ((err = lthip_stage_upload(ctx, g->d_btag.p, g->b_tag.data() + b0, cnt * 4, s)))
                return err;
            d_tags = (const uint32_t*)g->d_btag.p;
        }
        {
            LaunchTimer tm(ctx, LTHIP_K_OTHER);
            hipLaunchKernelGGL(k_stored_block_headers, dim3((uint32_t)cnt), dim3(64), 0, s, (const uint32_t*)g->d_bfirst.p, (uint32_t)cnt,
                               (const uint64_t*)g->d_mu_hash.p, (const uint32_t*)g->d_mu_len.p, (const uint64_t*)g->d_bhash.p + b0,
                               g->cfg.hash_identifier, g->cfg.compression_type, d_tags, (const uint32_t*)g->d_braw.p,
                               (const uint32_t*)g->d_comp.p + b0, (const uint64_t*)g->d_bimg.p, (uint8_t*)d_arena);
            LTHIP_LAUNCH_CHECK(ctx);
        }
        g->img_first = b0;
        g->img_offsets.assign(img_off.begin(), img_off.end());
        g->img_hdr.resize(cnt);
        for (size_t b = b0; b < b1; ++b)
            g->img_hdr[b - b0] = (uint32_t)lthip_stored_block_header_size((uint32_t)(g->b_first[b + 1] - g->b_first[b]));
        g->img_sizes = g->img_hdr; // (headers only until lthip_ingest_finish knows the payload sizes)
        b0 = b1;
    }
    if ((err = ingest_vi_start(g)) || (err = ingest_blocks_done(g))) // (nothing to write)
        return err;
    const size_t nb = g->b_size.size();
    // compressed sizes of all blocks: total on the device, list to the host for the caller's statistics
    LTHIP_CHECK(ctx, hipMemsetAsync(g->d_sum.p, 0, 8, s));
    if (nb)
    {
        hipLaunchKernelGGL(k_ing_sum_u32, dim3(64), dim3(256), 0, s, (const uint32_t*)g->d_comp.p, (uint32_t)nb, (unsigned long long*)g->d_sum.p);
        LTHIP_LAUNCH_CHECK(ctx);
        LTHIP_CHECK(ctx, hipMemcpyAsync((uint8_t*)g->h_comp.p + 8, g->d_comp.p, nb * 4, hipMemcpyDeviceToHost, s));
    }
    LTHIP_CHECK(ctx, hipMemcpyAsync(g->h_comp.p, g->d_sum.p, 8, hipMemcpyDeviceToHost, s));
    g->res.gathered_blocks = gathered_blocks;
    g->res.gathered_bytes = gathered_bytes;
    g->written = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// the serialized StoreIndex of what this rank wrote (Longtail_CreateStoreIndexFromBlocks :9060-9125, layout :8913-8931),
// laid out by the host while the codec is still running, then the one synchronisation of the session
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int lthip_ingest_finish(lthip_ingest* g, void* h_store_index, size_t store_index_capacity, lthip_ingest_result* out)
{
    if (!g || !g->indexed)
        return EINVAL;
    lthip_ctx* ctx = g->ctx;
    // (the caller says how large ITS struct is: a header older or newer than this library's never gets written past its end.  Checked
    // before any work: a call that is going to be refused changes nothing)
    if (out && (out->struct_size < 16 || out->struct_size > 4096))
        return lthip_fail(ctx, EINVAL, "lthip_ingest_finish", "out_result->struct_size must be set to sizeof(lthip_ingest_result)");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    {
        int err = ingest_blocks_done(g); // (an index without lthip_ingest_write)
        if (!err)
            err = ingest_vi_start(g);
        if (err)
            return err;
    }
    const size_t nb = g->b_size.size(), m = (size_t)g->n_mine;
    const size_t size = 16 + nb * 8 + m * 8 + nb * 12 + m * 4; // Longtail_GetStoreIndexDataSize
    g->res.store_index_size = size;
    int rc = 0;
    if (h_store_index)
    {
        if (store_index_capacity < size)
            rc = ENOMEM;
        else
        {
            LTHIP_CHECK(ctx, hipEventSynchronize(g->ev_index));  // block hashes ...
            LTHIP_CHECK(ctx, hipEventSynchronize(g->ev_hashes)); // ... and the owned chunks' hashes are on the host
            uint8_t* w = (uint8_t*)h_store_index;
            const uint32_t head[4] = {(1u << 24) /* LONGTAIL_STORE_INDEX_VERSION_1_0_0, :19-23 */, g->cfg.hash_identifier, (uint32_t)nb, (uint32_t)m};
            memcpy(w, head, 16);
            w += 16;
            memcpy(w, g->h_bhash.p, nb * 8); // m_BlockHashes
            w += nb * 8;
            memcpy(w, g->h_mu_hash.p, m * 8); // m_ChunkHashes
            w += m * 8;
            uint32_t* bo = (uint32_t*)w; // m_BlockChunksOffsets, m_BlockChunkCounts, m_BlockTags
            for (size_t b = 0; b < nb; ++b)
            {
                bo[b] = (uint32_t)g->b_first[b];
                bo[nb + b] = (uint32_t)(g->b_first[b + 1] - g->b_first[b]);
                bo[2 * nb + b] = g->b_tag[b];
            }
            w += nb * 12;
            memcpy(w, g->h_mu_len.p, m * 4); // m_ChunkSizes
        }
    }
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    LTHIP_CHECK(ctx, hipEventSynchronize(g->ev_hashes)); // (the side stream's copies)
    {
        const int vi_err = ingest_vi_join(g); // the VersionIndex is in the caller's buffer (or could not be made)
        if (vi_err)
            return lthip_fail(ctx, vi_err, "lthip_ingest_finish: VersionIndex", g->vi_ctx ? g->vi_ctx->err : "");
    }
    g->res.compressed_bytes = g->written ? *(const uint64_t*)g->h_comp.p : 0;
    if (g->written)
    {
        const uint32_t* comp = (const uint32_t*)((const uint8_t*)g->h_comp.p + 8);
        for (size_t i = 0; i < g->img_hdr.size(); ++i)
            g->img_sizes[i] = g->img_hdr[i] + comp[g->img_first + i]; // header (BlockIndex + [raw][compressed]) + payload
    }
    if (out)
    {
        const uint64_t have = out->struct_size;
        g->res.struct_size = have < sizeof g->res ? have : sizeof g->res;
        memcpy(out, &g->res, (size_t)g->res.struct_size);
    }
    return rc;
}

extern "C" int lthip_ingest_images(const lthip_ingest* g, uint64_t* out_first_block, uint64_t* out_count, const uint64_t** out_offsets,
                                   const uint32_t** out_sizes)
{
    if (!g || !g->written)
        return EINVAL;
    if (out_first_block)
        *out_first_block = g->img_first;
    if (out_count)
        *out_count = g->img_offsets.size();
    if (out_offsets)
        *out_offsets = g->img_offsets.data();
    if (out_sizes)
        *out_sizes = g->img_sizes.data();
    return 0;
}

extern "C" const uint32_t* lthip_ingest_compressed_sizes(const lthip_ingest* g) { return g && g->written ? (const uint32_t*)((const uint8_t*)g->h_comp.p + 8) : nullptr; }
