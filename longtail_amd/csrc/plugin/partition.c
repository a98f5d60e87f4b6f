/* partition.c -- multi-GPU work division of the ingest path, plain C host code.
 *
 * The unit of independence is the reference's own job: one (asset, target_chunk_size*1024-byte part) of
 * ChunkAssets (src/longtail.c:2396-2458) -- a fresh chunker per job, no state crosses jobs.  So any
 * assignment of jobs to ranks reproduces the single-process result as long as the per-job chunk lists are
 * put back into job order before the serial first-seen pass (:2499-2517, :2951-2970).  This file holds
 *   - the job list of a given tree (lthip_job_count / lthip_make_jobs),
 *   - three deterministic assignments every rank can compute for itself without communication
 *     (lthip_partition_jobs: byte-balanced contiguous ranges, longest-processing-time-first, job mod R),
 *   - the layout of the exchange (lthip_exchange_layout): where a job's run of chunks sits in the
 *     rank-major all-gathered arrays and where it belongs in job order.
 * The collective itself is two all-gathers (per-job counts, then padded hash / length arrays) done by the
 * caller (torch.distributed "nccl" == RCCL in bench.py); the reorder is one lthip_gather_ranges per array.
 */
#include "plugin_common.h"

uint64_t lthip_job_count(uint32_t asset_count, const uint64_t* asset_sizes, uint32_t target_chunk_size)
{
    if (!target_chunk_size || (!asset_sizes && asset_count))
        return 0;
    const uint64_t part = (uint64_t)target_chunk_size * 1024u; /* max_hash_size, src/longtail.c:2396 */
    uint64_t n = 0;
    for (uint32_t a = 0; a < asset_count; ++a)
        n += 1 + asset_sizes[a] / part; /* :2402 -- an exact multiple gets an empty trailing job, a directory one empty job */
    return n;
}

int lthip_make_jobs(uint32_t asset_count, const uint64_t* asset_sizes, uint32_t target_chunk_size, uint64_t capacity,
                    uint32_t* job_asset, uint64_t* job_offset, uint64_t* job_size)
{
    if (!target_chunk_size || (!asset_sizes && asset_count) || !job_asset || !job_offset || !job_size)
        return EINVAL;
    const uint64_t part = (uint64_t)target_chunk_size * 1024u;
    uint64_t j = 0;
    for (uint32_t a = 0; a < asset_count; ++a)
    {
        const uint64_t size = asset_sizes[a];
        const uint64_t parts = 1 + size / part;
        if (j + parts > capacity)
            return ENOMEM;
        for (uint64_t p = 0; p < parts; ++p, ++j)
        {
            const uint64_t start = p * part; /* :2439-2440 */
            job_asset[j] = a;
            job_offset[j] = start;
            job_size[j] = size - start > part ? part : size - start;
        }
    }
    return 0;
}

/* ---- longest processing time first: jobs by size descending (ties: lower job index first), each to the least loaded
 * rank (ties: lower rank).  Binary min-heap over (load, rank); O(J log J + J log R). ---- */
struct lpt_job
{
    uint64_t size;
    uint64_t index;
};

static int lpt_job_cmp(const void* pa, const void* pb)
{
    const struct lpt_job* a = (const struct lpt_job*)pa;
    const struct lpt_job* b = (const struct lpt_job*)pb;
    if (a->size != b->size)
        return a->size > b->size ? -1 : 1;
    return a->index < b->index ? -1 : (a->index > b->index ? 1 : 0);
}

struct lpt_rank
{
    uint64_t load;
    uint32_t rank;
};

static int lpt_rank_less(const struct lpt_rank* a, const struct lpt_rank* b)
{
    return a->load < b->load || (a->load == b->load && a->rank < b->rank);
}

static void lpt_sift_down(struct lpt_rank* h, uint32_t n, uint32_t i)
{
    for (;;)
    {
        uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && lpt_rank_less(&h[l], &h[m]))
            m = l;
        if (r < n && lpt_rank_less(&h[r], &h[m]))
            m = r;
        if (m == i)
            return;
        struct lpt_rank t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

int lthip_partition_jobs(uint64_t job_count, const uint64_t* job_sizes, uint32_t rank_count, int policy, uint32_t* job_rank,
                         uint64_t* rank_bytes)
{
    if (!rank_count || (job_count && (!job_sizes || !job_rank)))
        return EINVAL;
    if (rank_bytes)
        memset(rank_bytes, 0, sizeof(uint64_t) * rank_count);
    if (policy == LTHIP_PARTITION_MOD)
    {
        for (uint64_t j = 0; j < job_count; ++j)
            job_rank[j] = (uint32_t)(j % rank_count);
    }
    else if (policy == LTHIP_PARTITION_RANGE)
    {
        /* contiguous job ranges with equal byte shares: job j goes to the rank whose share holds the job's midpoint.  Every
         * job carries a small fixed weight besides its bytes so that trees of empty files / directories spread too. */
        const uint64_t job_weight = 4096;
        uint64_t total = 0;
        for (uint64_t j = 0; j < job_count; ++j)
            total += job_sizes[j] + job_weight;
        uint64_t before = 0;
        for (uint64_t j = 0; j < job_count; ++j)
        {
            const uint64_t w = job_sizes[j] + job_weight;
            /* rank = floor((before + w/2) * R / total), in 128-bit arithmetic */
            const unsigned __int128 mid = (unsigned __int128)(before + w / 2) * rank_count;
            uint32_t r = (uint32_t)(mid / total);
            if (r >= rank_count)
                r = rank_count - 1;
            job_rank[j] = r;
            before += w;
        }
    }
    else if (policy == LTHIP_PARTITION_LPT)
    {
        if (job_count > (uint64_t)(SIZE_MAX / sizeof(struct lpt_job)))
            return EOVERFLOW;
        struct lpt_job* jobs = (struct lpt_job*)ltp_alloc("lthip_partition_jobs", sizeof(struct lpt_job) * (size_t)(job_count ? job_count : 1));
        struct lpt_rank* heap = (struct lpt_rank*)ltp_alloc("lthip_partition_jobs", sizeof(struct lpt_rank) * rank_count);
        if (!jobs || !heap)
        {
            ltp_free(jobs);
            ltp_free(heap);
            return ENOMEM;
        }
        for (uint64_t j = 0; j < job_count; ++j)
        {
            jobs[j].size = job_sizes[j];
            jobs[j].index = j;
        }
        qsort(jobs, (size_t)job_count, sizeof *jobs, lpt_job_cmp);
        for (uint32_t r = 0; r < rank_count; ++r)
        {
            heap[r].load = 0;
            heap[r].rank = r; /* already a heap: equal loads, ascending ranks */
        }
        for (uint64_t k = 0; k < job_count; ++k)
        {
            job_rank[jobs[k].index] = heap[0].rank;
            heap[0].load += jobs[k].size + 1; /* + 1: empty jobs rotate over the ranks too */
            lpt_sift_down(heap, rank_count, 0);
        }
        ltp_free(jobs);
        ltp_free(heap);
    }
    else
        return EINVAL;
    if (rank_bytes)
        for (uint64_t j = 0; j < job_count; ++j)
            rank_bytes[job_rank[j]] += job_sizes[j];
    return 0;
}

int lthip_exchange_layout(uint64_t job_count, const uint32_t* job_rank, uint32_t rank_count, const uint32_t* gathered_counts,
                          uint64_t count_stride, uint64_t chunk_stride, uint64_t* job_src, uint64_t* job_dst, uint32_t* job_chunks)
{
    /* job_dst always receives job_count + 1 entries (the total at the end): it may never be NULL */
    if (!rank_count || !job_dst || (job_count && (!job_rank || !gathered_counts || !job_src)))
        return EINVAL;
    uint64_t* next_job = (uint64_t*)ltp_alloc("lthip_exchange_layout", sizeof(uint64_t) * 2 * rank_count);
    if (!next_job)
        return ENOMEM;
    uint64_t* next_chunk = next_job + rank_count;
    memset(next_job, 0, sizeof(uint64_t) * 2 * rank_count);
    uint64_t dst = 0;
    int err = 0;
    for (uint64_t j = 0; j < job_count; ++j)
    {
        const uint32_t r = job_rank[j];
        if (r >= rank_count || next_job[r] >= count_stride)
        {
            err = EINVAL;
            break;
        }
        const uint32_t c = gathered_counts[(uint64_t)r * count_stride + next_job[r]++]; /* rank r holds its jobs in ascending job order */
        if (next_chunk[r] + c > chunk_stride)
        {
            err = EINVAL; /* a rank announced more chunks than its slice of the gathered arrays holds */
            break;
        }
        job_src[j] = (uint64_t)r * chunk_stride + next_chunk[r];
        job_dst[j] = dst;
        if (job_chunks)
            job_chunks[j] = c;
        next_chunk[r] += c;
        dst += c;
    }
    if (!err)
        job_dst[job_count] = dst;
    ltp_free(next_job);
    return err;
}

/* The per-job runs of lthip_exchange_layout as the device reorder wants them: runs that are contiguous on both sides are merged
 * (range policy: one per rank; the other policies: one per maximal run of jobs of one rank) and cut into pieces of at most max_piece
 * elements -- lthip_exchange_reorder gives every piece a workgroup, so neither 500 000 ranges of 33 elements nor 8 ranges of 17 MB.
 * Returns the number of pieces (out arrays may be NULL: count only); more than `capacity` = the arrays were too small. */
uint64_t lthip_exchange_ranges(uint64_t job_count, const uint64_t* job_src, const uint64_t* job_dst, const uint32_t* job_chunks,
                               uint64_t max_piece, uint64_t capacity, uint64_t* out_src, uint64_t* out_dst, uint32_t* out_cnt)
{
    if (!max_piece || max_piece > 0x7FFFFFFFu || (job_count && (!job_src || !job_dst || !job_chunks)))
        return 0;
    uint64_t n = 0;
    uint64_t j = 0;
    while (j < job_count)
    {
        if (!job_chunks[j])
        {
            ++j;
            continue;
        }
        uint64_t src = job_src[j], dst = job_dst[j], len = job_chunks[j];
        for (++j; j < job_count; ++j)
        {
            if (!job_chunks[j])
                continue;
            if (job_src[j] != src + len || job_dst[j] != dst + len)
                break;
            len += job_chunks[j];
        }
        while (len)
        {
            const uint64_t k = len < max_piece ? len : max_piece;
            if (out_src && out_dst && out_cnt && n < capacity)
            {
                out_src[n] = src;
                out_dst[n] = dst;
                out_cnt[n] = (uint32_t)k;
            }
            ++n;
            src += k;
            dst += k;
            len -= k;
        }
    }
    return n;
}
