// k_gather.hip -- block assembly on the device: WriteContentBlockJob builds a stored block by
// reading every chunk of the block into one contiguous buffer (src/longtail.c:4640-4721).  With the assets already
// resident in HBM that is a gather of byte ranges: one workgroup per chunk, 16-byte stores, source realigned with
// v_alignbit.  Only needed when a block is not already one contiguous range of the asset buffer (dedup holes,
// assets whose sizes are not multiples of 16).
#include "lthip_internal.h"

namespace
{

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
constexpr int GT = 256;

__global__ __launch_bounds__(GT) void k_gather_ranges(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_offsets,
                                                      const uint32_t* __restrict__ lens, const uint64_t* __restrict__ dst_offsets,
                                                      uint64_t count, uint8_t* __restrict__ dst)
{
    const uint64_t i = blockIdx.x;
    if (i >= count)
        return;
    const int tid = threadIdx.x;
    const uint8_t* s = src + src_offsets[i];
    uint8_t* d = dst + dst_offsets[i];
    uint32_t n = lens[i];
    uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);
    if (head > n)
        head = n;
    if ((uint32_t)tid < head)
        d[tid] = s[tid];
    d += head;
    s += head;
    n -= head;
    const uint32_t nvec = n >> 4;
    const uint32_t mis = (uint32_t)((uintptr_t)s & 3u);
    const uint32_t sh = mis * 8u;
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s - mis);
    for (uint32_t v = tid; v < nvec; v += GT)
    {
        const uint32_t* q = s4 + v * 4u;
        const u32x4_a4 a = *reinterpret_cast<const u32x4_a4*>(q);
        const uint32_t e = mis ? q[4] : 0u;
        uint4 o;
        o.x = __builtin_amdgcn_alignbit(a.y, a.x, sh);
        o.y = __builtin_amdgcn_alignbit(a.z, a.y, sh);
        o.z = __builtin_amdgcn_alignbit(a.w, a.z, sh);
        o.w = __builtin_amdgcn_alignbit(e, a.w, sh);
        *reinterpret_cast<uint4*>(d + (uint64_t)v * 16u) = o;
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)tid < n - done)
        d[done + tid] = s[done + tid];
}

// Pinned host memory <-> HBM by the CUs.  The box's SDMA engines move ONE direction at a time (tools/pcie_duplex_probe.py,
// profiles/r05_pcie_duplex.json: hipMemcpyAsync h2d + d2h on two streams take the SUM of their times, 57 GB/s in total), a kernel
// that loads / stores over the link does not share them: kernel copies in both directions at once reach 46 GB/s EACH.  A host-fed
// ingest loop therefore fetches its slices and returns its block images with this kernel (and lthip_gather_ranges, whose
// destination may be pinned host memory too), on streams of their own beside the compute stream.
__global__ __launch_bounds__(GT) void k_link_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t nvec, uint32_t tail)
{
    const uint64_t stride = (uint64_t)gridDim.x * GT;
    uint64_t v = (uint64_t)blockIdx.x * GT + threadIdx.x;
    for (; v + 3u * stride < nvec; v += 4u * stride) // four 16-byte requests of every lane in flight (1 KiB per wave and request)
    {
        const uint4 a = src[v], b = src[v + stride], c = src[v + 2u * stride], d = src[v + 3u * stride];
        dst[v] = a;
        dst[v + stride] = b;
        dst[v + 2u * stride] = c;
        dst[v + 3u * stride] = d;
    }
    for (; v < nvec; v += stride)
        dst[v] = src[v];
    if (blockIdx.x == 0 && threadIdx.x < tail)
        reinterpret_cast<uint8_t*>(dst + nvec)[threadIdx.x] = reinterpret_cast<const uint8_t*>(src + nvec)[threadIdx.x];
}

// positions of a rank's chunks in job order (the multi-GPU exchange): chunk k of the rank lies in its own job m with
// part_first[m] <= k < part_first[m + 1] (lthip_chunk_hash's part table: one part per own job, ascending) and is the tree's chunk
// job_gfirst[m] + (k - part_first[m])
__global__ void k_job_ordinals(const uint32_t* __restrict__ part_first, uint32_t nparts, const uint32_t* __restrict__ job_gfirst,
                               uint32_t nlocal, uint32_t* __restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nlocal)
        return;
    uint32_t lo = 0, hi = nparts; // (empty parts: the last one that starts at or before k)
    while (hi - lo > 1)
    {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (part_first[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    out[k] = job_gfirst[lo] + (k - part_first[lo]);
}

// element ranges -> byte ranges of k_gather_ranges
__global__ void k_scale_ranges(const uint64_t* __restrict__ src, const uint64_t* __restrict__ dst, const uint32_t* __restrict__ cnt,
                               uint64_t count, uint32_t elem_bytes, uint64_t* __restrict__ src_b, uint64_t* __restrict__ dst_b,
                               uint32_t* __restrict__ len_b)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    src_b[i] = src[i] * elem_bytes;
    dst_b[i] = dst[i] * elem_bytes;
    len_b[i] = cnt[i] * elem_bytes;
}

} // namespace

extern "C" int lthip_exchange_reorder(lthip_ctx* ctx, const void* d_gathered, void* d_out, uint32_t elem_bytes, uint64_t range_count,
                                      const uint64_t* range_src, const uint64_t* range_dst, const uint32_t* range_cnt)
{
    if (!ctx || !elem_bytes || (range_count && (!d_gathered || !d_out || !range_src || !range_dst || !range_cnt)))
        return EINVAL;
    if (range_count == 0)
        return 0;
    if (range_count > 0x7FFFFFFFull)
        return lthip_fail(ctx, EINVAL, "lthip_exchange_reorder", "too many ranges");
    // k_scale_ranges holds a range's length in bytes in 32 bits (k_gather_ranges' table format): a range of count * elem_bytes above
    // that would be copied in part, silently -- refused instead (lthip_exchange_ranges cuts ranges at max_piece ELEMENTS; a caller
    // that passes a large max_piece with 8-byte elements lands here)
    for (uint64_t i = 0; i < range_count; ++i)
        if ((uint64_t)range_cnt[i] * elem_bytes > 0xFFFFFFFFull)
            return lthip_fail(ctx, EINVAL, "lthip_exchange_reorder", "a range of more than 4 GiB - 1 bytes: cut the ranges smaller (max_piece)");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    // tables: {src, dst} u64 + cnt u32 in elements, then the same in bytes
    const size_t n = (size_t)range_count, n8 = (n + 1) & ~(size_t)1;
    void* tab = nullptr;
    int err = lthip_scratch(ctx, S_XCHG, n8 * 40 + 64, &tab);
    if (err)
        return err;
    uint64_t* d_src = (uint64_t*)tab;
    uint64_t* d_dst = d_src + n8;
    uint64_t* d_src_b = d_dst + n8;
    uint64_t* d_dst_b = d_src_b + n8;
    uint32_t* d_cnt = (uint32_t*)(d_dst_b + n8);
    uint32_t* d_len_b = d_cnt + n8;
    hipStream_t s = ctx->stream;
    if ((err = lthip_stage_upload(ctx, d_src, range_src, n * 8, s)) || (err = lthip_stage_upload(ctx, d_dst, range_dst, n * 8, s)) ||
        (err = lthip_stage_upload(ctx, d_cnt, range_cnt, n * 4, s)))
        return err;
    LaunchTimer t(ctx, LTHIP_K_GATHER);
    hipLaunchKernelGGL(k_scale_ranges, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, d_src, d_dst, d_cnt, (uint64_t)n, elem_bytes,
                       d_src_b, d_dst_b, d_len_b);
    hipLaunchKernelGGL(k_gather_ranges, dim3((uint32_t)n), dim3(GT), 0, s, (const uint8_t*)d_gathered, d_src_b, d_len_b, d_dst_b,
                       (uint64_t)n, (uint8_t*)d_out);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

extern "C" int lthip_job_ordinals(lthip_ctx* ctx, uint64_t my_job_count, const uint32_t* local_first, const uint32_t* global_first,
                                  uint64_t local_chunks, uint32_t* d_out)
{
    if (!ctx || (local_chunks && (!my_job_count || !local_first || !global_first || !d_out)))
        return EINVAL;
    if (local_chunks == 0)
        return 0;
    if (local_chunks > 0x7FFFFFF0ull || my_job_count > 0x7FFFFFF0ull)
        return lthip_fail(ctx, EINVAL, "lthip_job_ordinals", "counts out of range");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t m = (size_t)my_job_count;
    void* tab = nullptr;
    int err = lthip_scratch(ctx, S_XCHG2, m * 8 + 64, &tab);
    if (err)
        return err;
    uint32_t* d_lf = (uint32_t*)tab;
    uint32_t* d_gf = d_lf + m;
    hipStream_t s = ctx->stream;
    if ((err = lthip_stage_upload(ctx, d_lf, local_first, m * 4, s)) || (err = lthip_stage_upload(ctx, d_gf, global_first, m * 4, s)))
        return err;
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_job_ordinals, dim3((uint32_t)((local_chunks + 255) / 256)), dim3(256), 0, s, d_lf, (uint32_t)m, d_gf,
                       (uint32_t)local_chunks, d_out);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

extern "C" int lthip_gather_ranges(lthip_ctx* ctx, const void* d_src, uint64_t range_count, const uint64_t* d_src_offsets,
                                   const uint32_t* d_lens, void* d_dst, const uint64_t* d_dst_offsets)
{
    if (!ctx || (range_count && (!d_src || !d_src_offsets || !d_lens || !d_dst || !d_dst_offsets)))
        return EINVAL;
    if (range_count == 0)
        return 0;
    if (range_count > 0x7FFFFFFFull)
        return lthip_fail(ctx, EINVAL, "gather", "too many ranges");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LaunchTimer t(ctx, LTHIP_K_GATHER);
    hipLaunchKernelGGL(k_gather_ranges, dim3((uint32_t)range_count), dim3(GT), 0, ctx->stream, (const uint8_t*)d_src,
                       d_src_offsets, d_lens, d_dst_offsets, range_count, (uint8_t*)d_dst);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

extern "C" int lthip_link_copy(lthip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src)))
        return EINVAL;
    if (bytes == 0)
        return 0;
    if (((uintptr_t)dst | (uintptr_t)src) & 15u)
        return lthip_fail(ctx, EINVAL, "lthip_link_copy", "source and destination must be 16-byte aligned");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t nvec = (uint64_t)bytes >> 4;
    uint64_t grid = div_up_u64(nvec ? nvec : 1, (uint64_t)GT * 4u);
    if (grid > 1024)
        grid = 1024; // (enough requests in flight for the link; the waves mostly wait, the compute stream's kernels run beside them)
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_link_copy, dim3((uint32_t)grid), dim3(GT), 0, ctx->stream, (const uint4*)src, (uint4*)dst, nvec, (uint32_t)(bytes & 15u));
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
