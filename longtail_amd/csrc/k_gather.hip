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

} // namespace

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
