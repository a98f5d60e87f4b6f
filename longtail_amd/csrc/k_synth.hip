// k_synth.hip -- synthetic asset generator (bench / test input), device side of include/longtail_synth.h.
// Not part of the hot path and not derived from the reference: it only fills HBM with the deterministic
// workload SURVEY.md §8(d) describes so that timed regions start with inputs already resident.
#include "lthip_internal.h"

#include "../../include/longtail_synth.h"

namespace
{

struct SynthAsset
{
    uint64_t off;  // 16-byte aligned
    uint64_t size;
    uint64_t seed;
    uint64_t vec_base; // first 16-byte vector of this asset in the launch's flat index space
    uint64_t vec_skip; // 16-byte vectors of the asset before the range that is generated (a part of a large asset)
};

__global__ __launch_bounds__(256) void k_synth_fill(uint8_t* __restrict__ dst, const SynthAsset* __restrict__ assets,
                                                    uint32_t nassets, uint64_t nvec, int kind)
{
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (uint64_t)gridDim.x * blockDim.x)
    {
        uint32_t lo = 0, hi = nassets;
        while (hi - lo > 1)
        {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (assets[mid].vec_base <= v)
                lo = mid;
            else
                hi = mid;
        }
        const SynthAsset a = assets[lo];
        const uint64_t b = (v - a.vec_base) * 16u;    // byte inside the generated range
        const uint64_t local = v - a.vec_base + a.vec_skip; // vector index inside the asset
        if (b >= a.size)
            continue;
        const uint64_t w0 = lt_synth_word(a.seed, local * 2u, kind);
        const uint64_t w1 = lt_synth_word(a.seed, local * 2u + 1u, kind);
        uint8_t* p = dst + a.off + b;
        if (b + 16u <= a.size)
        {
            *reinterpret_cast<uint4*>(p) = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
        }
        else
        {
            const uint32_t n = (uint32_t)(a.size - b);
            for (uint32_t i = 0; i < n; ++i)
                p[i] = (uint8_t)((i < 8 ? w0 : w1) >> (8u * (i & 7u)));
        }
    }
}

} // namespace

extern "C" int lthip_synth_fill(lthip_ctx* ctx, void* d_dst, uint32_t asset_count, const uint64_t* asset_offsets,
                                const uint64_t* asset_sizes, const uint64_t* asset_seeds, int kind)
{
    return lthip_synth_fill_ranges(ctx, d_dst, asset_count, asset_offsets, asset_sizes, asset_seeds, nullptr, kind);
}

extern "C" int lthip_synth_fill_ranges(lthip_ctx* ctx, void* d_dst, uint32_t asset_count, const uint64_t* asset_offsets,
                                       const uint64_t* asset_sizes, const uint64_t* asset_seeds, const uint64_t* asset_skips, int kind)
{
    if (!ctx || (asset_count && (!d_dst || !asset_offsets || !asset_sizes || !asset_seeds)))
        return EINVAL;
    if (asset_count == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<SynthAsset> h;
    h.reserve(asset_count);
    uint64_t nvec = 0;
    for (uint32_t i = 0; i < asset_count; ++i)
    {
        if (asset_offsets[i] & 15u)
            return lthip_fail(ctx, EINVAL, "lthip_synth_fill", "asset offsets must be 16-byte aligned");
        if (asset_sizes[i] == 0)
            continue;
        SynthAsset a;
        a.off = asset_offsets[i];
        a.size = asset_sizes[i];
        a.seed = asset_seeds[i];
        a.vec_base = nvec;
        a.vec_skip = 0;
        if (asset_skips)
        {
            if (asset_skips[i] & 15u)
                return lthip_fail(ctx, EINVAL, "lthip_synth_fill", "range starts must be multiples of 16 bytes");
            a.vec_skip = asset_skips[i] / 16u;
        }
        nvec += div_up_u64(a.size, 16);
        h.push_back(a);
    }
    if (h.empty())
        return 0;
    void* p;
    int err = lthip_scratch(ctx, S_TABLES, sizeof(SynthAsset) * h.size(), &p);
    if (err)
        return err;
    LTHIP_CHECK(ctx, hipMemcpyAsync(p, h.data(), sizeof(SynthAsset) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    uint64_t blocks = div_up_u64(nvec, 256);
    if (blocks > 256 * 32)
        blocks = 256 * 32;
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_synth_fill, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, (uint8_t*)d_dst,
                       (const SynthAsset*)p, (uint32_t)h.size(), nvec, kind);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
