// What a streaming copy reaches on this device, by how it is written: plain 16-byte loads / stores, non-temporal stores, non-temporal
// loads + stores, at several grid sizes and unroll depths -- the roof of the classification pass on incompressible data (N read + N
// written) and of the stitch copy.  Reports GB/s of read + write.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/copy_rates.hip -o build/copy_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t nvec)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; v + (U - 1) * stride < nvec; v += U * stride)
    {
        u32x4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            r[u] = MODE >= 2 ? __builtin_nontemporal_load(src + v + u * stride) : src[v + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (MODE >= 1)
                __builtin_nontemporal_store(r[u], dst + v + u * stride);
            else
                dst[v + u * stride] = r[u];
        }
    }
    for (; v < nvec; v += stride)
        dst[v] = src[v];
}
// contiguous tile per workgroup (what a block-structured kernel does): workgroup b copies tile b, b + grid, ...
template <int MODE>
__global__ __launch_bounds__(256) void k_copy_tiles(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t nvec, uint32_t tile_vec)
{
    for (uint64_t t = blockIdx.x; t * tile_vec < nvec; t += gridDim.x)
    {
        const uint64_t b = t * tile_vec;
        for (uint32_t i = threadIdx.x; i < tile_vec && b + i < nvec; i += 256 * 4)
        {
            u32x4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * 256 < tile_vec && b + i + u * 256 < nvec)
                    r[u] = MODE >= 2 ? __builtin_nontemporal_load(src + b + i + u * 256) : src[b + i + u * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * 256 < tile_vec && b + i + u * 256 < nvec)
                {
                    if (MODE >= 1)
                        __builtin_nontemporal_store(r[u], dst + b + i + u * 256);
                    else
                        dst[b + i + u * 256] = r[u];
                }
        }
    }
}
template <typename F>
static double best_ms(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < 5; ++i)
    {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    return best;
}
int main()
{
    const uint64_t bytes = 8ull << 30, nvec = bytes / 16;
    u32x4 *s, *d;
    hipMalloc(&s, bytes);
    hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes);
    hipMemset(d, 2, bytes);
    printf("hipMemcpy D2D: %.0f GB/s\n", 2.0 * bytes / best_ms([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }) / 1e6);
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536})
    {
        printf("grid %6d  grid-stride U=4: plain %.0f  nt-store %.0f  nt-load+store %.0f | U=8: plain %.0f nt-store %.0f nt-both %.0f | tiles of 64 KiB: plain %.0f nt-store %.0f nt-both %.0f GB/s\n", grid,
               2.0 * bytes / best_ms([&] { k_copy<0, 4><<<grid, 256>>>(s, d, nvec); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy<1, 4><<<grid, 256>>>(s, d, nvec); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy<2, 4><<<grid, 256>>>(s, d, nvec); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy<0, 8><<<grid, 256>>>(s, d, nvec); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy<1, 8><<<grid, 256>>>(s, d, nvec); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy<2, 8><<<grid, 256>>>(s, d, nvec); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy_tiles<0><<<grid, 256>>>(s, d, nvec, 4096); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy_tiles<1><<<grid, 256>>>(s, d, nvec, 4096); }) / 1e6,
               2.0 * bytes / best_ms([&] { k_copy_tiles<2><<<grid, 256>>>(s, d, nvec, 4096); }) / 1e6);
    }
    return 0;
}
