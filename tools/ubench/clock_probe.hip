// Does the shader clock depend on how busy the chip is?  A latency-bound kernel (one wave per workgroup, a dependent chain of
// LDS reads) is run with few and with many workgroups; s_memtime (shader clock) against s_memrealtime (100 MHz) gives the
// clock each wave actually ran at, the chain length gives cycles per dependent LDS round trip.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_probe.hip -o build/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void chain(uint32_t* out, unsigned long long* clk, int iters)
{
    __shared__ uint32_t s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64)
        s[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < iters; ++i)
        x = s[x]; // dependent LDS read
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0)
    {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}
int main()
{
    const int iters = 2000000;
    uint32_t* d;
    unsigned long long* c;
    hipMalloc(&d, 65536 * 64 * 4);
    hipMalloc(&c, 65536 * 16);
    for (int blocks : {256, 512, 2048, 8192, 512})
    {
        chain<<<blocks, 64>>>(d, c, iters);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        const double sec = (double)h[1] / 1e8;
        printf("%5d workgroups of one wave: %.1f ms, s_memtime %.0f MHz-equivalent, %.1f ns = %.0f shader ticks per dependent LDS read\n", blocks,
               sec * 1e3, (double)h[0] / sec / 1e6, sec / iters * 1e9, (double)h[0] / iters);
    }
    return 0;
}
