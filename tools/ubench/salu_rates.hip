// Micro-benchmark: issue rate of scalar instructions per CU (gfx950): how many SALU instructions per cycle a CU retires with 16 waves
// resident, for the instructions the decoders' scalar loops are made of.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/salu_rates.hip -o gpurun_out/salu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define UNROLL 16
template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t seed)
{
    uint32_t a = __builtin_amdgcn_readfirstlane(seed + blockIdx.x), b = __builtin_amdgcn_readfirstlane(seed ^ 0x9e3779b9u);
    uint64_t w = ((uint64_t)a << 32) | b, w2 = ~w;
    uint32_t sh = __builtin_amdgcn_readfirstlane(seed & 31u);
    for (int it = 0; it < N_ITER; ++it)
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            uint32_t& x = (u & 1) ? a : b;
            uint64_t& y = (u & 1) ? w : w2;
            if (OP == 0) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(sh) : "scc");
            if (OP == 1) asm volatile("s_lshl_b32 %0, %0, %1" : "+s"(x) : "s"(sh) : "scc");
            if (OP == 2) asm volatile("s_lshl_b64 %0, %0, %1" : "+s"(y) : "s"(sh) : "scc");
            if (OP == 3) asm volatile("s_lshr_b64 %0, %0, %1" : "+s"(y) : "s"(sh) : "scc");
            if (OP == 4) asm volatile("s_bfe_u32 %0, %0, 0x80008" : "+s"(x) : : "scc");
            if (OP == 5) asm volatile("s_bfe_u64 %0, %0, 0x200008" : "+s"(y) : : "scc");
            if (OP == 6) asm volatile("s_cmp_lt_u32 %0, %1\n\ts_cselect_b32 %0, %0, %1" : "+s"(x) : "s"(sh) : "scc");
            if (OP == 7) asm volatile("s_and_b64 %0, %0, %1" : "+s"(y) : "s"(w2) : "scc");
            if (OP == 8) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(x) : "s"(sh));
            if (OP == 9) asm volatile("s_nop 0");
        }
    }
    if (threadIdx.x == 0)
        out[blockIdx.x] = a ^ b ^ (uint32_t)w ^ (uint32_t)(w2 >> 32);
}
template <int OP> void run(const char* name, uint32_t* d, int waves_per_cu, int per)
{
    const int blocks = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)waves_per_cu * N_ITER * UNROLL * per;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %2d waves/CU  %8.3f ms  => %5.2f cycles per instruction per CU (2.4 GHz)\n", name, waves_per_cu, ms, cycles / instr_per_cu);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 1 << 20);
    for (int w : {4, 16})
    {
        run<0>("s_add_u32", d, w, 1);
        run<1>("s_lshl_b32", d, w, 1);
        run<2>("s_lshl_b64", d, w, 1);
        run<3>("s_lshr_b64", d, w, 1);
        run<4>("s_bfe_u32", d, w, 1);
        run<5>("s_bfe_u64", d, w, 1);
        run<6>("s_cmp + s_cselect", d, w, 2);
        run<7>("s_and_b64", d, w, 1);
        run<8>("s_mul_i32", d, w, 1);
        run<9>("s_nop 0", d, w, 1);
    }
    return 0;
}
