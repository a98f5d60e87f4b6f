// Micro-benchmark, second set (round 3): instructions the prefix-XOR K1 and its candidate reformulations are made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates2.hip -o tools/ubench/valu_rates2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define UNROLL 16
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    uint32_t a[8];
    uint64_t w[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; w[i] = a[i]; }
    uint32_t b = seed ^ 0x9e3779b9u, c = seed * 3 + 1;
    uint64_t c64 = c;
    for (int it = 0; it < N_ITER; ++it)
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            uint32_t& x = a[u & 7];
            uint64_t& y = w[u & 7];
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "+v"(y) : "v"(b), "v"(c), "v"(c64) : "vcc");
            if (OP == 1) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 2) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_xor_b32_dpp %0, %1, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
            if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 6) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(y) : "v"(c64));
            if (OP == 7) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "s"(seed));
            if (OP == 9) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(seed));
            if (OP == 10) asm volatile("v_xor_b32 %0, %1, %0\n\tv_alignbit_b32 %0, %0, %0, 5" : "+v"(x) : "v"(b));
            if (OP == 11) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 12) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 13) asm volatile("v_bfi_b32 %0, %0, 0, %1" : "+v"(x) : "s"(seed));
            if (OP == 14) asm volatile("v_xor_b32 %0, %1, %0\n\tv_xor_b32 %0, %2, %0" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 15) asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_add_u32 %0, %1, %0" : "+v"(x) : "s"(seed));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char* name, uint32_t* d, int per = 1)
{
    const int blocks = 256 * 8; // 8 workgroups of 256 per CU = 8 waves / SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * N_ITER * UNROLL * per;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-26s %8.3f ms  => %.2f cycles per wave64 instruction at 2.4 GHz (%d per step)\n", name, ms, 2.4e9 / per_simd_per_s, per);
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_mad_u64_u32", d); run<1>("v_min3_u32", d); run<2>("v_max3_u32", d); run<3>("v_xor_b32_dpp wave_ror:1", d);
    run<4>("v_mul_hi_u32", d); run<5>("v_alignbit_b32 (vgpr amt)", d); run<6>("v_lshl_add_u64", d); run<7>("v_mov_b32_dpp wave_shr:1", d);
    run<8>("v_mul_lo_u32 (sgpr)", d); run<9>("v_add_u32 (sgpr)", d); run<10>("xor + alignbit", d, 2); run<11>("v_pk_mul_lo_u16", d);
    run<12>("v_mad_u32_u16", d); run<13>("v_bfi_b32 (sgpr)", d); run<14>("xor + xor", d, 2); run<15>("mul_lo + add", d, 2);
    return 0;
}
