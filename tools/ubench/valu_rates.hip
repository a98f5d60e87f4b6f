// Micro-benchmark: issue rate of the integer VALU instructions the hot path is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define UNROLL 16
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 7 + i;
    uint32_t b = seed ^ 0x9e3779b9u, c = seed * 3 + 1;
    for (int it = 0; it < N_ITER; ++it)
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            uint32_t& x = a[u & 7];
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "v"(b));
            if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));
            if (OP == 2) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 4) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(b));
            if (OP == 5) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x));
            if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 7) { float f = __uint_as_float(x); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(__uint_as_float(b)), "v"(__uint_as_float(c))); x = __float_as_uint(f); }
            if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 10) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
            if (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 12) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
            if (OP == 13) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x));
            if (OP == 14) asm volatile("v_or_b32 %0, %1, %0" : "+v"(x) : "v"(b));
            if (OP == 15) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(b));
            if (OP == 16) asm volatile("v_min_u32 %0, %1, %0" : "+v"(x) : "v"(b));
            if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
            if (OP == 18) asm volatile("v_xor_b32_sdwa %0, %1, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(x) : "v"(b));
            if (OP == 19) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
            if (OP == 20) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 21) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 22) asm volatile("v_alignbyte_b32 %0, %0, %0, 1" : "+v"(x));
            if (OP == 23) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 24) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            if (OP == 25) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
            if (OP == 26) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(b));
            if (OP == 27) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 28) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            if (OP == 29) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(x) : "v"(b));
            if (OP == 30) asm volatile("v_xor_b32_dpp %0, %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b));
            if (OP == 31) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(x) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char* name, uint32_t* d)
{
    const int blocks = 256 * 8; // 8 workgroups of 256 per CU = 8 waves / SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * N_ITER * UNROLL;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-16s %8.3f ms  %7.1f G wave-instr/s/SIMD  => %.2f cycles per wave64 instruction at 2.4 GHz, %.1f T lane-ops/s\n", name, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, wave_instr * 64 / (ms * 1e-3) / 1e12);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", d); run<1>("v_alignbit_b32", d); run<2>("v_add3_u32", d); run<3>("v_mul_lo_u32", d);
    run<4>("v_lshl_add_u32", d); run<5>("v_bfe_u32", d); run<6>("v_add_u32", d); run<7>("v_fma_f32", d);
    run<8>("v_perm_b32", d); run<9>("v_and_or_b32", d); run<10>("v_cmp_le_u32", d); run<11>("v_mul_u32_u24", d);
    run<12>("v_lshlrev_b32", d); run<13>("v_lshrrev_b32", d); run<14>("v_or_b32", d); run<15>("v_and_b32", d);
    run<16>("v_min_u32", d); run<17>("v_cndmask_b32", d); run<18>("v_xor_b32_sdwa", d); run<19>("v_mov_b32_dpp", d);
    run<20>("v_or3_b32", d); run<21>("v_xad_u32", d); run<22>("v_alignbyte_b32", d); run<23>("v_pk_add_u16", d);
    run<24>("v_sub_u32", d); run<25>("v_mov_b32", d); run<26>("v_lshl_or_b32", d); run<27>("v_bfi_b32", d);
    run<28>("v_mad_u32_u24", d); run<29>("v_add_u32_sdwa", d); run<30>("v_xor_b32_dpp", d); run<31>("v_lshlrev_sdwa", d);
    return 0;
}
