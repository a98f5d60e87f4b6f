// Micro-benchmark, third set (round 3): does the ORDER of "fast" (VOP2, VGPR operands: v_xor_b32, v_add_u32 -- 2.5-2.8 cycles) and
// "slow" (VOP3: v_alignbit_b32, v_add3_u32 -- 4.2-4.3 cycles) instructions matter?  BLAKE3's G is 6 + 6 of them.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates3.hip -o tools/ubench/valu_rates3 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define X(r) "v_xor_b32 %" #r ", %8, %" #r "\n\t"
#define A(r) "v_add_u32 %" #r ", %8, %" #r "\n\t"
#define R(r) "v_alignbit_b32 %" #r ", %" #r ", %" #r ", 7\n\t"
#define T(r) "v_add3_u32 %" #r ", %" #r ", %8, %9\n\t"
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const uint32_t b = seed ^ 0x9e3779b9u, c = seed * 3 + 1;
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
    for (int it = 0; it < N_ITER; ++it)
    {
        if (OP == 0) asm volatile(X(0) R(0) X(1) R(1) X(2) R(2) X(3) R(3) X(4) R(4) X(5) R(5) X(6) R(6) X(7) R(7) : OPS);           // alternating
        if (OP == 1) asm volatile(X(0) X(1) R(0) R(1) X(2) X(3) R(2) R(3) X(4) X(5) R(4) R(5) X(6) X(7) R(6) R(7) : OPS);           // pairs
        if (OP == 2) asm volatile(X(0) X(1) X(2) X(3) R(0) R(1) R(2) R(3) X(4) X(5) X(6) X(7) R(4) R(5) R(6) R(7) : OPS);           // fours
        if (OP == 3) asm volatile(X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) : OPS);           // eights
        if (OP == 4) asm volatile(X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) : OPS);           // xor only
        if (OP == 5) asm volatile(R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) : OPS);           // alignbit only
        if (OP == 6) asm volatile(T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) : OPS);           // add3 only
        if (OP == 7) asm volatile(A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) : OPS);           // add only
        if (OP == 8) asm volatile(A(0) X(1) A(2) X(3) A(4) X(5) A(6) X(7) A(0) X(1) A(2) X(3) A(4) X(5) A(6) X(7) : OPS);           // add, xor
        if (OP == 11) asm volatile(X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) : OPS); // runs of 16
        if (OP == 12) asm volatile(X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
                                   R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) : OPS); // runs of 32
        if (OP == 13) asm volatile(X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) R(7) : OPS); // 31 fast, 1 slow
        // one BLAKE3 G on four columns, the compiler's kind of order (fast and slow interleaved) against grouped by kind
        if (OP == 9) asm volatile(T(0) X(4) T(1) R(4) X(5) T(2) R(5) X(6) T(3) R(6) X(7) A(0) R(7) A(1) X(4) A(2) : OPS);
        if (OP == 10) asm volatile(T(0) T(1) T(2) T(3) X(4) X(5) X(6) X(7) R(4) R(5) R(6) R(7) A(0) A(1) A(2) A(3) : OPS);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char* name, uint32_t* d, int per = 16)
{
    const int blocks = 256 * 8; // 8 workgroups of 256 per CU = 8 waves / SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * N_ITER * per;
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-44s %8.3f ms  => %.2f cycles per wave64 instruction at 2.4 GHz\n", name, ms, 2.4e9 / per_simd_per_s);
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("xor, alignbit alternating", d); run<1>("xor x2, alignbit x2", d); run<2>("xor x4, alignbit x4", d); run<3>("xor x8, alignbit x8", d);
    run<4>("xor only", d); run<5>("alignbit only", d); run<6>("add3 only", d); run<7>("add only", d); run<8>("add, xor alternating", d);
    run<11>("xor x16, alignbit x16", d, 32); run<12>("xor x32, alignbit x32", d, 64); run<13>("xor x31, alignbit x1", d, 32);
    run<9>("G on 4 columns, interleaved by kind", d); run<10>("G on 4 columns, grouped by kind", d);
    return 0;
}
