// Where does global_load_lds_dwordx4's instruction offset go?  (gfx950; build: hipcc --offload-arch=gfx950 -O2 -o glds_probe glds_probe.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned char* p, unsigned* out)
{
    __shared__ unsigned sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64)
        sm[i] = 0xDEAD0000u + i;
    __syncthreads();
    unsigned keep;
    unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned*)sm);
    unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(p)
                 : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64)
        out[i] = sm[i];
}
int main()
{
    unsigned char* d;
    unsigned* o;
    hipMalloc(&d, 65536);
    hipMalloc(&o, 4096 * 4);
    unsigned h[16384];
    for (int i = 0; i < 16384; ++i)
        h[i] = i; // dword i of global memory holds i
    hipMemcpy(d, h, 65536, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    unsigned r[4096];
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 4096; ++i)
        if ((r[i] & 0xFFFF0000u) != 0xDEAD0000u)
        {
            if (first < 0)
                first = i;
            last = i;
        }
    printf("LDS dwords written: [%d, %d]; first holds global dword %u (inst offset 2048 B = dword 512)\n", first, last, first >= 0 ? r[first] : 0);
    return 0;
}
