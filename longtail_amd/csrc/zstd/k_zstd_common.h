// k_zstd_common.h -- what the two zstd translation units share: k_zstd.hip (the encoder) and k_zstd_decode.hip (the decoders).
// The wave-level primitives zstd_block_core.h / zstd_decode_core.h are written against (ZB_FN, ZB_SYNC, scans, ballots), the stored
// block's descriptor, and the FRAME FORMAT both sides agree on: header size, the skippable trailer frames (versions 1-4) that tell the
// decoder how the pieces of a frame depend on each other.
#pragma once
#include <type_traits>

#define ZB_LANES 64u
#define ZB_UNROLL _Pragma("unroll")
#define ZB_FN __device__ __forceinline__ /* inlined so that LDS / global address spaces are known at every access */
#define ZB_SYNC() __syncthreads() /* the encoder runs in one-wave workgroups */
/* LDS traffic of ONE wave is executed in program order: only the compiler has to be kept from moving it */
#define ZB_SYNC_LDS()                                          \
    do                                                         \
    {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)
__device__ __forceinline__ void zb_atomic_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ void zb_atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
// exclusive prefix sum over the 64 lanes of the (single-wave) workgroup, lane 0 first
__device__ __forceinline__ uint32_t zb_scan_excl(uint32_t v, uint32_t* total)
{
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d)
            incl += o;
    }
    *total = __shfl(incl, 63, 64);
    return incl - v;
}
__device__ __forceinline__ uint64_t zb_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
// v of lane `lane` (any lane, also one that sits out a branch: every lane of the wave executes the call)
__device__ __forceinline__ uint32_t zb_shfl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lane << 2), (int)v); }
__device__ __forceinline__ uint32_t zb_reduce_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
    {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = v > o ? v : o;
    }
    return v;
}
#ifdef LTHIP_ZB_PROF /* debug build only: cycles per phase of zb_encode_block, summed over all pieces (lane 0) */
static __device__ unsigned long long g_zb_prof[32]; /* (one per translation unit: the encoder's and the decoder's) */
static __device__ unsigned long long g_zb_last[1 << 16];
#define ZB_MARK(i)                                                                                     \
    do                                                                                                 \
    {                                                                                                  \
        if (threadIdx.x == 0)                                                                          \
        {                                                                                              \
            const unsigned long long now__ = wall_clock64();                                           \
            atomicAdd(&g_zb_prof[i], now__ - g_zb_last[blockIdx.x]);                                   \
            g_zb_last[blockIdx.x] = now__;                                                             \
        }                                                                                              \
    } while (0)
#define ZD_MARK(i) ZB_MARK(i)
#endif
#ifdef K_ZSTD_DECODER /* (k_zstd_decode.hip) */
static __device__ uint32_t g_zd_ablate; /* timing experiments only (LTHIP_ZSTD_ABLATE): 1 = no sequence execution, 2 = no Huffman decode */
#define ZD_ABLATE g_zd_ablate
#include "../zstd_decode_core.h" /* includes zstd_block_core.h */
#include "../origin_exec.h"
#else
#include "../zstd_block_core.h"
#endif

namespace
{

struct ZBlock
{
    uint64_t src_off;
    uint64_t dst_off;
    uint32_t size;
    uint32_t dst_cap;
    uint32_t zb_base; // first 128 KiB piece of this stored block
    uint32_t nzb;
    uint32_t unit_base; // first 4 KiB match-finder unit of this stored block
    uint32_t pad;
};

constexpr uint32_t ZB = ZB_BLOCK_MAX;
static_assert(ZB_BLOCK_MAX == (128u << 10) && ZB_UNIT == 4096u, "k_lz4.hip's Z_PIECE and unit size");
constexpr size_t Z_WORK_SEQS = sizeof(uint64_t) * ZB_SEQ_MAX, Z_WORK_SBITS = sizeof(uint16_t) * 4 * ZB_SEQ_MAX;
constexpr size_t Z_WORK_STRIDE = Z_WORK_SEQS + Z_WORK_SBITS;
constexpr uint32_t ZHDR = 13u;
constexpr int ZT = 256;

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// The pieces (zstd blocks) of a frame written here are independent of each other: matches never leave their 64 KiB window group,
// offsets are never repeat codes, every block carries its own entropy tables.  A frame of two or more pieces says so in a trailing
// SKIPPABLE frame (magic 0x184D2A5D, 4 bytes of data "LTP\1": any zstd decoder skips it, zstd_decompress.c:1068-1085), which lets
// lthip_zstd_decompress_blocks decode the pieces on separate waves.
constexpr uint32_t ZTRAILER = 12u;
__device__ __forceinline__ void z_write_trailer(uint8_t* d)
{
    const uint8_t t[ZTRAILER] = {0x5D, 0x2A, 0x4D, 0x18, 4, 0, 0, 0, 'L', 'T', 'P', 1};
    for (uint32_t i = 0; i < ZTRAILER; ++i)
        d[i] = t[i];
}
__device__ __forceinline__ bool z_is_trailer(const uint8_t* d)
{
    const uint8_t t[ZTRAILER] = {0x5D, 0x2A, 0x4D, 0x18, 4, 0, 0, 0, 'L', 'T', 'P', 1};
    bool same = true;
    for (uint32_t i = 0; i < ZTRAILER; ++i)
        same &= d[i] == t[i];
    return same;
}

// Frames whose pieces are runs of SUB-BLOCKS (zb_encode_piece_sub; one zstd block per 4 KiB unit, entropy tables sent once per piece)
// end with a skippable frame that also carries a DIRECTORY: "LTP\2", then one u16 per 4 KiB unit of the content = the content size of
// the unit's block (| 0x8000: a Raw_Block); 0xFFFF / 0xFFFE for every unit of a piece that is one Raw_Block / one RLE_Block.  With it
// the decoder finds every block of the frame by prefix sums (and then checks each against its header) instead of walking 2 048
// headers per 8 MiB one after the other.
constexpr uint16_t ZDIR_RAW_PIECE = 0xFFFFu, ZDIR_RLE_PIECE = 0xFFFEu;
__host__ __device__ __forceinline__ uint32_t z_units(uint64_t content) { return (uint32_t)((content + ZB_UNIT - 1u) / ZB_UNIT); }
__host__ __device__ __forceinline__ uint32_t z_trailer2_size(uint64_t content) { return ZTRAILER + 2u * z_units(content); }
// version 2: plain offsets only; version 3 (round 4, LTHIP_ZSTD_REP=1): blocks may use repeat-offset codes for history entries set inside
// the block (zb_encode_piece_sub, ZB_F_REPCODES) -- the lane decoder then carries a block-local history (zs_seq_lanes<2>), which costs it
// 5-9 % (321 -> 294 GB/s on "mixed"): frames say which they are so that the others keep the cheaper loop
__device__ __forceinline__ void z_write_trailer2_head(uint8_t* d, uint64_t content, uint32_t version)
{
    const uint32_t n = 4u + 2u * z_units(content);
    const uint8_t t[ZTRAILER] = {0x5D, 0x2A, 0x4D, 0x18, (uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24), 'L', 'T', 'P', (uint8_t)version};
    for (uint32_t i = 0; i < ZTRAILER; ++i)
        d[i] = t[i];
}
// version 4 (round 5, the "max" setting): the match finder gave every redundant half the 32 KiB in front of it as history, also a piece's
// first half -- matches reach into the piece before -- except in every ZCHAIN-th piece of the frame: pieces k ZCHAIN .. k ZCHAIN + 7 are
// a CHAIN for the decoder (a piece is executed when the one before it is complete), the chains of a frame are independent of each other.
// (One chain per frame was measured first: a frame of 64 pieces then decodes in 14-40 ms however many waves idle -- 100 / 78 GB/s on
// mixed / tokens at 512 blocks, 0.2-0.6 GB/s for one block; chains of eight keep 7/8 of the ratio gain.)
constexpr uint32_t ZCHAIN = LTHIP_ZSTD_CHAIN;
// 0: not a directory trailer; else its version (2, 3 or 4)
__device__ __forceinline__ uint32_t z_is_trailer2_head(const uint8_t* d, uint64_t content)
{
    const uint32_t n = 4u + 2u * z_units(content);
    const uint8_t t[ZTRAILER] = {0x5D, 0x2A, 0x4D, 0x18, (uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24), 'L', 'T', 'P', 2};
    bool same = true;
    for (uint32_t i = 0; i + 1u < ZTRAILER; ++i)
        same &= d[i] == t[i];
    const uint32_t ver = d[ZTRAILER - 1u];
    return same && (ver >= 2u && ver <= 4u) ? ver : 0u;
}
// a workgroup of ZT threads copies n bytes, 16-byte stores, source of any alignment
__device__ __forceinline__ void wg_copy16(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, int tid)
{
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > n)
        head = n;
    if ((uint32_t)tid < head)
        dst[tid] = src[tid];
    dst += head;
    src += head;
    n -= head;
    const uint32_t nvec = n >> 4;
    const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t sh = mis * 8u;
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src - mis);
    for (uint32_t v = tid; v < nvec; v += ZT)
    {
        const uint32_t* q = s4 + v * 4u;
        const u32x4_a4 a = *reinterpret_cast<const u32x4_a4*>(q);
        const uint32_t e = mis ? q[4] : 0u;
        uint4 o;
        o.x = __builtin_amdgcn_alignbit(a.y, a.x, sh);
        o.y = __builtin_amdgcn_alignbit(a.z, a.y, sh);
        o.z = __builtin_amdgcn_alignbit(a.w, a.z, sh);
        o.w = __builtin_amdgcn_alignbit(e, a.w, sh);
        *reinterpret_cast<uint4*>(dst + (uint64_t)v * 16u) = o;
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)tid < n - done)
        dst[done + tid] = src[done + tid];
}

} // namespace
