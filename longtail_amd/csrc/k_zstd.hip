// k_zstd.hip -- ZStd CompressionAPI, GPU encoder.  Every stored block becomes ONE zstd frame
// (RFC 8878 §3.1.1: magic 0xFD2FB528 | FHD 0xE0 = single segment, 8-byte content size, no checksum, no dictionary |
// u64 content size | blocks, each with a 3-byte header {last:1, type:2, size:21}) whose 128 KiB pieces are
//   RLE_Block         all bytes equal (per-unit flags from the match finder, checked in k_zstd_encode)
//   Compressed_Block  LZ sequences from the LZ4 match finder run with sequence output (k_lz4.hip, FMT 1), literals
//                     Huffman-coded in 4 streams, the three symbol streams FSE-coded (predefined / RLE / described
//                     tables) -- zstd_block_core.h, one wavefront per piece (k_zstd_encode)
//   Raw_Block         whenever that would not be smaller.
// The reference's ZSTD_decompressDCtx (lib/zstd/longtail_zstd.c:144-177) decodes these like any other frame; the
// settings ('ztd1'..'ztd5', longtail_zstd.c:12-22) select nothing here: there is one parse.
#include "lthip_internal.h"

#define ZB_LANES 64u
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
#ifdef LTHIP_ZB_PROF /* debug build only: cycles per phase of zb_encode_block, summed over all pieces (lane 0) */
__device__ unsigned long long g_zb_prof[16];
__device__ unsigned long long g_zb_last[1 << 16];
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
#include "zstd_decode_core.h" /* includes zstd_block_core.h */

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
constexpr size_t Z_WORK_SEQS = sizeof(uint64_t) * ZB_SEQ_MAX, Z_WORK_SBITS = sizeof(uint16_t) * 3 * ZB_SEQ_MAX;
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

// serial per stored block: destination offset of every piece, total size
__global__ void k_zstd_scan(const ZBlock* __restrict__ blocks, uint32_t nblocks, const uint8_t* __restrict__ is_rle,
                            const uint32_t* __restrict__ enc_size, uint32_t* __restrict__ zb_dst,
                            uint32_t* __restrict__ out_sizes, uint8_t* __restrict__ dst, uint32_t dbg)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks)
        return;
    const ZBlock blk = blocks[b];
    uint64_t pos = ZHDR;
    if (blk.nzb == 0)
        pos += 3; // one empty raw block
    for (uint32_t i = 0; i < blk.nzb; ++i)
    {
        const uint32_t len = blk.size - i * ZB < ZB ? blk.size - i * ZB : ZB;
        zb_dst[blk.zb_base + i] = (uint32_t)pos;
        const uint32_t enc = enc_size[blk.zb_base + i];
        pos += 3u + ((is_rle[blk.zb_base + i] & 1u) ? 1u : enc ? enc : len);
    }
    if (blk.nzb >= 2u && pos + ZTRAILER <= (uint64_t)blk.dst_cap && !(dbg & 2u))
    {
        z_write_trailer(dst + blk.dst_off + pos);
        pos += ZTRAILER;
    }
    out_sizes[b] = pos <= (uint64_t)blk.dst_cap ? (uint32_t)pos : 0u;
}

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

__global__ __launch_bounds__(ZT) void k_zstd_emit(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks,
                                                  uint32_t nblocks, const uint8_t* __restrict__ is_rle,
                                                  const uint32_t* __restrict__ enc_size, const uint8_t* __restrict__ enc,
                                                  const uint32_t* __restrict__ zb_dst, const uint32_t* __restrict__ out_sizes,
                                                  uint8_t* __restrict__ dst)
{
    const uint32_t zb = blockIdx.x;
    uint32_t lo = 0, hi = nblocks;
    while (hi - lo > 1)
    {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (blocks[mid].zb_base <= zb)
            lo = mid;
        else
            hi = mid;
    }
    const ZBlock b = blocks[lo];
    if (out_sizes[lo] == 0)
        return;
    const uint32_t i = zb - b.zb_base;
    const uint32_t start = i * ZB;
    const uint32_t len = b.size - start < ZB ? b.size - start : ZB;
    const uint8_t* p = src + b.src_off + start;
    uint8_t* d = dst + b.dst_off + zb_dst[zb];
    const uint32_t flags = is_rle[zb]; // 1 = RLE_Block, 2 = every unit without a sequence (bytes placed by the match finder)
    const uint32_t rle = flags & 1u;
    const uint32_t csize = rle ? 0u : enc_size[zb];
    if (threadIdx.x == 0)
    {
        const uint32_t h = (i + 1 == b.nzb ? 1u : 0u) | ((csize ? 2u : rle) << 1) | ((csize ? csize : len) << 3);
        d[0] = (uint8_t)h;
        d[1] = (uint8_t)(h >> 8);
        d[2] = (uint8_t)(h >> 16);
        if (rle)
            d[3] = p[0];
    }
    if (csize)
        wg_copy16(d + 3, enc + (uint64_t)zb * ZB_OUT_BYTES, csize, threadIdx.x);
    else if (!rle && !((flags & 2u) && zb_dst[zb] == ZHDR + i * (ZB + 3u)))
        wg_copy16(d + 3, p, len, threadIdx.x); // (otherwise the Raw_Block's bytes are already in place, k_lz4.hip)
}

// One wavefront per 128 KiB piece, persistent over the pieces: the entropy stage of zstd_block_core.h.
__global__ __launch_bounds__(64, 4) void k_zstd_encode(const ZBlock* __restrict__ blocks, uint32_t nblocks, uint32_t npieces,
                                                    const uint8_t* __restrict__ src, uint8_t* __restrict__ is_rle,
                                                    const ZbUnitMeta* __restrict__ unit_meta,
                                                    const uint8_t* __restrict__ unit_lits, const uint64_t* __restrict__ unit_recs,
                                                    uint8_t* __restrict__ work, uint8_t* __restrict__ enc,
                                                    uint32_t* __restrict__ enc_size)
{
    __shared__ ZbShared sh;
    ZbScratch sc;
    uint8_t* w = work + (uint64_t)blockIdx.x * Z_WORK_STRIDE;
    sc.seqs = reinterpret_cast<uint64_t*>(w);
    sc.sbits = reinterpret_cast<uint16_t*>(w + Z_WORK_SEQS);
    for (uint32_t zb = blockIdx.x; zb < npieces; zb += gridDim.x)
    {
        uint32_t lo = 0, hi = nblocks;
        while (hi - lo > 1)
        {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (blocks[mid].zb_base <= zb)
                lo = mid;
            else
                hi = mid;
        }
        const ZBlock b = blocks[lo];
        const uint32_t i = zb - b.zb_base;
        const uint32_t len = b.size - i * ZB < ZB ? b.size - i * ZB : ZB;
        const uint64_t u0 = (uint64_t)b.unit_base + (uint64_t)i * ZB_MAX_UNITS;
        ZbInput in;
        in.meta = unit_meta + u0;
        in.unit_lits = unit_lits + u0 * ZB_UNIT;
        in.unit_recs = unit_recs + u0 * ZB_UNIT_SEQ_MAX;
        in.nunits = (len + ZB_UNIT - 1u) / ZB_UNIT;
        in.raw_size = len;
        in.src = src + b.src_off + (uint64_t)i * ZB;
        {
            // RLE_Block: every unit of the piece is one repeated byte (flagged by the match finder) and it is the same one
            const uint32_t f = threadIdx.x < in.nunits ? in.meta[threadIdx.x].uniform : in.meta[0].uniform;
            const uint32_t f0 = __builtin_amdgcn_readfirstlane(f);
            const bool rle = f0 != 0u && __builtin_amdgcn_ballot_w64(f != f0) == 0ull;
            const bool matchless = __builtin_amdgcn_ballot_w64(threadIdx.x < in.nunits && in.meta[threadIdx.x].nseq != 0u) == 0ull;
            if (threadIdx.x == 0)
            {
                is_rle[zb] = (rle ? 1 : 0) | (matchless ? 2 : 0);
                if (rle)
                    enc_size[zb] = 0;
            }
            if (rle)
                continue;
        }
        sc.out = reinterpret_cast<uint32_t*>(enc + (uint64_t)zb * ZB_OUT_BYTES);
#ifdef LTHIP_ZB_PROF
        if (threadIdx.x == 0)
            g_zb_last[blockIdx.x] = wall_clock64();
#endif
        const uint32_t n = zb_encode_block(&in, &sc, &sh, threadIdx.x);
        if (threadIdx.x == 0)
            enc_size[zb] = n;
        __syncthreads();
    }
}

// frame headers (and the lone empty block of empty inputs)
__global__ void k_zstd_headers(const ZBlock* __restrict__ blocks, uint32_t nblocks, const uint32_t* __restrict__ out_sizes,
                               uint8_t* __restrict__ dst)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks || out_sizes[b] == 0)
        return;
    const ZBlock blk = blocks[b];
    uint8_t* d = dst + blk.dst_off;
    d[0] = 0x28;
    d[1] = 0xB5;
    d[2] = 0x2F;
    d[3] = 0xFD;
    d[4] = 0xE0;
    uint64_t n = blk.size;
    for (int i = 0; i < 8; ++i)
        d[5 + i] = (uint8_t)(n >> (8 * i));
    if (blk.nzb == 0)
    {
        d[13] = 1; // last block, raw, size 0
        d[14] = 0;
        d[15] = 0;
    }
}

} // namespace

// ZSTD_COMPRESSBOUND, lib/zstd/ext/zstd.h:231-232
extern "C" size_t lthip_zstd_bound(size_t n)
{
    if (n >= 0xFF00FF00FF00FF00ull)
        return 0;
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

static int zstd_compress_batch(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                               const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                               uint32_t* d_out_sizes);

extern "C" int lthip_zstd_compress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                                          const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets,
                                          const uint32_t* dst_caps, uint32_t* d_out_sizes)
{
    if (!ctx || !d_out_sizes || (block_count && (!src_offsets || !src_sizes || !dst_offsets || !dst_caps || !d_dst)))
        return EINVAL;
    const uint64_t budget = lthip_codec_batch_bytes(); // scratch (unit literals + records + piece slots): about three times that
    for (uint32_t b0 = 0; b0 < block_count;)
    {
        uint64_t bytes = src_sizes[b0];
        uint32_t b1 = b0 + 1;
        while (b1 < block_count && bytes + src_sizes[b1] <= budget)
            bytes += src_sizes[b1++];
        const int err = zstd_compress_batch(ctx, d_src, b1 - b0, src_offsets + b0, src_sizes + b0, d_dst, dst_offsets + b0, dst_caps + b0,
                                            d_out_sizes + b0);
        if (err)
            return err;
        b0 = b1;
    }
    return 0;
}

static int zstd_compress_batch(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                               const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                               uint32_t* d_out_sizes)
{
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<ZBlock> hb(block_count);
    std::vector<uint32_t> unit_base(block_count);
    uint64_t nzb = 0;
    for (uint32_t b = 0; b < block_count; ++b)
    {
        hb[b].src_off = src_offsets[b];
        hb[b].dst_off = dst_offsets[b];
        hb[b].size = src_sizes[b];
        hb[b].dst_cap = dst_caps[b];
        hb[b].zb_base = (uint32_t)nzb;
        hb[b].nzb = (uint32_t)div_up_u64(src_sizes[b], ZB);
        hb[b].pad = 0;
        nzb += hb[b].nzb;
    }
    if (nzb > 0x7FFFF0ull)
        return lthip_fail(ctx, EINVAL, "zstd", "batch too large");
    // ---- match finder: sequences + literals per 4 KiB unit ----
    uint8_t* d_lits = nullptr;
    uint64_t* d_recs = nullptr;
    void* d_meta = nullptr;
    uint64_t nunits = 0;
    int err;
    if ((err = lthip_launch_lz_sequences(ctx, d_src, block_count, src_offsets, src_sizes, d_dst, dst_offsets, dst_caps, &d_lits, &d_recs,
                                         &d_meta, unit_base.data(), &nunits)))
        return err;
    for (uint32_t b = 0; b < block_count; ++b)
        hb[b].unit_base = unit_base[b];
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const uint32_t nwg = (uint32_t)(nzb < (uint64_t)ncu * 16 ? nzb : (uint64_t)ncu * 16);
    void *d_blocks, *d_rle, *d_zdst, *d_enc, *d_encsz, *d_work;
    if ((err = lthip_scratch(ctx, S_LZ4_BLOCKS, sizeof(ZBlock) * (size_t)block_count, &d_blocks)))
        return err;
    if ((err = lthip_scratch(ctx, S_TABLES2, (size_t)nzb + 16, &d_rle)))
        return err;
    if ((err = lthip_scratch(ctx, S_LZ4_SEGS, ((size_t)nzb + 4) * 4, &d_zdst)))
        return err;
    if ((err = lthip_scratch(ctx, S_MISC, ((size_t)nzb + 4) * 4, &d_encsz)))
        return err;
    if ((err = lthip_scratch(ctx, S_Z_ENC, (size_t)ZB_OUT_BYTES * ((size_t)nzb + 1), &d_enc)))
        return err;
    if ((err = lthip_scratch(ctx, S_Z_WORK, Z_WORK_STRIDE * ((size_t)nwg + 1), &d_work)))
        return err;
    if ((err = lthip_stage_upload(ctx, d_blocks, hb.data(), sizeof(ZBlock) * (size_t)block_count, ctx->stream)))
        return err;
    if (nzb)
    {
        LaunchTimer t(ctx, LTHIP_K_ZSTD_ENC);
        hipLaunchKernelGGL(k_zstd_encode, dim3(nwg), dim3(64), 0, ctx->stream, (const ZBlock*)d_blocks, block_count, (uint32_t)nzb,
                           (const uint8_t*)d_src, (uint8_t*)d_rle, (const ZbUnitMeta*)d_meta, (const uint8_t*)d_lits, (const uint64_t*)d_recs,
                           (uint8_t*)d_work, (uint8_t*)d_enc, (uint32_t*)d_encsz);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_zstd_scan, dim3((block_count + 63) / 64), dim3(64), 0, ctx->stream, (const ZBlock*)d_blocks,
                       block_count, (const uint8_t*)d_rle, (const uint32_t*)d_encsz, (uint32_t*)d_zdst, d_out_sizes, (uint8_t*)d_dst,
                       (uint32_t)(getenv("LTHIP_ZSTD_DBG") ? atoi(getenv("LTHIP_ZSTD_DBG")) : 0)); // bit 1: no independence marker
    hipLaunchKernelGGL(k_zstd_headers, dim3((block_count + 63) / 64), dim3(64), 0, ctx->stream, (const ZBlock*)d_blocks,
                       block_count, (const uint32_t*)d_out_sizes, (uint8_t*)d_dst);
    if (nzb)
        hipLaunchKernelGGL(k_zstd_emit, dim3((uint32_t)nzb), dim3(ZT), 0, ctx->stream, (const uint8_t*)d_src,
                           (const ZBlock*)d_blocks, block_count, (const uint8_t*)d_rle, (const uint32_t*)d_encsz,
                           (const uint8_t*)d_enc, (const uint32_t*)d_zdst, (const uint32_t*)d_out_sizes, (uint8_t*)d_dst);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

#ifdef LTHIP_ZB_PROF
extern "C" __attribute__((visibility("default"))) int lthip_zb_prof_dump(void)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_zb_prof), sizeof(h)) != hipSuccess)
        return -1;
    for (int i = 0; i < 16; ++i)
        if (h[i])
            fprintf(stderr, "zb phase ending at mark %2d: %.3f ms wave-time (100 MHz clock)\n", i, (double)h[i] / 1e5);
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_zb_prof), h, sizeof(h));
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------------
// decoder (zstd_decode_core.h): one wavefront per WORK ITEM, persistent over the items.  A payload is one item -- decoded serially,
// frame by frame, block by block -- unless it is a frame of this library's own encoder that carries the independence marker
// (z_write_trailer): then every 128 KiB piece is an item of its own (k_zstd_split lists them), and a stored block of 8 MiB is
// decoded by 64 waves instead of one.  Anything about a marked payload that is not exactly what the encoder writes (header form,
// block count, sizes) sends it down the serial path, which accepts and rejects what it always did.
// ---------------------------------------------------------------------------------------------------
namespace
{
struct ZItem
{
    uint64_t src_off; // absolute, of the block header (piece) or the payload (whole)
    uint32_t size;    // bytes of the item's source
    uint32_t out0;    // piece: first output byte inside the payload's destination; whole: unused
    uint32_t payload;
    uint32_t kind;    // 0 nothing, 1 whole payload, 2 piece
};

// one thread per payload: is it a marked frame of ours?  then list its pieces, else list the payload
// The list is DENSE (items are appended through a counter): with one slot per possible piece the whole-payload items of equal-sized
// payloads sit a power of two apart and land on a handful of the persistent workgroups (measured: 128 payloads on 32 of 2048).
__global__ void k_zstd_split(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, uint32_t nblocks, ZItem* __restrict__ items,
                             uint32_t* __restrict__ item_count, uint32_t* __restrict__ out_sizes, uint32_t dbg)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks)
        return;
    const ZBlock blk = blocks[b];
    const uint8_t* p = src + blk.src_off;
    bool pieces = false;
    uint32_t np = 0;
    uint64_t content = 0;
    if (!(dbg & 1u) && blk.size >= ZHDR + 3u + ZTRAILER && z_is_trailer(p + blk.size - ZTRAILER) && p[0] == 0x28 && p[1] == 0xB5 &&
        p[2] == 0x2F && p[3] == 0xFD && p[4] == 0xE0)
    {
        for (int i = 0; i < 8; ++i)
            content |= (uint64_t)p[5 + i] << (8 * i);
        const uint32_t end = blk.size - ZTRAILER;
        const uint64_t want = (content + ZB - 1u) / ZB;
        if (content != 0 && content <= (uint64_t)blk.dst_cap && want <= (uint64_t)blk.nzb)
        {
            // first walk: is the block structure what the encoder writes?  second walk (below): list the pieces
            uint32_t ip = ZHDR;
            bool ok = true, last = false;
            while (ok && !last && np < (uint32_t)want)
            {
                if (end - ip < 3u)
                {
                    ok = false;
                    break;
                }
                const uint32_t bh = (uint32_t)p[ip] | ((uint32_t)p[ip + 1] << 8) | ((uint32_t)p[ip + 2] << 16);
                const uint32_t type = (bh >> 1) & 3u, bsize = bh >> 3;
                last = (bh & 1u) != 0u;
                const uint32_t body = type == 1u ? 1u : bsize;
                if (type == 3u || body > end - ip - 3u)
                {
                    ok = false;
                    break;
                }
                ip += 3u + body;
                ++np;
            }
            pieces = ok && last && np == (uint32_t)want && ip == end;
        }
    }
    if (pieces)
    {
        ZItem* out = items + atomicAdd(item_count, np);
        uint32_t ip = ZHDR;
        for (uint32_t i = 0; i < np; ++i)
        {
            const uint32_t bh = (uint32_t)p[ip] | ((uint32_t)p[ip + 1] << 8) | ((uint32_t)p[ip + 2] << 16);
            const uint32_t body = ((bh >> 1) & 3u) == 1u ? 1u : bh >> 3;
            out[i].src_off = blk.src_off + ip;
            out[i].size = 3u + body;
            out[i].out0 = i * ZB;
            out[i].payload = b;
            out[i].kind = 2;
            ip += 3u + body;
        }
        out_sizes[b] = (uint32_t)content; // a piece that fails replaces it by ZD_ERROR
    }
    else
    {
        ZItem* it = items + atomicAdd(item_count, 1u);
        it->src_off = blk.src_off;
        it->size = blk.size;
        it->out0 = 0;
        it->payload = b;
        it->kind = 1;
    }
}

// PIECES selects the item kind the launch works on: the mode of the decoder core is then a compile-time constant (with a run-time
// mode the whole-payload path ran 4.6x slower per wave -- measured; the two flavours are launched back to back)
template <bool PIECES>
__global__ __launch_bounds__(64) void k_zstd_decode(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, const ZItem* __restrict__ items,
                                                    const uint32_t* __restrict__ item_count, uint8_t* __restrict__ dst,
                                                    uint8_t* __restrict__ lit_scratch, uint32_t* __restrict__ out_sizes)
{
    __shared__ ZdShared sh;
    uint8_t* lits = lit_scratch + (uint64_t)blockIdx.x * (ZD_LIT_MAX + 64u);
    const uint32_t nitems = *item_count;
    for (uint32_t i = blockIdx.x; i < nitems; i += gridDim.x)
    {
        const ZItem it = items[i];
        if (it.kind != (PIECES ? 2u : 1u))
            continue;
        const ZBlock blk = blocks[it.payload];
#ifdef LTHIP_ZB_PROF
        if (threadIdx.x == 0)
            g_zb_last[blockIdx.x] = wall_clock64();
#endif
        if constexpr (!PIECES)
        {
            const uint32_t n = zd_decode_payload_ex(src + it.src_off, it.size, dst + blk.dst_off, blk.dst_cap, lits, &sh, threadIdx.x, ZD_WHOLE);
            if (threadIdx.x == 0)
                out_sizes[it.payload] = n; // ZD_ERROR (0xFFFFFFFF) for malformed input, like lthip_lz4_decompress_blocks
        }
        else
        {
            const uint32_t content = out_sizes[it.payload] == ZD_ERROR ? 0u : out_sizes[it.payload]; // (another piece may have failed)
            const uint32_t n = zd_decode_payload_ex(src + it.src_off, it.size, dst + blk.dst_off, blk.dst_cap, lits, &sh, threadIdx.x, it.out0);
            const uint32_t expect = content > it.out0 ? (content - it.out0 < ZB ? content - it.out0 : ZB) : 0u;
            if (threadIdx.x == 0 && (n != expect || content == 0u))
                atomicExch(&out_sizes[it.payload], ZD_ERROR);
        }
        __syncthreads();
    }
}
} // namespace

extern "C" int lthip_zstd_decompress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                                            const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets,
                                            const uint32_t* dst_caps, uint32_t* d_out_sizes)
{
    if (!ctx || !d_out_sizes || (block_count && (!src_offsets || !src_sizes || !dst_offsets || !dst_caps)))
        return EINVAL;
    if (block_count == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<ZBlock> hb(block_count);
    uint64_t nitems = 0;
    for (uint32_t b = 0; b < block_count; ++b)
    {
        hb[b].src_off = src_offsets[b];
        hb[b].dst_off = dst_offsets[b];
        hb[b].size = src_sizes[b];
        hb[b].dst_cap = dst_caps[b];
        hb[b].zb_base = (uint32_t)nitems; // item slots of the payload: one per 128 KiB of destination, at least one
        hb[b].nzb = dst_caps[b] ? (uint32_t)(((uint64_t)dst_caps[b] + ZB - 1u) / ZB) : 1u;
        hb[b].unit_base = hb[b].pad = 0;
        nitems += hb[b].nzb;
    }
    if (nitems > 0x7FFFFFF0ull)
        return lthip_fail(ctx, EINVAL, "zstd decode", "too many pieces in one call");
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    // 8 single-wave workgroups per CU (12 would be resident at 143 VGPRs, measured slower: 190 vs 160 ms for 512 blocks)
    uint32_t nwg = nitems < (uint64_t)ncu * 8u ? (uint32_t)nitems : (uint32_t)ncu * 8u;
    if (getenv("LTHIP_ZSTD_NWG"))
        nwg = (uint32_t)atoi(getenv("LTHIP_ZSTD_NWG"));
    void *d_blocks, *d_lits, *d_items;
    int err;
    if ((err = lthip_scratch(ctx, S_LZ4_BLOCKS, sizeof(ZBlock) * (size_t)block_count, &d_blocks)))
        return err;
    if ((err = lthip_scratch(ctx, S_Z_WORK, (size_t)(ZD_LIT_MAX + 64u) * nwg, &d_lits)))
        return err;
    if ((err = lthip_scratch(ctx, S_Z_ENC, sizeof(ZItem) * (size_t)nitems + 16, &d_items)))
        return err;
    uint32_t* d_count = (uint32_t*)((uint8_t*)d_items + sizeof(ZItem) * (size_t)nitems);
    LTHIP_CHECK(ctx, hipMemsetAsync(d_count, 0, 4, ctx->stream));
    if ((err = lthip_stage_upload(ctx, d_blocks, hb.data(), sizeof(ZBlock) * (size_t)block_count, ctx->stream)))
        return err;
    const uint32_t dbg = (uint32_t)(getenv("LTHIP_ZSTD_DBG") ? atoi(getenv("LTHIP_ZSTD_DBG")) : 0); // 1: never decode by pieces
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_zstd_split, dim3((block_count + 63) / 64), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                       block_count, (ZItem*)d_items, d_count, d_out_sizes, dbg);
    hipLaunchKernelGGL(k_zstd_decode<false>, dim3(nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                       (const ZItem*)d_items, (const uint32_t*)d_count, (uint8_t*)d_dst, (uint8_t*)d_lits, d_out_sizes);
    hipLaunchKernelGGL(k_zstd_decode<true>, dim3(nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                       (const ZItem*)d_items, (const uint32_t*)d_count, (uint8_t*)d_dst, (uint8_t*)d_lits, d_out_sizes);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

// Diagnostics for the parity tests: the match finder's output of the LAST lthip_zstd_compress_blocks call on this
// context (units [first, first+count)), so that the host model of the entropy stage can be run on the same input.
extern "C" int lthip_zstd_debug_units(lthip_ctx* ctx, uint64_t first, uint64_t count, void* h_meta, void* h_lits, void* h_recs)
{
    if (!ctx || !ctx->scratch[S_Z_LITS] || !ctx->scratch[S_Z_RECS] || !ctx->scratch[S_LZ4_META])
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (h_meta)
        LTHIP_CHECK(ctx, hipMemcpy(h_meta, (const uint8_t*)ctx->scratch[S_LZ4_META] + first * sizeof(ZbUnitMeta), count * sizeof(ZbUnitMeta),
                                   hipMemcpyDeviceToHost));
    if (h_lits)
        LTHIP_CHECK(ctx, hipMemcpy(h_lits, (const uint8_t*)ctx->scratch[S_Z_LITS] + first * ZB_UNIT, count * ZB_UNIT, hipMemcpyDeviceToHost));
    if (h_recs)
        LTHIP_CHECK(ctx, hipMemcpy(h_recs, (const uint8_t*)ctx->scratch[S_Z_RECS] + first * ZB_UNIT_SEQ_MAX * 8, count * ZB_UNIT_SEQ_MAX * 8,
                                   hipMemcpyDeviceToHost));
    return 0;
}
