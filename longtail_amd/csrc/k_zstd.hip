// k_zstd.hip -- ZStd CompressionAPI, GPU encoder.  Every stored block becomes ONE zstd frame
// (RFC 8878 §3.1.1: magic 0xFD2FB528 | FHD 0xE0 = single segment, 8-byte content size, no checksum, no dictionary |
// u64 content size | blocks, each with a 3-byte header {last:1, type:2, size:21}) whose 128 KiB pieces are
//   RLE_Block         all bytes equal (per-unit flags from the match finder, checked in k_zstd_encode)
//   Compressed        LZ sequences from the LZ4 match finder run with sequence output (k_lz4.hip, FMT 1), literals
//                     Huffman-coded, the three symbol streams FSE-coded (predefined / RLE / described tables) --
//                     zstd_block_core.h, one wavefront per piece (k_zstd_encode).  By default a RUN OF SUB-BLOCKS, one
//                     Compressed_Block per 4 KiB match-finder unit sharing the piece's entropy tables (first block: tree and
//                     table descriptions, the others Treeless / Repeat_Mode; zb_encode_piece_sub), so that encoder and
//                     decoder have 32 sequence streams and up to 128 literal streams per piece, one per lane, instead of
//                     1 and 4; LTHIP_ZSTD_SUB=0: one Compressed_Block per piece (zb_encode_block)
//   Raw_Block         whenever that would not be smaller.
// A skippable frame at the end carries the directory of block sizes (sub-block layout) or just says that the pieces are
// independent of each other (one block per piece): see z_write_trailer / z_write_trailer2_head.
// The reference's ZSTD_decompressDCtx (lib/zstd/longtail_zstd.c:144-177) decodes these like any other frame; the
// settings ('ztd1'..'ztd5', longtail_zstd.c:12-22) select one of three parses of the match finder (k_lz4.hip: default, history
// halves, + re-read after inserts; lthip_zstd_quality_of_settings) -- the entropy stage below is the same for all of them.
#include "lthip_internal.h"

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
__device__ unsigned long long g_zb_prof[32];
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
__device__ uint32_t g_zd_ablate; /* timing experiments only (LTHIP_ZSTD_ABLATE): 1 = no sequence execution, 2 = no Huffman decode */
#define ZD_ABLATE g_zd_ablate
#include "zstd_decode_core.h" /* includes zstd_block_core.h */
#include "origin_exec.h"

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

// serial per stored block: destination offset of every piece, total size
__global__ void k_zstd_scan(const ZBlock* __restrict__ blocks, uint32_t nblocks, const uint8_t* __restrict__ is_rle,
                            const uint32_t* __restrict__ enc_size, uint32_t* __restrict__ zb_dst,
                            uint32_t* __restrict__ out_sizes, uint8_t* __restrict__ dst, uint32_t dbg, uint32_t sub,
                            uint32_t* __restrict__ trailer_at)
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
        // (a piece of sub-blocks brings its Block_Headers along)
        pos += (is_rle[blk.zb_base + i] & 1u) ? 4u : enc ? ((is_rle[blk.zb_base + i] & 4u) ? enc : 3u + enc) : 3u + len;
    }
    trailer_at[b] = 0;
    const uint32_t tsize = sub ? z_trailer2_size(blk.size) : ZTRAILER;
    if (blk.nzb >= (sub ? 1u : 2u) && pos + tsize <= (uint64_t)blk.dst_cap && !(dbg & 2u))
    {
        if (sub)
        {
            z_write_trailer2_head(dst + blk.dst_off + pos, blk.size, sub == 2u ? 3u : sub == 3u ? 4u : 2u); // (the directory: k_zstd_emit, every piece its own entries)
            trailer_at[b] = (uint32_t)pos;
        }
        else
            z_write_trailer(dst + blk.dst_off + pos);
        pos += tsize;
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
                                                  uint8_t* __restrict__ dst, const uint16_t* __restrict__ sub_sizes,
                                                  const uint32_t* __restrict__ trailer_at)
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
    const uint32_t flags = is_rle[zb]; // 1 = RLE_Block, 2 = every unit without a sequence (bytes placed by the match finder),
                                       // 4 = a run of sub-blocks with their own headers
    const uint32_t rle = flags & 1u;
    const uint32_t csize = rle ? 0u : enc_size[zb];
    const bool subs = csize && (flags & 4u);
    const uint32_t nunits = (len + ZB_UNIT - 1u) / ZB_UNIT;
    const uint16_t* my_sub = sub_sizes + (uint64_t)zb * ZB_MAX_UNITS;
    if (trailer_at && trailer_at[lo] && threadIdx.x < nunits)
    {
        // this piece's entries of the frame's directory (unaligned: bytes)
        uint8_t* e = dst + b.dst_off + trailer_at[lo] + ZTRAILER + 2u * ((uint64_t)i * ZB_MAX_UNITS + threadIdx.x);
        const uint32_t v = subs ? my_sub[threadIdx.x] : rle ? ZDIR_RLE_PIECE : ZDIR_RAW_PIECE;
        e[0] = (uint8_t)v;
        e[1] = (uint8_t)(v >> 8);
    }
    if (subs)
    {
        wg_copy16(d, enc + (uint64_t)zb * ZB_OUT_BYTES, csize, threadIdx.x);
        if (i + 1 == b.nzb)
        {
            __syncthreads();
            if (threadIdx.x == 0) // Last_Block on the frame's very last block
                d[csize - 3u - (my_sub[nunits - 1u] & 0x7FFFu)] |= 1u;
        }
        return;
    }
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
                                                    uint32_t* __restrict__ enc_size, uint16_t* __restrict__ sub_sizes,
                                                    uint32_t* __restrict__ ticket, uint32_t zflags)
{
    __shared__ ZbShared sh;
    ZbScratch sc;
    uint8_t* w = work + (uint64_t)blockIdx.x * Z_WORK_STRIDE;
    sc.seqs = reinterpret_cast<uint64_t*>(w);
    sc.sbits = reinterpret_cast<uint16_t*>(w + Z_WORK_SEQS);
    // The pieces differ in what they cost (raw and RLE pieces next to nothing, pieces full of short matches the most) and a wave gets
    // only ~16 of them per launch: the waves DRAW their pieces (`ticket`, zero at launch; the first gridDim.x by index) -- with a fixed
    // stride the unluckiest of 4096 waves set the launch's time.  ticket == nullptr (LTHIP_ZSTD_TICKETS=0): the stride.
    auto next_piece = [&](uint32_t zb) -> uint32_t {
        if (!ticket)
            return zb + gridDim.x;
        uint32_t t = 0;
        if (threadIdx.x == 0)
            t = atomicAdd(ticket, 1u);
        return gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    for (uint32_t zb = blockIdx.x; zb < npieces; zb = next_piece(zb))
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
        in.flags = zflags;
        {
            // RLE_Block: every unit of the piece is one repeated byte (flagged by the match finder) and it is the same one
            const uint32_t f = threadIdx.x < in.nunits ? in.meta[threadIdx.x].uniform : in.meta[0].uniform;
            const uint32_t f0 = __builtin_amdgcn_readfirstlane(f);
            const bool rle = f0 != 0u && __builtin_amdgcn_ballot_w64(f != f0) == 0ull;
            const bool matchless = __builtin_amdgcn_ballot_w64(threadIdx.x < in.nunits && in.meta[threadIdx.x].nseq != 0u) == 0ull;
            if (threadIdx.x == 0)
            {
                is_rle[zb] = (rle ? 1 : 0) | (matchless ? 2 : 0) | (sub_sizes ? 4 : 0);
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
#ifdef LTHIP_ABLATIONS /* (one block per 128 KiB piece, round 2's frame layout: LTHIP_ZSTD_SUB=0) */
        const uint32_t n = sub_sizes ? zb_encode_piece_sub(&in, &sc, &sh, threadIdx.x, sub_sizes + (uint64_t)zb * ZB_MAX_UNITS)
                                     : zb_encode_block(&in, &sc, &sh, threadIdx.x);
#else
        const uint32_t n = zb_encode_piece_sub(&in, &sc, &sh, threadIdx.x, sub_sizes + (uint64_t)zb * ZB_MAX_UNITS);
#endif
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
                               uint32_t* d_out_sizes, int quality);

extern "C" int lthip_zstd_quality_of_settings(uint32_t settings_id)
{
    // lib/zstd/longtail_zstd.c:11-28, 43-60: the id's last character picks the level ('1' -> 0 = default 3, '2' -> 3, '3' -> 22, '4' -> 8,
    // '5' -> its own id, clamped to 22); anything else compresses at the default there, and does here
    const uint32_t low = settings_id & 0xFFu;
    if ((settings_id >> 8) != (((uint32_t)'z' << 16) | ((uint32_t)'t' << 8) | (uint32_t)'d'))
        return LTHIP_ZSTD_Q_DEFAULT;
    return low == '4' ? LTHIP_ZSTD_Q_HIGH : (low == '3' || low == '5') ? LTHIP_ZSTD_Q_MAX : LTHIP_ZSTD_Q_DEFAULT;
}

extern "C" int lthip_zstd_compress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                                          const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets,
                                          const uint32_t* dst_caps, uint32_t* d_out_sizes)
{
    return lthip_zstd_compress_blocks_q(ctx, d_src, block_count, src_offsets, src_sizes, d_dst, dst_offsets, dst_caps, d_out_sizes,
                                        LTHIP_ZSTD_Q_DEFAULT);
}

extern "C" int lthip_zstd_compress_blocks_q(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                                            const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets,
                                            const uint32_t* dst_caps, uint32_t* d_out_sizes, int quality)
{
    if (!ctx || !d_out_sizes || (block_count && (!src_offsets || !src_sizes || !dst_offsets || !dst_caps || !d_dst)) ||
        quality < LTHIP_ZSTD_Q_DEFAULT || quality > LTHIP_ZSTD_Q_MAX)
        return EINVAL;
    const uint64_t budget = lthip_codec_batch_bytes(); // scratch (unit literals + records + piece slots): about three times that
    for (uint32_t b0 = 0; b0 < block_count;)
    {
        uint64_t bytes = src_sizes[b0];
        uint32_t b1 = b0 + 1;
        while (b1 < block_count && bytes + src_sizes[b1] <= budget)
            bytes += src_sizes[b1++];
        const int err = zstd_compress_batch(ctx, d_src, b1 - b0, src_offsets + b0, src_sizes + b0, d_dst, dst_offsets + b0, dst_caps + b0,
                                            d_out_sizes + b0, quality);
        if (err)
            return err;
        b0 = b1;
    }
    return 0;
}

static int zstd_compress_batch(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                               const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                               uint32_t* d_out_sizes, int quality)
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
                                         &d_meta, unit_base.data(), &nunits, quality)))
        return err;
    for (uint32_t b = 0; b < block_count; ++b)
        hb[b].unit_base = unit_base[b];
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const uint32_t nwg = (uint32_t)(nzb < (uint64_t)ncu * 16 ? nzb : (uint64_t)ncu * 16);
    void *d_blocks, *d_rle, *d_zdst, *d_enc, *d_encsz, *d_work, *d_sub, *d_trail;
    // sub-blocks (default): one zstd block per 4 KiB unit and a directory in the trailer; LTHIP_ZSTD_SUB=0: one block per 128 KiB piece
    LTHIP_ABLATION_ENV(env_sub, "LTHIP_ZSTD_SUB");
    const bool sub = env_sub.get() != 0;
    // LTHIP_ZSTD_REP=1: block-local repeat-offset codes (sub-block layout only; the trailer says so: version 3).  Off by default -- on
    // every kind measured they are worth < 0.1 % of the compressed size (offsets rarely repeat inside a 4 KiB block:
    // profiles/r04_zstd_ratio_table*.txt) and cost the encoder 4 % and the lane decoder 5-9 %
    LTHIP_ABLATION_ENV(env_rep, "LTHIP_ZSTD_REP");
    const bool rep = sub && env_rep.get() == 1;
    if ((err = lthip_scratch(ctx, S_LZ4_BLOCKS, sizeof(ZBlock) * (size_t)block_count, &d_blocks)))
        return err;
    if ((err = lthip_scratch(ctx, S_Z_SUB, sizeof(uint16_t) * ZB_MAX_UNITS * ((size_t)nzb + 1) + 4 * ((size_t)block_count + 1), &d_sub)))
        return err;
    d_trail = (uint8_t*)d_sub + sizeof(uint16_t) * ZB_MAX_UNITS * ((size_t)nzb + 1);
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
        LTHIP_ABLATION_ENV(env_tickets, "LTHIP_ZSTD_TICKETS");

        uint32_t* d_ticket = env_tickets.get() == 0 ? nullptr : (uint32_t*)d_encsz + nzb + 1; // (the size list has four spare words)
        if (d_ticket)
            LTHIP_CHECK(ctx, hipMemsetAsync(d_ticket, 0, 4, ctx->stream));
        LaunchTimer t(ctx, LTHIP_K_ZSTD_ENC);
        hipLaunchKernelGGL(k_zstd_encode, dim3(nwg), dim3(64), 0, ctx->stream, (const ZBlock*)d_blocks, block_count, (uint32_t)nzb,
                           (const uint8_t*)d_src, (uint8_t*)d_rle, (const ZbUnitMeta*)d_meta, (const uint8_t*)d_lits, (const uint64_t*)d_recs,
                           (uint8_t*)d_work, (uint8_t*)d_enc, (uint32_t*)d_encsz, sub ? (uint16_t*)d_sub : (uint16_t*)nullptr, d_ticket,
                           rep ? (uint32_t)ZB_F_REPCODES : 0u);
        LTHIP_LAUNCH_CHECK(ctx);
    }
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    LTHIP_ABLATION_ENV(env_zdbg, "LTHIP_ZSTD_DBG");
    hipLaunchKernelGGL(k_zstd_scan, dim3((block_count + 63) / 64), dim3(64), 0, ctx->stream, (const ZBlock*)d_blocks,
                       block_count, (const uint8_t*)d_rle, (const uint32_t*)d_encsz, (uint32_t*)d_zdst, d_out_sizes, (uint8_t*)d_dst,
                       env_zdbg.get() > 0 ? (uint32_t)env_zdbg.get() : 0u, // (ablation build) bit 1: no trailer
                       sub ? (rep ? 2u : (quality >= LTHIP_ZSTD_Q_MAX ? 3u : 1u)) : 0u, (uint32_t*)d_trail);
    hipLaunchKernelGGL(k_zstd_headers, dim3((block_count + 63) / 64), dim3(64), 0, ctx->stream, (const ZBlock*)d_blocks,
                       block_count, (const uint32_t*)d_out_sizes, (uint8_t*)d_dst);
    if (nzb)
        hipLaunchKernelGGL(k_zstd_emit, dim3((uint32_t)nzb), dim3(ZT), 0, ctx->stream, (const uint8_t*)d_src,
                           (const ZBlock*)d_blocks, block_count, (const uint8_t*)d_rle, (const uint32_t*)d_encsz,
                           (const uint8_t*)d_enc, (const uint32_t*)d_zdst, (const uint32_t*)d_out_sizes, (uint8_t*)d_dst,
                           (const uint16_t*)d_sub, (const uint32_t*)d_trail);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

#ifdef LTHIP_ZB_PROF
extern "C" __attribute__((visibility("default"))) int lthip_zb_prof_dump(void)
{
    unsigned long long h[32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_zb_prof), sizeof(h)) != hipSuccess)
        return -1;
    for (int i = 0; i < 32; ++i)
        if (h[i])
            fprintf(stderr, "zb phase ending at mark %2d: %.3f ms wave-time (100 MHz clock)\n", i, (double)h[i] / 1e5);
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_zb_prof), h, sizeof(h));
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------------
// decoder (zstd_decode_core.h): one wavefront per WORK ITEM, persistent over the items.  A payload is one item -- decoded serially,
// frame by frame, block by block -- unless it is a frame of this library's own encoder that says so in its trailer: then every
// 128 KiB piece is an item of its own (k_zstd_split lists them), and a stored block of 8 MiB is decoded by 64 waves instead of
// one -- with the directory (sub-block layout) every lane of such a wave has a block of its own (k_zstd_sub_entropy, further
// down).  Anything about such a payload that is not exactly what the encoder writes (header form, block count, sizes) sends it
// down the serial path, which accepts and rejects what it always did.
// ---------------------------------------------------------------------------------------------------
namespace
{
struct ZItem
{
    uint64_t src_off; // absolute, of the block header (piece) or the payload (whole)
    uint32_t size;    // bytes of the item's source
    uint32_t out0;    // piece: first output byte inside the payload's destination; whole: unused
    uint32_t payload;
    uint32_t kind;    // 0 nothing, 1 whole payload, 2 piece (one block), 3 piece (a run of sub-blocks)
    uint32_t aux;     // kind 3: offset (inside the payload) of the piece's first directory entry
    uint32_t pad;
};

// Is the payload a marked frame of ours?  then list its pieces, else list the payload.  One wave per payload.
// The list is DENSE (items are appended through a counter): with one slot per possible piece the whole-payload items of equal-sized
// payloads sit a power of two apart and land on a handful of the persistent workgroups (measured: 128 payloads on 32 of 2048).
//
// Frames with the DIRECTORY (sub-block layout): every lane adds up the entries of one piece, a wave scan places the pieces; nothing
// of the frame's 2 048 block headers is read here -- every piece checks its own headers against the directory when it is decoded,
// and the sum of all sizes must land exactly on the trailer.
// Frames with the plain marker (one block per piece): lane 0 walks the block headers.
__device__ void z_split_walk(const uint8_t* p, const ZBlock blk, uint32_t b, ZItem* items, uint32_t* item_count, uint32_t* out_sizes, uint32_t dbg)
{
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
            out[i].aux = out[i].pad = 0;
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
        it->aux = it->pad = 0;
    }
}

__device__ __forceinline__ uint32_t z_wave_scan_excl(uint32_t v, int lane, uint32_t* total)
{
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const uint32_t x = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d)
            incl += x;
    }
    *total = (uint32_t)__shfl((int)incl, 63, 64);
    return incl - v;
}

// Frames of OTHER encoders (the reference's: blocks that depend on each other through the window, repeat offsets, repeated tables):
// one frame that fills the payload, content size stated, no dictionary.  Its blocks are listed at FIXED places (slot zb_base + k of
// `fitems`): k_zstd_blk_entropy decodes the streams of every block on a wave of its own (literals and sequence records to scratch),
// k_zstd_execute<2> executes a payload's blocks one after the other.  Frame header: RFC 8878 3.1.1.1.
constexpr uint32_t ZF_SLOTS = 8u; // block slots of a payload of another encoder per 128 KiB of its capacity
struct ZFrameHdr
{
    uint32_t size;      // bytes of the frame header
    uint64_t content;   // Frame_Content_Size
    uint32_t checksum;  // a 4-byte content checksum follows the last block
    bool ok;
};
__device__ __forceinline__ ZFrameHdr z_frame_header(const uint8_t* p, uint32_t avail)
{
    ZFrameHdr h;
    h.size = 0;
    h.content = 0;
    h.checksum = 0;
    h.ok = false;
    if (avail < 6u || p[0] != 0x28 || p[1] != 0xB5 || p[2] != 0x2F || p[3] != 0xFD)
        return h;
    const uint32_t fhd = p[4];
    const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1u;
    if ((fhd & 8u) || (fhd & 3u)) // reserved bit; a dictionary
        return h;
    uint32_t pos = 5u + (single ? 0u : 1u);
    const uint32_t fcs = fcs_flag == 0u ? (single ? 1u : 0u) : fcs_flag == 1u ? 2u : fcs_flag == 2u ? 4u : 8u;
    if (fcs == 0u || pos + fcs > avail)
        return h;
    uint64_t c = 0;
    for (uint32_t i = 0; i < fcs; ++i)
        c |= (uint64_t)p[pos + i] << (8u * i);
    if (fcs == 2u)
        c += 256u;
    h.size = pos + fcs;
    h.content = c;
    h.checksum = (fhd >> 2) & 1u;
    h.ok = true;
    return h;
}
__device__ bool z_split_foreign(const uint8_t* p, const ZBlock blk, uint32_t b, ZItem* fitems, uint32_t* f_nblocks, uint32_t* out_sizes, uint32_t* totals,
                                uint32_t* flist)
{
    const ZFrameHdr h = z_frame_header(p, blk.size);
    if (!h.ok || h.content == 0u || h.content > (uint64_t)blk.dst_cap || h.content > 0x7F000000ull)
        return false;
    uint32_t pos = h.size, k = 0;
    bool last = false;
    while (!last)
    {
        if (k >= ZF_SLOTS * blk.nzb || blk.size - pos < 3u) // (blocks of 16 KiB on average still fit: the block splitter of the high levels)
            return false;
        const uint32_t bh = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16);
        const uint32_t type = (bh >> 1) & 3u, bsize = bh >> 3;
        last = (bh & 1u) != 0u;
        const uint32_t body = type == 1u ? 1u : bsize;
        if (type == 3u || bsize > ZB || body > blk.size - pos - 3u)
            return false;
        ZItem it;
        it.src_off = blk.src_off;
        it.size = blk.size;
        it.out0 = 0;
        it.payload = b;
        it.kind = 4;
        it.aux = k;
        it.pad = pos;
        fitems[blk.pad + k] = it;
        pos += 3u + body;
        ++k;
    }
    if (pos + 4u * h.checksum != blk.size)
    {
        return false; // more frames behind this one: the serial decoder (nothing is listed yet)
    }
    f_nblocks[b] = k;
    out_sizes[b] = (uint32_t)h.content; // (replaced when the payload goes back to the serial decoder)
    {
        const uint32_t at = atomicAdd(&totals[1], k); // the work list of k_zstd_blk_entropy
        for (uint32_t j = 0; j < k; ++j)
            flist[at + j] = blk.pad + j;
    }
    atomicMax(totals - 3, k);                                          // (item_count[1]: the most blocks any such frame has)
    atomicAdd(&totals[0], k);                                          // blocks listed this way, and the bytes they regenerate:
    atomicAdd((unsigned long long*)&totals[2], (unsigned long long)h.content); // the host sizes the literal and record arenas from these
    return true;
}

__global__ __launch_bounds__(64) void k_zstd_split(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, uint32_t nblocks,
                                                   ZItem* __restrict__ items, uint32_t* __restrict__ item_count,
                                                   uint32_t* __restrict__ out_sizes, uint32_t dbg, ZItem* __restrict__ fitems,
                                                   uint32_t* __restrict__ f_nblocks, uint32_t* __restrict__ flist)
{
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const ZBlock blk = blocks[b];
    const uint8_t* p = src + blk.src_off;
    uint64_t content = 0;
    uint32_t dir = 0; // the directory trailer's version, 0 = none
    if (!(dbg & 1u) && blk.size >= ZHDR + 3u + ZTRAILER && p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD && p[4] == 0xE0)
    {
        for (int i = 0; i < 8; ++i)
            content |= (uint64_t)p[5 + i] << (8 * i);
        const uint64_t want = (content + ZB - 1u) / ZB;
        if (content != 0 && content <= (uint64_t)blk.dst_cap && want <= (uint64_t)blk.nzb &&
            (uint64_t)blk.size >= (uint64_t)ZHDR + 3u + z_trailer2_size(content))
            dir = z_is_trailer2_head(p + blk.size - z_trailer2_size(content), content);
    }
    if (dir == 4u && blk.dst_cap >= 0x7FFFFFFFu)
    {
        // pieces that depend on each other, positions beyond 31 bits: the payload's blocks in order, on one wave (the serial decoder)
        if (lane == 0)
            z_split_walk(p, blk, b, items, item_count, out_sizes, dbg | 1u);
        return;
    }
    if (!dir)
    {
        if (lane == 0)
        {
            const bool marked = blk.size >= ZHDR + 3u + ZTRAILER && z_is_trailer(p + blk.size - ZTRAILER);
            if (marked || (dbg & 9u) || !z_split_foreign(p, blk, b, fitems, f_nblocks, out_sizes, item_count + 4, flist)) // (dbg 8: no block-parallel path for foreign frames)
                z_split_walk(p, blk, b, items, item_count, out_sizes, dbg);
        }
        return;
    }
    const uint32_t tsize = z_trailer2_size(content);
    const uint8_t* d = p + blk.size - tsize + ZTRAILER; // u16 entries, any alignment
    const uint32_t np = (uint32_t)((content + ZB - 1u) / ZB);
    // pass 1: do the sizes add up?  pass 2: the items
    uint32_t base = 0;
    ZItem* out = nullptr;
    for (int pass = 0; pass < 2; ++pass)
    {
        uint32_t pos = ZHDR;
        bool ok = true;
        for (uint32_t i0 = 0; i0 < np; i0 += 64u)
        {
            const uint32_t i = i0 + (uint32_t)lane;
            uint32_t size = 0, kind = 0, rle = 0;
            if (i < np)
            {
                const uint32_t len = (uint32_t)(content - (uint64_t)i * ZB < ZB ? content - (uint64_t)i * ZB : ZB);
                const uint32_t nu = (len + ZB_UNIT - 1u) / ZB_UNIT;
                const uint8_t* e = d + 2u * (uint64_t)i * ZB_MAX_UNITS;
                const uint32_t e0 = (uint32_t)e[0] | ((uint32_t)e[1] << 8);
                if (e0 == ZDIR_RAW_PIECE || e0 == ZDIR_RLE_PIECE)
                {
                    size = e0 == ZDIR_RAW_PIECE ? 3u + len : 4u;
                    rle = e0 == ZDIR_RLE_PIECE ? 1u : 0u;
                    kind = 2;
                    for (uint32_t u = 1; u < nu; ++u)
                        ok = ok && ((uint32_t)e[2u * u] | ((uint32_t)e[2u * u + 1u] << 8)) == e0;
                }
                else
                {
                    kind = 3;
                    for (uint32_t u = 0; u < nu; ++u)
                    {
                        const uint32_t eu = (uint32_t)e[2u * u] | ((uint32_t)e[2u * u + 1u] << 8);
                        ok = ok && eu < ZDIR_RLE_PIECE;
                        size += 3u + (eu & 0x7FFFu);
                    }
                }
            }
            uint32_t total;
            const uint32_t off = z_wave_scan_excl(size, lane, &total);
            if (pass == 0 && kind == 2u) // a piece of one Raw_Block / RLE_Block: its header is checked here (runs of sub-blocks: by their decoder)
            {
                const uint64_t at = (uint64_t)pos + off;
                if (at + size > (uint64_t)blk.size - tsize)
                    ok = false;
                else
                {
                    const uint32_t len = (uint32_t)(content - (uint64_t)i * ZB < ZB ? content - (uint64_t)i * ZB : ZB);
                    const uint32_t bh = (uint32_t)p[at] | ((uint32_t)p[at + 1] << 8) | ((uint32_t)p[at + 2] << 16);
                    ok = ok && bh == ((i + 1u == np ? 1u : 0u) | (rle << 1) | (len << 3));
                }
            }
            if (pass == 1 && i < np)
            {
                ZItem it;
                it.src_off = blk.src_off + pos + off;
                it.size = size;
                it.out0 = i * ZB;
                it.payload = b;
                it.kind = kind;
                it.aux = (uint32_t)(d - p) + 2u * i * ZB_MAX_UNITS;
                // kind 3: bit 0 the frame's blocks may use block-local repeat-offset codes; bit 1 (version 4) the piece's matches may
                // reach into the pieces before it: k_zstd_execute runs such a frame's pieces as a chain
                it.pad = (dir == 3u ? 1u : 0u) | (dir == 4u ? 2u : 0u);
                out[i] = it;
            }
            if ((uint64_t)pos + total > (uint64_t)blk.size)
                ok = false;
            pos += total;
        }
        ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull && pos == blk.size - tsize;
        if (pass == 0)
        {
            if (!ok)
            {
                if (lane == 0) // not what the directory promises: the serial decoder says what the payload is
                {
                    ZItem* it = items + atomicAdd(item_count, 1u);
                    it->src_off = blk.src_off;
                    it->size = blk.size;
                    it->out0 = 0;
                    it->payload = b;
                    it->kind = 1;
                    it->aux = it->pad = 0;
                }
                return;
            }
            if (lane == 0)
            {
                base = atomicAdd(item_count, np);
                out_sizes[b] = (uint32_t)content; // a piece that fails replaces it
            }
            base = (uint32_t)__builtin_amdgcn_readfirstlane(base);
            out = items + base;
        }
    }
}

// The item list in LINK-MAJOR order (round 5): row k = the pieces whose index in their frame is k modulo ZCHAIN, of every payload
// (whole-payload items: row 0) -- all chain heads first, then every chain's second piece, ...  The rounds below go over the items in
// this order, so that the chains of the frames whose pieces depend on each other (trailer version 4) all run side by side: the heads
// fill the machine, the workgroups of a launch are dispatched in order, a piece only ever waits for a workgroup that was dispatched
// before it (or belongs to an earlier launch), and by the time a row's workgroups get a slot most of the row before is done.  (Measured
// with the rows = piece indices: one link of one chain per frame at a time, 512 waves at work: 134 GB/s on "mixed" at 512 blocks.)
// Three small kernels: count the rows, scan them, fill (the order inside a row is whatever the atomics give: every kernel of a round
// uses the same table).
__global__ void k_zstd_rows(const ZItem* __restrict__ items, const uint32_t* __restrict__ item_count, uint32_t nrows, uint32_t* __restrict__ row_cnt,
                            uint32_t* __restrict__ row_start, uint32_t* __restrict__ perm, uint32_t phase)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = *item_count;
    if (phase == 1u) // exclusive scan of the row counts, one wave
    {
        uint32_t base = 0;
        for (uint32_t r0 = 0; r0 < nrows; r0 += 64u)
        {
            const uint32_t r = r0 + threadIdx.x;
            const uint32_t c = r < nrows ? row_cnt[r] : 0u;
            uint32_t incl = c;
            for (int d = 1; d < 64; d <<= 1)
            {
                const uint32_t o = __shfl_up(incl, d, 64);
                if ((int)threadIdx.x >= d)
                    incl += o;
            }
            if (r < nrows)
            {
                row_start[r] = base + incl - c;
                row_cnt[r] = 0u; // (the fill counts again)
            }
            base += __shfl(incl, 63, 64);
        }
        return;
    }
    if (i >= n)
        return;
    const ZItem it = items[i];
    const uint32_t k = it.kind == 2u || it.kind == 3u ? (it.out0 / ZB) % nrows : 0u;
    if (phase == 0u)
        atomicAdd(&row_cnt[k], 1u);
    else
        perm[row_start[k] + atomicAdd(&row_cnt[k], 1u)] = i;
}

// What k_zstd_prepare leaves for k_zstd_execute about one piece
enum : uint32_t { ZP_READY = 0u, ZP_DONE = 1u, ZP_SERIAL = 2u };
struct ZPrep
{
    uint64_t bits_off;  // absolute offset (in the source arena) of the sequences' bit-stream
    uint32_t bits_size;
    uint32_t nbseq;
    uint32_t nlit;
    uint32_t status;    // ZP_READY: literals + tables exported; ZP_DONE: nothing left to do; ZP_SERIAL: the serial piece decoder takes it
    uint32_t log[3];    // table logs (0: an RLE table, one entry)
    uint32_t expect;    // bytes the piece has to produce
    // blocks of other encoders' frames (k_zstd_blk_entropy -> k_zstd_blk_sequences -> k_zstd_execute_payload): bits_off = where the
    // block's literals are in the literal arena (Raw / RLE blocks: where its bytes are in the source), log[0] = 0 compressed | 1 raw |
    // 2 RLE, log[1] = the three table logs (LL | OF << 8 | ML << 16), and:
    uint64_t rec_at;    // first record of the block in the record arena
    uint32_t seq_off;   // the sequences' bit-stream: offset inside the payload ...
    uint32_t seq_size;  // ... and bytes
};
constexpr uint32_t ZREC_MAX = ZB_MAX_UNITS * ZB_UNIT_SEQ_MAX; // sequence records per piece of sub-blocks
constexpr uint32_t ZT_ENTRIES = 512u; // per table and piece: u64 {BYTE OFFSET (within the piece's three tables) of the new state's base entry:16 |
                                       // state bits:8 | extra bits:8 | baseline:32}

// PIECES selects the item kind the launch works on: the mode of the decoder core is then a compile-time constant (with a run-time
// mode the whole-payload path ran 4.6x slower per wave -- measured; the two flavours are launched back to back)
template <bool PIECES>
__global__ __launch_bounds__(64) void k_zstd_decode(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, const ZItem* __restrict__ items,
                                                    const uint32_t* __restrict__ item_count, uint8_t* __restrict__ dst,
                                                    uint8_t* __restrict__ lit_scratch, uint32_t* __restrict__ out_sizes,
                                                    const ZPrep* __restrict__ prep)
{
    __shared__ ZdShared sh;
    uint8_t* lits = lit_scratch + (uint64_t)blockIdx.x * (ZD_LIT_MAX + 64u);
    const uint32_t nitems = *item_count;
    if (threadIdx.x == 0)
        sh.v[ZDV_PREP] = 0;
    for (uint32_t i = blockIdx.x; i < nitems; i += gridDim.x)
    {
        const ZItem it = items[i];
        if (it.kind != (PIECES ? 2u : 1u))
            continue;
        if (PIECES && prep && prep[i].status != ZP_SERIAL) // the two-stage path below has done it (or will report it)
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
#ifdef LTHIP_ABLATIONS
#include "ablations/k_zstd_prepare.inc"
#endif

// ---------------------------------------------------------------------------------------------------
// Pieces that are runs of SUB-BLOCKS (kind 3): k_zstd_sub_entropy + k_zstd_execute<true>.
// One wave per piece.  Lane u owns unit u's block: it finds it through the frame's directory, checks the Block_Header against the
// directory, parses the literals and sequences section headers.  The Huffman tree and the three FSE tables come with the first
// block that needs them (everything else is treeless / Predefined or Repeat_Mode) and are built once, by the decoder core's
// own readers.  Then every lane decodes streams of its own: the up to 128 Huffman streams of the piece's literals (into the piece's
// literal buffer, back to back in block order), and the up to 32 sequence bit-streams (lane u the sequences of block u, as records
// {literal length, match length, offset value} into the piece's record array).  A block's literals after its last sequence join
// the literal length of the next sequence of the piece, so what k_zstd_execute<true> sees is ONE run of sequences over ONE run of
// literals, exactly what it executes for a one-block piece.
// STRICT like the one-block piece decoder: a block may not use repeat offsets; more than that, anything that is not what
// zb_encode_piece_sub writes (a second tree, tables sent twice, a block that does not regenerate exactly its unit, a header that
// disagrees with the directory) sends the whole payload to the serial decoder (retry[payload]), which accepts and rejects what it
// always did.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zs_bits(const uint8_t* base, uint32_t bitpos, uint32_t n) // bits [bitpos, bitpos + n), n <= 32
{
    const uint32_t b1 = (bitpos + n + 7u) >> 3; // the 8 bytes that END with the byte holding the field's top bit: never past the stream
    uint64_t w;
    __builtin_memcpy(&w, base + (int64_t)b1 - 8, 8);
    return (uint32_t)(w >> (bitpos + 64u - 8u * b1)) & (uint32_t)((1ull << n) - 1ull);
}

// A lane's backward bit reader with the stream ahead of it in registers: 128 bits [top - 128, top) plus the 64 below them already
// on their way, so that no load sits on the chain state -> bits -> next state (one 8-byte load per 64 bits consumed, issued a
// window ahead).  Loads may reach up to 24 bytes below the stream: `reach` = how many bytes below it are readable (what lies
// there is never used).
struct ZsWin
{
    const uint8_t* base;
    uint64_t hi, lo, nx;
    int32_t top;   // bit index (relative to base) one past the window
    int32_t reach; // lowest readable byte offset relative to base (<= 0)
    __device__ __forceinline__ uint64_t load(int32_t byte_off) const
    {
        uint64_t w;
        const int32_t o = byte_off < reach ? reach : byte_off;
        __builtin_memcpy(&w, base + o, 8);
        return w;
    }
    __device__ __forceinline__ void open(const uint8_t* b, uint32_t size, uint32_t below)
    {
        base = b;
        reach = -(int32_t)below;
        top = (int32_t)(8u * size);
        hi = load((int32_t)size - 8);
        lo = load((int32_t)size - 16);
        nx = load((int32_t)size - 24);
    }
    // make bits [pos - need, pos) part of the window (need <= 64; pos <= top)
    __device__ __forceinline__ void ensure(uint32_t pos, uint32_t need)
    {
        if ((int32_t)pos - (int32_t)need < top - 128)
        {
            hi = lo;
            lo = nx;
            top -= 64;
            nx = load((top >> 3) - 24);
        }
    }
    __device__ __forceinline__ uint64_t get64(uint32_t bitpos) const // bits [bitpos, bitpos + 64) (zeros above the window's top)
    {
        const uint32_t s = (uint32_t)((int32_t)bitpos - (top - 128)); // 0 .. 127
        return s >= 64u ? hi >> (s - 64u) : (s ? (lo >> s) | (hi << (64u - s)) : lo);
    }
    // the 64 bits below `pos`, the next bit to read on top (fewer than 64 bits left: zeros below them)
    __device__ __forceinline__ uint64_t below(uint32_t pos) const
    {
        return pos >= 64u ? get64(pos - 64u) : (pos ? get64(0u) << (64u - pos) : 0ull);
    }
    __device__ __forceinline__ uint32_t get(uint32_t bitpos, uint32_t n) const // bits [bitpos, bitpos + n), n <= 32, inside the window
    {
        const uint32_t s = (uint32_t)((int32_t)bitpos - (top - 128)); // 0 .. 127
        const uint64_t v = s >= 64u ? hi >> (s - 64u) : (s ? (lo >> s) | (hi << (64u - s)) : lo);
        return (uint32_t)v & (uint32_t)((1ull << n) - 1ull);
    }
};

// The sequence lanes: every lane with `act` decodes ONE bit-stream of nbseq sequences (tables packed in shared memory, one 8-byte
// entry per state) into records {literal length:20 | match length:20 | offset value:24}.  One loop for the wave, a lane takes part
// while its stream has sequences left (no lane leaves early: a lane that finds an error only stops decoding).  Per sequence: the
// three entries, ONE 64-bit view of the bits below the position (a second one only when offset + lengths + states exceed 64 bits),
// the fields shifted off its top.  STRICT: offset values 1..3 (repeat offsets) are errors; otherwise they stay in the record for
// whoever executes it in order.
// STRICT 0: offset values 1..3 stay in the record; 1: they are errors; 2: resolved with a block-local history (version-3 frames)
template <int STRICT>
__device__ __forceinline__ void zs_seq_lanes(bool act, const uint8_t* stream, uint32_t ssize, uint32_t below, uint32_t nbseq, uint32_t log_l,
                                             uint32_t log_o, uint32_t log_m, const uint64_t* pk_ll, const uint64_t* pk_of, const uint64_t* pk_ml,
                                             uint64_t* recs, bool& bad, uint32_t& sum_ll, uint32_t& sum_ml)
{
    uint32_t pos = 0, sl = 0, so = 0, sm = 0;
    [[maybe_unused]] uint32_t r1 = 0, r2 = 0, r3 = 0; // STRICT: the block's own offset history, 0 = unknown
    ZsWin w;
    w.base = stream;
    w.hi = w.lo = w.nx = 0;
    w.top = 0;
    w.reach = 0;
    if (act)
    {
        if (ssize == 0u || stream[ssize - 1u] == 0u)
            bad = true;
        else
        {
            pos = (ssize - 1u) * 8u + (31u - (uint32_t)__builtin_clz((uint32_t)stream[ssize - 1u]));
            w.open(stream, ssize, below);
            if (log_l + log_o + log_m > pos)
                bad = true;
            else
            {
                pos -= log_l;
                sl = w.get(pos, log_l);
                pos -= log_o;
                so = w.get(pos, log_o);
                pos -= log_m;
                sm = w.get(pos, log_m);
            }
        }
        act = !bad;
    }
    const bool had = act;
    for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(act && k < nbseq) != 0ull; ++k)
    {
        if (act && k < nbseq)
        {
            const uint64_t el = pk_ll[sl], eo = pk_of[so], em = pk_ml[sm];
            const uint32_t l0 = (uint32_t)el, o0 = (uint32_t)eo, m0 = (uint32_t)em;
            const uint32_t ob = o0 >> 24, mb = m0 >> 24, lb = l0 >> 24;
            const uint32_t nbl = (l0 >> 16) & 255u, nbm = (m0 >> 16) & 255u, nbo = (o0 >> 16) & 255u;
            const bool more = k + 1u < nbseq;
            const uint32_t n1 = ob + mb + lb, n2 = more ? nbl + nbm + nbo : 0u; // <= 63, <= 26
            if (n1 + n2 > pos)
            {
                bad = true;
                act = false;
            }
            else
            {
                w.ensure(pos, 64u);
                uint64_t acc = w.below(pos);
                const uint32_t ov = (uint32_t)(eo >> 32) + (uint32_t)((acc >> 1) >> (63u - ob));
                acc <<= ob;
                const uint32_t t2 = (uint32_t)((acc >> 1) >> (63u - (mb + lb))); // match-length and literal-length bits are adjacent
                acc <<= mb + lb;
                const uint32_t ml = (uint32_t)(em >> 32) + (t2 >> lb);
                const uint32_t ll = (uint32_t)(el >> 32) + (t2 & ((1u << lb) - 1u));
                if (n1 + n2 > 64u) // (rare: the view does not reach the state fields)
                {
                    w.ensure(pos - n1, 32u);
                    acc = w.below(pos - n1);
                }
                if (more)
                {
                    const uint32_t t3 = (uint32_t)((acc >> 1) >> (63u - n2)); // LL, ML, OF from the top
                    sl = (l0 & 0xFFFFu) + (t3 >> (nbm + nbo));
                    sm = (m0 & 0xFFFFu) + ((t3 >> nbo) & ((1u << nbm) - 1u));
                    so = (o0 & 0xFFFFu) + (t3 & ((1u << nbo) - 1u));
                }
                pos -= n1 + n2;
                uint32_t ovr = ov;
                if constexpr (STRICT == 1)
                {
                    if (ov <= 3u)
                        bad = true; // (a version-2 frame writes plain offsets only)
                }
                if constexpr (STRICT == 2)
                {
                    // Repeat offsets, resolved here: this library's encoder only refers to history entries that the block's own
                    // sequences have set (zb_encode_piece_sub, ZB_F_REPCODES), so the lane starts with an unknown history (0) and a
                    // code that would read an unknown entry is what "needs the block before" now means.
                    const uint32_t used = ov > 3u ? 0u : (ll != 0u ? ov : ov + 1u); // entry 1..3, 4 = entry 1 minus one, 0 = a new offset
                    uint32_t off = ov - 3u;
                    if (used)
                    {
                        off = used == 1u ? r1 : used == 2u ? r2 : used == 3u ? r3 : r1 - 1u;
                        if (off == 0u || (used == 4u && r1 == 0u))
                            bad = true;
                    }
                    if (used == 2u)
                    {
                        r2 = r1;
                        r1 = off;
                    }
                    else if (used != 1u)
                    {
                        r3 = r2;
                        r2 = r1;
                        r1 = off;
                    }
                    ovr = off + 3u;
                }
                if (ovr >= (1u << 24))
                    bad = true;
                sum_ll += ll;
                sum_ml += ml;
                recs[k] = (uint64_t)ll | ((uint64_t)ml << 20) | ((uint64_t)ovr << 40);
            }
        }
    }
    if (had && !bad && pos != 0u)
        bad = true; // the bit-stream must be consumed exactly
}

__global__ __launch_bounds__(64) void k_zstd_sub_entropy(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, const ZItem* __restrict__ items,
                                                         const uint32_t* __restrict__ item_count, uint32_t item0, uint32_t item1,
                                                         uint8_t* __restrict__ lit_scratch, uint64_t* __restrict__ rec_scratch,
                                                         ZPrep* __restrict__ prep, const uint32_t* __restrict__ out_sizes,
                                                         uint32_t* __restrict__ retry, uint32_t* __restrict__ ticket, const uint32_t* __restrict__ perm)
{
    __shared__ ZdShared sh;
    __shared__ uint4 s_streams[4 * ZB_MAX_UNITS]; // {source offset inside the piece, bytes, literal offset, symbols}
    __shared__ __attribute__((aligned(16))) uint8_t s_desc[512 + 256]; // the table descriptions and the tree description, staged: lane 0 parses
                                                                      // them bit by bit, and a byte from global memory costs it a round trip
    const uint32_t nitems = *item_count < item1 ? *item_count : item1;
    const int lane = threadIdx.x;
    uint64_t* const pk_ll = reinterpret_cast<uint64_t*>(sh.huf); // packed tables {base:16 | state bits:8 | extra bits:8 | baseline:32},
    uint64_t* const pk_ml = pk_ll + 512;                         // over the Huffman table (done with by then) ...
    uint64_t* const pk_of = reinterpret_cast<uint64_t*>(&sh.wtab); // ... and the weights' table
    static_assert(sizeof(sh.huf) >= 2 * 512 * 8 && sizeof(sh.wtab) >= 256 * 8, "room for the packed tables");
    for (;;)
    {
        // Pieces cost very different amounts (raw units next to units full of sequences): the persistent waves draw them from a
        // counter.  (Every lane takes part in the draw -- lane 0 adds one, the others zero -- see k_lz4_pd_units for why.)
        __builtin_amdgcn_wave_barrier();
        uint32_t t = atomicAdd(ticket, lane == 0 ? 1u : 0u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (item0 + t >= nitems)
            break;
        const uint32_t i = perm[item0 + t]; // (link-major order: k_zstd_rows)
        const ZItem it = items[i];
        if (it.kind != 3u)
            continue;
        __syncthreads();
#ifdef LTHIP_ZB_PROF
        if (lane == 0)
            g_zb_last[blockIdx.x] = wall_clock64();
#endif
        const ZBlock blk = blocks[it.payload];
        const uint32_t slot = t;
        uint8_t* lits = lit_scratch + (uint64_t)slot * (ZD_LIT_MAX + 64u);
        uint64_t* recs = rec_scratch + (uint64_t)slot * ZREC_MAX;
        const uint8_t* p = src + it.src_off;
        const uint8_t* dir = src + blk.src_off + it.aux;
        const uint32_t content = out_sizes[it.payload] == ZD_ERROR ? 0u : out_sizes[it.payload];
        const uint32_t expect = content > it.out0 ? (content - it.out0 < ZB ? content - it.out0 : ZB) : 0u;
        const uint32_t nunits = (expect + ZB_UNIT - 1u) / ZB_UNIT;
        const bool mine = (uint32_t)lane < nunits;
        bool bad = expect == 0u;
        if (lane == 0)
        {
            sh.v[ZDV_ERR] = 0;
            sh.v[ZDV_PREP] = 0;
            sh.huf_valid = 0;
        }
        // ---- my block: where, what ----
        uint32_t e = 0;
        if (mine)
            e = (uint32_t)dir[2 * lane] | ((uint32_t)dir[2 * lane + 1] << 8);
        const uint32_t raw = e >> 15, csz = e & 0x7FFFu;
        uint32_t total;
        const uint32_t off = z_wave_scan_excl(mine ? 3u + csz : 0u, lane, &total);
        bad = bad || total != it.size;
        const uint32_t ubytes = mine ? ((uint32_t)lane + 1u == nunits ? expect - (uint32_t)lane * ZB_UNIT : ZB_UNIT) : 0u;
        const uint8_t* c = p + off + 3u; // my block's content
        uint32_t lmode = 0, nlit = 0, lhdr = 0, lcs = 0, nstr = 0, nbseq = 0, shdr = 0, modes = 0;
        if (mine && !bad)
        {
            const uint32_t h = (uint32_t)p[off] | ((uint32_t)p[off + 1u] << 8) | ((uint32_t)p[off + 2u] << 16);
            const uint32_t is_last = (it.out0 + expect == content && (uint32_t)lane + 1u == nunits) ? 1u : 0u;
            if ((h & 1u) != is_last || ((h >> 1) & 3u) != (raw ? 0u : 2u) || (h >> 3) != csz || (raw && csz != ubytes))
                bad = true;
            else if (raw)
                nlit = ubytes; // the unit's bytes are its literals
            else if (csz < 2u)
                bad = true;
            else
            {
                const uint32_t b0 = c[0], sf = (b0 >> 2) & 3u;
                lmode = b0 & 3u;
                if (lmode < 2u)
                {
                    lhdr = (sf & 1u) == 0u ? 1u : sf == 1u ? 2u : 3u;
                    if (lhdr > csz)
                        bad = true;
                    else
                    {
                        nlit = lhdr == 1u ? b0 >> 3 : lhdr == 2u ? ((uint32_t)c[0] | ((uint32_t)c[1] << 8)) >> 4
                                                                : ((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16)) >> 4;
                        lcs = lmode == 0u ? nlit : 1u;
                    }
                }
                else
                {
                    lhdr = sf < 2u ? 3u : sf == 2u ? 4u : 5u;
                    if (lhdr > csz)
                        bad = true;
                    else if (lhdr == 3u)
                    {
                        const uint32_t hh = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
                        nlit = (hh >> 4) & 0x3FFu;
                        lcs = hh >> 14;
                    }
                    else if (lhdr == 4u)
                    {
                        const uint32_t hh = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
                        nlit = (hh >> 4) & 0x3FFFu;
                        lcs = hh >> 18;
                    }
                    else
                    {
                        const uint32_t hh = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
                        nlit = (hh >> 4) & 0x3FFFFu;
                        lcs = (hh >> 22) | ((uint32_t)c[4] << 10);
                    }
                    nstr = sf == 0u ? 1u : 4u;
                }
                if (!bad && (nlit > ubytes || lcs >= csz - lhdr)) // (at least one byte of sequences section follows)
                    bad = true;
                if (!bad)
                {
                    const uint32_t sp = lhdr + lcs;
                    const uint32_t n0 = c[sp];
                    if (n0 == 0u)
                    {
                        shdr = 1;
                        bad = sp + 1u != csz;
                    }
                    else
                    {
                        if (n0 < 128u)
                        {
                            nbseq = n0;
                            shdr = 1;
                        }
                        else if (n0 < 255u)
                        {
                            bad = sp + 2u > csz;
                            nbseq = bad ? 0u : ((n0 - 128u) << 8) + c[sp + 1u];
                            shdr = 2;
                        }
                        else
                            bad = true; // 0x7F00 sequences and more: not in a 4 KiB unit
                        if (!bad && (sp + shdr + 1u >= csz || nbseq > ZB_UNIT_SEQ_MAX || nbseq == 0u))
                            bad = true;
                        if (!bad)
                        {
                            modes = c[sp + shdr];
                            shdr += 1u;
                            bad = (modes & 3u) != 0u;
                        }
                    }
                }
            }
        }
#ifdef LTHIP_ZB_PROF
        ZB_MARK(19);
#endif
        // ---- who brings the tree, who the tables; is everybody else consistent with them? ----
        const uint64_t treem = __builtin_amdgcn_ballot_w64(mine && !raw && lmode == 2u);
        const uint64_t lessm = __builtin_amdgcn_ballot_w64(mine && !raw && lmode == 3u);
        const uint64_t seqm = __builtin_amdgcn_ballot_w64(mine && nbseq != 0u);
        const int tree_lane = treem ? __builtin_ctzll(treem) : -1;
        const int tab_lane = seqm ? __builtin_ctzll(seqm) : -1;
        if (treem & (treem - 1ull))
            bad = true; // a second tree
        if (lessm && (tree_lane < 0 || (lessm & ((1ull << tree_lane) - 1ull))))
            bad = true; // treeless before the tree
        const uint32_t modes0 = tab_lane >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)modes, tab_lane) : 0u;
        {
            uint32_t want = 0;
            for (int t = 0; t < 3; ++t)
            {
                const uint32_t m = (modes0 >> (6 - 2 * t)) & 3u;
                if (m == 3u)
                    bad = true; // nothing to repeat at the head of a piece
                want |= (m ? 3u : 0u) << (6 - 2 * t);
            }
            if (mine && nbseq != 0u && lane != tab_lane && modes != want)
                bad = true;
        }
        if (__builtin_amdgcn_ballot_w64(bad))
            bad = true;
        // ---- the table descriptions (lane 0), the tables (all lanes; their scratch lies over the Huffman table, so:), THEN the tree
        // (lane 0), all by the decoder core's own readers ----
        uint32_t tree_bytes = 0, desc_bytes = 0;
        if (!bad)
        {
            const uint32_t t_off = tree_lane >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)(off + 3u + lhdr), tree_lane) : 0u;
            const uint32_t t_size = tree_lane >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)lcs, tree_lane) : 0u;
            const uint32_t d_off = tab_lane >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)(off + 3u + lhdr + lcs + shdr), tab_lane) : 0u;
            const uint32_t d_end = tab_lane >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)(off + 3u + csz), tab_lane) : 0u;
            const uint32_t d_staged = d_end - d_off < 512u ? d_end - d_off : 512u, t_staged = t_size < 256u ? t_size : 256u;
            for (uint32_t k = lane; k < d_staged; k += 64)
                s_desc[k] = p[d_off + k];
            for (uint32_t k = lane; k < t_staged; k += 64)
                s_desc[512u + k] = p[t_off + k];
            __syncthreads();
            if (lane == 0)
            {
                sh.v[ZDV_LEN] = 0;
                sh.v[ZDV_LL] = 0;
                if (tab_lane >= 0)
                {
                    uint32_t q = 0; // (a description that ran past the staged bytes fails the readers' bounds checks: the serial decoder then)
                    for (int t = 0; t < 3 && !sh.v[ZDV_ERR]; ++t) // LL, OF, ML
                    {
                        uint32_t used = 0;
                        if (zd_set_table(&sh, t, (modes0 >> (6 - 2 * t)) & 3u, s_desc + q, d_staged - q, &used))
                            sh.v[ZDV_ERR] = 1;
                        q += used;
                    }
                    sh.v[ZDV_LL] = q;
                    if (d_off + q >= d_end)
                        sh.v[ZDV_ERR] = 1;
                }
            }
            __syncthreads();
            bad = sh.v[ZDV_ERR] != 0u;
            desc_bytes = sh.v[ZDV_LL];
            if (!bad && tab_lane >= 0)
            {
                for (int t = 0; t < 3; ++t)
                    if (sh.tb_build[t])
                    {
                        if (zd_build_fse_par(&sh.fse[t], sh.norm + 64 * t, sh.tb_maxsym[t], sh.tb_log[t], sh.cum, (uint32_t*)sh.huf,
                                             sh.huf + 2u * ZD_FSE_PAR_MASK_WORDS, (uint32_t)lane))
                            bad = true;
                        ZB_SYNC_LDS();
                    }
            }
            __syncthreads();
            if (lane == 0 && !bad && tree_lane >= 0)
            {
                const uint32_t tr = zd_read_huf_tree(&sh, s_desc + 512, t_staged);
                if (tr == ZD_ERROR)
                    sh.v[ZDV_ERR] = 1;
                else
                    sh.v[ZDV_LEN] = tr;
            }
            __syncthreads();
            bad = bad || sh.v[ZDV_ERR] != 0u;
            tree_bytes = sh.v[ZDV_LEN];
        }
#ifdef LTHIP_ZB_PROF
        ZB_MARK(20);
#endif
        // ---- literals: offsets, the stream list, raw runs ----
        uint32_t nlit_total, nstr_total, nseq_total;
        const uint32_t lo = z_wave_scan_excl(mine ? nlit : 0u, lane, &nlit_total);
        const bool huf = mine && !raw && lmode >= 2u;
        const uint32_t st0 = z_wave_scan_excl(huf ? nstr : 0u, lane, &nstr_total);
        const uint32_t rec0 = z_wave_scan_excl(mine ? nbseq : 0u, lane, &nseq_total);
        bad = bad || nlit_total > ZD_LIT_MAX || nlit_total > expect;
        if (!bad && huf)
        {
            const uint32_t tb = lane == tree_lane ? tree_bytes : 0u;
            const uint32_t at = off + 3u + lhdr + tb;
            if (lcs < tb)
                bad = true;
            else if (nstr == 1u)
                s_streams[st0] = make_uint4(at, lcs - tb, lo, nlit);
            else if (lcs - tb < 10u)
                bad = true;
            else
            {
                const uint8_t* j = p + at;
                const uint32_t s1 = (uint32_t)j[0] | ((uint32_t)j[1] << 8), s2 = (uint32_t)j[2] | ((uint32_t)j[3] << 8),
                               s3 = (uint32_t)j[4] | ((uint32_t)j[5] << 8);
                const uint32_t body = lcs - tb - 6u, seg = (nlit + 3u) >> 2;
                if (s1 + s2 + s3 >= body || 3u * seg > nlit)
                    bad = true;
                else
                {
                    s_streams[st0] = make_uint4(at + 6u, s1, lo, seg);
                    s_streams[st0 + 1u] = make_uint4(at + 6u + s1, s2, lo + seg, seg);
                    s_streams[st0 + 2u] = make_uint4(at + 6u + s1 + s2, s3, lo + 2u * seg, seg);
                    s_streams[st0 + 3u] = make_uint4(at + 6u + s1 + s2 + s3, body - s1 - s2 - s3, lo + 3u * seg, nlit - 3u * seg);
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(bad))
            bad = true;
        __syncthreads();
        if (!bad)
        {
            // raw / RLE literals and raw blocks: all lanes, unit after unit
            uint64_t plain = __builtin_amdgcn_ballot_w64(mine && nlit != 0u && (raw || lmode < 2u));
            while (plain)
            {
                const int u = __builtin_ctzll(plain);
                plain &= plain - 1ull;
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)nlit, u);
                const uint32_t from = (uint32_t)__builtin_amdgcn_readlane((int)(off + 3u + lhdr), u);
                const uint32_t to = (uint32_t)__builtin_amdgcn_readlane((int)lo, u);
                const uint32_t rle = (uint32_t)__builtin_amdgcn_readlane((int)((!raw && lmode == 1u) ? 1u : 0u), u);
                if (rle)
                    for (uint32_t k = lane; k < n; k += 64)
                        lits[to + k] = p[from];
                else
                {
                    typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
                    const uint32_t nv = n >> 4; // 16 bytes per lane and trip, any alignment on both sides
                    for (uint32_t k = lane; k < nv; k += 64)
                        *reinterpret_cast<u32x4_a1*>(lits + to + 16u * k) = *reinterpret_cast<const u32x4_a1*>(p + from + 16u * k);
                    for (uint32_t k = 16u * nv + (uint32_t)lane; k < n; k += 64)
                        lits[to + k] = p[from + k];
                }
            }
            // Huffman streams: TWO per lane, interleaved (a stream is a chain of dependent table reads: two chains hide half of
            // the latency), each with its next 128 bits in registers (ZsWin: no load on the chain).  Four symbols per step while a
            // stream has 64 bits and four symbols left; the core's careful loop finishes it and gives the verdict.
            const uint32_t tl = sh.huf_log;
            for (uint32_t k0 = 0; k0 < nstr_total; k0 += 128u)
            {
                const uint32_t ka = k0 + (uint32_t)lane, kb = ka + 64u;
                const bool has_a = ka < nstr_total, has_b = kb < nstr_total;
                const uint4 sa = has_a ? s_streams[ka] : make_uint4(0, 0, 0, 0), sb2 = has_b ? s_streams[kb] : make_uint4(0, 0, 0, 0);
                uint32_t pa = 0, pb = 0, ia = 0, ib = 0;
                ZsWin wa, wb;
                bool go_a = false, go_b = false;
                if (has_a)
                {
                    const uint32_t last = sa.y ? p[sa.x + sa.y - 1u] : 0u;
                    if (sa.w == 0u || last == 0u)
                        bad = true;
                    else
                    {
                        pa = (sa.y - 1u) * 8u + (31u - (uint32_t)__builtin_clz(last));
                        wa.open(p + sa.x, sa.y, (uint32_t)(it.src_off + sa.x > 64u ? 64u : it.src_off + sa.x));
                        go_a = true;
                    }
                }
                if (has_b)
                {
                    const uint32_t last = sb2.y ? p[sb2.x + sb2.y - 1u] : 0u;
                    if (sb2.w == 0u || last == 0u)
                        bad = true;
                    else
                    {
                        pb = (sb2.y - 1u) * 8u + (31u - (uint32_t)__builtin_clz(last));
                        wb.open(p + sb2.x, sb2.y, (uint32_t)(it.src_off + sb2.x > 64u ? 64u : it.src_off + sb2.x));
                        go_b = true;
                    }
                }
                uint8_t* const oa = lits + sa.z;
                uint8_t* const ob2 = lits + sb2.z;
                for (;;)
                {
                    const bool fa = go_a && sa.w - ia >= 4u && pa >= 64u, fb = go_b && sb2.w - ib >= 4u && pb >= 64u;
                    if (!__builtin_amdgcn_ballot_w64(fa || fb))
                        break;
                    uint64_t ta = 0, tb2 = 0;
                    if (fa)
                    {
                        wa.ensure(pa, 64u);
                        ta = wa.get64(pa - 64u);
                    }
                    if (fb)
                    {
                        wb.ensure(pb, 64u);
                        tb2 = wb.get64(pb - 64u);
                    }
                    uint32_t ua = 0, ub = 0, qa = 0, qb = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j)
                    {
                        const uint32_t ea = sh.huf[(uint32_t)(ta >> (64u - tl))], eb = sh.huf[(uint32_t)(tb2 >> (64u - tl))];
                        const uint32_t na = ea >> 8, nb2 = eb >> 8;
                        ta <<= na;
                        tb2 <<= nb2;
                        ua += na;
                        ub += nb2;
                        qa |= (ea & 255u) << (8u * j);
                        qb |= (eb & 255u) << (8u * j);
                    }
                    if (fa)
                    {
                        __builtin_memcpy(oa + ia, &qa, 4);
                        pa -= ua;
                        ia += 4u;
                    }
                    if (fb)
                    {
                        __builtin_memcpy(ob2 + ib, &qb, 4);
                        pb -= ub;
                        ib += 4u;
                    }
                }
                if (go_a && zd_huf_stream_from(&sh, p + sa.x, sa.y, oa, sa.w, pa, ia))
                    bad = true;
                if (go_b && zd_huf_stream_from(&sh, p + sb2.x, sb2.y, ob2, sb2.w, pb, ib))
                    bad = true;
            }
        }
        if (__builtin_amdgcn_ballot_w64(bad))
            bad = true;
        __syncthreads();
#ifdef LTHIP_ZB_PROF
        ZB_MARK(21);
#endif
        // ---- the tables, packed: one 8-byte read per state ----
        uint32_t log_l = 0, log_o = 0, log_m = 0;
        if (!bad && tab_lane >= 0)
        {
            for (int t = 0; t < 3; ++t)
            {
                const ZdFse* f = &sh.fse[t];
                uint64_t* pk = t == ZT_LL ? pk_ll : t == ZT_ML ? pk_ml : pk_of;
                const uint32_t size = f->valid == 2u ? 1u : f->valid == 1u ? 1u << f->log : 0u;
                if (size == 0u || size > (t == ZT_OF ? 256u : 512u))
                    bad = true;
                const uint32_t lg = f->valid == 2u ? 0u : f->log;
                if (t == ZT_LL)
                    log_l = lg;
                else if (t == ZT_OF)
                    log_o = lg;
                else
                    log_m = lg;
                for (uint32_t x = lane; x < size && !bad; x += 64)
                {
                    const uint32_t sym = f->sym[x];
                    uint32_t baseline, ebits;
                    if (t == ZT_LL)
                    {
                        baseline = zb_ll_base(sym & 63u);
                        ebits = zb_ll_bits(sym & 63u);
                        bad = sym > 35u;
                    }
                    else if (t == ZT_ML)
                    {
                        baseline = zb_ml_base(sym & 63u) + 3u;
                        ebits = zb_ml_bits(sym & 63u);
                        bad = sym > 52u;
                    }
                    else
                    {
                        baseline = 1u << (sym & 31u);
                        ebits = sym & 31u;
                        bad = sym > 31u; // (codes above 23 may sit in a table -- the predefined one has 29 --: a sequence that USES one is stopped in zs_seq_lanes)
                    }
                    // the table's own fields are read before its packed form lands on them?  No: the packed tables lie over the
                    // Huffman table and the weights' table, never over sh.fse
                    pk[x] = (uint64_t)(f->valid == 2u ? 0u : f->base[x]) | ((uint64_t)(f->valid == 2u ? 0u : f->nb[x]) << 16) | ((uint64_t)ebits << 24) |
                            ((uint64_t)baseline << 32);
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(bad))
            bad = true;
        __syncthreads();
#ifdef LTHIP_ZB_PROF
        ZB_MARK(22);
#endif
        // ---- sequences: lane u the bit-stream of block u ----
        uint32_t sum_ll = 0, sum_ml = 0;
        {
            const uint32_t at = off + 3u + lhdr + lcs + shdr + (lane == tab_lane ? desc_bytes : 0u);
            const uint32_t end = off + 3u + csz;
            const bool had = !bad && mine && nbseq != 0u;
            if (it.pad) // (wave-uniform: the frame's trailer version)
                zs_seq_lanes<2>(had, p + at, at < end ? end - at : 0u, (uint32_t)(it.src_off + at > 64u ? 64u : it.src_off + at), nbseq, log_l, log_o,
                                log_m, pk_ll, pk_of, pk_ml, recs + rec0, bad, sum_ll, sum_ml);
            else
                zs_seq_lanes<1>(had, p + at, at < end ? end - at : 0u, (uint32_t)(it.src_off + at > 64u ? 64u : it.src_off + at), nbseq, log_l, log_o,
                                log_m, pk_ll, pk_of, pk_ml, recs + rec0, bad, sum_ll, sum_ml);
            // the block must regenerate exactly its unit
            if (had && !bad && (sum_ll > nlit || nlit + sum_ml != ubytes))
                bad = true;
        }
        if (!bad && mine && nbseq == 0u && nlit != ubytes)
            bad = true; // no sequences: all literals
        if (__builtin_amdgcn_ballot_w64(bad))
            bad = true;
#ifdef LTHIP_ZB_PROF
        ZB_MARK(23);
#endif
        // ---- a block's last literals go with the next sequence of the piece ----
        if (!bad)
        {
            const uint32_t tail = mine ? nlit - sum_ll : 0u;
            uint32_t run = 0, carry = 0;
            for (uint32_t u = 0; u < nunits; ++u)
            {
                const uint32_t tu = (uint32_t)__builtin_amdgcn_readlane((int)tail, (int)u);
                const bool has = (seqm >> u) & 1ull;
                if ((uint32_t)lane == u)
                    carry = run;
                run = has ? tu : run + tu;
            }
            if (mine && nbseq != 0u && carry != 0u)
                recs[rec0] += carry; // (20 bits hold a whole piece of literals)
        }
        ZPrep pr;
        pr.bits_off = 0;
        pr.bits_size = 0;
        pr.nbseq = nseq_total;
        pr.nlit = nlit_total;
        pr.log[0] = pr.log[1] = pr.log[2] = 0;
        pr.expect = expect;
        pr.status = bad ? ZP_SERIAL : ZP_READY;
        if (lane == 0)
        {
            prep[i] = pr;
            if (bad)
                retry[it.payload] = 1u;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Frames of other encoders, block-parallel (fitems, kind 4): k_zstd_blk_entropy, one wave per BLOCK.
// A block's streams need nothing from the blocks before it except their entropy tables (Treeless literals, Repeat_Mode), and those
// are found without decoding anything: the wave walks the headers of the blocks before its own (section headers only), notes which
// block last SET the Huffman tree and each of the three FSE tables, and builds them from there with the decoder core's readers.
// Then it decodes its own block's literal streams (two per lane) into the block's literal buffer and its sequence bit-stream (one
// lane) into records -- offset values 1..3 (repeat offsets) stay in the records; k_zstd_execute<2> resolves them when it executes
// the payload's blocks in order.  Whatever is unusual gives the payload back to the serial decoder.
// ---------------------------------------------------------------------------------------------------
struct ZfBlk
{
    uint32_t lmode, nlit, lhdr, lcs, nstr, nbseq, shdr, modes;
    bool ok;
};
// section headers of a Compressed_Block's content c[0 .. csz)
__device__ __forceinline__ ZfBlk zf_parse(const uint8_t* c, uint32_t csz)
{
    ZfBlk f;
    f.lmode = f.nlit = f.lhdr = f.lcs = f.nstr = f.nbseq = f.shdr = f.modes = 0;
    f.ok = false;
    if (csz < 2u)
        return f;
    const uint32_t b0 = c[0], sf = (b0 >> 2) & 3u;
    f.lmode = b0 & 3u;
    if (f.lmode < 2u)
    {
        f.lhdr = (sf & 1u) == 0u ? 1u : sf == 1u ? 2u : 3u;
        if (f.lhdr > csz)
            return f;
        f.nlit = f.lhdr == 1u ? b0 >> 3 : f.lhdr == 2u ? ((uint32_t)c[0] | ((uint32_t)c[1] << 8)) >> 4
                                                       : ((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16)) >> 4;
        f.lcs = f.lmode == 0u ? f.nlit : 1u;
    }
    else
    {
        f.lhdr = sf < 2u ? 3u : sf == 2u ? 4u : 5u;
        if (f.lhdr > csz)
            return f;
        if (f.lhdr == 3u)
        {
            const uint32_t hh = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
            f.nlit = (hh >> 4) & 0x3FFu;
            f.lcs = hh >> 14;
        }
        else if (f.lhdr == 4u)
        {
            const uint32_t hh = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            f.nlit = (hh >> 4) & 0x3FFFu;
            f.lcs = hh >> 18;
        }
        else
        {
            const uint32_t hh = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            f.nlit = (hh >> 4) & 0x3FFFFu;
            f.lcs = (hh >> 22) | ((uint32_t)c[4] << 10);
        }
        f.nstr = sf == 0u ? 1u : 4u;
    }
    if (f.nlit > ZD_LIT_MAX || f.lcs >= csz - f.lhdr) // (at least one byte of sequences section follows)
        return f;
    const uint32_t sp = f.lhdr + f.lcs;
    const uint32_t n0 = c[sp];
    if (n0 == 0u)
    {
        f.shdr = 1;
        f.ok = sp + 1u == csz;
        return f;
    }
    if (n0 < 128u)
    {
        f.nbseq = n0;
        f.shdr = 1;
    }
    else if (n0 < 255u)
    {
        if (sp + 2u > csz)
            return f;
        f.nbseq = ((n0 - 128u) << 8) + c[sp + 1u];
        f.shdr = 2;
    }
    else
    {
        if (sp + 3u > csz)
            return f;
        f.nbseq = (uint32_t)c[sp + 1u] + ((uint32_t)c[sp + 2u] << 8) + 0x7F00u;
        f.shdr = 3;
    }
    if (sp + f.shdr + 1u >= csz || f.nbseq == 0u)
        return f;
    f.modes = c[sp + f.shdr];
    f.shdr += 1u;
    f.ok = (f.modes & 3u) == 0u;
    return f;
}

__global__ __launch_bounds__(64) void k_zstd_blk_entropy(const uint8_t* __restrict__ src, const ZItem* __restrict__ fitems, const uint32_t* __restrict__ flist,
                                                         uint32_t nlist, uint32_t* __restrict__ slist, uint32_t* __restrict__ scount,
                                                         uint8_t* __restrict__ lit_scratch, uint64_t* __restrict__ rec_scratch,
                                                         ZPrep* __restrict__ fprep, uint32_t* __restrict__ retry, uint32_t* __restrict__ ticket,
                                                         unsigned long long* __restrict__ bump, uint64_t lit_cap, uint64_t rec_cap,
                                                         uint64_t* __restrict__ tabs, uint32_t inline_seqs)
{
    __shared__ ZdShared sh;
    __shared__ uint4 s_streams[4];
    __shared__ __attribute__((aligned(16))) uint8_t s_desc[512 + 256];
    const int lane = threadIdx.x;
    uint64_t* const pk_ll = reinterpret_cast<uint64_t*>(sh.huf);
    uint64_t* const pk_ml = pk_ll + 512;
    uint64_t* const pk_of = reinterpret_cast<uint64_t*>(&sh.wtab);
    for (;;)
    {
        __builtin_amdgcn_wave_barrier();
        uint32_t tk = atomicAdd(ticket, lane == 0 ? 1u : 0u);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= nlist)
            break;
        const uint32_t i = flist[tk];
        const ZItem it = fitems[i];
        if (it.kind != 4u)
            continue;
        __syncthreads();
#ifdef LTHIP_ZB_PROF
        if (lane == 0)
            g_zb_last[blockIdx.x] = wall_clock64();
#endif
        const uint8_t* p = src + it.src_off; // the payload
        ZPrep pr;
        pr.bits_off = 0;
        pr.bits_size = pr.nbseq = pr.nlit = 0;
        pr.log[0] = pr.log[1] = pr.log[2] = 0;
        pr.expect = 0;
        pr.status = ZP_SERIAL;
        pr.rec_at = 0;
        pr.seq_off = pr.seq_size = 0;
        uint32_t why = 0;
        bool bad = false;
        if (lane == 0)
        {
            sh.v[ZDV_ERR] = 0;
            sh.v[ZDV_PREP] = 0;
            sh.huf_valid = 0;
        }
        const uint32_t bh = (uint32_t)p[it.pad] | ((uint32_t)p[it.pad + 1u] << 8) | ((uint32_t)p[it.pad + 2u] << 16);
        const uint32_t type = (bh >> 1) & 3u, bsize = bh >> 3;
        if (type != 2u)
        {
            // Raw_Block / RLE_Block: where the bytes are (the executor copies / fills)
            pr.log[0] = type == 0u ? 1u : 2u;
            pr.bits_off = it.src_off + it.pad + 3u;
            pr.expect = bsize;
            pr.status = ZP_DONE;
            if (lane == 0)
                fprep[i] = pr;
            continue;
        }
        const uint32_t c0 = it.pad + 3u; // my block's content inside the payload
        const ZfBlk me = zf_parse(p + c0, bsize);
        bad = !me.ok;
        // room for my literals and my records in the two arenas (a wave-uniform draw; an arena that is full sends the payload to the
        // serial decoder: the host sizes them for a record per six bytes of output)
        unsigned long long lit_at = 0, rec_at = 0;
        if (!bad)
        {
            const unsigned long long want_l = ((unsigned long long)me.nlit + 79ull) & ~15ull;
            lit_at = atomicAdd(&bump[0], lane == 0 ? want_l : 0ull);
            rec_at = atomicAdd(&bump[1], lane == 0 ? (unsigned long long)me.nbseq : 0ull);
            lit_at = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(lit_at >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)lit_at);
            rec_at = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(rec_at >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)rec_at);
            if (lit_at + want_l > lit_cap || rec_at + me.nbseq > rec_cap)
                { bad = true; why = __LINE__; }
        }
        uint8_t* lits = lit_scratch + lit_at;
        // ---- who set the tree / the tables last?  (the section headers of the blocks up to mine) ----
        uint32_t tree_at = 0, tree_size = 0, tab_at[3] = {0, 0, 0}, tab_end[3] = {0, 0, 0}, tab_modes[3] = {0, 0, 0};
        bool have_tree = false, have_tab[3] = {false, false, false};
        if (!bad)
        {
            const ZFrameHdr fh = z_frame_header(p, it.size);
            uint32_t q = fh.size;
            for (uint32_t j = 0; j <= it.aux && !bad; ++j)
            {
                const uint32_t h = (uint32_t)p[q] | ((uint32_t)p[q + 1u] << 8) | ((uint32_t)p[q + 2u] << 16);
                const uint32_t ty = (h >> 1) & 3u, sz = h >> 3;
                if (ty == 2u)
                {
                    const ZfBlk f = j == it.aux ? me : zf_parse(p + q + 3u, sz);
                    if (!f.ok)
                        { bad = true; why = __LINE__; }
                    else
                    {
                        if (f.lmode == 2u)
                        {
                            have_tree = true;
                            tree_at = q + 3u + f.lhdr;
                            tree_size = f.lcs;
                        }
                        if (f.nbseq)
                            for (int t = 0; t < 3; ++t)
                                if (((f.modes >> (6 - 2 * t)) & 3u) != 3u)
                                {
                                    have_tab[t] = true;
                                    tab_at[t] = q + 3u + f.lhdr + f.lcs + f.shdr; // the first description of that block
                                    tab_end[t] = q + 3u + sz;
                                    tab_modes[t] = f.modes;
                                }
                    }
                }
                q += 3u + (ty == 1u ? 1u : sz);
            }
            if ((me.lmode == 3u && !have_tree) || (me.nbseq && !(have_tab[0] && have_tab[1] && have_tab[2])))
                { bad = true; why = __LINE__; } // nothing to repeat: the serial decoder says so
        }
#ifdef LTHIP_ZB_PROF
        ZB_MARK(19);
#endif
        // ---- the three tables (descriptions by lane 0 from staged bytes, tables by all lanes), then the tree ----
        if (!bad && me.nbseq)
        {
            for (int t = 0; t < 3 && !bad; ++t)
            {
                const uint32_t staged = tab_end[t] - tab_at[t] < 512u ? tab_end[t] - tab_at[t] : 512u;
                __syncthreads();
                for (uint32_t k = lane; k < staged; k += 64)
                    s_desc[k] = p[tab_at[t] + k];
                __syncthreads();
                if (lane == 0)
                {
                    // the descriptions of a block lie one behind the other (LL, OF, ML): skip the ones before mine
                    uint32_t q = 0;
                    for (int tt = 0; tt <= t && !sh.v[ZDV_ERR]; ++tt)
                    {
                        const uint32_t m = (tab_modes[t] >> (6 - 2 * tt)) & 3u;
                        uint32_t used = 0;
                        if ((m == 1u || m == 2u) && q >= staged) // a description past what was staged (also the one-byte RLE form):
                        {                                        // `staged - q` must never wrap -- the payload goes to the serial decoder
                            sh.v[ZDV_ERR] = 1;
                            break;
                        }
                        if (tt == t)
                        {
                            if (zd_set_table(&sh, t, m, s_desc + q, staged - q, &used))
                                sh.v[ZDV_ERR] = 1;
                        }
                        else if (m == 1u)
                            used = 1;
                        else if (m == 2u)
                        {
                            uint32_t maxsym = tt == ZT_LL ? 35u : tt == ZT_ML ? 52u : 31u, tl = 0;
                            used = q < staged ? zd_read_ncount(s_desc + q, staged - q, sh.norm + 192, &maxsym, zb_table_max_log(tt), &tl) : ZD_ERROR;
                            if (used == ZD_ERROR)
                            {
                                sh.v[ZDV_ERR] = 1;
                                used = 0;
                            }
                        }
                        q += used;
                        if (q > staged)
                            sh.v[ZDV_ERR] = 1;
                    }
                }
                __syncthreads();
                bad = sh.v[ZDV_ERR] != 0u;
                if (!bad && sh.tb_build[t])
                {
                    if (zd_build_fse_par(&sh.fse[t], sh.norm + 64 * t, sh.tb_maxsym[t], sh.tb_log[t], sh.cum, (uint32_t*)sh.huf,
                                         sh.huf + 2u * ZD_FSE_PAR_MASK_WORDS, (uint32_t)lane))
                        { bad = true; why = __LINE__; }
                    ZB_SYNC_LDS();
                }
            }
            __syncthreads();
        }
        uint32_t tree_bytes = 0;
        if (!bad && me.lmode >= 2u)
        {
            const uint32_t staged = tree_size < 256u ? tree_size : 256u;
            for (uint32_t k = lane; k < staged; k += 64)
                s_desc[512u + k] = p[tree_at + k];
            __syncthreads();
            if (lane == 0)
            {
                const uint32_t tr = zd_read_huf_tree(&sh, s_desc + 512, staged);
                if (tr == ZD_ERROR)
                    sh.v[ZDV_ERR] = 1;
                else
                    sh.v[ZDV_LEN] = tr;
            }
            __syncthreads();
            bad = sh.v[ZDV_ERR] != 0u;
            tree_bytes = me.lmode == 2u ? sh.v[ZDV_LEN] : 0u;
        }
#ifdef LTHIP_ZB_PROF
        ZB_MARK(20);
#endif
        // ---- literals ----
        uint32_t nstr_total = 0;
        if (!bad)
        {
            const uint32_t at0 = c0 + me.lhdr;
            if (me.lmode == 0u)
            {
                typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
                const uint32_t nv = me.nlit >> 4;
                for (uint32_t k = lane; k < nv; k += 64)
                    *reinterpret_cast<u32x4_a1*>(lits + 16u * k) = *reinterpret_cast<const u32x4_a1*>(p + at0 + 16u * k);
                for (uint32_t k = 16u * nv + (uint32_t)lane; k < me.nlit; k += 64)
                    lits[k] = p[at0 + k];
            }
            else if (me.lmode == 1u)
                for (uint32_t k = lane; k < me.nlit; k += 64)
                    lits[k] = p[at0];
            else
            {
                const uint32_t at = at0 + tree_bytes;
                if (me.lcs < tree_bytes)
                    { bad = true; why = __LINE__; }
                else if (me.nstr == 1u)
                {
                    if (lane == 0)
                        s_streams[0] = make_uint4(at, me.lcs - tree_bytes, 0u, me.nlit);
                    nstr_total = 1;
                }
                else if (me.lcs - tree_bytes < 10u)
                    { bad = true; why = __LINE__; }
                else
                {
                    const uint8_t* j = p + at;
                    const uint32_t s1 = (uint32_t)j[0] | ((uint32_t)j[1] << 8), s2 = (uint32_t)j[2] | ((uint32_t)j[3] << 8),
                                   s3 = (uint32_t)j[4] | ((uint32_t)j[5] << 8);
                    const uint32_t body = me.lcs - tree_bytes - 6u, seg = (me.nlit + 3u) >> 2;
                    if (s1 + s2 + s3 >= body || 3u * seg > me.nlit)
                        { bad = true; why = __LINE__; }
                    else
                    {
                        if (lane == 0)
                        {
                            s_streams[0] = make_uint4(at + 6u, s1, 0u, seg);
                            s_streams[1] = make_uint4(at + 6u + s1, s2, seg, seg);
                            s_streams[2] = make_uint4(at + 6u + s1 + s2, s3, 2u * seg, seg);
                            s_streams[3] = make_uint4(at + 6u + s1 + s2 + s3, body - s1 - s2 - s3, 3u * seg, me.nlit - 3u * seg);
                        }
                        nstr_total = 4;
                    }
                }
            }
        }
        __syncthreads();
        if (!bad && nstr_total)
        {
            // the streams: one per lane (a block has at most four; the loop is k_zstd_sub_entropy's, its second stream unused: this
            // wave has lanes to spare, and a lane alone with one chain finishes it sooner than two lanes with two chains each)
            const uint32_t tl = sh.huf_log;
            const uint32_t ka = (uint32_t)lane, kb = ka + 64u;
            const bool has_a = ka < nstr_total, has_b = kb < nstr_total;
            const uint4 sa = has_a ? s_streams[ka] : make_uint4(0, 0, 0, 0), sb2 = has_b ? s_streams[kb] : make_uint4(0, 0, 0, 0);
            uint32_t pa = 0, pb = 0, ia = 0, ib = 0;
            ZsWin wa, wb;
            wa.base = wb.base = p;
            wa.hi = wa.lo = wa.nx = wb.hi = wb.lo = wb.nx = 0;
            wa.top = wb.top = 0;
            wa.reach = wb.reach = 0;
            bool go_a = false, go_b = false;
            if (has_a)
            {
                const uint32_t last = sa.y ? p[sa.x + sa.y - 1u] : 0u;
                if (sa.w == 0u || last == 0u)
                    { bad = true; why = __LINE__; }
                else
                {
                    pa = (sa.y - 1u) * 8u + (31u - (uint32_t)__builtin_clz(last));
                    wa.open(p + sa.x, sa.y, (uint32_t)(it.src_off + sa.x > 64u ? 64u : it.src_off + sa.x));
                    go_a = true;
                }
            }
            if (has_b)
            {
                const uint32_t last = sb2.y ? p[sb2.x + sb2.y - 1u] : 0u;
                if (sb2.w == 0u || last == 0u)
                    { bad = true; why = __LINE__; }
                else
                {
                    pb = (sb2.y - 1u) * 8u + (31u - (uint32_t)__builtin_clz(last));
                    wb.open(p + sb2.x, sb2.y, (uint32_t)(it.src_off + sb2.x > 64u ? 64u : it.src_off + sb2.x));
                    go_b = true;
                }
            }
            uint8_t* const oa = lits + sa.z;
            uint8_t* const ob2 = lits + sb2.z;
            for (;;)
            {
                const bool fa = go_a && sa.w - ia >= 4u && pa >= 64u, fb = go_b && sb2.w - ib >= 4u && pb >= 64u;
                if (!__builtin_amdgcn_ballot_w64(fa || fb))
                    break;
                uint64_t ta = 0, tb2 = 0;
                if (fa)
                {
                    wa.ensure(pa, 64u);
                    ta = wa.get64(pa - 64u);
                }
                if (fb)
                {
                    wb.ensure(pb, 64u);
                    tb2 = wb.get64(pb - 64u);
                }
                uint32_t ua = 0, ub = 0, qa = 0, qb = 0;
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j)
                {
                    const uint32_t ea = sh.huf[(uint32_t)(ta >> (64u - tl))], eb = sh.huf[(uint32_t)(tb2 >> (64u - tl))];
                    const uint32_t na = ea >> 8, nb2 = eb >> 8;
                    ta <<= na;
                    tb2 <<= nb2;
                    ua += na;
                    ub += nb2;
                    qa |= (ea & 255u) << (8u * j);
                    qb |= (eb & 255u) << (8u * j);
                }
                if (fa)
                {
                    __builtin_memcpy(oa + ia, &qa, 4);
                    pa -= ua;
                    ia += 4u;
                }
                if (fb)
                {
                    __builtin_memcpy(ob2 + ib, &qb, 4);
                    pb -= ub;
                    ib += 4u;
                }
            }
            if (go_a && zd_huf_stream_from(&sh, p + sa.x, sa.y, oa, sa.w, pa, ia))
                { bad = true; why = __LINE__; }
            if (go_b && zd_huf_stream_from(&sh, p + sb2.x, sb2.y, ob2, sb2.w, pb, ib))
                { bad = true; why = __LINE__; }
        }
        if (__builtin_amdgcn_ballot_w64(bad))
            { bad = true; why = __LINE__; }
        __syncthreads();
#ifdef LTHIP_ZB_PROF
        ZB_MARK(21);
#endif
        // ---- the tables, packed; the sequences (lane 0) ----
        uint32_t log_l = 0, log_o = 0, log_m = 0, seq_ml = 0;
        if (!bad && me.nbseq)
        {
            for (int t = 0; t < 3; ++t)
            {
                const ZdFse* f = &sh.fse[t];
                uint64_t* pk = t == ZT_LL ? pk_ll : t == ZT_ML ? pk_ml : pk_of;
                const uint32_t size = f->valid == 2u ? 1u : f->valid == 1u ? 1u << f->log : 0u;
                if (size == 0u || size > (t == ZT_OF ? 256u : 512u))
                    { bad = true; why = __LINE__; }
                const uint32_t lg = f->valid == 2u ? 0u : f->log;
                if (t == ZT_LL)
                    log_l = lg;
                else if (t == ZT_OF)
                    log_o = lg;
                else
                    log_m = lg;
                for (uint32_t x = lane; x < size && !bad; x += 64)
                {
                    const uint32_t sym = f->sym[x];
                    uint32_t baseline, ebits;
                    if (t == ZT_LL)
                    {
                        baseline = zb_ll_base(sym & 63u);
                        ebits = zb_ll_bits(sym & 63u);
                        bad = sym > 35u;
                    }
                    else if (t == ZT_ML)
                    {
                        baseline = zb_ml_base(sym & 63u) + 3u;
                        ebits = zb_ml_bits(sym & 63u);
                        bad = sym > 52u;
                    }
                    else
                    {
                        baseline = 1u << (sym & 31u);
                        ebits = sym & 31u;
                        bad = sym > 31u; // (codes above 23 may sit in a table -- the predefined one has 29 --: a sequence that USES one is stopped in zs_seq_lanes)
                    }
                    pk[x] = (uint64_t)(f->valid == 2u ? 0u : f->base[x]) | ((uint64_t)(f->valid == 2u ? 0u : f->nb[x]) << 16) | ((uint64_t)ebits << 24) |
                            ((uint64_t)baseline << 32);
                }
            }
            if (__builtin_amdgcn_ballot_w64(bad))
                { bad = true; why = __LINE__; }
            __syncthreads();
            const uint32_t at = c0 + me.lhdr + me.lcs + me.shdr + (uint32_t)0;
            // my own block's descriptions precede the bit-stream: their size = the bytes its non-repeated tables took
            uint32_t skip = 0;
            if (!bad)
            {
                if (lane == 0)
                {
                    uint32_t q = 0;
                    const uint32_t avail = c0 + bsize - at;
                    for (int tt = 0; tt < 3 && !sh.v[ZDV_ERR]; ++tt)
                    {
                        const uint32_t m = (me.modes >> (6 - 2 * tt)) & 3u;
                        if (m == 1u)
                            q += 1u;
                        else if (m == 2u)
                        {
                            uint32_t maxsym = tt == ZT_LL ? 35u : tt == ZT_ML ? 52u : 31u, tl2 = 0;
                            const uint32_t used = q < avail ? zd_read_ncount(p + at + q, avail - q, sh.norm + 192, &maxsym, zb_table_max_log(tt), &tl2) : ZD_ERROR;
                            if (used == ZD_ERROR)
                                sh.v[ZDV_ERR] = 1;
                            else
                                q += used;
                        }
                    }
                    sh.v[ZDV_LL] = q;
                    if (q >= avail)
                        sh.v[ZDV_ERR] = 1;
                }
                __syncthreads();
                bad = sh.v[ZDV_ERR] != 0u;
                skip = sh.v[ZDV_LL];
            }
            const uint32_t sat = at + skip, send = c0 + bsize;
            // the packed tables go to the block's place in the table arena: k_zstd_blk_sequences decodes the bit-streams of 64 blocks
            // per wave, a lane each, reading its states' entries from there.  ONE lane of this wave takes 2.3 ms per block with the
            // tables in shared memory -- which is the faster way as long as the call has no more blocks than the machine has waves for
            // (a lane of k_zstd_blk_sequences takes 14 ms for its block, however few there are): inline_seqs.
            if (!bad && inline_seqs)
            {
                uint32_t sum_ll = 0, sum_ml = 0;
                zs_seq_lanes<0>(lane == 0, p + sat, sat < send ? send - sat : 0u, (uint32_t)(it.src_off + sat > 64u ? 64u : it.src_off + sat),
                                    me.nbseq, log_l, log_o, log_m, pk_ll, pk_of, pk_ml, rec_scratch + rec_at, bad, sum_ll, sum_ml);
                sum_ll = (uint32_t)__builtin_amdgcn_readfirstlane(sum_ll);
                sum_ml = (uint32_t)__builtin_amdgcn_readfirstlane(sum_ml);
                if (__builtin_amdgcn_ballot_w64(bad))
                    bad = true;
                if (!bad && (sum_ll > me.nlit || me.nlit + sum_ml > ZB))
                    bad = true;
                seq_ml = sum_ml;
            }
            else if (!bad)
            {
                // (its place in the table arena = its place in the list of blocks k_zstd_blk_sequences has to visit)
                uint32_t si = atomicAdd(scount, lane == 0 ? 1u : 0u);
                si = __builtin_amdgcn_readfirstlane(si);
                if (lane == 0)
                    slist[si] = i;
                uint64_t* tp = tabs + (uint64_t)si * 1280u;
                for (uint32_t x = lane; x < 512u; x += 64)
                {
                    tp[x] = pk_ll[x];
                    tp[512u + x] = pk_ml[x];
                }
                for (uint32_t x = lane; x < 256u; x += 64)
                    tp[1024u + x] = pk_of[x];
                pr.seq_off = sat;
                pr.seq_size = sat < send ? send - sat : 0u;
                pr.log[1] = log_l | (log_o << 8) | (log_m << 16);
            }
        }
        else if (!bad && me.nlit > ZB)
            { bad = true; why = __LINE__; }
#ifdef LTHIP_ZB_PROF
        ZB_MARK(23);
#endif
        pr.nbseq = me.nbseq;
        pr.nlit = me.nlit;
        pr.expect = me.nlit + seq_ml; // (k_zstd_blk_sequences adds the match lengths when the sequences are its)
        pr.bits_off = lit_at;
        pr.rec_at = rec_at;
        pr.status = bad ? ZP_SERIAL : ZP_READY;
        if (lane == 0)
        {
            fprep[i] = pr;
            if (bad)
                retry[it.payload] = why ? why : __LINE__;
        }
    }
}

// The sequence bit-streams of other encoders' blocks: 64 blocks per wave, a lane each (zs_seq_lanes), every lane reading the entries of
// ITS block's tables from the table arena (three 8-byte gathers per sequence: the machine has the lanes and the L2 for them; one lane
// with its tables in shared memory, the other 63 idle, is what made k_zstd_blk_entropy take 2.3 ms per block).
__global__ __launch_bounds__(64) void k_zstd_blk_sequences(const uint8_t* __restrict__ src, const ZItem* __restrict__ fitems, const uint32_t* __restrict__ slist,
                                                          const uint32_t* __restrict__ scount,
                                                          const uint64_t* __restrict__ tabs, uint64_t* __restrict__ rec_scratch,
                                                          ZPrep* __restrict__ fprep, uint32_t* __restrict__ retry)
{
    const uint32_t g = blockIdx.x * 64u + threadIdx.x;
    if (blockIdx.x * 64u >= *scount)
        return;
    ZItem it;
    it.kind = 0;
    it.src_off = 0;
    it.payload = 0;
    uint32_t i = 0;
    if (g < *scount)
    {
        i = slist[g];
        it = fitems[i];
    }
    ZPrep pr;
    pr.status = ZP_SERIAL;
    pr.nbseq = pr.nlit = pr.seq_off = pr.seq_size = 0;
    pr.log[0] = 1;
    pr.log[1] = 0;
    pr.rec_at = 0;
    if (it.kind == 4u)
        pr = fprep[i];
    const bool act = it.kind == 4u && pr.status == ZP_READY && pr.log[0] == 0u && pr.nbseq != 0u;
    const uint64_t* tp = tabs + (uint64_t)g * 1280u;
    bool bad = false;
    uint32_t sum_ll = 0, sum_ml = 0;
    const uint64_t at = it.src_off + pr.seq_off;
    zs_seq_lanes<0>(act, src + at, pr.seq_size, (uint32_t)(at > 64u ? 64u : at), pr.nbseq, pr.log[1] & 255u, (pr.log[1] >> 8) & 255u,
                        (pr.log[1] >> 16) & 255u, tp, tp + 1024, tp + 512, rec_scratch + pr.rec_at, bad, sum_ll, sum_ml);
    if (act)
    {
        if (!bad && (sum_ll > pr.nlit || pr.nlit + sum_ml > ZB))
            bad = true;
        if (bad)
        {
            fprep[i].status = ZP_SERIAL;
            retry[it.payload] = __LINE__;
        }
        else
            fprep[i].expect = pr.nlit + sum_ml; // what the block regenerates
    }
}

// Pieces of directory frames that are ONE Raw_Block or RLE_Block (kind 2 with aux != 0: k_zstd_split has checked the header against
// the directory): a copy / a fill, 256 threads per piece -- no reason to send them through the decoder core.
__global__ __launch_bounds__(256) void k_zstd_plain_pieces(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, const ZItem* __restrict__ items,
                                                           const uint32_t* __restrict__ item_count, uint32_t item0, uint32_t item1,
                                                           uint8_t* __restrict__ dst, ZPrep* __restrict__ prep, const uint32_t* __restrict__ perm)
{
    if (item0 + blockIdx.x >= item1 || item0 + blockIdx.x >= *item_count)
        return;
    const uint32_t i = perm[item0 + blockIdx.x];
    const ZItem it = items[i];
    if (it.kind != 2u || it.aux == 0u)
        return;
    const ZBlock blk = blocks[it.payload];
    const uint8_t* p = src + it.src_off;
    const uint32_t bh = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    const uint32_t len = bh >> 3;
    uint8_t* out = dst + blk.dst_off + it.out0;
    if ((uint64_t)it.out0 + len > (uint64_t)blk.dst_cap)
        return; // (cannot happen: the content size fits the capacity; the piece then stays for the serial piece decoder)
    if (((bh >> 1) & 3u) == 1u)
    {
        const uint8_t b = p[3];
        for (uint32_t k = threadIdx.x; k < len; k += 256)
            out[k] = b;
    }
    else
        wg_copy16(out, p + 3, len, threadIdx.x);
    if (threadIdx.x == 0)
    {
        ZPrep pr;
        pr.bits_off = 0;
        pr.bits_size = pr.nbseq = pr.nlit = 0;
        pr.log[0] = pr.log[1] = pr.log[2] = 0;
        pr.expect = len;
        pr.status = ZP_DONE;
        prep[i] = pr;
    }
}

// payloads the sub-block decoder gave back: the whole payload, serially (the plain decoder's verdict and bytes)
__global__ __launch_bounds__(64) void k_zstd_decode_retry(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, uint32_t nblocks,
                                                          const uint32_t* __restrict__ retry, uint8_t* __restrict__ dst,
                                                          uint8_t* __restrict__ lit_scratch, uint32_t* __restrict__ out_sizes)
{
    __shared__ ZdShared sh;
    uint8_t* lits = lit_scratch + (uint64_t)blockIdx.x * (ZD_LIT_MAX + 64u);
    if (threadIdx.x == 0)
        sh.v[ZDV_PREP] = 0;
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x)
    {
        if (!retry[b])
            continue;
        const ZBlock blk = blocks[b];
        __syncthreads();
        const uint32_t n = zd_decode_payload_ex(src + blk.src_off, blk.size, dst + blk.dst_off, blk.dst_cap, lits, &sh, threadIdx.x, ZD_WHOLE);
        if (threadIdx.x == 0)
            out_sizes[b] = n;
        __syncthreads();
    }
}

constexpr uint32_t ZX_RING = 8192u, ZX_FLUSH = 2048u, ZX_LIT = 2048u;
constexpr uint32_t ZX_ML_LANE = 64u;                         // the longest match a sequence's own lane copies
constexpr uint32_t ZX_RING_SAFE = ZX_RING - 64u * 80u - 64u; // a run appends at most 64 x (16 + 64) bytes ahead of `op`
constexpr uint32_t ZX_BATCH_ADV = 64u * 80u, ZX_BATCH_LL = 1100u; // a batch of 64 sequences executed in one pass: at most as much as a run
constexpr uint32_t ZX_LL_OWN = 32u;                              // ... literal runs up to this by the sequence's own lane

// output of one piece: LDS ring + flush, literal stream through an LDS window (piece-local 32-bit positions)
struct ZxOut
{
    uint8_t* s_ring;
    uint8_t* s_lit;
    uint8_t* out_al;       // piece output byte q is out_al[q + g]
    const uint8_t* lit_al; // literal p is lit_al[p + lh]
    uint32_t g, lh, nlit, cap;
    uint32_t lo; // aligned offset of the first byte that is this piece's to write (g; a chain member: the piece before it lies below)
    int lane;
    uint32_t op, flushed, drained;
    int32_t lwa; // aligned literal offset of s_lit[0]

    __device__ __forceinline__ uint32_t ring(uint32_t q) const { return (q + g) & (ZX_RING - 1u); }
    __device__ __forceinline__ void flush(uint32_t upto) // aligned offsets [flushed, upto)
    {
        while (flushed < upto)
        {
            const uint32_t stop = upto - flushed < ZX_FLUSH ? upto : flushed + ZX_FLUSH;
            const uint32_t lim = stop < cap + g ? stop : cap + g;
#pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                const uint32_t P = flushed + 16u * (uint32_t)(u * 64 + lane);
                if (P >= stop)
                    continue;
                const uint8_t* r = s_ring + (P & (ZX_RING - 1u));
                if (P >= lo && P + 16u <= lim)
                    *reinterpret_cast<uint4*>(out_al + P) = *reinterpret_cast<const uint4*>(r);
                else
                    for (uint32_t k = 0; k < 16u; ++k)
                        if (P + k >= lo && P + k < lim)
                            out_al[P + k] = r[k];
            }
            flushed = stop;
        }
    }
    __device__ __forceinline__ void maybe_flush()
    {
        if (op + g - flushed >= ZX_FLUSH)
            flush((op + g) & ~(ZX_FLUSH - 1u));
    }
    // literals [p, p + k) resident in s_lit (k <= 1100); returns the index of p
    __device__ __forceinline__ uint32_t need_lit(uint32_t p, uint32_t k)
    {
        const int32_t a = (int32_t)(p + lh);
        if (a < lwa || a + (int32_t)k > lwa + (int32_t)ZX_LIT)
        {
            lwa = a & ~15;
            __builtin_amdgcn_wave_barrier();
            uint4 q[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                q[u] = *reinterpret_cast<const uint4*>(lit_al + lwa + 16 * (u * 64 + lane)); // the scratch slot is padded: always readable
#pragma unroll
            for (int u = 0; u < 2; ++u)
                reinterpret_cast<uint4*>(s_lit)[u * 64 + lane] = q[u];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return (uint32_t)(a - lwa);
    }
    __device__ __forceinline__ void copy_lits(uint32_t p, uint32_t len) // the whole wave, appended at op
    {
        while (len)
        {
            const uint32_t i = need_lit(p, 1);
            uint32_t c = ZX_LIT - i;
            c = c < ZX_FLUSH ? c : ZX_FLUSH;
            c = c < len ? c : len;
            for (uint32_t j = lane; j < c; j += 64)
                s_ring[ring(op + j)] = s_lit[i + j];
            p += c;
            op += c;
            len -= c;
            maybe_flush();
        }
    }
    __device__ __forceinline__ void copy_match(uint32_t off, uint32_t ml) // the whole wave, appended at op; 1 <= off <= op
    {
        while (ml)
        {
            const uint32_t seg = ml < ZX_FLUSH ? ml : ZX_FLUSH;
            if (off <= ZX_RING)
            {
                const uint32_t base = op - off;
                if (off >= 64u)
                    for (uint32_t j = lane; j < seg; j += 64)
                        s_ring[ring(op + j)] = s_ring[ring(base + j)];
                else
                    for (uint32_t j0 = 0; j0 < seg; j0 += 64) // 64 bytes at a time read only what is final: byte j = seed byte j mod off
                    {
                        const uint32_t j = j0 + (uint32_t)lane;
                        if (j < seg)
                            s_ring[ring(op + j)] = s_ring[ring(base + j % off)];
                    }
            }
            else
            {
                if (op - off + seg + g > drained) // the source was flushed: have those stores landed?
                {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_s_waitcnt(0);
                    drained = flushed;
                }
                for (uint32_t j = lane; j < seg; j += 64) // off > 8192 >= seg: no overlap
                    s_ring[ring(op + j)] = out_al[op - off + j + g];
            }
            op += seg;
            ml -= seg;
            maybe_flush();
        }
    }
};

__device__ __forceinline__ uint32_t zx_u(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// inclusive prefix sum over the 64 lanes with DPP moves (row shifts inside the rows of 16, then the two row broadcasts): twelve
// VALU instructions, nothing through the LDS crossbar (__shfl_up is a ds_bpermute per step, each a round trip on the chain)
__device__ __forceinline__ uint32_t zx_scan_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return v;
}

// The sequences' bit-stream, read backwards, on the scalar unit.  Per sequence ONE aligned 16-byte load that ends with the dword holding
// the next bit (a sequence takes at most 89 bits: offset 31, lengths 16 + 16, states 9 + 9 + 8), then two 64-bit accumulators with
// the next bits on top -- the three extra-bit fields (<= 63 bits) come out of the first, the three state fields out of the second --
// so that a field costs two shifts.  Values only, no indexing: everything stays in SGPRs.
struct ZxBits
{
    const uint32_t* arena; // dword view of the source
    uint64_t base;         // arena bit of the stream's bit 0
    uint64_t lo, hi;       // window bits 0..63 / 64..127
    uint64_t wbit0;        // arena bit of window bit 0
    uint64_t acc;          // the next bits to read, first one on top
    uint32_t pos;          // bits of the stream not consumed yet
    uint32_t rel;          // window bit one past the next bit to read (= bits of the window still unread)
    __device__ __forceinline__ void load_window()
    {
        const uint64_t top = base + (uint64_t)pos; // arena bit one past the next bit to read
        uint64_t d = (top + 31ull) >> 5;           // dword one past the one holding bit top - 1
        d = d >= 4ull ? d - 4ull : 0ull;
        // a SCALAR load (the compiler picks a vector load here -- it cannot prove the payload read-only -- whose latency is on the
        // critical path of every sequence: three times that of the scalar cache, which these sequential reads hit 19 times in 20)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 w;
        const uint32_t* at = arena + d;
        asm volatile("s_nop 4\n\ts_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(at) : "memory");
        lo = ((uint64_t)w.y << 32) | w.x;
        hi = ((uint64_t)w.w << 32) | w.z;
        wbit0 = d * 32ull;
        rel = (uint32_t)(top - wbit0); // 97 .. 128 (less only at the very start of the arena)
    }
    // a sequence takes at most 89 bits: the window of the one before usually still holds them
    __device__ __forceinline__ void ensure_window()
    {
        if (rel < 89u)
            load_window();
    }
    // acc = the 64 bits below pos (zeros below the window's bit 0, which only happens where the stream has no bits either)
    __device__ __forceinline__ void normalize()
    {
        const uint32_t s = 128u - rel; // shift the 128-bit window left by s, keep the upper half
        const uint64_t up = (hi << (s & 63u)) | ((s & 63u) ? lo >> (64u - (s & 63u)) : 0ull);
        acc = s < 64u ? up : lo << (s & 63u);
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) // n <= 32 bits off the top of acc (at most 64 between two normalize())
    {
        const uint32_t v = (uint32_t)((acc >> 1) >> (63u - n));
        acc <<= n;
        pos -= n;
        rel -= n;
        return v;
    }
};

// One batch of up to 64 sequences, one per lane (r_ll, r_ml, r_off = offset + 3 in lanes 0 .. cnt-1): the serial decoder's checks for
// all of them at once, then their execution through the ring.  litpos / produced: literals consumed / bytes produced so far in the
// literal buffer / the output the ring belongs to (updated); nlit_total / out_limit: how many there are / may be.
__device__ __forceinline__ void zx_batch(ZxOut& zx, uint8_t* s_ring, uint8_t* s_lit, const int lane, const uint32_t cnt, const uint32_t r_ll,
                                         const uint32_t r_ml, const uint32_t r_off, uint32_t& litpos, uint32_t& produced, const uint32_t nlit_total,
                                         const uint32_t out_limit, bool& bad, const uint32_t frame_pos = 0u)
{
    // ---- vector unit: the serial decoder's checks for all of them at once, then execution ----
    const uint32_t ll = (uint32_t)lane < cnt ? r_ll : 0u, ml = (uint32_t)lane < cnt ? r_ml : 0u, off = r_off - 3u;
    uint32_t batch_ll, batch_adv;
    uint32_t i_l = zx_scan_incl(ll), i_a = zx_scan_incl(ll + ml); // inclusive prefix sums: literals / output up to and including my sequence
    {
        batch_ll = (uint32_t)__builtin_amdgcn_readlane((int)i_l, 63);
        batch_adv = (uint32_t)__builtin_amdgcn_readlane((int)i_a, 63);
        // a piece may not use repeat offsets (ov <= 3); literals and output must fit; an offset may not reach below the piece
        const bool wrong = (uint32_t)lane < cnt && (r_off <= 3u || ll > 131072u || ml > 131075u || litpos + i_l > nlit_total ||
                                                     produced + i_a > out_limit || off > frame_pos + produced + i_a - ml);
        if (__builtin_amdgcn_ballot_w64(wrong))
        {
            bad = true;
            return;
        }
    }
    // ---- execution: as many sequences as fit the ring's margins in ONE pass (usually the whole batch): the prefix sums of the
    // checks place everything; ALL literals first (they depend on nothing), then the matches in dependency rounds -- short ones
    // by their own lanes, a long one by the whole wave when its turn comes.  A sequence too big for a pass (a raw unit's 4 KiB
    // of literals) goes through the whole wave alone. ----
    uint32_t start = 0, base_l = 0, base_a = 0; // literals / output of the batch's sequences before `start`
    while (start < cnt)
    {
        const bool fit = (uint32_t)lane >= start && (uint32_t)lane < cnt && i_a - base_a <= ZX_BATCH_ADV && i_l - base_l <= ZX_BATCH_LL;
        const uint64_t fm = __builtin_amdgcn_ballot_w64(fit) >> start; // (the sums grow: the bits are a run from bit 0)
        const uint32_t k = fm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fm);
        if (k == 0u)
        {
            const uint32_t gl = zx_u((uint32_t)__builtin_amdgcn_readlane(r_ll, (int)start)), gm = zx_u((uint32_t)__builtin_amdgcn_readlane(r_ml, (int)start)),
                           go = zx_u((uint32_t)__builtin_amdgcn_readlane(r_off, (int)start)) - 3u;
            zx.copy_lits(litpos + base_l, gl);
            zx.copy_match(go, gm);
            base_l += gl;
            base_a += gl + gm;
            ++start;
            continue;
        }
        const bool in = (uint32_t)lane >= start && (uint32_t)lane < start + k;
        const uint32_t t_ll = zx_u((uint32_t)__builtin_amdgcn_readlane((int)i_l, (int)(start + k - 1u))) - base_l,
                       t_adv = zx_u((uint32_t)__builtin_amdgcn_readlane((int)i_a, (int)(start + k - 1u))) - base_a;
        const uint32_t o_l = zx.op + (i_a - ll - ml - base_a); // where my literals go
        const uint32_t o_m = o_l + ll;                          // where my match goes
        const uint32_t li = zx.need_lit(litpos + base_l, t_ll + 1u) + (i_l - ll - base_l);
        typedef uint32_t u32_a1 __attribute__((aligned(1)));
        if (in && ll <= ZX_LL_OWN)
        {
            uint32_t b = 0;
            {
                const uint32_t rl0 = zx.ring(o_l);
                if (rl0 + ll <= ZX_RING) // (no wrap: plain pointers)
                {
                    uint8_t* dp = s_ring + rl0;
                    const uint8_t* sp = s_lit + li;
                    for (; b + 16u <= ll; b += 16u)
                    {
                        const uint32_t v0 = *reinterpret_cast<const u32_a1*>(sp + b), v1 = *reinterpret_cast<const u32_a1*>(sp + b + 4u),
                                       v2 = *reinterpret_cast<const u32_a1*>(sp + b + 8u), v3 = *reinterpret_cast<const u32_a1*>(sp + b + 12u);
                        *reinterpret_cast<u32_a1*>(dp + b) = v0;
                        *reinterpret_cast<u32_a1*>(dp + b + 4u) = v1;
                        *reinterpret_cast<u32_a1*>(dp + b + 8u) = v2;
                        *reinterpret_cast<u32_a1*>(dp + b + 12u) = v3;
                    }
                    for (; b + 4u <= ll; b += 4u)
                        *reinterpret_cast<u32_a1*>(dp + b) = *reinterpret_cast<const u32_a1*>(sp + b);
                    for (; b < ll; ++b)
                        dp[b] = sp[b];
                }
            }
            for (; b + 4u <= ll; b += 4u) // four bytes per trip of this lane-divergent loop (unaligned LDS dwords)
            {
                const uint32_t r = zx.ring(o_l + b);
                const uint32_t v = *reinterpret_cast<const u32_a1*>(s_lit + li + b);
                if (r <= ZX_RING - 4u)
                    *reinterpret_cast<u32_a1*>(s_ring + r) = v;
                else
                    for (uint32_t j = 0; j < 4u; ++j)
                        s_ring[zx.ring(o_l + b + j)] = (uint8_t)(v >> (8u * j));
            }
            for (; b < ll; ++b)
                s_ring[zx.ring(o_l + b)] = s_lit[li + b];
        }
        for (uint64_t big = __builtin_amdgcn_ballot_w64(in && ll > ZX_LL_OWN); big; big &= big - 1ull)
        {
            const int u = __builtin_ctzll(big);
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)ll, u), from = (uint32_t)__builtin_amdgcn_readlane((int)li, u),
                           to = (uint32_t)__builtin_amdgcn_readlane((int)o_l, u);
            for (uint32_t j = 4u * (uint32_t)lane; j < n; j += 256u)
            {
                const uint32_t r = zx.ring(to + j);
                if (j + 4u <= n && r <= ZX_RING - 4u)
                    *reinterpret_cast<u32_a1*>(s_ring + r) = *reinterpret_cast<const u32_a1*>(s_lit + from + j);
                else
                    for (uint32_t q = 0; q < 4u && j + q < n; ++q)
                        s_ring[zx.ring(to + j + q)] = s_lit[from + j + q];
            }
        }
        const uint32_t end = zx.op + t_adv;
        const bool own = in && ml <= ZX_ML_LANE && !(ml > 20u && off > ZX_RING_SAFE); // my lane copies my match
        const uint64_t ownm = __builtin_amdgcn_ballot_w64(own);
        // a source the ring loses while this pass appends (it holds the 8 KiB below `end`) was flushed long ago: from global memory
        const bool glob = in && ml != 0u && o_m - off + ZX_RING < end;
        if (__builtin_amdgcn_ballot_w64(glob && o_m - off + ml + zx.g > zx.drained))
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            zx.drained = zx.flushed;
        }
        uint64_t pend = __builtin_amdgcn_ballot_w64(in && ml != 0u);
        const uint64_t globm = __builtin_amdgcn_ballot_w64(glob && own);
        if (globm)
        {
            if (glob && own) // at most 20 bytes, final data: five dwords
            {
                uint32_t w[5];
#pragma unroll
                for (uint32_t b = 0; b < 5u; ++b)
                    w[b] = 4u * b < ml ? *reinterpret_cast<const u32_a1*>(zx.out_al + (o_m - off + 4u * b + zx.g)) : 0u;
#pragma unroll
                for (uint32_t b = 0; b < 20u; ++b)
                    if (b < ml)
                        s_ring[zx.ring(o_m + b)] = (uint8_t)(w[b >> 2] >> (8u * (b & 3u)));
            }
            pend &= ~globm;
        }
        while (pend)
        {
            const int first = __builtin_ctzll(pend);
            if (!((ownm >> first) & 1ull))
            {
                // a long match (or a far source with more than 20 bytes): the whole wave
                const uint32_t gm = (uint32_t)__builtin_amdgcn_readlane((int)ml, first), go = (uint32_t)__builtin_amdgcn_readlane((int)off, first),
                               gd = (uint32_t)__builtin_amdgcn_readlane((int)o_m, first);
                if (go >= 64u)
                    for (uint32_t j = lane; j < gm; j += 64)
                    {
                        // a source byte the ring has lost by the end of this pass was flushed long ago (it lies more than 3 KiB
                        // below `op`); everything younger -- unflushed bytes, this very match's own output -- is in the ring
                        const uint32_t sp = gd - go + j;
                        s_ring[zx.ring(gd + j)] = sp + ZX_RING >= end ? s_ring[zx.ring(sp)] : zx.out_al[sp + zx.g];
                    }
                else
                    for (uint32_t j0 = 0; j0 < gm; j0 += 64) // byte j = seed byte j mod off
                    {
                        const uint32_t j = j0 + (uint32_t)lane;
                        if (j < gm)
                            s_ring[zx.ring(gd + j)] = s_ring[zx.ring(gd - go + j % go)];
                    }
                pend &= ~(1ull << first);
                continue;
            }
            const int32_t rel_m = (int32_t)(o_m - zx.op);
            const int32_t frontier = (int32_t)__builtin_amdgcn_readlane((uint32_t)rel_m, first);
            const bool ready = ((pend >> lane) & 1ull) && own && (lane == first || rel_m - (int32_t)off + (int32_t)ml <= frontier);
            if (ready)
            {
                const uint32_t so2 = o_m - off;
                uint32_t b = 0;
                {
                    // neither range wraps around the ring's end (all but one copy in a hundred): plain pointers, no index arithmetic
                    // and no wrap tests per dword
                    const uint32_t ra0 = zx.ring(so2), rb0 = zx.ring(o_m);
                    if (off >= 4u && ra0 + ml <= ZX_RING && rb0 + ml <= ZX_RING)
                    {
                        const uint8_t* sp = s_ring + ra0;
                        uint8_t* dp = s_ring + rb0;
                        if (off >= 16u) // sixteen bytes read, then written: one wait per sixteen instead of one per four
                            for (; b + 16u <= ml; b += 16u)
                            {
                                const uint32_t v0 = *reinterpret_cast<const u32_a1*>(sp + b), v1 = *reinterpret_cast<const u32_a1*>(sp + b + 4u),
                                               v2 = *reinterpret_cast<const u32_a1*>(sp + b + 8u), v3 = *reinterpret_cast<const u32_a1*>(sp + b + 12u);
                                *reinterpret_cast<u32_a1*>(dp + b) = v0;
                                *reinterpret_cast<u32_a1*>(dp + b + 4u) = v1;
                                *reinterpret_cast<u32_a1*>(dp + b + 8u) = v2;
                                *reinterpret_cast<u32_a1*>(dp + b + 12u) = v3;
                            }
                        for (; b + 4u <= ml; b += 4u)
                            *reinterpret_cast<u32_a1*>(dp + b) = *reinterpret_cast<const u32_a1*>(sp + b);
                        for (; b < ml; ++b)
                            dp[b] = sp[b];
                    }
                }
                if (off >= 4u)
                    for (; b + 4u <= ml; b += 4u)
                    {
                        const uint32_t ra = zx.ring(so2 + b), rb = zx.ring(o_m + b);
                        if (ra <= ZX_RING - 4u && rb <= ZX_RING - 4u)
                            *reinterpret_cast<u32_a1*>(s_ring + rb) = *reinterpret_cast<const u32_a1*>(s_ring + ra);
                        else
                            for (uint32_t j = 0; j < 4u; ++j)
                                s_ring[zx.ring(o_m + b + j)] = s_ring[zx.ring(so2 + b + j)];
                    }
                for (; b < ml; ++b)
                    s_ring[zx.ring(o_m + b)] = s_ring[zx.ring(so2 + b)];
            }
            pend &= ~__builtin_amdgcn_ballot_w64(ready);
        }
        zx.op = end;
        zx.maybe_flush();
        base_l += t_ll;
        base_a += t_adv;
        start += k;
    }
    litpos += batch_ll;
    produced += batch_adv;
}

// The sequence bit-stream of ONE block of another encoder's frame on the SCALAR unit (a wave per block): what k_zstd_blk_sequences does
// with a lane per block -- same tables (the block's packed tables in the table arena), same records, same verdicts -- for calls with so
// few blocks that a block's chain of sequences IS the call's time: ONE reference-made 8 MiB frame has 64 blocks of up to 16 000
// sequences each; a lane walks them at 0.42 us a sequence (~150 dependent vector instructions), the scalar unit at about half of that
// (the state machine of k_zstd_execute<false>: entries by s_load from the scalar cache, two 64-bit accumulators, every sequence dropped
// into its lane with v_writelane, the records leave 64 at a time).  Worth it while a CU's scalar pipe serves one or two such waves.
__global__ __launch_bounds__(64) void k_zstd_blk_seq_scalar(const uint8_t* __restrict__ src, const ZItem* __restrict__ fitems, const uint32_t* __restrict__ slist,
                                                           const uint32_t* __restrict__ scount, const uint64_t* __restrict__ tabs,
                                                           uint64_t* __restrict__ rec_scratch, ZPrep* __restrict__ fprep, uint32_t* __restrict__ retry)
{
    const uint32_t g = blockIdx.x;
    if (g >= *scount)
        return;
    const uint32_t i = slist[g];
    const ZItem it = fitems[i];
    if (it.kind != 4u)
        return;
    const ZPrep pr = fprep[i];
    if (!(pr.status == ZP_READY && pr.log[0] == 0u && pr.nbseq != 0u))
        return;
    const int lane = threadIdx.x;
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tabs + (uint64_t)g * 1280u); // LL at 0, ML at 512, OF at 1024 entries of 8 bytes
    const uint64_t at = it.src_off + pr.seq_off;
    const uint32_t ssize = zx_u(pr.seq_size), nbseq = zx_u(pr.nbseq);
    const uint32_t log_l = zx_u(pr.log[1] & 255u), log_o = zx_u((pr.log[1] >> 8) & 255u), log_m = zx_u((pr.log[1] >> 16) & 255u);
    uint64_t* recs = rec_scratch + pr.rec_at;
    bool bad = false;
    uint32_t sum_ll = 0, sum_ml = 0;
    ZxBits br;
    br.arena = reinterpret_cast<const uint32_t*>(src - ((uintptr_t)src & 3u));
    br.base = at * 8ull + 8ull * ((uintptr_t)src & 3u);
    br.pos = 0;
    br.lo = br.hi = br.acc = 0;
    br.wbit0 = 0;
    br.rel = 0;
    if (ssize == 0u)
        bad = true;
    else
    {
        const uint32_t last = zx_u((uint32_t)src[at + ssize - 1u]);
        if (last == 0u)
            bad = true;
        else
            br.pos = (ssize - 1u) * 8u + (31u - (uint32_t)__builtin_clz(last));
    }
    uint32_t sl = 0, so = 0, sm = 0; // byte offsets into tb
    if (!bad)
    {
        if (log_l + log_o + log_m > br.pos)
            bad = true;
        else
        {
            br.load_window();
            br.normalize();
            sl = br.take(log_l) * 8u;
            so = (1024u + br.take(log_o)) * 8u;
            sm = (512u + br.take(log_m)) * 8u;
        }
    }
    for (uint32_t s0 = 0; s0 < nbseq && !bad; s0 += 64u)
    {
        const uint32_t cnt = nbseq - s0 < 64u ? nbseq - s0 : 64u;
        uint32_t r_ll = 0, r_ml = 0, r_off = 0;
        for (uint32_t k = 0; k < cnt; ++k)
        {
            const uint2 ql = *reinterpret_cast<const uint2*>(tb + sl), qo = *reinterpret_cast<const uint2*>(tb + so),
                        qm = *reinterpret_cast<const uint2*>(tb + sm);
            const uint32_t l0 = zx_u(ql.x), o0 = zx_u(qo.x), m0 = zx_u(qm.x);
            const uint32_t ob = o0 >> 24, mb = m0 >> 24, lb = l0 >> 24;
            const uint32_t nbl = (l0 >> 16) & 255u, nbm = (m0 >> 16) & 255u, nbo = (o0 >> 16) & 255u;
            const bool more = s0 + k + 1u < nbseq;
            const uint32_t n1 = ob + mb + lb, n2 = more ? nbl + nbm + nbo : 0u; // <= 63, <= 26
            if (n1 + n2 > br.pos)
            {
                bad = true; // the stream runs out
                break;
            }
            br.ensure_window();
            br.normalize();
            const uint32_t ov = zx_u(qo.y) + br.take(ob);
            const uint32_t t2 = br.take(mb + lb); // match-length and literal-length extra bits are adjacent
            const uint32_t ml = zx_u(qm.y) + (t2 >> lb);
            const uint32_t ll = zx_u(ql.y) + (t2 & ((1u << lb) - 1u));
            if (more)
            {
                if (n1 + n2 > 64u)
                    br.normalize();
                const uint32_t t3 = br.take(n2); // LL, ML, OF from the top
                sl = ((l0 & 0xFFFFu) + (t3 >> (nbm + nbo))) * 8u;
                sm = (512u + (m0 & 0xFFFFu) + ((t3 >> nbo) & ((1u << nbm) - 1u))) * 8u;
                so = (1024u + (o0 & 0xFFFFu) + (t3 & ((1u << nbo) - 1u))) * 8u;
            }
            sum_ll += ll;
            sum_ml += ml;
            {
                const uint32_t a = zx_u(ll), b = zx_u(ml), c = zx_u(ov), kk = zx_u(k);
                uint32_t keep;
                asm volatile("s_mov_b32 %3, m0\n\ts_mov_b32 m0, %7\n\ts_nop 4\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\t"
                             "v_writelane_b32 %2, %6, m0\n\ts_mov_b32 m0, %3"
                             : "+v"(r_ll), "+v"(r_ml), "+v"(r_off), "=&s"(keep)
                             : "s"(a), "s"(b), "s"(c), "s"(kk));
            }
        }
        if (!bad)
        {
            const bool mine = (uint32_t)lane < cnt;
            if (__builtin_amdgcn_ballot_w64(mine && r_off >= (1u << 24)))
                bad = true; // (as zs_seq_lanes: an offset value the record cannot hold)
            if (mine)
                recs[s0 + (uint32_t)lane] = (uint64_t)r_ll | ((uint64_t)r_ml << 20) | ((uint64_t)r_off << 40);
        }
    }
    if (!bad && br.pos != 0u)
        bad = true; // the bit-stream must be consumed exactly
    if (!bad && (sum_ll > pr.nlit || pr.nlit + sum_ml > ZB))
        bad = true;
    if (lane == 0)
    {
        if (bad)
        {
            fprep[i].status = ZP_SERIAL;
            retry[it.payload] = __LINE__;
        }
        else
            fprep[i].expect = pr.nlit + sum_ml; // what the block regenerates
    }
}

// RECS: the sequences come as records {literal length:20 | match length:20 | offset value:24} from k_zstd_sub_entropy (`tables` is
// then the record array, ZREC_MAX per slot) instead of from the bit-stream; everything after that is the same.
template <bool RECS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_zstd_execute(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, const ZItem* __restrict__ items,
                                                     const uint32_t* __restrict__ item_count, uint32_t item0, uint32_t item1,
                                                     uint8_t* __restrict__ dst, const uint8_t* __restrict__ lit_scratch,
                                                     const uint64_t* __restrict__ tables, const ZPrep* __restrict__ prep,
                                                     uint32_t* __restrict__ status_out, uint32_t* __restrict__ retry, uint32_t px,
                                                     const uint32_t* __restrict__ perm, uint32_t* __restrict__ done)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[ZX_RING];
    __shared__ __attribute__((aligned(16))) uint8_t s_lit[ZX_LIT];
    __shared__ uint32_t s_ia[64], s_om[64]; // (px: the batch's prefix sums, zo_batch_bytes)
    if (item0 + blockIdx.x >= item1 || item0 + blockIdx.x >= *item_count)
        return;
    const uint32_t i = perm ? perm[item0 + blockIdx.x] : item0 + blockIdx.x; // (link-major order: k_zstd_rows)
    const ZItem it = items[i];
    if (it.kind != (RECS ? 3u : 2u))
        return;
    const ZPrep pr = prep[i];
    if (pr.status != ZP_READY)
        return;
    const int lane = threadIdx.x;
    const ZBlock blk = blocks[it.payload];
    const uint32_t slot = blockIdx.x;
    // A piece of a frame whose pieces depend on each other (trailer version 4: it.pad bit 1) may copy from the pieces before it: it is
    // executed with positions relative to the start of the piece before it (through memory that is an address like any other; the ring
    // starts out holding that piece's last 8 KiB), and not before that piece is complete --
    // the items of a payload are consecutive, the piece before is item i - 1, and in the link-major order its workgroup was dispatched
    // before this one (an earlier row), or belongs to an earlier launch.  Pieces the executor does not run (Raw / RLE pieces of
    // k_zstd_plain_pieces: an earlier kernel of the round; pieces given back by k_zstd_sub_entropy: the payload goes to the serial decoder
    // anyway) are not waited for.
    const bool in_chain = RECS && done != nullptr && (it.pad & 2u) != 0u;  // (tells the piece behind it when it is done)
    const uint32_t link = in_chain ? (it.out0 / ZB) % ZCHAIN : 0u;          // my place in the chain: 0 = its head, which waits for nobody
    const bool chained = link != 0u;
    // position of the piece's first byte in what zo_batch_bytes addresses: the piece BEFORE it and itself.  (The encoder's matches reach
    // less than 64 KiB back; a frame that claims more -- forged, damaged -- goes to the serial decoder: bytes further back would need the
    // flag of a piece this one does not wait for when the piece in between is a Raw / RLE piece.)
    const uint32_t frame_pos = chained ? ZB : 0u;
    bool pred_failed = false;
    if (chained)
    {
        const ZItem before = items[i - 1u];
        if (before.kind == 3u && before.payload == it.payload)
        {
            const uint32_t st = prep[i - 1u].status;
            if (st == ZP_READY)
            {
                // (relaxed polls: an acquire per poll would invalidate the CU's vector cache under the waves that are at work; one
                // acquire fence once the flag is up)
                // The wait is BOUNDED.  Forward progress here rests on the piece before having been dispatched already (lower
                // blockIdx in the link-major order, or an earlier launch) -- true of the command processor's in-order dispatch on
                // this hardware, but not a promise of HIP: a different partition mode, queue preemption or a debugger may hold the
                // predecessor's workgroup back while waiters occupy the slots it needs.  ZX_WAIT_TICKS of the 100 MHz wall clock
                // (1 s: a piece executes in well under a millisecond, a chain of ZCHAIN in a few) and the piece gives up: it reports
                // failure like any other (status ZP_SERIAL, retry[payload], done = 2 for the piece behind it), the payload goes to
                // the serial decoder, and the slot is free again -- a slow restore instead of a hung GPU.
                constexpr unsigned long long ZX_WAIT_TICKS = 100000000ull;
                uint32_t f;
                const unsigned long long t_wait = wall_clock64();
                bool gave_up = false;
                while ((f = __hip_atomic_load(&done[i - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u)
                {
                    __builtin_amdgcn_s_sleep(32);
                    if (wall_clock64() - t_wait > ZX_WAIT_TICKS)
                    {
                        gave_up = true;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                pred_failed = gave_up || f != 1u;
            }
            else if (st != ZP_DONE)
                pred_failed = true;
        }
    }
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tables + (uint64_t)slot * 3u * ZT_ENTRIES); // states are byte offsets into this
    const uint64_t* recs = tables + (uint64_t)slot * ZREC_MAX;
    const uint8_t* lits = lit_scratch + (uint64_t)slot * (ZD_LIT_MAX + 64u);
    uint8_t* out = dst + blk.dst_off + it.out0;

    ZxOut zx;
    zx.s_ring = s_ring;
    zx.s_lit = s_lit;
    zx.g = (uint32_t)((uintptr_t)out & 15u);
    zx.out_al = out - frame_pos - zx.g; // (ZB is a multiple of the ring: positions keep their place in it)
    zx.lh = (uint32_t)((uintptr_t)lits & 15u);
    zx.lit_al = lits - zx.lh;
    zx.nlit = pr.nlit;
    zx.cap = frame_pos + pr.expect;
    zx.lo = frame_pos + zx.g;
    zx.lane = lane;
    zx.op = frame_pos;
    zx.flushed = zx.drained = frame_pos; // (what lies below was written, and released, by the piece before)
    zx.lwa = -(int32_t)ZX_LIT;

    // ---- the bit-stream, read backwards: bit k of the stream is bit (8 * bits_off + k) of the arena; a sequence takes at most 89 bits
    // (offset 31 + lengths 16 + 16 + states 9 + 9 + 8): four aligned dwords that end with the dword holding the next bit cover it ----
    const uint64_t base_bit = pr.bits_off * 8ull;
    const uint32_t* arena = reinterpret_cast<const uint32_t*>(src - ((uintptr_t)src & 3u)); // dword view; bit b of src = bit b + 8 * skew here
    const uint64_t skew_bits = 8ull * ((uintptr_t)src & 3u);
    bool bad = pred_failed;
    uint32_t pos = 0; // bits of the stream not consumed yet
    if (!RECS)
    {
        const uint32_t last = src[pr.bits_off + pr.bits_size - 1u];
        if (last == 0u)
            bad = true;
        else
            pos = (pr.bits_size - 1u) * 8u + (31u - (uint32_t)__builtin_clz(last));
    }
    pos = zx_u(pos);
    ZxBits br;
    br.arena = arena;
    br.base = base_bit + skew_bits;
    br.pos = pos;
    br.lo = br.hi = br.acc = 0;
    br.wbit0 = 0;
    br.rel = 0;
    uint32_t sl = 0, so = 0, sm = 0;
    if (!bad && !RECS)
    {
        if (pr.log[ZT_LL] + pr.log[ZT_OF] + pr.log[ZT_ML] > br.pos)
            bad = true;
        else
        {
            br.load_window();
            br.normalize();
            sl = (ZT_LL * ZT_ENTRIES + br.take(pr.log[ZT_LL])) * 8u;
            so = (ZT_OF * ZT_ENTRIES + br.take(pr.log[ZT_OF])) * 8u;
            sm = (ZT_ML * ZT_ENTRIES + br.take(pr.log[ZT_ML])) * 8u;
        }
    }
    uint32_t litpos = 0, produced = 0; // what the decoded sequences consume / produce (scalar bookkeeping of the checks)
#ifdef LTHIP_ZB_PROF
    unsigned long long t_prof = wall_clock64();
    (void)t_prof;
#endif
    if (RECS && px == 1u)
    {
        // Which executor?  The ring (zx_batch) copies a match of more than 20 bytes whose source has left its 8 KiB with the whole wave,
        // one such match at a time; bytes through memory (zo_batch_bytes) do not care where a source lies but pay a memory round trip
        // per dependency round.  A piece with many such matches (records: the group's first occurrence of a 24-byte field, up to
        // 64 KiB back) goes through memory -- 502 against 295 GB/s --, the others through the ring (lines 565 against 328, tokens 223
        // against 195): one pass over the records decides.
        uint32_t far_long = 0;
        for (uint32_t s0 = (uint32_t)lane; s0 < pr.nbseq; s0 += 64u)
        {
            const uint64_t r = recs[s0];
            far_long += ((uint32_t)(r >> 40) > ZX_RING_SAFE + 3u && ((uint32_t)(r >> 20) & 0xFFFFFu) > 20u) ? 1u : 0u;
        }
        far_long = zx_scan_incl(far_long);
        far_long = (uint32_t)__builtin_amdgcn_readlane((int)far_long, 63);
        px = far_long * 8u > pr.nbseq ? 2u : 0u;
    }
    if (chained && !pred_failed && !(RECS && px))
    {
        // the ring as the piece before left it: its last 8 KiB (the aligned lines below and around the seam; the bytes of the seam's line
        // that are this piece's are written before they are read)
        const uint32_t a0 = frame_pos + 16u - ZX_RING;
#pragma unroll
        for (uint32_t u = 0; u < ZX_RING / 1024u; ++u)
        {
            const uint32_t A = a0 + 16u * (u * 64u + (uint32_t)lane);
            *reinterpret_cast<uint4*>(s_ring + (A & (ZX_RING - 1u))) = *reinterpret_cast<const uint4*>(zx.out_al + A);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t s0 = 0; s0 < pr.nbseq && !bad; s0 += 64u)
    {
        const uint32_t cnt = pr.nbseq - s0 < 64u ? pr.nbseq - s0 : 64u;
        // ---- scalar unit: the next `cnt` sequences into lanes 0 .. cnt-1 ----
        uint32_t r_ll = 0, r_ml = 0, r_off = 0;
        // one sequence: the three entries, the three extra-bit fields, (MORE) the three state updates, the record into lane k.
        // GUARD: test that the stream still holds the bits (without it the caller has made sure of 89 bits per sequence).
        auto one = [&](uint32_t k, auto more_c, auto guard_c) -> bool {
            constexpr bool MORE = decltype(more_c)::value, GUARD = decltype(guard_c)::value;
            const uint2 ql = *reinterpret_cast<const uint2*>(tb + sl), qo = *reinterpret_cast<const uint2*>(tb + so),
                        qm = *reinterpret_cast<const uint2*>(tb + sm);
            const uint32_t l0 = zx_u(ql.x), o0 = zx_u(qo.x), m0 = zx_u(qm.x);
            const uint32_t ob = o0 >> 24, mb = m0 >> 24, lb = l0 >> 24;
            const uint32_t nbl = (l0 >> 16) & 255u, nbm = (m0 >> 16) & 255u, nbo = (o0 >> 16) & 255u;
            if (GUARD && ob + mb + lb + (MORE ? nbl + nbm + nbo : 0u) > br.pos)
                return false; // the stream runs out: the serial decoder says how
            br.ensure_window();
            br.normalize();
            const uint32_t ov = zx_u(qo.y) + br.take(ob);
            const uint32_t t2 = br.take(mb + lb); // match-length and literal-length extra bits are adjacent (16 + 16 at most)
            const uint32_t ml = zx_u(qm.y) + (t2 >> lb);
            const uint32_t ll = zx_u(ql.y) + (t2 & ((1u << lb) - 1u));
            if (MORE)
            {
                // the three state fields are adjacent (LL, ML, OF from the top; at most 26 bits): one take, split with 32-bit shifts;
                // the accumulator still holds them unless this sequence has taken more than 64 bits in all (rare)
                if (ob + mb + lb + nbl + nbm + nbo > 64u)
                    br.normalize();
                const uint32_t t3 = br.take(nbl + nbm + nbo);
                sl = (l0 & 0xFFFFu) + ((t3 >> (nbm + nbo)) << 3);
                sm = (m0 & 0xFFFFu) + (((t3 >> nbo) & ((1u << nbm) - 1u)) << 3);
                so = (o0 & 0xFFFFu) + ((t3 & ((1u << nbo) - 1u)) << 3);
            }
            // into lane k of the three record registers (gfx9 allows one SGPR per VOP3: the lane select goes through M0);
            // what the serial decoder checks per sequence is checked for the whole batch by the vector unit below
            const uint32_t a = zx_u(ll), b = zx_u(ml), c = zx_u(ov), kk = zx_u(k);
            uint32_t keep;
            asm volatile("s_mov_b32 %3, m0\n\ts_mov_b32 m0, %7\n\ts_nop 4\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\t"
                         "v_writelane_b32 %2, %6, m0\n\ts_mov_b32 m0, %3"
                         : "+v"(r_ll), "+v"(r_ml), "+v"(r_off), "=&s"(keep)
                         : "s"(a), "s"(b), "s"(c), "s"(kk));
            return true;
        };
        if (RECS)
        {
            const uint64_t r = (uint32_t)lane < cnt ? recs[s0 + (uint32_t)lane] : 0ull;
            r_ll = (uint32_t)r & 0xFFFFFu;
            r_ml = (uint32_t)(r >> 20) & 0xFFFFFu;
            r_off = (uint32_t)(r >> 40);
        }
        else
        {
            const bool last_batch = s0 + cnt == pr.nbseq;
            const uint32_t with_more = last_batch ? cnt - 1u : cnt; // the block's very last sequence updates no state
            bool ok = true;
            if (br.pos >= 64u * 89u)
                for (uint32_t k = 0; k < with_more; ++k)
                    (void)one(k, std::true_type{}, std::false_type{});
            else
                for (uint32_t k = 0; k < with_more && ok; ++k)
                    ok = one(k, std::true_type{}, std::true_type{});
            if (ok && last_batch)
                ok = one(cnt - 1u, std::false_type{}, std::true_type{});
            if (!ok)
                bad = true;
        }
        if (bad)
            break;
        if (!RECS && s0 + cnt == pr.nbseq && br.pos != 0u)
        {
            bad = true; // the bit-stream must be consumed exactly
            break;
        }
        if (RECS && px)
        {
            // executed on bytes through memory (origin_exec.h) instead of through the LDS ring: the serial decoder's checks for the 64
            // sequences at once, then literals and matches in dependency rounds straight into the piece's output
            const bool act = (uint32_t)lane < cnt;
            const uint32_t ll = act ? r_ll : 0u, ml = act ? r_ml : 0u, off = r_off - 3u;
            const uint32_t i_l = zx_scan_incl(ll), i_a = zx_scan_incl(ll + ml);
            const uint32_t batch_ll = (uint32_t)__builtin_amdgcn_readlane((int)i_l, 63), batch_adv = (uint32_t)__builtin_amdgcn_readlane((int)i_a, 63);
            const bool wrong = act && (r_off <= 3u || ll > 131072u || ml > 131075u || litpos + i_l > pr.nlit || produced + i_a > pr.expect ||
                                       off > frame_pos + produced + i_a - ml);
            if (__builtin_amdgcn_ballot_w64(wrong) ||
                !zo_batch_bytes(out - frame_pos, lits, lane, act, ll, litpos + (i_l - ll), ml, off, i_a, frame_pos + produced, s_ia, s_om))
            {
                bad = true;
                break;
            }
            litpos += batch_ll;
            produced += batch_adv;
        }
        else
            zx_batch(zx, s_ring, s_lit, lane, cnt, r_ll, r_ml, r_off, litpos, produced, pr.nlit, pr.expect, bad, frame_pos);
        if (bad)
            break;
    }
    if (!bad)
    {
        // literals after the last sequence; the piece must come out at exactly its size
        const uint32_t rest = pr.nlit - litpos;
        if (rest != pr.expect - produced)
            bad = true;
        else if (RECS && px)
        {
            typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
            uint32_t j = 16u * (uint32_t)lane;
            for (; j + 16u <= rest; j += 1024u)
                *reinterpret_cast<u32x4_a1*>(out + produced + j) = *reinterpret_cast<const u32x4_a1*>(lits + litpos + j);
            if (j < rest)
                for (uint32_t k = j; k < rest && k < j + 16u; ++k)
                    out[produced + k] = lits[litpos + k];
        }
        else
        {
            zx.copy_lits(litpos, rest);
            zx.flush(zx.op + zx.g);
        }
    }
    if (bad && lane == 0)
    {
        status_out[(size_t)i * (sizeof(ZPrep) / 4u)] = ZP_SERIAL; // = prep[i].status (a second view: `prep` itself is read-only here)
        if (RECS)
            retry[it.payload] = 1u; // a run of sub-blocks has no serial piece decoder: the whole payload, serially
    }
    if (in_chain)
    {
        // the piece's bytes are in memory before the next piece of the chain is told so (1), or that it need not bother (2)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0)
            __hip_atomic_store(&done[i], bad ? 2u : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifdef LTHIP_ABLATIONS
#include "ablations/k_zstd_execute_payload.inc"
#endif


// ---------------------------------------------------------------------------------------------------------------------------------
// Frames of other encoders, EXECUTED block-parallel (round 3).  A frame is one chain of dependent copies -- a block's matches reach
// into the blocks before it, and through them into everything older -- so executing VALUES in parallel would have every block wait
// for the end of the one before.  What CAN run in parallel is the execution of ORIGINS: where does every byte of a block come from?
//   k_zstd_fr_reps     one wave per block: repeat offsets turned into offsets, with the history at the block's start SYMBOLIC
//                      (offset values 1..3 in the records it rewrites then mean "entry 0..2 of the history this block starts with",
//                      minus the two bits above the match length); leaves the block's final history, entries symbolic or not
//   k_zstd_fr_chain    one lane per frame: composes the histories block by block, adds up where every block starts
//   k_zstd_fr_trace    one wave per block, all blocks of all frames at once: the block's sequences executed on 32-bit origins
//                      instead of bytes -- a literal's origin is its index in the block's literal buffer, a match copies the origins
//                      of its source, and a source byte that lies in an EARLIER block is recorded as that frame position.  Chains
//                      inside the block collapse as it goes; what is left per byte is "literal i" or "byte p of an earlier block"
//   k_zstd_fr_gather   launch k fills block k of every frame: out[q] = literal or out[p]; blocks below k are final by then.
// Same checks as the serial decoder (RFC 8878 3.1.1.3-5); whatever fails sends the payload there (retry).
// ---------------------------------------------------------------------------------------------------------------------------------
// (histories: ZO_FLAG marks a symbolic entry)
struct ZFr
{
    uint32_t in[3];  // the repeat offsets the block starts with
    uint32_t start;  // frame position of the block's first byte
    uint32_t out[3]; // ... and ends with: offsets, or ZO_FLAG | slot << 24 | d = entry `slot` of in[] minus d
    uint32_t pad;
};
__device__ __forceinline__ uint32_t zo_minus1(uint32_t v) { return (v & ZO_FLAG) ? v + 1u : (v ? v - 1u : 0u); }
__device__ __forceinline__ uint32_t zo_bind(uint32_t v, const uint32_t in0, const uint32_t in1, const uint32_t in2)
{
    if (!(v & ZO_FLAG))
        return v;
    const uint32_t slot = (v >> 24) & 3u, d = v & 0xFFFFFFu;
    const uint32_t e = slot == 0u ? in0 : slot == 1u ? in1 : in2;
    return e > d ? e - d : 0u; // (0: not an offset; whoever uses it is stopped)
}
__global__ __launch_bounds__(64) void k_zstd_fr_reps(const uint32_t* __restrict__ flist, uint32_t n, uint64_t* __restrict__ rec_scratch,
                                                    const ZPrep* __restrict__ fprep, ZFr* __restrict__ fr, const ZItem* __restrict__ fitems,
                                                    uint32_t* __restrict__ retry)
{
    if (blockIdx.x >= n)
        return;
    const uint32_t fi = flist[blockIdx.x];
    const ZPrep pr = fprep[fi];
    const int lane = threadIdx.x;
    uint32_t rep0 = ZO_FLAG, rep1 = ZO_FLAG | (1u << 24), rep2 = ZO_FLAG | (2u << 24);
    bool bad = false;
    if (pr.status == ZP_READY && pr.log[0] == 0u && pr.nbseq != 0u)
    {
        uint64_t* recs = rec_scratch + pr.rec_at;
        uint64_t r_next = (uint32_t)lane < pr.nbseq ? recs[lane] : 0ull;
        for (uint32_t s0 = 0; s0 < pr.nbseq; s0 += 64u)
        {
            const uint32_t cnt = pr.nbseq - s0 < 64u ? pr.nbseq - s0 : 64u;
            const uint64_t r = r_next;
            r_next = s0 + 64u + (uint32_t)lane < pr.nbseq ? recs[s0 + 64u + (uint32_t)lane] : 0ull;
            const uint32_t r_ll = (uint32_t)r & 0xFFFFFu;
            const uint32_t r_off = (uint32_t)(r >> 40); // Offset_Value: 1..3 = repeat offsets
            const uint64_t repm = __builtin_amdgcn_ballot_w64((uint32_t)lane < cnt && r_off <= 3u);
            if (repm == 0ull)
            {
                // no repeat offset in the batch: the history is simply its last three offsets
                const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)r_off, (int)(cnt - 1u)) - 3u;
                if (cnt >= 3u)
                {
                    rep2 = (uint32_t)__builtin_amdgcn_readlane((int)r_off, (int)(cnt - 3u)) - 3u;
                    rep1 = (uint32_t)__builtin_amdgcn_readlane((int)r_off, (int)(cnt - 2u)) - 3u;
                }
                else if (cnt == 2u)
                {
                    rep2 = rep0;
                    rep1 = (uint32_t)__builtin_amdgcn_readlane((int)r_off, 0) - 3u;
                }
                else
                {
                    rep2 = rep1;
                    rep1 = rep0;
                }
                rep0 = o1;
                continue;
            }
            uint32_t mine = 0; // what my sequence's repeat code stands for
            for (uint32_t q = 0; q < cnt; ++q)
            {
                const uint32_t ov = (uint32_t)__builtin_amdgcn_readlane((int)r_off, (int)q);
                if (ov > 3u)
                {
                    rep2 = rep1;
                    rep1 = rep0;
                    rep0 = ov - 3u;
                    continue;
                }
                const uint32_t idx = ov + ((uint32_t)__builtin_amdgcn_readlane((int)r_ll, (int)q) == 0u ? 1u : 0u); // 1..4 (0: not a value)
                uint32_t o = rep0;
                if (idx != 1u)
                {
                    o = idx == 4u ? zo_minus1(rep0) : idx == 2u ? rep1 : rep2;
                    if (idx >= 3u)
                        rep2 = rep1;
                    rep1 = rep0;
                    rep0 = o;
                }
                if (ov == 0u)
                    o = 0u;
                if ((uint32_t)lane == q)
                    mine = o;
            }
            rep0 = zx_u(rep0);
            rep1 = zx_u(rep1);
            rep2 = zx_u(rep2);
            if ((repm >> lane) & 1ull)
            {
                // an offset: value offset + 3 (0: none -- the trace stops there); symbolic: value 1 + slot, the decrement above the match length
                uint32_t ov = 0, dd = 0;
                if (mine & ZO_FLAG)
                {
                    ov = 1u + ((mine >> 24) & 3u);
                    dd = mine & 0xFFFFFFu;
                    if (dd > 3u)
                        bad = true; // (four "repeat offset 1 minus one" in a row on a history nobody has seen yet: the serial decoder)
                }
                else if (mine != 0u && mine < 0xFFFFFCu)
                    ov = mine + 3u;
                recs[s0 + (uint32_t)lane] = (r & 0x0000003FFFFFFFFFull) | ((uint64_t)(dd & 3u) << 38) | ((uint64_t)ov << 40);
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0)
        retry[fitems[fi].payload] = __LINE__;
    if (lane == 0)
    {
        fr[fi].out[0] = rep0;
        fr[fi].out[1] = rep1;
        fr[fi].out[2] = rep2;
    }
}

__global__ __launch_bounds__(64) void k_zstd_fr_chain(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, uint32_t nblocks,
                                                     const uint32_t* __restrict__ f_nblocks, const ZPrep* __restrict__ fprep, ZFr* __restrict__ fr,
                                                     uint32_t* __restrict__ retry)
{
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= nblocks)
        return;
    const uint32_t nb = f_nblocks[b];
    if (nb == 0u || retry[b])
        return;
    const ZBlock blk = blocks[b];
    const ZFrameHdr fh = z_frame_header(src + blk.src_off, blk.size);
    if (!fh.ok)
    {
        retry[b] = __LINE__;
        return;
    }
    const uint32_t content = (uint32_t)fh.content;
    uint32_t h0 = 1, h1 = 4, h2 = 8, produced = 0; // Repeated_Offsets at the start of a frame
    for (uint32_t k = 0; k < nb; ++k)
    {
        const uint32_t fi = blk.pad + k;
        const ZPrep pr = fprep[fi];
        if (pr.status == ZP_SERIAL || pr.expect > content - produced || pr.expect > ZB)
        {
            retry[b] = __LINE__;
            return;
        }
        ZFr f = fr[fi];
        f.in[0] = h0;
        f.in[1] = h1;
        f.in[2] = h2;
        f.start = produced;
        fr[fi] = f;
        const uint32_t n0 = zo_bind(f.out[0], h0, h1, h2), n1 = zo_bind(f.out[1], h0, h1, h2), n2 = zo_bind(f.out[2], h0, h1, h2);
        h0 = n0;
        h1 = n1;
        h2 = n2;
        produced += pr.expect;
    }
    if (produced != content)
        retry[b] = __LINE__;
}

// (one wave per block slot [slot0, slot0 + gridDim.x); origins of payload b at org + (zb_base(b) - item0) * ZB, one u32 per byte)
__global__ __launch_bounds__(64) void k_zstd_fr_trace(const ZBlock* __restrict__ blocks, const ZItem* __restrict__ fitems, uint32_t slot0,
                                                     const uint32_t* __restrict__ f_nblocks, const uint64_t* __restrict__ rec_scratch,
                                                     const ZPrep* __restrict__ fprep, const ZFr* __restrict__ fr, uint32_t* __restrict__ org_arena,
                                                     uint32_t item0, uint32_t* __restrict__ retry)
{
    __shared__ uint32_t s_ia[64], s_om[64];
    const uint32_t fi = slot0 + blockIdx.x;
    const ZItem it = fitems[fi];
    if (it.kind != 4u)
        return;
    const uint32_t b = it.payload;
    if (f_nblocks[b] == 0u || retry[b])
        return;
    const ZPrep pr = fprep[fi];
    if (pr.log[0] != 0u)
        return; // Raw_Block / RLE_Block: the gather copies / fills
    const int lane = threadIdx.x;
    const ZFr f = fr[fi];
    const uint32_t start = f.start;
    uint32_t* org = org_arena + (uint64_t)(blocks[b].zb_base - item0) * ZB + start; // origin of the block's byte q: org[q]
    const uint64_t* recs = rec_scratch + pr.rec_at;
    uint32_t litpos = 0, produced = 0;
    bool bad = false;
    uint32_t why = 0;
    uint64_t r_next = (uint32_t)lane < pr.nbseq ? recs[lane] : 0ull;
    for (uint32_t s0 = 0; s0 < pr.nbseq && !bad; s0 += 64u)
    {
        const uint32_t cnt = pr.nbseq - s0 < 64u ? pr.nbseq - s0 : 64u;
        const uint64_t r = r_next;
        r_next = s0 + 64u + (uint32_t)lane < pr.nbseq ? recs[s0 + 64u + (uint32_t)lane] : 0ull;
        const bool act = (uint32_t)lane < cnt;
        const uint32_t ll = act ? (uint32_t)r & 0xFFFFFu : 0u, ml = act ? (uint32_t)(r >> 20) & 0x3FFFFu : 0u, dd = (uint32_t)(r >> 38) & 3u;
        const uint32_t ov = (uint32_t)(r >> 40);
        uint32_t off;
        if (ov > 3u)
            off = ov - 3u;
        else if (ov == 0u)
            off = 0u;
        else
        {
            const uint32_t e = ov == 1u ? f.in[0] : ov == 2u ? f.in[1] : f.in[2];
            off = e > dd ? e - dd : 0u;
        }
        const uint32_t i_l = zx_scan_incl(ll), i_a = zx_scan_incl(ll + ml);
        const uint32_t batch_ll = (uint32_t)__builtin_amdgcn_readlane((int)i_l, 63), batch_adv = (uint32_t)__builtin_amdgcn_readlane((int)i_a, 63);
        const uint32_t o_l = produced + (i_a - ll - ml), o_m = o_l + ll; // where my literals / my match go (block positions)
        {
            const bool wrong = act && (off == 0u || ll > 131072u || ml > 131075u || ml < 3u || litpos + i_l > pr.nlit || produced + i_a > pr.expect ||
                                       off > start + o_m);
            if (__builtin_amdgcn_ballot_w64(wrong))
            {
                bad = true;
                why = __LINE__;
                break;
            }
        }
        zo_batch(org, start, lane, act, ll, litpos + (i_l - ll), ml, off, i_a, produced, s_ia, s_om);
        litpos += batch_ll;
        produced += batch_adv;
    }
    if (!bad)
    {
        // the block's last literals; it must regenerate what k_zstd_blk_entropy counted
        const uint32_t rest = pr.nlit - litpos;
        if (produced + rest != pr.expect)
        {
            bad = true;
            why = __LINE__;
        }
        else
            for (uint32_t j = lane; j < rest; j += 64)
                org[produced + j] = litpos + j;
    }
    if (bad && lane == 0)
        retry[b] = why ? why : 1u;
}

// block k of the payloads [pb0, pb0 + gridDim.y): 256 threads x 16 bytes per workgroup
__global__ __launch_bounds__(256) void k_zstd_fr_gather(const uint8_t* __restrict__ src, const ZBlock* __restrict__ blocks, uint32_t pb0, uint32_t k,
                                                       const uint32_t* __restrict__ f_nblocks, uint8_t* __restrict__ dst,
                                                       const uint8_t* __restrict__ lit_scratch, const ZPrep* __restrict__ fprep,
                                                       const ZFr* __restrict__ fr, const uint32_t* __restrict__ org_arena, uint32_t item0,
                                                       const uint32_t* __restrict__ retry)
{
    const uint32_t b = pb0 + blockIdx.y;
    if (k >= f_nblocks[b] || retry[b])
        return;
    const ZBlock blk = blocks[b];
    const uint32_t fi = blk.pad + k;
    const ZPrep pr = fprep[fi];
    const uint32_t q = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (q >= pr.expect)
        return;
    const uint32_t n = pr.expect - q < 16u ? pr.expect - q : 16u;
    const uint32_t start = fr[fi].start;
    uint8_t* out = dst + blk.dst_off;
    uint32_t w[4] = {0, 0, 0, 0}; // the sixteen bytes
    if (pr.log[0] == 2u)
    {
        const uint32_t x = src[pr.bits_off];
        w[0] = w[1] = w[2] = w[3] = x * 0x01010101u;
    }
    else if (pr.log[0] == 1u)
    {
        const uint8_t* from = src + pr.bits_off + q;
        for (uint32_t i = 0; i < n; ++i)
            w[i >> 2] |= (uint32_t)from[i] << (8u * (i & 3u));
    }
    else
    {
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t* org = org_arena + (uint64_t)(blk.zb_base - item0) * ZB + start + q;
        const uint8_t* lits = lit_scratch + pr.bits_off;
        uint32_t o[16];
        if (n == 16u)
        {
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(org + 4 * i);
                o[4 * i] = v.x;
                o[4 * i + 1] = v.y;
                o[4 * i + 2] = v.z;
                o[4 * i + 3] = v.w;
            }
        }
        else
        {
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i)
                o[i] = i < n ? org[i] : 0u;
        }
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i)
        {
            const uint32_t x = (o[i] & ZO_FLAG) ? out[o[i] & ~ZO_FLAG] : lits[o[i]];
            w[i >> 2] |= x << (8u * (i & 3u));
        }
    }
    uint8_t* to = out + start + q;
    if (n == 16u)
    {
        typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
        u32x4_a1 v;
        v.x = w[0];
        v.y = w[1];
        v.z = w[2];
        v.w = w[3];
        *reinterpret_cast<u32x4_a1*>(to) = v;
    }
    else
        for (uint32_t i = 0; i < n; ++i)
            to[i] = (uint8_t)(w[i >> 2] >> (8u * (i & 3u)));
}
} // namespace

// what the last lthip_zstd_decompress_blocks call did (diagnostics for the tests: which decoder the payloads went to)
// (kept in the context: one context per calling thread, so concurrent callers do not share them)
extern "C" int lthip_zstd_last_decode_stats(lthip_ctx* ctx, uint32_t out[4])
{
    if (!ctx || !out)
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> r(ctx->z_last_payloads);
    if (ctx->z_last_payloads)
        LTHIP_CHECK(ctx, hipMemcpy(r.data(), ctx->z_last_retry, 4 * (size_t)ctx->z_last_payloads, hipMemcpyDeviceToHost));
    uint32_t back = 0, first = 0;
    for (uint32_t v : r)
    {
        back += v ? 1u : 0u;
        first = first ? first : v;
    }
    out[3] = first; // (where the first of them was sent back: a source line of k_zstd.hip, 1 = not recorded)
    out[0] = ctx->z_last_payloads;       // payloads of the call
    out[1] = ctx->z_last_foreign_blocks; // blocks of other encoders' frames listed for the block-parallel path
    out[2] = back;                  // payloads a lane-parallel decoder gave back to the serial one
    return 0;
}

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
        hb[b].unit_base = 0;
        hb[b].pad = ZF_SLOTS * (uint32_t)nitems; // first block slot of the payload should it be another encoder's frame
        nitems += hb[b].nzb;
    }
    if (nitems > 0x7FFFFFF0ull / ZF_SLOTS)
        return lthip_fail(ctx, EINVAL, "zstd decode", "too many pieces in one call");
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    // 8 single-wave workgroups per CU (12 would be resident at 143 VGPRs, measured slower: 190 vs 160 ms for 512 blocks)
    uint32_t nwg = nitems < (uint64_t)ncu * 8u ? (uint32_t)nitems : (uint32_t)ncu * 8u;
    LTHIP_ABLATION_ENV(env_nwg, "LTHIP_ZSTD_NWG");
    LTHIP_ABLATION_ENV(env_dbg, "LTHIP_ZSTD_DBG");
    LTHIP_ABLATION_ENV(env_ablate, "LTHIP_ZSTD_ABLATE");
    LTHIP_ABLATION_ENV(env_zpx, "LTHIP_ZSTD_PX");
    // the sub-block pieces' sequences: 1 (default) = the kernel chooses per piece between the LDS ring (zx_batch) and bytes through memory
    // (zo_batch_bytes), 0 = always the ring (round 2), 2 = always through memory
    const uint32_t zpx = env_zpx.get() < 0 ? 1u : (uint32_t)env_zpx.get();
    if (env_nwg.get() >= 0)
        nwg = (uint32_t)env_nwg.get();
    void *d_blocks, *d_lits, *d_items;
    int err;
    if ((err = lthip_scratch(ctx, S_LZ4_BLOCKS, sizeof(ZBlock) * (size_t)block_count, &d_blocks)))
        return err;
    if ((err = lthip_scratch(ctx, S_Z_WORK, (size_t)(ZD_LIT_MAX + 64u) * nwg, &d_lits)))
        return err;
    // frames of other encoders, block-parallel: their blocks sit at FIXED slots (zb_base + k) of a second item list
    constexpr uint32_t ZROUND = 8192u;
    const size_t nrounds = (size_t)((nitems + ZROUND - 1) / ZROUND);
    const size_t nfslots = (size_t)ZF_SLOTS * nitems;
    const size_t ncounters = 8 + (size_t)block_count * 2 + nrounds + 10; // item count, totals | retry | f_nblocks | tickets | foreign ticket, list count, arenas
    if ((err = lthip_scratch(ctx, S_Z_ENC, sizeof(ZItem) * ((size_t)nitems + nfslots) + sizeof(ZPrep) * nfslots + 8 * nfslots + 4 * ncounters + 64, &d_items)))
        return err;
    ZItem* d_fitems = (ZItem*)d_items + nitems;
    ZPrep* d_fprep = (ZPrep*)(d_fitems + nfslots);
    uint32_t* d_flist = (uint32_t*)(d_fprep + nfslots); // blocks of other encoders' frames (slots), in no order
    uint32_t* d_slist = d_flist + nfslots;              // those of them that have sequences
    uint32_t* d_count = d_slist + nfslots;
    uint32_t* d_retry = d_count + 8; // per payload: the lane-parallel decoders give it back to the serial one
    uint32_t* d_fnb = d_retry + block_count; // per payload: blocks of a frame of another encoder (0: not decoded that way)
    uint32_t* d_tickets = d_fnb + block_count; // one work counter per round
    uint32_t* d_ftickets = d_tickets + nrounds + (((nrounds + (size_t)block_count * 2) & 1) ? 1 : 0); // (8-byte aligned: the arena counters follow)
    uint32_t* d_scount = d_ftickets + 1;
    unsigned long long* d_bump = (unsigned long long*)(d_ftickets + 2);
    ctx->z_last_retry = d_retry;
    ctx->z_last_payloads = block_count;
    ctx->z_last_foreign_blocks = 0;
    LTHIP_CHECK(ctx, hipMemsetAsync(d_fitems, 0, sizeof(ZItem) * nfslots, ctx->stream));
    LTHIP_CHECK(ctx, hipMemsetAsync(d_count, 0, 4 * ncounters, ctx->stream));
    if ((err = lthip_stage_upload(ctx, d_blocks, hb.data(), sizeof(ZBlock) * (size_t)block_count, ctx->stream)))
        return err;
    const uint32_t dbg = env_dbg.get() > 0 ? (uint32_t)env_dbg.get() : 0u; // 1: never decode by pieces
    if (env_ablate.get() >= 0)
    {
        const uint32_t a = (uint32_t)env_ablate.get();
        LTHIP_CHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_zd_ablate), &a, sizeof(a)));
    }
    // the items in link-major order (k_zstd_rows), and a done flag per item for the frames whose pieces form a chain
    const uint32_t nrows = ZCHAIN;
    void* d_pm;
    if ((err = lthip_scratch(ctx, S_Z_PERM, 4 * (2 * (size_t)nitems + 2 * (size_t)nrows + 16), &d_pm)))
        return err;
    uint32_t* d_perm = (uint32_t*)d_pm;
    uint32_t* d_done = d_perm + nitems;
    uint32_t* d_row_cnt = d_done + nitems;
    uint32_t* d_row_start = d_row_cnt + nrows;
    LTHIP_CHECK(ctx, hipMemsetAsync(d_done, 0, 4 * ((size_t)nitems + nrows), ctx->stream));
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_zstd_split, dim3(block_count), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                       block_count, (ZItem*)d_items, d_count, d_out_sizes, dbg, d_fitems, d_fnb, d_flist);
    {
        const uint32_t g = (uint32_t)((nitems + 255) / 256);
        hipLaunchKernelGGL(k_zstd_rows, dim3(g), dim3(256), 0, ctx->stream, (const ZItem*)d_items, (const uint32_t*)d_count, nrows, d_row_cnt, d_row_start, d_perm, 0u);
        hipLaunchKernelGGL(k_zstd_rows, dim3(1), dim3(64), 0, ctx->stream, (const ZItem*)d_items, (const uint32_t*)d_count, nrows, d_row_cnt, d_row_start, d_perm, 1u);
        hipLaunchKernelGGL(k_zstd_rows, dim3(g), dim3(256), 0, ctx->stream, (const ZItem*)d_items, (const uint32_t*)d_count, nrows, d_row_cnt, d_row_start, d_perm, 2u);
    }
    hipLaunchKernelGGL(k_zstd_decode<false>, dim3(nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                       (const ZItem*)d_items, (const uint32_t*)d_count, (uint8_t*)d_dst, (uint8_t*)d_lits, d_out_sizes, (const ZPrep*)nullptr);
    LTHIP_LAUNCH_CHECK(ctx);
    // pieces, in rounds of ZROUND whose literals and tables / sequence records live in scratch:
    //   one block per piece (kind 2):        the serial piece decoder (k_zstd_decode<true>).  Only the ablation build's encoder writes such
    //                                        frames (LTHIP_ZSTD_SUB=0); that build decodes them in two stages, k_zstd_prepare +
    //                                        k_zstd_execute<false>, and what they leave (ZP_SERIAL) goes to the serial piece decoder
    //   a run of sub-blocks per piece (3):   k_zstd_sub_entropy + k_zstd_execute<true>; what they leave goes, payload-wise, to the serial decoder
    // LTHIP_ZSTD_DBG & 4: the serial piece decoder for every one-block piece.
    ZPrep* d_prep = nullptr;
    {
        const uint32_t per_round = nitems < ZROUND ? (uint32_t)nitems : ZROUND;
        const uint32_t slots = per_round;
        void *d_plits, *d_tabs, *d_pr, *d_recs;
        if ((err = lthip_scratch(ctx, S_Z_LITS, (size_t)(ZD_LIT_MAX + 64u) * slots + 4096, &d_plits)))
            return err;
        if ((err = lthip_scratch(ctx, S_Z_RECS, (size_t)3u * ZT_ENTRIES * 8u * per_round, &d_tabs)))
            return err;
        if ((err = lthip_scratch(ctx, S_Z_SUB, (size_t)ZREC_MAX * 8u * slots, &d_recs)))
            return err;
        if ((err = lthip_scratch(ctx, S_LZ4_META, sizeof(ZPrep) * (size_t)nitems, &d_pr)))
            return err;
        d_prep = (ZPrep*)d_pr;
        LTHIP_CHECK(ctx, hipMemsetAsync(d_pr, 0xFF, sizeof(ZPrep) * (size_t)nitems, ctx->stream)); // (status of items nobody prepares: none of the three)
        for (uint64_t i0 = 0; i0 < nitems; i0 += per_round)
        {
            const uint32_t i1 = (uint32_t)(i0 + per_round < nitems ? i0 + per_round : nitems);
            const uint32_t n = i1 - (uint32_t)i0;
            hipLaunchKernelGGL(k_zstd_plain_pieces, dim3(n), dim3(256), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                               (const ZItem*)d_items, (const uint32_t*)d_count, (uint32_t)i0, i1, (uint8_t*)d_dst, d_prep, (const uint32_t*)d_perm);
            LTHIP_LAUNCH_CHECK(ctx);
#ifdef LTHIP_ABLATIONS
            if (!(dbg & 4u))
            {
                hipLaunchKernelGGL(k_zstd_prepare, dim3(n < nwg ? n : nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                                   (const ZItem*)d_items, (const uint32_t*)d_count, (uint32_t)i0, i1, (uint8_t*)d_dst, (uint8_t*)d_plits,
                                   (uint64_t*)d_tabs, d_prep, (const uint32_t*)d_out_sizes, (const uint32_t*)d_perm);
                LTHIP_LAUNCH_CHECK(ctx);
                hipLaunchKernelGGL(k_zstd_execute<false>, dim3(n), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                                   (const ZItem*)d_items, (const uint32_t*)d_count, (uint32_t)i0, i1, (uint8_t*)d_dst, (const uint8_t*)d_plits,
                                   (const uint64_t*)d_tabs, (const ZPrep*)d_prep, &d_prep->status, d_retry, 0u, (const uint32_t*)d_perm, (uint32_t*)nullptr);
                LTHIP_LAUNCH_CHECK(ctx);
            }
#else
            (void)d_tabs;
#endif
            hipLaunchKernelGGL(k_zstd_sub_entropy, dim3(n < nwg ? n : nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                               (const ZItem*)d_items, (const uint32_t*)d_count, (uint32_t)i0, i1, (uint8_t*)d_plits, (uint64_t*)d_recs, d_prep,
                               (const uint32_t*)d_out_sizes, d_retry, d_tickets + (size_t)(i0 / per_round), (const uint32_t*)d_perm);
            LTHIP_LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL(k_zstd_execute<true>, dim3(n), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                               (const ZItem*)d_items, (const uint32_t*)d_count, (uint32_t)i0, i1, (uint8_t*)d_dst, (const uint8_t*)d_plits,
                               (const uint64_t*)d_recs, (const ZPrep*)d_prep, &d_prep->status, d_retry, zpx, (const uint32_t*)d_perm, d_done);
            LTHIP_LAUNCH_CHECK(ctx);
        }
    }
    // frames of other encoders: the streams of every block on a wave of its own, then a payload's blocks in order on one wave -- all
    // payloads at once (a payload is ONE chain of dependent copies: only many of them fill the machine).  How many there are, and how
    // large the literal and record arenas must be, is known after k_zstd_split: the one place where this call waits for the device.
    if (!(dbg & 9u))
    {
        uint32_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        LTHIP_CHECK(ctx, hipMemcpyAsync(counters, d_count, sizeof(counters), hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t* totals = counters + 4;
        const uint32_t f_most = counters[1]; // the most blocks a frame of another encoder has
        const uint64_t f_blocks = totals[0], f_bytes = ((uint64_t)totals[3] << 32) | totals[2];
        ctx->z_last_foreign_blocks = totals[0];
        if (f_blocks)
        {
            const uint64_t lit_cap = f_bytes + 96ull * f_blocks + 4096ull, rec_cap = f_bytes / 6ull + 64ull * f_blocks + 4096ull;
            void *d_flits, *d_frecs;
            if ((err = lthip_scratch(ctx, S_Z_LITS, (size_t)lit_cap + 4096, &d_flits)))
                return err;
            if ((err = lthip_scratch(ctx, S_Z_SUB, (size_t)rec_cap * 8u, &d_frecs)))
                return err;
            const uint32_t n = (uint32_t)f_blocks;
            // few blocks: their sequences on the block's own wave; VERY few (a wave or two per CU: one to eight frames of 8 MiB): on the
            // scalar unit of a wave of their own, which walks a block's chain of sequences twice as fast as a lane (LTHIP_ZSTD_SEQ_SCALAR=0: off)
            LTHIP_ABLATION_ENV(env_scal, "LTHIP_ZSTD_SEQ_SCALAR");
            int ncu = 256;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
            const bool scalar_seqs = env_scal.get() != 0 && (n <= 2u * (uint32_t)ncu || env_scal.get() == 2); // (2: always -- tests)
            const uint32_t inline_seqs = !scalar_seqs && n <= 4u * nwg ? 1u : 0u;
            void* d_ftabs;
            if ((err = lthip_scratch(ctx, S_Z_RECS, (size_t)n * 1280u * 8u, &d_ftabs))) // (10 KiB of packed tables per block)
                return err;
            hipLaunchKernelGGL(k_zstd_blk_entropy, dim3(n < nwg ? n : nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZItem*)d_fitems,
                               (const uint32_t*)d_flist, n, d_slist, d_scount, (uint8_t*)d_flits, (uint64_t*)d_frecs, d_fprep, d_retry, d_ftickets, d_bump,
                               lit_cap, rec_cap, (uint64_t*)d_ftabs, inline_seqs);
            LTHIP_LAUNCH_CHECK(ctx);
            if (scalar_seqs)
                hipLaunchKernelGGL(k_zstd_blk_seq_scalar, dim3(n), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZItem*)d_fitems,
                                   (const uint32_t*)d_slist, (const uint32_t*)d_scount, (const uint64_t*)d_ftabs, (uint64_t*)d_frecs, d_fprep, d_retry);
            else if (!inline_seqs)
            hipLaunchKernelGGL(k_zstd_blk_sequences, dim3((n + 63u) / 64u), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZItem*)d_fitems,
                               (const uint32_t*)d_slist, (const uint32_t*)d_scount, (const uint64_t*)d_ftabs, (uint64_t*)d_frecs, d_fprep, d_retry);
            LTHIP_LAUNCH_CHECK(ctx);
#ifdef LTHIP_ABLATIONS
            if (dbg & 16u) // (round 2's way: a payload's blocks one after the other on ONE wave)
            {
                hipLaunchKernelGGL(k_zstd_execute_payload, dim3(block_count), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks, 0u,
                                   block_count, (const uint32_t*)d_fnb, (uint8_t*)d_dst, (const uint8_t*)d_flits, (const uint64_t*)d_frecs,
                                   (const ZPrep*)d_fprep, d_retry);
                LTHIP_LAUNCH_CHECK(ctx);
            }
            else
#endif
            {
                // execution on origins, all blocks at once; the origins (4 bytes per byte of output) of as many payloads at a time as
                // the arena's budget allows (LTHIP_ORIGIN_MIB, default: lthip_origin_budget_mib)
                void *d_fr, *d_org;
                if ((err = lthip_scratch(ctx, S_Z_FR, sizeof(ZFr) * nfslots, &d_fr)))
                    return err;
                const uint64_t budget_items = (lthip_origin_budget_mib() << 20) / ((uint64_t)ZB * 4u);
                uint64_t most = 0;
                for (uint32_t p0 = 0; p0 < block_count;)
                {
                    uint64_t items = hb[p0].nzb;
                    uint32_t p1 = p0 + 1;
                    while (p1 < block_count && items + hb[p1].nzb <= budget_items)
                        items += hb[p1++].nzb;
                    most = items > most ? items : most;
                    p0 = p1;
                }
                if ((err = lthip_scratch(ctx, S_Z_ORG, (size_t)most * ZB * 4u + 256, &d_org)))
                    return err;
                hipLaunchKernelGGL(k_zstd_fr_reps, dim3(n), dim3(64), 0, ctx->stream, (const uint32_t*)d_flist, n, (uint64_t*)d_frecs,
                                   (const ZPrep*)d_fprep, (ZFr*)d_fr, (const ZItem*)d_fitems, d_retry);
                hipLaunchKernelGGL(k_zstd_fr_chain, dim3((block_count + 63u) / 64u), dim3(64), 0, ctx->stream, (const uint8_t*)d_src,
                                   (const ZBlock*)d_blocks, block_count, (const uint32_t*)d_fnb, (const ZPrep*)d_fprep, (ZFr*)d_fr, d_retry);
                LTHIP_LAUNCH_CHECK(ctx);
                for (uint32_t p0 = 0; p0 < block_count;)
                {
                    uint64_t items = hb[p0].nzb;
                    uint32_t p1 = p0 + 1;
                    while (p1 < block_count && items + hb[p1].nzb <= budget_items)
                        items += hb[p1++].nzb;
                    hipLaunchKernelGGL(k_zstd_fr_trace, dim3((uint32_t)(items * ZF_SLOTS)), dim3(64), 0, ctx->stream, (const ZBlock*)d_blocks,
                                       (const ZItem*)d_fitems, hb[p0].pad, (const uint32_t*)d_fnb, (const uint64_t*)d_frecs, (const ZPrep*)d_fprep,
                                       (const ZFr*)d_fr, (uint32_t*)d_org, hb[p0].zb_base, d_retry);
                    LTHIP_LAUNCH_CHECK(ctx);
                    for (uint32_t k = 0; k < f_most; ++k)
                        hipLaunchKernelGGL(k_zstd_fr_gather, dim3(ZB / 4096u, p1 - p0), dim3(256), 0, ctx->stream, (const uint8_t*)d_src,
                                           (const ZBlock*)d_blocks, p0, k, (const uint32_t*)d_fnb, (uint8_t*)d_dst, (const uint8_t*)d_flits,
                                           (const ZPrep*)d_fprep, (const ZFr*)d_fr, (const uint32_t*)d_org, hb[p0].zb_base, (const uint32_t*)d_retry);
                    LTHIP_LAUNCH_CHECK(ctx);
                    p0 = p1;
                }
            }
        }
    }
    hipLaunchKernelGGL(k_zstd_decode<true>, dim3(nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, (const ZBlock*)d_blocks,
                       (const ZItem*)d_items, (const uint32_t*)d_count, (uint8_t*)d_dst, (uint8_t*)d_lits, d_out_sizes,
#ifdef LTHIP_ABLATIONS
                       (dbg & 4u) ? (const ZPrep*)nullptr : (const ZPrep*)d_prep);
#else
                       (const ZPrep*)nullptr);
#endif
    hipLaunchKernelGGL(k_zstd_decode_retry, dim3(block_count < nwg ? block_count : nwg), dim3(64), 0, ctx->stream, (const uint8_t*)d_src,
                       (const ZBlock*)d_blocks, block_count, (const uint32_t*)d_retry, (uint8_t*)d_dst, (uint8_t*)d_lits, d_out_sizes);
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
