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

#include "zstd/k_zstd_common.h"

namespace
{

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


// Diagnostics for the parity tests: the match finder's output of the LAST lthip_zstd_compress_blocks call on this
// context (units [first, first+count)), so that the host model of the entropy stage can be run on the same input.
extern "C" int lthip_zstd_debug_units(lthip_ctx* ctx, uint64_t first, uint64_t count, void* h_meta, void* h_lits, void* h_recs)
{
    if (!ctx || !ctx->scratch[S_Z_LITS] || !ctx->scratch[S_Z_RECS] || !ctx->scratch[S_LZ4_META])
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
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
