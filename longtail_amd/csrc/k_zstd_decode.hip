// k_zstd_decode.hip -- ZStd CompressionAPI, GPU decoders (SURVEY.md §8 a6 Decompress, row f3): lthip_zstd_decompress_blocks.
// The encoder is k_zstd.hip; the frame format both agree on is zstd/k_zstd_common.h; the serial decoder core every path falls back
// to is zstd_decode_core.h.  This translation unit, in the order of its parts:
//   here                       work items (a payload, or the pieces of one of this library's own frames): k_zstd_split, k_zstd_rows,
//                              k_zstd_decode (one wave per item, the serial core)
//   zstd/zd_sub_blocks.inc     own frames whose pieces are runs of sub-blocks: every lane decodes streams of its own
//   zstd/zd_foreign_blocks.inc frames of other encoders (the reference's): one wave per BLOCK for the entropy stage
//   zstd/zd_execute.inc        executing a piece's sequences through an LDS ring (own frames, chains of pieces of the "max" setting)
//   zstd/zd_foreign_frames.inc executing other encoders' frames block-parallel on origins
//   here                       the launcher
#include "lthip_internal.h"

#include <type_traits>

#define K_ZSTD_DECODER
#include "zstd/k_zstd_common.h"

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

#include "zstd/zd_sub_blocks.inc"
#include "zstd/zd_foreign_blocks.inc"
#include "zstd/zd_execute.inc"
#include "zstd/zd_foreign_frames.inc"
}
} // namespace

// what the last lthip_zstd_decompress_blocks call did (diagnostics for the tests: which decoder the payloads went to)
// (kept in the context: one context per calling thread, so concurrent callers do not share them)
extern "C" int lthip_zstd_last_decode_stats(lthip_ctx* ctx, uint32_t out[4])
{
    if (!ctx || !out)
        return EINVAL;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
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
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
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
