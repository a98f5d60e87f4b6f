// k_lz4.hip -- per-block LZ4 (block format) compression and decompression on gfx950.
//
// Reference behaviour: LZ4CompressionAPI_Compress/_Decompress (lib/lz4/longtail_lz4.c:52-102) =
// LZ4_compress_fast(acc 1) / LZ4_decompress_safe of the vendored LZ4 1.10.0 (lib/lz4/ext/lz4.c:930-1338,
// 2016-2445).  The contract kept is the FORMAT one: every payload is one LZ4 block that the
// reference decoder expands to the original bytes; the parse itself is re-designed for a GPU:
//
//   K5 lz4_segments  a stored block (<= 8.8 MiB) is cut into 4 KiB UNITS; consecutive units form a window group that is staged
//                    once into LDS and parsed by one workgroup, one wave per unit, each with a PRIVATE u16 hash table (results
//                    do not depend on timing).  Two passes (launch_match_finder):
//                      1. classification, batch geometry (8 units / 32 KiB, 1280-entry tables, 24 waves per CU): four probe
//                         batches of 64 positions decide whether the group holds redundancy.  If not -- incompressible
//                         data -- the same workgroup skims it (probe stride grows with the misses, lz4.c:1044-1053; units
//                         without a match put their bytes where an all-literal block wants them); if so the 16-unit group is
//                         put on a list and left alone.
//                      2. lane parser (16 units / 64 KiB, 2560-entry tables, one persistent workgroup per CU) over the list:
//                         every LANE owns a 64-byte sub-unit and runs the reference's greedy loop on it (probe; on a hit extend
//                         forwards 36 bytes on its own -- longer ones with the whole wave -- and 8 backwards, record, jump;
//                         else step 1 + misses / 4): 64 parsers in lock step, no selection among hits.  History enters the
//                         table every 4th position (aligned dwords, one 16-byte read per four inserts); lanes map to
//                         sub-units in reverse so that the lowest position survives a write conflict; matches may cross
//                         sub-units, what they cover is dropped from the later lanes' records (prefix maximum); three wave
//                         scans place the sequences and every lane writes its own bytes.
//                    LTHIP_LZ4_PARSER=batch keeps the round-1 batch parser for everything (64 consecutive positions per step,
//                    scalar selection among the hits).  Sequences go to a per-unit stream.
//   K6 lz4_stitch    per block: three wave scans over the units' results (literal carry, output position, run count) turn
//                    trailing literals of one unit into leading literals of the next sequence and assign exact output offsets,
//                    honouring the end-of-block rules (last 5 bytes literal, last match starts >= 12 bytes before the end;
//                    lz4.c:242-246, 2279-2329, 2421-2423); then one wave per unit, persistent over a work list, copies header,
//                    literals (from the source) and sequence body (from the stream) into the single final block with 16-byte
//                    stores.  Blocks without a single match were laid out by K5: only their header is written.
//   decode           k_lz4_decode.hip
#include "lthip_internal.h"

#include <string>

#include <stdlib.h>

namespace
{

struct Lz4Block
{
    uint64_t src_off;
    uint64_t dst_off;
    uint32_t size;
    uint32_t dst_cap;
    uint32_t seg_base; // first stitch unit (sub-segment) of the block
    uint32_t nseg;     // number of units
    uint32_t grp_base; // first window group of the block (groups of `gunits` units: the parser's and the stitch's)
    uint32_t ngrp;
    uint32_t cgrp_base; // first group of the classification pass (LZ4_G_BATCH units per group), see lz4_compress_batch
    uint32_t ncgrp;
};

struct Lz4Meta // result of one segment
{
    uint32_t seq_bytes;       // bytes of complete sequences in the segment stream (0 = no match found)
    uint32_t tail_lits;       // trailing literal bytes not covered by a sequence
    uint32_t first_lit_len;   // literal length of the first sequence
    uint32_t first_hdr_bytes; // token + literal-length bytes of the first sequence
};

struct Lz4Plan // where one segment's pieces go, dst offsets relative to the block's output
{
    uint32_t hdr_pos;       // rewritten first token position (valid when the segment has a match)
    uint32_t hdr_lits;      // literal length to encode there (carry-in + first_lit_len)
    uint32_t first_lit_dst; // destination of the segment's own leading literals
    uint32_t body_dst;      // destination of the rest of the stream
    uint32_t tail_rel;      // offset of the trailing literals inside the literal run they belong to
    uint32_t run;           // index of that run in the run table (its literal-area start)
};

struct Lz4BlockOut
{
    uint32_t final_hdr_pos;
    uint32_t final_lits;
    uint32_t total; // 0 = does not fit
    uint32_t pad;
};

constexpr uint32_t LZ4_EMPTY = 0xFFFFu;

__host__ __device__ __forceinline__ uint32_t lz4_stream_stride(uint32_t seg) { return (seg + seg / 255u + 16u + 15u) & ~15u; }
__host__ __device__ __forceinline__ uint32_t lz4_len_bytes(uint32_t len) { return len >= 15u ? (len - 15u) / 255u + 1u : 0u; }
// ... of a length below 65 536 (what a lane's record holds) without a branch and without a 32-bit multiply: (len + 240) / 255 is the
// same number for every len (0 below 15), and n / 255 = n * 0x8081 >> 23 is exact for n < 66 299
__device__ __forceinline__ uint32_t lz4_len_bytes16(uint32_t len) { return __umul24(len + 240u, 0x8081u) >> 23; }

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// ---------------------------------------------------------------------------------------------------
// K5
// ---------------------------------------------------------------------------------------------------
// four bytes at any byte position of the LDS window: two aligned reads (one ds_read2_b32) and a v_alignbyte.  A single
// unaligned ds_read_b32 is legal on gfx950 but measured slower (K5 +15 %: the LDS splits misaligned dwords expensively).
__device__ __forceinline__ uint32_t lds_read32(const uint32_t* sdata, uint32_t byte_idx)
{
    const uint32_t w = byte_idx >> 2;
    return __builtin_amdgcn_alignbyte(sdata[w + 1], sdata[w], byte_idx & 3u);
}

// PADDED window (the second formulation of the lane parser, PV 2).  Lanes own consecutive 64-byte sub-units, so whatever the lanes of a
// wave read "at their own position" lies 16 dwords apart: two of the 32 LDS banks serve a 32-lane group, a 16-way conflict on every
// such read while the lanes run in phase (any data made of aligned structures; SQ_LDS_BANK_CONFLICT was 62 % of the LDS-array cycles on
// "tokens", and half of that went away with an odd sub-unit pitch -- which costs ratio and lanes).  Here the window keeps its 64-byte
// sub-units and is stored in ROWS of 32 dwords (128 bytes) at a pitch of 35: two neighbouring sub-units share a row, rows shift by
// three banks each -- the 32 lanes of a group land on 32 different banks (3 j + 16 b mod 32, j < 16, b < 2: all distinct).  The three
// extra dwords of a row REPEAT the first three dwords of the next row, so that a run of up to four dwords starting anywhere in a row is
// contiguous in LDS: one address computation (x + 12 (x >> 7): two instructions) and ds_read2_b32 with constant offsets, no matter
// where the row ends.
constexpr uint32_t LZ4_ROW_DUP = 3; // repeated dwords per 32-dword row
template <bool PAD>
__device__ __forceinline__ uint32_t lds_pidx(uint32_t D)
{
    return PAD ? D + LZ4_ROW_DUP * (D >> 5) : D;
}
// the dword that holds window byte x, as a pointer: x + 12 (x >> 7) in bytes -- a shift and one v_mad_u32_u24 (a 32-bit multiply is
// a quarter-rate instruction), the runs then use the LDS instructions' immediate offsets
template <bool PAD>
__device__ __forceinline__ const uint32_t* lds_ptr(const uint32_t* sdata, uint32_t x)
{
    const uint32_t xa = x & ~3u;
    const uint32_t a = PAD ? __umul24(xa >> 7, 4u * LZ4_ROW_DUP) + xa : xa;
    if constexpr (PAD)
    {
        // The padded window is the lane parser's, and there it is the kernel's dynamic LDS, which starts at LDS address 0 (k_lz4_lanes2
        // has no static LDS and checks this once): the byte offset IS the address.  Through `sdata` the compiler adds the window's base
        // to every address -- a constant it learns too late to fold --, one more vector instruction per address in a kernel that is
        // bound by their number.
        typedef __attribute__((address_space(3))) const uint32_t* lds_cptr_t;
        return (const uint32_t*)(lds_cptr_t)(uintptr_t)a;
    }
    return reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(sdata) + a);
}
// (what lds_ptr<true> relies on; a kernel with a padded window calls this first)
__device__ __forceinline__ void lds_window_must_start_at_zero(const uint32_t* sdata)
{
    typedef __attribute__((address_space(3))) const uint32_t* lds_cptr_t;
    if ((uint32_t)(uintptr_t)(lds_cptr_t)sdata != 0u)
        __builtin_trap();
}
template <bool PAD>
__device__ __forceinline__ uint32_t lds_dw(const uint32_t* sdata, uint32_t D)
{
    return *lds_ptr<PAD>(sdata, D << 2);
}
// N <= 4 consecutive dwords of the window from dword D
template <bool PAD, int N>
__device__ __forceinline__ void lds_run(const uint32_t* sdata, uint32_t D, uint32_t* out)
{
    static_assert(N >= 1 && N <= 1 + (int)LZ4_ROW_DUP, "a run is contiguous up to 1 + LZ4_ROW_DUP dwords");
    const uint32_t* q = lds_ptr<PAD>(sdata, D << 2);
#pragma unroll
    for (int k = 0; k < N; ++k)
        out[k] = q[k];
}
template <bool PAD>
__device__ __forceinline__ uint32_t lds_byte(const uint32_t* sdata, uint32_t x)
{
    if constexpr (PAD) // (the offset is the address: see lds_ptr)
    {
        typedef __attribute__((address_space(3))) const uint8_t* lds_bptr_t;
        return *(const uint8_t*)(lds_bptr_t)(uintptr_t)(x + __umul24(x >> 7, 4u * LZ4_ROW_DUP));
    }
    return reinterpret_cast<const uint8_t*>(sdata)[x];
}
template <bool PAD>
__device__ __forceinline__ uint32_t lds_read32x(const uint32_t* sdata, uint32_t byte_idx)
{
    if constexpr (!PAD)
        return lds_read32(sdata, byte_idx);
    uint32_t d[2];
    lds_run<true, 2>(sdata, byte_idx >> 2, d);
    return __builtin_amdgcn_alignbyte(d[1], d[0], byte_idx); // (v_alignbyte_b32 shifts by the selector's two low bits: no mask)
}
// bytes of LDS a window of n data bytes occupies
__host__ __device__ constexpr uint32_t lz4_window_lds_bytes(uint32_t n, bool pad)
{
    return pad ? ((((n + 127u) >> 7) * (128u + 4u * LZ4_ROW_DUP) + 16u + 15u) & ~15u) : n;
}

// n bytes of the LDS window (from byte `sbyte`) to global memory, one wave, 16-byte stores on the aligned part
template <bool PAD = false>
__device__ __forceinline__ void wave_copy_lds_to_global(uint8_t* __restrict__ dst, const uint32_t* sdata, uint32_t sbyte, uint32_t n,
                                                        int lane)
{
    uint32_t headb = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (headb > n)
        headb = n;
    if ((uint32_t)lane < headb)
        dst[lane] = (uint8_t)lds_byte<PAD>(sdata, sbyte + (uint32_t)lane);
    dst += headb;
    sbyte += headb;
    n -= headb;
    const uint32_t nvec = n >> 4;
    for (uint32_t v = lane; v < nvec; v += 64)
    {
        const uint32_t b = sbyte + 16u * v;
        uint4 o;
        if constexpr (PAD)
        {
            const uint32_t w = b >> 2, sh = b & 3u;
            uint32_t d[4];
            lds_run<true, 4>(sdata, w, d);
            const uint32_t d4 = lds_dw<true>(sdata, w + 4u);
            o.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
            o.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
            o.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
            o.w = __builtin_amdgcn_alignbyte(d4, d[3], sh);
        }
        else
        {
            o.x = lds_read32(sdata, b);
            o.y = lds_read32(sdata, b + 4u);
            o.z = lds_read32(sdata, b + 8u);
            o.w = lds_read32(sdata, b + 12u);
        }
        *reinterpret_cast<uint4*>(dst + (uint64_t)v * 16u) = o;
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)lane < n - done)
        dst[done + (uint32_t)lane] = (uint8_t)lds_byte<PAD>(sdata, sbyte + done + (uint32_t)lane);
}

// wave-cooperative emission of a length (already reduced by 15) as 255,255,...,rem
__device__ __forceinline__ void emit_len(uint8_t* out, uint32_t len, int lane)
{
    const uint32_t n = len / 255u + 1u;
    for (uint32_t j = lane; j < n; j += 64)
        out[j] = j + 1 == n ? (uint8_t)(len % 255u) : (uint8_t)255;
}

// Parser state shared by the cooperative and the lane-parallel paths (all wave-uniform).
struct Lz4Seq
{
    uint32_t op;        // bytes written to the unit's stream
    uint32_t anchor;    // first byte not yet covered by a sequence
    uint32_t first_lit; // literal length / header bytes of the unit's first sequence (for the stitcher)
    uint32_t first_hdr;
    uint32_t nseq; // FMT 1: sequences emitted so far
    bool have_first;
};

// One sequence handled by the whole wave: optionally extend the match (forwards from `mlen` bytes already known to
// be equal, backwards down to the anchor, lz4.c:1104-1109) and emit  token | literal length | literals | offset |
// match length  (lz4.c:1111-1226), every output byte produced by "its" lane.
// FMT 0 writes the LZ4 byte stream; FMT 1 (zstd front end) appends the literals to `out` and one
// {literals, match length, offset} record to `recs` (zstd_block_core.h ZB_REC).
template <int FMT>
__device__ __forceinline__ void lz4_coop_sequence(const uint8_t* __restrict__ sbytes, uint32_t head, uint8_t* __restrict__ out,
                                                  uint64_t* __restrict__ recs, int lane, uint32_t end_limit, uint32_t pf, uint32_t cf,
                                                  uint32_t mlen, bool extend_fwd, bool extend_back, Lz4Seq& st)
{
    if (extend_fwd || extend_back)
    {
        // first round: lanes 0..31 compare forwards, lanes 32..63 backwards; addresses are selected, not branched
        // on, so both directions share ONE pair of LDS reads
        const uint32_t room = extend_back ? (pf - st.anchor < cf ? pf - st.anchor : cf) : 0u;
        uint32_t nf, nb;
        {
            const uint32_t j = (uint32_t)lane & 31u;
            const bool fwd = lane < 32;
            const uint32_t a1 = fwd ? pf + mlen + j : pf - 1u - j;
            const uint32_t a2 = fwd ? cf + mlen + j : cf - 1u - j;
            const bool inr = fwd ? (extend_fwd && a1 < end_limit) : j < room;
            bool same = false;
            if (inr)
                same = sbytes[a1 + head] == sbytes[a2 + head];
            const uint64_t diff = __builtin_amdgcn_ballot_w64(!same);
            const uint32_t dl = (uint32_t)diff, dh = (uint32_t)(diff >> 32);
            nf = dl ? (uint32_t)__builtin_ctz(dl) : 32u;
            nb = dh ? (uint32_t)__builtin_ctz(dh) : 32u;
        }
        mlen += nf;
        if (nf == 32u)
        {
            // the match goes on: 256 bytes per LDS round trip (a dword pair per lane)
            const uint32_t* sdata = reinterpret_cast<const uint32_t*>(sbytes);
            for (;;)
            {
                const uint32_t i = pf + mlen + 4u * (uint32_t)lane;
                uint32_t cnt = 0; // equal bytes of my four, as far as the unit goes
                if (i < end_limit)
                {
                    const uint32_t x = lds_read32(sdata, i + head) ^ lds_read32(sdata, cf + mlen + 4u * (uint32_t)lane + head);
                    const uint32_t lim = end_limit - i < 4u ? end_limit - i : 4u;
                    cnt = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
                    cnt = cnt < lim ? cnt : lim;
                }
                const uint64_t diff = __builtin_amdgcn_ballot_w64(cnt < 4u);
                if (diff)
                {
                    const int f = __builtin_ctzll(diff);
                    mlen += 4u * (uint32_t)f + __builtin_amdgcn_readlane(cnt, f);
                    break;
                }
                mlen += 256u;
            }
        }
        if (nb == 32u && room > 32u)
        {
            uint32_t back = 32u;
            for (;;)
            {
                const uint32_t j = back + (uint32_t)lane;
                const bool same = j < room && sbytes[pf - 1u - j + head] == sbytes[cf - 1u - j + head];
                const uint64_t diff = __builtin_amdgcn_ballot_w64(!same);
                if (diff)
                {
                    back += (uint32_t)__builtin_ctzll(diff);
                    break;
                }
                back += 64u;
            }
            nb = back;
        }
        pf -= nb;
        cf -= nb;
        mlen += nb;
    }
    const uint32_t lit = pf - st.anchor;
    const uint32_t mcode = mlen - 4u;
    const uint32_t off = pf - cf;
    if constexpr (FMT == 1)
    {
        for (uint32_t j = lane; j < lit; j += 64)
            out[st.op + j] = sbytes[st.anchor + j + head];
        if (lane == 0)
            recs[st.nseq] = (uint64_t)lit | ((uint64_t)mlen << 16) | ((uint64_t)off << 32);
        st.have_first = true;
        st.nseq += 1u;
        st.op += lit;
        st.anchor = pf + mlen;
        return;
    }
    const uint32_t hdr = 1u + lz4_len_bytes(lit);
    const uint32_t mext = lz4_len_bytes(mcode);
    const uint32_t total = hdr + lit + 2u + mext;
    const uint32_t token = ((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u);
    for (uint32_t j = lane; j < total; j += 64)
    {
        uint32_t b;
        if (j >= hdr && j < hdr + lit)
            b = sbytes[st.anchor + (j - hdr) + head];
        else if (j == 0)
            b = token;
        else if (j < hdr)
            b = j + 1 == hdr ? (lit - 15u) % 255u : 255u;
        else if (j == hdr + lit)
            b = off & 255u;
        else if (j == hdr + lit + 1u)
            b = off >> 8;
        else
            b = j + 1 == total ? (mcode - 15u) % 255u : 255u;
        out[st.op + j] = (uint8_t)b;
    }
    if (!st.have_first)
    {
        st.have_first = true;
        st.first_lit = lit;
        st.first_hdr = hdr;
    }
    st.op += total;
    st.anchor = pf + mlen;
}

// K5.  A workgroup of G waves owns one WINDOW GROUP = G consecutive sub-segments ("units", default 8 x 4 KiB) of one
// block, staged once into LDS.  Wave w parses unit w sequentially (LZ4 parsing is a chain), may match against ANY
// earlier byte of the group (units 0..w-1 are plain history for it, like the preceding bytes of a 32 KiB segment),
// and uses a PRIVATE table (TAB entries) so the result does not depend on the other waves' timing.  The table learns
// the history lazily: every wave first parses PROBE batches with an empty table; if any wave of the group finds a
// match the data is taken to be compressible and each wave inserts the positions before its unit (newer entries
// kept), otherwise (incompressible data) nobody pays for it.  24 waves per CU instead of 4 at the same window.
constexpr int LZ4_G_BATCH = 8;   // MODE 0 (batch parser): 8 x 4 KiB units per 32 KiB window group
constexpr int LZ4_G_LANES = 16;  // MODE 1 (lane parser): 16 x 4 KiB units per 64 KiB window group, one workgroup per CU
constexpr int LZ4_PROBE_BATCHES = 4;
constexpr uint32_t LZ4_Z_CHAIN = LTHIP_ZSTD_CHAIN;
constexpr uint32_t Z_PIECE = 128u << 10; // zstd Block_Maximum_Size (ZB_BLOCK_MAX of zstd_block_core.h, asserted in k_zstd.hip)
// table entries per wave.  MODE 0: 32 KiB window + 8 x 2.5 KiB tables = 52 KiB: THREE workgroups (24 waves) per CU.
// MODE 1: 64 KiB window + 16 x 5 KiB tables = 144 KiB: one workgroup of 16 waves per CU (the lane parser is not issue bound, it
// wants history: tools/lz4_lane_model.c -- mixed 1.72 at 8 x 1280, 1.85 at 16 x 2048, 1.89 at 16 x 2560; reference 1.92)
constexpr int LZ4_TAB_LZ4 = 1024 + 256;
constexpr int LZ4_TAB_ZSTD = 1024 + 256;
[[maybe_unused]] constexpr int LZ4_TAB_LANES = 2560; // (round 2's lane kernel: ablations/)
constexpr int LZ4_TAB_SHARED = 1536, LZ4_SH_LOG2 = 13; // lane parser with the group's shared table: 64 + 48 + 32 KiB of LDS

#include "lz4/lz4_lane_parse.inc"
#include "lz4/lz4_classify.inc"
#include "lz4/lz4_lanes_kernel.inc"
#include "lz4/lz4_stitch.inc"


} // namespace

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------

extern "C" size_t lthip_lz4_bound(size_t size) { return size > 0x7E000000u ? 0 : size + size / 255 + 16; }

static int upload_blocks(lthip_ctx* ctx, uint32_t block_count, const uint64_t* src_offsets, const uint32_t* src_sizes,
                         const uint64_t* dst_offsets, const uint32_t* dst_caps, uint32_t seg_bytes, uint32_t gunits, Lz4Block** d_blocks,
                         uint64_t* out_nseg, uint64_t* out_ngrp = nullptr, uint64_t* out_ncgrp = nullptr)
{
    std::vector<Lz4Block> hb(block_count);
    uint64_t nseg = 0, ngrp = 0, ncgrp = 0;
    for (uint32_t b = 0; b < block_count; ++b)
    {
        if (src_sizes[b] > 0x7E000000u)
            return lthip_fail(ctx, EINVAL, "lz4", "block larger than LZ4_MAX_INPUT_SIZE");
        hb[b].src_off = src_offsets[b];
        hb[b].dst_off = dst_offsets[b];
        hb[b].size = src_sizes[b];
        hb[b].dst_cap = dst_caps[b];
        hb[b].seg_base = (uint32_t)nseg;
        hb[b].nseg = seg_bytes ? (uint32_t)(((uint64_t)src_sizes[b] + seg_bytes - 1) / seg_bytes) : 0;
        hb[b].grp_base = (uint32_t)ngrp;
        hb[b].ngrp = (hb[b].nseg + gunits - 1) / gunits;
        ngrp += hb[b].ngrp;
        hb[b].cgrp_base = (uint32_t)ncgrp;
        hb[b].ncgrp = (hb[b].nseg + LZ4_G_BATCH - 1) / LZ4_G_BATCH;
        ncgrp += hb[b].ncgrp;
        nseg += hb[b].nseg;
    }
    if (nseg > 0x7FFFFFF0ull)
        return lthip_fail(ctx, EINVAL, "lz4", "too many segments in one batch");
    void* p;
    int err = lthip_scratch(ctx, S_LZ4_BLOCKS, sizeof(Lz4Block) * (size_t)block_count, &p);
    if (err)
        return err;
    if ((err = lthip_stage_upload(ctx, p, hb.data(), sizeof(Lz4Block) * (size_t)block_count, ctx->stream))) // no host stall
        return err;
    *d_blocks = (Lz4Block*)p;
    *out_nseg = nseg;
    if (out_ngrp)
        *out_ngrp = ngrp;
    if (out_ncgrp)
        *out_ncgrp = ncgrp;
    return 0;
}

// The parser of the match finder: "lanes" (default, MODE 1: 16 x 4 KiB units share a 64 KiB window, every lane parses its own
// 64-byte sub-unit; 144 KiB of LDS, one workgroup of 16 waves per CU) or "batch" (LTHIP_LZ4_PARSER=batch, MODE 0: the round-1
// parser, 8 x 4 KiB units per 32 KiB window, 52 KiB of LDS, 24 waves per CU).
#ifdef LTHIP_K5_PROF
extern "C" __attribute__((visibility("default"))) int lthip_k5_prof_dump(int reset)
{
    unsigned long long h[32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_k5_prof), sizeof(h)) != hipSuccess)
        return -1;
    static const char* names[16] = {"stage+clear+barrier", "prefetch issue", "pre-seed", "loop: probe", "loop: extend fwd/back", "loop: long+record",
                                    "cover scans", "emission", "unit end", "group barrier", "#probe iterations", "#extension rounds", "#hits", "#long matches", "", ""};
    unsigned long long tot = 0;
    for (int i = 0; i < 10; ++i)
        tot += h[i];
    for (int i = 0; i < 14; ++i)
        printf("k5prof %-24s %14llu%s\n", names[i], h[i], i < 10 ? (std::string("  ") + std::to_string(100.0 * (double)h[i] / (double)(tot ? tot : 1)).substr(0, 5) + " %").c_str() : "");
    if (reset)
    {
        memset(h, 0, sizeof(h));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k5_prof), h, sizeof(h));
    }
    return 0;
}
#endif

static bool lz4_lane_parser()
{
#ifdef LTHIP_ABLATIONS
    static const bool v = [] {
        const char* e = getenv("LTHIP_LZ4_PARSER"); // "batch": the round-1 batch parser for everything
        return !(e && strcmp(e, "batch") == 0);
    }();
    return v;
#else
    return true;
#endif
}

// the batch parser's geometry: classification pass (CLS 1) or the parser for everything (CLS 0)
template <int FMT, int CLS>
static int launch_segments(lthip_ctx* ctx, uint32_t groups, uint32_t SEG, const void* d_src, const Lz4Block* d_blocks, uint32_t block_count,
                           uint32_t g0, uint8_t* streams, Lz4Meta* meta, uint64_t* zrecs, uint8_t* spec_dst, uint32_t dbg, uint32_t ngroups,
                           uint32_t* worklist)
{
    constexpr int TAB = FMT == 1 ? LZ4_TAB_ZSTD : LZ4_TAB_LZ4;
    const size_t lds = (size_t)lz4_window_lds_bytes(LZ4_G_BATCH * SEG + 64u, false) + 16 + (size_t)LZ4_G_BATCH * TAB * 2;
    if (lds > 64u * 1024u && !ctx->k5_lds_enabled[FMT][CLS]) // (8 KiB units)
    {
        LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lz4_segments<LZ4_G_BATCH, TAB, FMT, CLS>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->k5_lds_enabled[FMT][CLS] = true;
    }
    hipLaunchKernelGGL((k_lz4_segments<LZ4_G_BATCH, TAB, FMT, CLS>), dim3(groups), dim3(64 * LZ4_G_BATCH), lds, ctx->stream, (const uint8_t*)d_src,
                       d_blocks, block_count, g0, SEG, streams, meta, zrecs, spec_dst, dbg, nullptr, ngroups, worklist);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

#ifdef LTHIP_ABLATIONS
// the lane parser inside the segments kernel (rounds 2-4: ablations/k_lz4_segments_modes.inc), persistent, one workgroup per CU
template <int TAB, int FMT, int SH = 0, int PV = 0>
static int launch_segments_lane_mode(lthip_ctx* ctx, uint32_t groups, uint32_t SEG, const void* d_src, const Lz4Block* d_blocks, uint32_t block_count,
                                     uint8_t* streams, Lz4Meta* meta, uint64_t* zrecs, uint8_t* spec_dst, uint32_t dbg, uint64_t* lane_recs,
                                     uint32_t ngroups, uint32_t* worklist)
{
    constexpr int G = LZ4_G_LANES;
    const size_t lds = (size_t)lz4_window_lds_bytes(G * SEG + 64u + (PV == 2 ? LZ4_LPAD : 0u), PV == 2) + 16 + (size_t)G * TAB * 2 + (SH ? (size_t)4 << SH : 0);
    static bool granted[64] = {}; // per device and instance: more than 64 KiB of dynamic LDS has to be granted explicitly
    if (ctx->device < 0 || ctx->device >= 64 || !granted[ctx->device])
    {
        LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lz4_segments_modes<G, TAB, FMT, 1, 0, SH, PV>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (ctx->device >= 0 && ctx->device < 64)
            granted[ctx->device] = true;
    }
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const uint32_t grid = groups < (uint32_t)ncu ? groups : (uint32_t)ncu;
    hipLaunchKernelGGL((k_lz4_segments_modes<G, TAB, FMT, 1, 0, SH, PV>), dim3(grid), dim3(64 * G), lds, ctx->stream, (const uint8_t*)d_src, d_blocks,
                       block_count, 0u, SEG, streams, meta, zrecs, spec_dst, dbg, lane_recs, ngroups, worklist);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
#endif

// The match finder of one batch (FMT 0: LZ4 streams, FMT 1: zstd sequences), two passes: the batch parser's geometry (8 units per group,
// 24 waves per CU) classifies every 32 KiB half-group with its four probe batches and skims the ones without redundancy on the spot
// (incompressible data never sees the slower one-workgroup-per-CU geometry); the 16-unit groups that hold redundancy are listed, the
// list becomes items of two half-groups (k_lz4_pair_halves), and the lane kernel (k_lz4_lanes2) is persistent over the items.
// `lanes` false (unit sizes other than 4 KiB): the batch parser for everything.
template <int FMT>
static int launch_match_finder(lthip_ctx* ctx, bool lanes, uint32_t SEG, const void* d_src, const Lz4Block* d_blocks, uint32_t block_count,
                               uint32_t g0, uint32_t g1, uint64_t ngrp, uint64_t ncgrp, uint8_t* streams, Lz4Meta* meta, uint64_t* zrecs,
                               uint8_t* spec_dst, uint32_t dbg, [[maybe_unused]] uint64_t* lane_recs)
{
    if (!lanes)
        return launch_segments<FMT, 0>(ctx, g1 - g0, SEG, d_src, d_blocks, block_count, g0, streams, meta, zrecs, spec_dst, dbg, 0, nullptr);
#ifdef LTHIP_ABLATIONS
    // LTHIP_LZ4_SHARED=0: round 2/3a's history (prefix maximum over the waves' 2560-entry tables) instead of the shared table
    LTHIP_ABLATION_ENV(env_shared, "LTHIP_LZ4_SHARED");
    const bool shared = env_shared.get() != 0;
    LTHIP_ABLATION_ENV(env_pv, "LTHIP_LZ4_PV"); // 0: the round-3 formulation of the lane parse (lz4_lane_parse), default: the round-4 one
    const bool pv2 = shared && env_pv.get() != 0;
    if (dbg & 16384u) // the lane kernel alone, with its own probe
        return pv2      ? launch_segments_lane_mode<LZ4_TAB_SHARED, FMT, LZ4_SH_LOG2, 2>(ctx, (uint32_t)ngrp, SEG, d_src, d_blocks, block_count, streams, meta, zrecs,
                                                                                        spec_dst, dbg, lane_recs, (uint32_t)ngrp, nullptr)
               : shared ? launch_segments_lane_mode<LZ4_TAB_SHARED, FMT, LZ4_SH_LOG2>(ctx, (uint32_t)ngrp, SEG, d_src, d_blocks, block_count, streams, meta, zrecs,
                                                                                     spec_dst, dbg, lane_recs, (uint32_t)ngrp, nullptr)
                        : launch_segments_lane_mode<LZ4_TAB_LANES, FMT>(ctx, (uint32_t)ngrp, SEG, d_src, d_blocks, block_count, streams, meta, zrecs, spec_dst,
                                                                        dbg, lane_recs, (uint32_t)ngrp, nullptr);
#endif
    void* wl;
    int err = lthip_scratch(ctx, S_LZ4_CLASSIFY, 4 * (11 * (size_t)ngrp + 16), &wl);
    if (err)
        return err;
    LTHIP_CHECK(ctx, hipMemsetAsync(wl, 0, 4 * (2 * (size_t)ngrp + 4), ctx->stream));
    if ((err = launch_segments<FMT, 1>(ctx, (uint32_t)ncgrp, SEG, d_src, d_blocks, block_count, 0, streams, meta, zrecs, spec_dst, dbg, (uint32_t)ngrp,
                                       (uint32_t*)wl)))
        return err;
#ifdef LTHIP_ABLATIONS
    LTHIP_ABLATION_ENV(env_halves, "LTHIP_LZ4_HALVES"); // 0: whole groups as the unit of work (a group's incompressible half idles eight waves)
    if (!pv2 || env_halves.get() == 0)
        return pv2      ? launch_segments_lane_mode<LZ4_TAB_SHARED, FMT, LZ4_SH_LOG2, 2>(ctx, (uint32_t)ngrp, SEG, d_src, d_blocks, block_count, streams, meta, zrecs,
                                                                                        spec_dst, dbg, lane_recs, (uint32_t)ngrp, (uint32_t*)wl)
               : shared ? launch_segments_lane_mode<LZ4_TAB_SHARED, FMT, LZ4_SH_LOG2>(ctx, (uint32_t)ngrp, SEG, d_src, d_blocks, block_count, streams, meta, zrecs,
                                                                                     spec_dst, dbg, lane_recs, (uint32_t)ngrp, (uint32_t*)wl)
                        : launch_segments_lane_mode<LZ4_TAB_LANES, FMT>(ctx, (uint32_t)ngrp, SEG, d_src, d_blocks, block_count, streams, meta, zrecs, spec_dst,
                                                                        dbg, lane_recs, (uint32_t)ngrp, (uint32_t*)wl);
#endif
    // the list of groups -> items of two half-groups (k_lz4_pair_halves), then the lane kernel over the items
    const uint32_t nthreads = 256, ng = (uint32_t)ngrp;
    // history halves: the zstd flavour at its "high" and "max" settings
    uint32_t hist = (FMT == 1 && (dbg & LZ4_DBG_Q_MAX)) ? 3u : (FMT == 1 && (dbg & LZ4_DBG_Q_HIGH)) ? 1u : 0u;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    LTHIP_ABLATION_ENV(env_far, "LTHIP_LZ4_FAR");
    const uint32_t farlog = env_far.get() >= 0 ? (uint32_t)env_far.get() : (10u | (12u << 8) | (8u << 16));
#ifdef LTHIP_ABLATIONS
    LTHIP_ABLATION_ENV(env_split, "LTHIP_LZ4_SPLITWG"); // experiment (round 5): half-items on two 8-wave workgroups per CU
    if (env_split.get() > 0 && hist == 0u)
    {
        hist = 2u;
        hipLaunchKernelGGL(k_lz4_pair_halves, dim3((ng + nthreads - 1) / nthreads), dim3(nthreads), 0, ctx->stream, (uint32_t*)wl, ng, 0u, d_blocks, block_count, hist);
        hipLaunchKernelGGL(k_lz4_pair_halves, dim3((2u * ng + nthreads) / nthreads), dim3(nthreads), 0, ctx->stream, (uint32_t*)wl, ng, 1u, d_blocks, block_count, hist);
        LTHIP_LAUNCH_CHECK(ctx);
        const size_t lds8 = (size_t)lz4_window_lds_bytes(8u * SEG + 64u + LZ4_LPAD + 32u, true) + 16 + (size_t)8 * LZ4_TAB_SHARED * 2 + ((size_t)2 << LZ4_SH_LOG2);
        static bool granted8[2] = {false, false};
        if (!granted8[FMT])
        {
            LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lz4_lanes2<FMT, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            granted8[FMT] = true;
        }
        hipLaunchKernelGGL((k_lz4_lanes2<FMT, 8>), dim3(2u * (uint32_t)ncu), dim3(64 * 8), lds8, ctx->stream, (const uint8_t*)d_src, d_blocks, block_count, SEG,
                           streams, meta, zrecs, spec_dst, dbg, ng, (uint32_t*)wl, farlog);
        LTHIP_LAUNCH_CHECK(ctx);
        return 0;
    }
#endif
    hipLaunchKernelGGL(k_lz4_pair_halves, dim3((ng + nthreads - 1) / nthreads), dim3(nthreads), 0, ctx->stream, (uint32_t*)wl, ng, 0u, d_blocks, block_count, hist);
    hipLaunchKernelGGL(k_lz4_pair_halves, dim3((ng / 2 + nthreads) / nthreads), dim3(nthreads), 0, ctx->stream, (uint32_t*)wl, ng, 1u, d_blocks, block_count, hist);
    LTHIP_LAUNCH_CHECK(ctx);
    const size_t lds = (size_t)lz4_window_lds_bytes(LZ4_G_LANES * SEG + 64u + LZ4_LPAD + 32u, true) + 16 + (size_t)LZ4_G_LANES * LZ4_TAB_SHARED * 2 +
                       ((size_t)4 << LZ4_SH_LOG2);
    if (!ctx->k5h_lds_enabled[FMT])
    {
        LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lz4_lanes2<FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->k5h_lds_enabled[FMT] = true;
    }
    // zstd flavour (farlog above): a match must be one byte longer from 2^a bytes away and two from 2^b, and c bytes long when it reaches
    // into a history half (a + 256 b + 65536 c = 10, 12, 8; a = b = 31 and c = 4: no rule; the ablation build reads LTHIP_LZ4_FAR)
    hipLaunchKernelGGL((k_lz4_lanes2<FMT>), dim3((uint32_t)ncu), dim3(64 * LZ4_G_LANES), lds, ctx->stream, (const uint8_t*)d_src, d_blocks,
                       block_count, SEG, streams, meta, zrecs, spec_dst, dbg, ng, (uint32_t*)wl, farlog);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

static int lz4_compress_batch(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                              const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                              uint32_t* d_out_sizes, int segment_log2);

// The per-unit scratch of one launch sequence is about as large as its input, so a call of any size is cut into batches of
// LTHIP_BATCH_BYTES (default 8 GiB, the size bench.py uses) that reuse the same scratch one after the other on the stream.
uint64_t lthip_codec_batch_bytes()
{
    static const uint64_t v = [] {
        const char* e = getenv("LTHIP_BATCH_BYTES");
        const uint64_t x = e ? strtoull(e, nullptr, 10) : 0;
        return x ? x : (8ull << 30);
    }();
    return v;
}

// Budget (MiB) of the arena the restore paths execute dependent payloads on ORIGINS in (4 bytes per byte of output of the payloads in
// flight; k_lz4_decode.hip, k_zstd.hip).  The payloads of a call go through it in rounds, one small launch per unit row and round: 512
// sliding-window LZ4 blocks of 8 MiB take 25.2 ms with 4 GiB (four rounds), 21.7 with 8, 20.1 with 16 (one round).  An explicit
// setting (LTHIP_ORIGIN_MIB, one of the product's documented switches) wins; otherwise a quarter of what the device has free when
// the question is first asked, between 4 and 16 GiB.
uint64_t lthip_origin_budget_mib()
{
    static LthipEnvInt env_org{"LTHIP_ORIGIN_MIB"};
    if (env_org.get() > 0)
        return (uint64_t)env_org.get();
    static const uint64_t v = [] {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
            return (uint64_t)4096;
        uint64_t mib = (uint64_t)(free_b >> 20) / 4u;
        mib = mib < 4096u ? 4096u : mib > 16384u ? 16384u : mib;
        return mib;
    }();
    return v;
}

extern "C" int lthip_lz4_compress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                                         const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets,
                                         const uint32_t* dst_caps, uint32_t* d_out_sizes, int segment_log2)
{
    if (!ctx || !d_out_sizes || (block_count && (!src_offsets || !src_sizes || !dst_offsets || !dst_caps || !d_dst)))
        return EINVAL;
    const uint64_t budget = lthip_codec_batch_bytes();
    for (uint32_t b0 = 0; b0 < block_count;)
    {
        uint64_t bytes = src_sizes[b0];
        uint32_t b1 = b0 + 1;
        while (b1 < block_count && bytes + src_sizes[b1] <= budget)
            bytes += src_sizes[b1++];
        const int err = lz4_compress_batch(ctx, d_src, b1 - b0, src_offsets + b0, src_sizes + b0, d_dst, dst_offsets + b0, dst_caps + b0,
                                           d_out_sizes + b0, segment_log2);
        if (err)
            return err;
        b0 = b1;
    }
    return 0;
}

static int lz4_compress_batch(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                              const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                              uint32_t* d_out_sizes, int segment_log2)
{
    if (segment_log2 == 0)
        segment_log2 = 12;
    if (segment_log2 < 10 || segment_log2 > 13) // units are multiples of 1024 bytes (pre-seed loop)
        return lthip_fail(ctx, EINVAL, "lz4", "segment_log2 (unit size) must be 0 (default = 12) or 10..13");
    const uint32_t SEG = 1u << segment_log2; // stitch unit; the match window is LZ4_G units
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    Lz4Block* d_blocks = nullptr;
    uint64_t nseg64 = 0, ngrp64 = 0;
    const bool lanes = lz4_lane_parser() && SEG == 4096u; // other unit sizes (tests, ablations) keep the batch parser's geometry
    const uint32_t GU = lanes ? LZ4_G_LANES : LZ4_G_BATCH;
    uint64_t ncgrp64 = 0;
    int err = upload_blocks(ctx, block_count, src_offsets, src_sizes, dst_offsets, dst_caps, SEG, GU, &d_blocks, &nseg64, &ngrp64, &ncgrp64);
    if (err)
        return err;
    const uint32_t nseg = (uint32_t)nseg64;
    void *meta, *plan, *bout, *streams, *runs, *lrecs = nullptr;
    if (lanes && (err = lthip_scratch(ctx, S_LZ4_LANE_RECS, (size_t)64 * LZ4_LANE_MAXREC * 8 * ((size_t)nseg + 1), &lrecs)))
        return err;
    if ((err = lthip_scratch(ctx, S_TABLES2, 4 * ((size_t)nseg + block_count + 1), &runs)))
        return err;
    if ((err = lthip_scratch(ctx, S_LZ4_META, sizeof(Lz4Meta) * ((size_t)nseg + 1), &meta)))
        return err;
    if ((err = lthip_scratch(ctx, S_LZ4_SEGS, sizeof(Lz4Plan) * ((size_t)nseg + 1), &plan)))
        return err;
    if ((err = lthip_scratch(ctx, S_MISC, sizeof(Lz4BlockOut) * (size_t)block_count, &bout)))
        return err;
    if ((err = lthip_scratch(ctx, S_LZ4_STREAM, (size_t)lz4_stream_stride(SEG) * ((size_t)nseg + 1), &streams)))
        return err;
    // Optional (LTHIP_LZ4_DBG bit 5): the match finder is latency / LDS bound, the stitch copy HBM bound, so the batch can
    // be cut into up to four slices by bytes with slice i's stitch on the context's second stream while slice i+1 is
    // parsed on the main one.  MEASURED SLOWER on MI355X (64 GiB random: 151.6 vs 137.6 ms per step: the copy's traffic
    // lengthens every probe's LDS-fill latency and both kernels lose more than the overlap wins), hence off by default.
    LTHIP_ABLATION_ENV(env_dbg, "LTHIP_LZ4_DBG"); // (ablation build: bits that switch parts of the match finder off or over; 0 in the product)
    const uint32_t dbg = env_dbg.get() > 0 ? (uint32_t)env_dbg.get() : 0u;
    uint64_t total_bytes = 0;
    for (uint32_t b = 0; b < block_count; ++b)
        total_bytes += src_sizes[b];
    const uint32_t want = (!(dbg & 32u) || lanes) ? 1u : (total_bytes >= (1ull << 30) ? 4u : (total_bytes >= (256ull << 20) ? 2u : 1u));
    std::vector<uint32_t> cut{0};
    {
        uint64_t acc = 0;
        for (uint32_t b = 0; b < block_count; ++b)
        {
            acc += src_sizes[b];
            if (cut.size() < want && acc * want >= total_bytes * cut.size() && b + 1 < block_count)
                cut.push_back(b + 1);
        }
        cut.push_back(block_count);
    }
    // per-block first segment / group (same arithmetic as upload_blocks)
    std::vector<uint32_t> grp_first(block_count + 1, 0);
    for (uint32_t b = 0; b < block_count; ++b)
        grp_first[b + 1] = grp_first[b] + (uint32_t)(((((uint64_t)src_sizes[b] + SEG - 1) / SEG) + GU - 1) / GU);
    const bool overlap = cut.size() > 2;
    void* worklist;
    if ((err = lthip_scratch(ctx, S_LZ4_WORKLIST, 4 * (2 * (size_t)ngrp64 + 1) * (cut.size() - 1), &worklist)))
        return err;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    // The stitch copy is persistent over the work list with a fixed stride, so its grid has to be RESIDENT at once: a multiple of what
    // the kernel's registers let a CU hold (72 VGPRs: 7 workgroups of 4 waves, not 8 -- with 8 per CU the eighth ran its share of the list
    // after the others were through: 23.9 instead of 18.1 ms on the compressible 64 GiB tree; twice that many, half the share each, evens
    // the tail: 17.8).  LTHIP_LZ4_STITCH_WGS = workgroups per CU.
    LTHIP_ABLATION_ENV(env_swg, "LTHIP_LZ4_STITCH_WGS");
    int stitch_wgs = 7;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&stitch_wgs, k_lz4_stitch_copy, K6_THREADS, 0) != hipSuccess || stitch_wgs < 1)
        stitch_wgs = 7;
    const uint32_t copy_grid = (uint32_t)ncu * (env_swg.get() > 0 ? (uint32_t)env_swg.get() : 2u * (uint32_t)stitch_wgs);
    hipStream_t s2 = ctx->stream;
    if (overlap)
    {
        if ((err = lthip_second_stream(ctx, &s2)))
            return err;
    }
    for (size_t i = 0; i + 1 < cut.size(); ++i)
    {
        const uint32_t b0 = cut[i], b1 = cut[i + 1];
        const uint32_t g0 = grp_first[b0], g1 = grp_first[b1];
        if (g1 > g0)
        {
            LaunchTimer t(ctx, LTHIP_K_LZ4_SEG);
            uint8_t* spec = (dbg & 64u) ? (uint8_t*)nullptr : (uint8_t*)d_dst;
            if ((err = launch_match_finder<0>(ctx, lanes, SEG, d_src, d_blocks, block_count, g0, g1, ngrp64, ncgrp64, (uint8_t*)streams, (Lz4Meta*)meta,
                                              nullptr, spec, dbg, (uint64_t*)lrecs)))
                return err;
        }
        if (overlap)
        {
            hipEvent_t e = lthip_sync_event(ctx);
            LTHIP_CHECK(ctx, hipEventRecord(e, ctx->stream));
            LTHIP_CHECK(ctx, hipStreamWaitEvent(s2, e, 0));
        }
        {
            LaunchTimer t(ctx, LTHIP_K_LZ4_STITCH, s2);
            uint32_t* wl = (uint32_t*)worklist + (size_t)i * (2 * (size_t)ngrp64 + 1); // one list per slice: count, group ids, their blocks
            LTHIP_CHECK(ctx, hipMemsetAsync(wl, 0, 4, s2));
            hipLaunchKernelGGL(k_lz4_stitch_scan, dim3(b1 - b0), dim3(64), 0, s2, d_blocks, b0, b1, SEG, (const Lz4Meta*)meta,
                               (Lz4Plan*)plan, (uint32_t*)runs, (Lz4BlockOut*)bout, d_out_sizes, (uint8_t*)d_dst, (dbg & 64u) ? 0u : 1u, wl, (uint32_t)ngrp64);
            if (g1 > g0)
                hipLaunchKernelGGL(k_lz4_stitch_copy, dim3(g1 - g0 < copy_grid ? g1 - g0 : copy_grid), dim3(K6_THREADS), 0, s2,
                                   (const uint8_t*)d_src, d_blocks, block_count, SEG, (const uint8_t*)streams, (const Lz4Meta*)meta,
                                   (const Lz4Plan*)plan, (const uint32_t*)runs, (const Lz4BlockOut*)bout, (uint8_t*)d_dst,
                                   (const uint32_t*)wl, GU, (uint32_t)ngrp64);
            if (i + 2 == cut.size())
                hipLaunchKernelGGL(k_lz4_empty_blocks, dim3((block_count + 255) / 256), dim3(256), 0, s2, d_blocks, block_count,
                                   (const Lz4BlockOut*)bout, (uint8_t*)d_dst);
            LTHIP_LAUNCH_CHECK(ctx);
        }
    }
    if (overlap)
    {
        hipEvent_t e = lthip_sync_event(ctx);
        LTHIP_CHECK(ctx, hipEventRecord(e, s2));
        LTHIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, e, 0));
    }
    return 0;
}

int lthip_launch_lz_sequences(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                              const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                              uint8_t** d_lits, uint64_t** d_recs, void** d_meta, uint32_t* unit_base, uint64_t* total_units, int quality)
{
    const uint32_t SEG = 4096u; // ZB_UNIT
    Lz4Block* d_blocks = nullptr;
    uint64_t nseg64 = 0, ngrp64 = 0;
    const bool lanes = lz4_lane_parser();
    uint64_t ncgrp64 = 0;
    int err = upload_blocks(ctx, block_count, src_offsets, src_sizes, dst_offsets, dst_caps, SEG, lanes ? LZ4_G_LANES : LZ4_G_BATCH, &d_blocks,
                            &nseg64, &ngrp64, &ncgrp64);
    if (err)
        return err;
    uint64_t nseg = 0;
    for (uint32_t b = 0; b < block_count; ++b)
    {
        unit_base[b] = (uint32_t)nseg;
        nseg += ((uint64_t)src_sizes[b] + SEG - 1) / SEG;
    }
    *total_units = nseg;
    void *lits, *recs, *meta;
    if ((err = lthip_scratch(ctx, S_Z_LITS, (size_t)SEG * (nseg + 1), &lits)))
        return err;
    if ((err = lthip_scratch(ctx, S_Z_RECS, (size_t)SEG * 2 * (nseg + 1), &recs)))
        return err;
    if ((err = lthip_scratch(ctx, S_LZ4_META, sizeof(Lz4Meta) * ((size_t)nseg + 1), &meta)))
        return err;
    void* lrecs = nullptr;
    if (lanes && (err = lthip_scratch(ctx, S_LZ4_LANE_RECS, (size_t)64 * LZ4_LANE_MAXREC * 8 * ((size_t)nseg + 1), &lrecs)))
        return err;
    if (nseg)
    {
        LaunchTimer t(ctx, LTHIP_K_LZ4_SEG);
        LTHIP_ABLATION_ENV(env_dbg, "LTHIP_LZ4_DBG");
        uint32_t dbg = env_dbg.get() > 0 ? (uint32_t)env_dbg.get() : 0u;
        // the parse the zstd setting asks for (LTHIP_ZSTD_Q_*): bit 15 = "high" (history halves: k_lz4_pair_halves), bit 31 = "max" (high +
        // the private table read again after the step's inserts); launch_match_finder and lz4_lane_parse2 read them
        if (quality >= 1)
            dbg |= LZ4_DBG_Q_HIGH;
        if (quality >= 2)
            dbg |= LZ4_DBG_Q_MAX;
        if ((err = launch_match_finder<1>(ctx, lanes, SEG, d_src, d_blocks, block_count, 0u, (uint32_t)ngrp64, ngrp64, ncgrp64, (uint8_t*)lits,
                                          (Lz4Meta*)meta, (uint64_t*)recs, (uint8_t*)d_dst, dbg, (uint64_t*)lrecs)))
            return err;
    }
    *d_lits = (uint8_t*)lits;
    *d_recs = (uint64_t*)recs;
    *d_meta = meta;
    return 0;
}
