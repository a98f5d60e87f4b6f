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

// ---------------------------------------------------------------------------------------------------
// K5, lane-sequential parse (MODE 1).  The batch parser above looks at 64 CONSECUTIVE positions per step and then has to choose
// among the hits -- a scalar walk that costs ~400 wave instructions per 64 positions on compressible data.  Here every lane owns
// a SUB-UNIT (unit / 64 bytes) and runs the reference's greedy loop on it (lz4.c:1019-1110: probe; on a hit extend, record,
// jump; else step one byte): 64 independent parsers in lock step, one probe per lane per iteration, no selection at all.
//   * the table is the wave's private one; positions of all lanes go into it.  Lanes are mapped to sub-units in REVERSE
//     (lane 63 = first sub-unit) so that when two lanes write one slot in the same instruction the surviving entry is the lower
//     position, which every later sub-unit can use (tools/lz4_lane_model.c: "lines" 8.4 -> 20.7, mixed 1.66 -> 1.72)
//   * a match may run past its lane's sub-unit (up to the unit's end); what it covers is dropped from the later lanes'
//     records afterwards (exclusive prefix maximum of the lanes' match ends) and lanes that are covered while a long match
//     is being extended stop parsing
//   * records {start, length, offset} go to a per-unit scratch area (8 per lane); after the parse three wave scans (cover,
//     previous kept end, output offset) place every lane's sequences and each lane writes its own bytes; literal runs longer
//     than 16 bytes are copied by the whole wave
// Results are a function of the data only (the table is private, the wave runs in lock step).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t LZ4_LANE_MAXREC = 8; // sequences a lane may record (tools/lz4_lane_model.c: 8 costs nothing, 6 does)

#ifdef LTHIP_K5_PROF /* debug build only (make prof, tools/k5_prof.sh): shader-clock cycles per phase of the lane parser, summed over all waves */
__device__ unsigned long long g_k5_prof[32];
struct K5Prof
{
    unsigned long long last, acc[16];
};
#define K5P_DECL K5Prof k5p; k5p.last = __builtin_readcyclecounter(); for (int i__ = 0; i__ < 16; ++i__) k5p.acc[i__] = 0;
#define K5P_ARG , K5Prof& k5p
#define K5P_PASS , k5p
#define K5P(i) do { const unsigned long long n__ = __builtin_readcyclecounter(); k5p.acc[i] += n__ - k5p.last; k5p.last = n__; } while (0)
#define K5P_COUNT(i, n) do { k5p.acc[i] += (n); } while (0)
#define K5P_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i__ = 0; i__ < 16; ++i__) atomicAdd(&g_k5_prof[i__], k5p.acc[i__]); } while (0)
#else
#define K5P_DECL
#define K5P_ARG
#define K5P_PASS
#define K5P(i) do { } while (0)
#define K5P_COUNT(i, n) do { } while (0)
#define K5P_FLUSH do { } while (0)
#endif

// value held by the lane that owns sub-unit `s` (s outside 0..63: `ident`)
__device__ __forceinline__ uint32_t sub_shfl(uint32_t x, int s, bool rev, uint32_t ident)
{
    const int src = rev ? 63 - s : s;
    const uint32_t v = __shfl(x, src & 63, 64);
    return (s < 0 || s > 63) ? ident : v;
}

#ifdef LTHIP_ABLATIONS
#include "ablations/k_lz4_lane_parse_r3.inc" // lz4_lane_parse: the round-3 formulation of the parse below (same payloads)
#endif

// Scans in SUB-UNIT order with the reverse mapping (sub-unit 63 - lane): a prefix over the sub-units is a SUFFIX over the lanes.  Round 5:
// with DPP moves -- four row shifts inside the rows of 16 (a lane takes the value n lanes above it; beyond the row: nothing), then the
// rows above a lane's row through three v_readlane -- instead of six ds_bpermute round trips with their index arithmetic (36 vector
// instructions and six LDS operations a scan; the three or four scans behind a unit's parse were 3 % of the kernel).
template <bool MAX>
__device__ __forceinline__ uint32_t lane_suffix_scan(uint32_t v, int lane)
{
    auto op = [](uint32_t a, uint32_t b) { return MAX ? (a > b ? a : b) : a + b; };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true)); // row_shl:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true)); // row_shl:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true)); // row_shl:4
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xf, 0xf, true)); // row_shl:8
    // a row's first lane holds the row: rows above mine
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16), t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32),
                   t3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t s1 = op(t2, t3), s0 = op(t1, s1);
    const uint32_t above = lane < 16 ? s0 : (lane < 32 ? s1 : (lane < 48 ? t3 : 0u));
    return op(v, above);
}
// the value of the sub-unit before mine (the lane above; the first sub-unit: 0)
__device__ __forceinline__ uint32_t lane_from_sub_before(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); // wave_shl:1
}

// ---------------------------------------------------------------------------------------------------
// K5, lane-sequential parse, second formulation (round 4; PV = 2 of k_lz4_segments).  The parse is the one above -- same probe rule,
// same candidates, same extension limits, same cover rule: the payloads are byte-identical -- restated around what the per-phase
// cycle counters said it costs (profiles/r04_k5_prof.txt: a probe step took 1500 cycles for ~100 instructions):
//   * PROBE in three LDS round trips instead of four: the private and the shared candidate's bytes are read by ONE pair of
//     unconditional reads (an invalid candidate reads the lane's own position and is masked) -- the compiler had serialised the two
//     predicated reads, each with its own s_waitcnt.  (Round 4 also kept the lane's next bytes in a three-dword register window that
//     slid with p, for two round trips; round 5 took it out again: the kernel had become bound by the number of instructions it
//     issues, and the window's bookkeeping was a fifth of the probe round's.)
//   * EXTENSION in one round trip: the 28 bytes p-8 .. p+20 of both sides are eight aligned dwords each (the window is staged 16
//     bytes into LDS so that "8 bytes before position 0" is a legal address), read together; the backward count and the first 16
//     forward bytes come out of the same registers.  Matches of 20 bytes and more take a second 16-byte round, then the wave.
//   * RECORDS in registers (8 x {start | length << 16, offset}) instead of a global scratch area of 4 KiB per unit that the cover
//     scans, the size pass and the emission each read back through the memory system (1 B/B written and read on "tokens").
// Round 5 (tools/isa_blocks.py): the same parse, byte for byte, in a quarter fewer instructions -- vector AND
// scalar, which issue at the same aggregate rate.  The rules that came out of it: a ballot wants ONE compare (a combination of
// conditions costs a select and a second compare); wave-uniform state wants the scalar unit; switches of experiments want to be
// compile-time zeros in the product; code that every wave walks anyway wants | and & and selects instead of || / && / else-if (an
// exec-mask region is three scalar instructions), code that whole waves usually skip wants the branch.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t LZ4_LPAD = 16; // PV 2: LDS byte offset of the staged window
constexpr uint32_t LZ4_QUIET = 3;  // lane parser: probe rounds without a hit at an unaligned position before the one-byte steps stop
                                   // (profiles/r05_adaptive_stepping.txt: 6 / 4 / 3 / 2 / 1 rounds: match finder 161.6 / 160.3 / 159.0 / 158.1 /
                                   // 156.4 ms per 64 GiB of "mixed", word-soup text 2.0284 / 2.0284 / 2.0283 / 2.0282 / 2.0249; fixed rule 169.9, 2.0284)
constexpr uint32_t LZ4_DBG_Q_HIGH = 1u << 15, LZ4_DBG_Q_MAX = 1u << 31; // quality bits of `dbg` (zstd settings, lthip_launch_lz_sequences)

// slot of a private table of TAB entries for the (multiplied) hash `prod`.  PV 0: mulhi(prod, TAB), a quarter-rate 32-bit multiply;
// PV 2: the upper 16 bits of prod times TAB, a 24-bit multiply and a shift (TAB < 2^16) -- another function of the same bits, so the two
// formulations fill their tables differently (ratios agree to the fourth digit)
template <int TAB, int PV>
__device__ __forceinline__ uint32_t lz4_tab_slot(uint32_t prod)
{
    if constexpr (PV == 2)
        return __umul24(prod >> 16, (uint32_t)TAB) >> 16;
    else
        return __umulhi(prod, (uint32_t)TAB);
}

// the first nonzero byte of four XOR words (16: none) without a branch: as an if / else-if chain every level is an exec-mask region
// of its own, three scalar instructions each, and all levels are walked anyway as soon as one lane of the wave gets there
__device__ __forceinline__ uint32_t lz4_first_diff16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    const uint64_t lo = (uint64_t)x0 | ((uint64_t)x1 << 32), hi = (uint64_t)x2 | ((uint64_t)x3 << 32);
    const uint32_t nlo = (uint32_t)__builtin_ctzll(lo | (1ull << 63)) >> 3, nhi = 8u + ((uint32_t)__builtin_ctzll(hi | (1ull << 63)) >> 3);
    return lo ? nlo : (hi ? nhi : 16u);
}
// equal leading bytes (0..16) of the 16 bytes at LDS byte addresses qa and qb (any alignment): five aligned dwords per side
template <bool PAD>
__device__ __forceinline__ uint32_t lds_cmp16(const uint32_t* sdata, uint32_t qa, uint32_t qb)
{
    const uint32_t wa = qa >> 2, wb = qb >> 2, da = qa, db = qb; // (v_alignbyte_b32 takes the two low bits)
    uint32_t ra[4], rb[4];
    lds_run<PAD, 4>(sdata, wa, ra);
    lds_run<PAD, 4>(sdata, wb, rb);
    const uint32_t a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3], a4 = lds_dw<PAD>(sdata, wa + 4u);
    const uint32_t b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3], b4 = lds_dw<PAD>(sdata, wb + 4u);
    const uint32_t x0 = __builtin_amdgcn_alignbyte(a1, a0, da) ^ __builtin_amdgcn_alignbyte(b1, b0, db);
    const uint32_t x1 = __builtin_amdgcn_alignbyte(a2, a1, da) ^ __builtin_amdgcn_alignbyte(b2, b1, db);
    const uint32_t x2 = __builtin_amdgcn_alignbyte(a3, a2, da) ^ __builtin_amdgcn_alignbyte(b3, b2, db);
    const uint32_t x3 = __builtin_amdgcn_alignbyte(a4, a3, da) ^ __builtin_amdgcn_alignbyte(b4, b3, db);
    return lz4_first_diff16(x0, x1, x2, x3);
}

template <int TAB, int FMT, int SH>
__device__ __forceinline__ void lz4_lane_parse2(const uint32_t* sdata, uint32_t head /* incl. LZ4_LPAD */, uint16_t* tab, const uint32_t* shr,
                                                uint32_t sh_base, int lane, uint32_t my_start, uint32_t my_len, int32_t start_limit,
                                                uint32_t end_limit, uint32_t sub, uint8_t* __restrict__ out, uint64_t* __restrict__ zrecs,
                                                Lz4Seq& st, uint32_t dbg K5P_ARG, uint32_t lo_bound = 0u, uint32_t sh_shift = 32u - (uint32_t)SH,
                                                uint32_t sh_off = 0u, uint32_t not_private = 0xFFFFFFFFu, uint32_t far1 = 0xFFFFFFFFu,
                                                uint32_t far2 = 0xFFFFFFFFu, uint32_t hist = 0u, uint32_t far3 = 4u)
{
    // hist != 0: the window below position `hist` is HISTORY (k_lz4_pair_halves): the shared table's lower half holds, per key, its LATEST
    // position there (stored inverted, so that the table's minimum is the nearest one) -- a third candidate, asked last; a match into the
    // history must have far3 bytes

    // not_private: a position the PRIVATE table never offers (an unlinked half's first position: in the window's lower half that is
    // position 0, which the table cannot tell from "empty", so the upper half must not have it either -- what a half compresses to must
    // not depend on which slot of the window it was given)
    // lo_bound: candidates below this window position are not this unit's history (two unrelated half-groups share the window, see
    // k_lz4_lanes2); sh_shift / sh_off: the part of the shared table that is this half's
    static_assert(SH != 0, "the second formulation is the shared-table parser's");
    static_assert(LZ4_LANE_MAXREC == 8, "records are eight register pairs");
    constexpr bool PAD = true; // the window is the padded one (lds_dw)
    constexpr bool rev = true;
    const int sidx = 63 - lane; // my sub-unit (reverse mapping: the lowest position survives a same-instruction write conflict)
    const uint32_t unit_end = my_start + my_len;
    const uint32_t s0 = my_start + (uint32_t)sidx * sub;
    const uint32_t lend = s0 + sub < unit_end ? s0 + sub : unit_end;
    uint32_t p = s0, anchor = s0, nrec = 0, last_end = 0, nmiss = 0;
    // LTHIP_LZ4_DBG bit 13, "deep": every byte position is probed and, where both tables' candidates verify, the longer match wins --
    // round 4's first "high" setting; with the history halves it measures WORSE than without on every synthetic kind (and a third slower)
    // the experiments' switches (LTHIP_LZ4_DBG) exist in the ablation build; in the product they are compile-time zeros, not
    // wave-uniform branches and selects in the probe loop
#ifdef LTHIP_ABLATIONS
    const uint32_t xdbg = dbg;
#else
    constexpr uint32_t xdbg = 0u;
#endif
    const bool q_high = (xdbg & 8192u) != 0u, q_max = FMT == 1 && (dbg & LZ4_DBG_Q_MAX) != 0u;
    const uint32_t dense = q_high ? 0xFFFFu : 4u >> ((xdbg >> 29) & 3u);
    // Round 5, ADAPTIVE miss stepping.  After a hit (and at a sub-unit's start) a lane steps `dense` single bytes before it probes
    // aligned dwords only -- history enters the tables at aligned dwords, so a probe at an unaligned position is what finds a repeat
    // whose distance is not a multiple of four.  Data made of aligned structures (records, tables, tokens: what compresses in an asset
    // store) never answers such a probe: on bench.py's compressible tree the one-byte steps were 19 ms of the match finder's 166 and
    // bought nothing (profiles/r05_lane_dense_sweep.txt: none of them 314 GB/s at ratio 2.011, four of them 288 at 2.000; on word-soup
    // text none of them costs 5 % of the ratio, tools/text_ratio_probe.py).  So the WAVE
    // keeps count: after LZ4_QUIET probe rounds without a single hit at an unaligned position the one-byte steps stop; the first such
    // hit -- the probe behind a match's end stands wherever the match ended -- brings them back.  Wave-uniform, from ballots: a function
    // of the unit's data alone.  (LTHIP_LZ4_DBG bit 16, or an explicit step count in bits 29-30: the fixed rule.)
    const bool adaptive = !q_high && !(xdbg & 65536u) && ((xdbg >> 29) & 3u) == 0u;
    const uint32_t quiet_rounds = (xdbg >> 17) & 7u ? (xdbg >> 17) & 7u : LZ4_QUIET; // (bits 17-19: the sweep of profiles/r05_adaptive_stepping.txt)
    // (the state: bit k of qhist = an unaligned hit k probe rounds ago; bit 0 also stands for the unit's start)
    uint32_t qhist = 1u;
    bool dense_now = true;
    // lanes that must hold a hit before the wave turns to the extension: 4 (round 4, VALU bound: 2 GiB mixed 4.25 ms at 8, 4.12 at 4,
    // 4.16 / 4.37 / 4.76 at 16 / 32 / 48; tokens 5.97 / 5.93 / 6.03 / 6.80 / 7.78 -- waiting lanes are idle lanes)
    // (round 5, with the probe round at half its instructions: mixed 147.2 / 146.2 / 145.6 / 145.0 / 145.4 / 148.4 ms of match finder per
    // 64 GiB at 4 / 6 / 8 / 12 / 16 / 24, tokens 165.8 / 163.4 / 162.0 / 160.5 / 161.3 / 169.5 -- tools/wait_for_sweep.sh: 12)
    const uint32_t wait_for = (xdbg >> 20) & 63u ? (xdbg >> 20) & 63u : 12u;
    uint32_t rsl[8], roff[8]; // records: start | length << 16, offset
    uint32_t cand2 = 0xFFFFFFFFu; // "high": the other verified candidate of the probe (0xFFFFFFFF: none)
#pragma unroll
    for (int k = 0; k < 8; ++k)
        rsl[k] = roff[k] = 0u;
    uint32_t cand = 0u;
    // The loop's conditions, one compare each (a ballot of a compare IS the compare; a ballot of a combination of conditions costs a
    // select and a second compare -- and the kernel is bound by the number of vector instructions it issues): a lane probes while
    // p < plim (the end of its sub-unit, the unit's last start; 0 once its records are full), and a hit that waits for its extension
    // sets p's top bit -- which also takes the lane out of p < plim.
    constexpr uint32_t PEND = 0x80000000u;
    const uint32_t headm1 = head - 1u;
    const uint32_t stop0 = start_limit < 0 ? 0u : ((uint32_t)start_limit + 1u < lend ? (uint32_t)start_limit + 1u : lend);
    uint32_t plim = (xdbg & 2048u) ? 0u : stop0;
    for (;;)
    {
        // ---- probe rounds, until enough hits wait (or nobody can probe any more) ----
        uint64_t pm = 0ull;
        for (;;)
        {
            const bool act = p < plim;
            const uint64_t am = __builtin_amdgcn_ballot_w64(act);
            if (am == 0ull)
                break;
            const uint32_t x = p + head; // (the LDS byte address of position p)
            const uint64_t unm = __builtin_amdgcn_ballot_w64((x & 3u) != 0u); // lanes that stand at an unaligned position
            if (am)
            {
                if (act)
                {
                    // the four bytes at p: two aligned dwords, read where they are needed (round 4 kept the lane's next dwords in a register
                    // window that slid with p to save this round trip; since the kernel is bound by the vector instructions it issues, the
                    // window's bookkeeping -- reload after every jump, shift at every crossing, and the copies the three versions of
                    // every register cost at the loop's joins -- was 20 of a probe round's 110 instructions, the round trip is hidden)
                    uint32_t d2[2];
                    lds_run<PAD, 2>(sdata, x >> 2, d2);
                    const uint32_t v = __builtin_amdgcn_alignbyte(d2[1], d2[0], x);
                    const uint32_t prod = v * 2654435761u;
                    const uint32_t h = lz4_tab_slot<TAB, 2>(prod);
                    uint32_t c = tab[h];
                    const uint32_t c2 = shr[(prod >> sh_shift) + sh_off] - sh_base; // (another group's entry: far above any position)
                    tab[h] = (uint16_t)p;
                    if ((xdbg & (1u << 28)) || q_max) // "max": the slot is read again after the step's inserts (entries of lanes in phase)
                    {
                        uint32_t hr = h;
                        asm volatile("" : "+v"(hr));
                        const uint32_t fresh = tab[hr];
                        if (fresh < p)
                            c = fresh;
                    }
                    // both candidates' bytes in one round trip (an invalid one reads my own position and is masked)
                    const bool v1 = c < p && c >= lo_bound && c != not_private, v2 = c2 < p && c2 >= lo_bound;
                    const uint32_t r1 = lds_read32x<PAD>(sdata, (v1 ? c : p) + head);
                    const uint32_t r2 = lds_read32x<PAD>(sdata, (v2 ? c2 : p) + head);
                    const bool h1 = v1 && r1 == v, h2 = v2 && r2 == v;
                    bool h3 = false;
                    uint32_t c3 = 0u;
                    if (FMT == 1 && hist)
                    {
                        const uint32_t e3 = shr[prod >> sh_shift] - sh_base; // (another item's entry: above 0xFFFF)
                        c3 = 0xFFFFu - e3;
                        const bool v3 = e3 <= 0xFFFFu;
                        h3 = v3 && lds_read32x<PAD>(sdata, (v3 ? c3 : p) + head) == v;
                    }
                    if (h1 || h2 || h3)
                    {
                        p |= PEND;
                        cand = h1 ? c : (h2 ? c2 : c3); // the private table's (the nearest one) first, the history's last
                        cand2 = (q_high && h1 && h2 && c != c2) ? c2 : 0xFFFFFFFFu;
                    }
                    else
                    {
                        p = ((dense_now && nmiss < dense) ? x : (x | 3u)) - headm1;
                        ++nmiss;
                    }
                }
            }
            K5P(3);
            K5P_COUNT(10, 1);
            pm = __builtin_amdgcn_ballot_w64((int32_t)p < 0);
            if (adaptive) // (scalar: the hits of this round are the waiting lanes that probed)
            {
                const uint64_t t = unm & pm & am;
                uint32_t bit; // t != 0 as 0 / 1 (written out: the compiler takes the truth value through a vector register)
                asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(bit) : "s"(t) : "scc");
                qhist = (qhist << 1) | bit;
                dense_now = (qhist & ((1u << quiet_rounds) - 1u)) != 0u;
            }
            if ((uint32_t)__builtin_popcountll(pm) >= wait_for)
                break; // (fewer: somebody may still probe -- let the hits pile up)
        }
        pm = __builtin_amdgcn_ballot_w64((int32_t)p < 0);
        if (pm == 0ull)
            break; // nobody probes, nobody waits: the unit is parsed
        bool ok = (int32_t)p < 0;
        p &= ~PEND;
        K5P_COUNT(11, 1);
        K5P_COUNT(12, (unsigned)__builtin_popcountll(pm));
        // ---- one round trip: 8 bytes backwards and the first 16 bytes forwards of every hit ----
        uint32_t mlen = 0, nbk = 0;
        bool grow = false;
        const uint32_t maxlen = ok ? end_limit - p : 0u; // p <= start_limit: at least 4
        if (ok)
        {
            const uint32_t qo = p + head - 8u, qc = cand + head - 8u;
            const uint32_t bo = qo >> 2, bc = qc >> 2, dlo = qo, dlc = qc; // (v_alignbyte_b32 takes the two low bits)
            uint32_t Do[8], Dc[8];
            lds_run<PAD, 4>(sdata, bo, Do);
            lds_run<PAD, 4>(sdata, bo + 4u, Do + 4);
            lds_run<PAD, 4>(sdata, bc, Dc);
            lds_run<PAD, 4>(sdata, bc + 4u, Dc + 4);
            uint32_t X[7];
#pragma unroll
            for (int k = 0; k < 7; ++k)
                X[k] = __builtin_amdgcn_alignbyte(Do[k + 1], Do[k], dlo) ^ __builtin_amdgcn_alignbyte(Dc[k + 1], Dc[k], dlc);
            const uint32_t add = lz4_first_diff16(X[3], X[4], X[5], X[6]); // bytes p + 4 .. p + 20
            mlen = 4u + add;
            grow = add == 16u;
            if (mlen >= maxlen)
            {
                mlen = maxlen;
                grow = false;
            }
            {
                // (unconditional: the distance to the anchor bounds it -- 0 at the anchor --, a candidate too close to the window's
                // lower bound takes none)
                const uint64_t bw = ((uint64_t)X[1] << 32) | (uint64_t)X[0];
                nbk = (uint32_t)__builtin_clzll(bw | 1ull) >> 3;
                nbk = bw ? nbk : 8u;
                nbk = nbk < p - anchor ? nbk : p - anchor;
                nbk = cand >= lo_bound + 8u ? nbk : 0u;
            }
        }
        // ---- "high": where the probe verified BOTH candidates, the other one's first 16 bytes too; the longer match wins (the nearer on a
        // tie).  One more round trip for the lanes concerned. ----
        if (q_high && __builtin_amdgcn_ballot_w64(ok && cand2 != 0xFFFFFFFFu))
        {
            if (ok && cand2 != 0xFFFFFFFFu)
            {
                uint32_t a2 = 4u + lds_cmp16<PAD>(sdata, p + 4u + head, cand2 + 4u + head);
                bool g2 = a2 == 20u;
                if (a2 >= maxlen)
                {
                    a2 = maxlen;
                    g2 = false;
                }
                if (a2 > mlen)
                {
                    // the backward count belongs to the candidate: redo it for the new one
                    nbk = 0;
                    if (cand2 >= lo_bound + 8u && p - anchor != 0u)
                    {
                        const uint32_t x1 = lds_read32x<PAD>(sdata, p - 4u + head) ^ lds_read32x<PAD>(sdata, cand2 - 4u + head);
                        const uint32_t x0 = lds_read32x<PAD>(sdata, p - 8u + head) ^ lds_read32x<PAD>(sdata, cand2 - 8u + head);
                        nbk = x1 ? (uint32_t)__builtin_clz(x1) >> 3 : (x0 ? 4u + ((uint32_t)__builtin_clz(x0) >> 3) : 8u);
                        nbk = nbk < p - anchor ? nbk : p - anchor;
                    }
                    cand = cand2;
                    mlen = a2;
                    grow = g2;
                }
            }
        }
        // ---- matches of 20 bytes and more: one more 16-byte round of their own (at most 36 bytes), then the whole wave ----
        if (grow)
        {
            const uint32_t add = lds_cmp16<PAD>(sdata, p + mlen + head, cand + mlen + head);
            mlen += add;
            if (add != 16u)
                grow = false;
            if (mlen >= maxlen)
            {
                mlen = maxlen;
                grow = false;
            }
        }
        uint64_t longs = __builtin_amdgcn_ballot_w64(grow);
        K5P(4);
        while (longs)
        {
            K5P_COUNT(13, 1);
            const int f = rev ? 63 - __builtin_clzll(longs) : __builtin_ctzll(longs);
            longs &= ~(1ull << f);
            const uint32_t pf = __builtin_amdgcn_readlane(p, f), cf = __builtin_amdgcn_readlane(cand, f);
            uint32_t ml = __builtin_amdgcn_readlane(mlen, f);
            for (;;)
            {
                const uint32_t i = pf + ml + 4u * (uint32_t)lane;
                uint32_t cnt = 0; // equal bytes of my four, as far as the unit goes
                if (i < end_limit)
                {
                    const uint32_t x = lds_read32x<PAD>(sdata, i + head) ^ lds_read32x<PAD>(sdata, cf + ml + 4u * (uint32_t)lane + head);
                    const uint32_t lim = end_limit - i < 4u ? end_limit - i : 4u;
                    cnt = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
                    cnt = cnt < lim ? cnt : lim;
                }
                const uint64_t diff = __builtin_amdgcn_ballot_w64(cnt < 4u);
                if (diff)
                {
                    const int g = __builtin_ctzll(diff);
                    ml += 4u * (uint32_t)g + __builtin_amdgcn_readlane(cnt, g);
                    break;
                }
                ml += 256u;
            }
            if (lane == f)
                mlen = ml;
            const uint32_t cov = pf + ml;
            const bool cv = p - (pf + 1u) < ml - 1u; // pf < p < cov, one compare (lane f itself stands at pf; waiting lanes' p carries no flag here)
            if (cv)
            {
                p = cov;
                anchor = anchor > cov ? anchor : cov;
                ok = false;
            }
            longs &= ~__builtin_amdgcn_ballot_w64(cv);
        }
        // the zstd flavour prices a match by its distance (the offset's bits are written out): four or five bytes from far away cost more
        // than the literals they replace -- such a hit counts as a miss (T1, T2: LTHIP_LZ4_FAR, measured in profiles/r04_zstd_ratio_table.txt)
        if (FMT == 1 && ok && mlen + nbk < (cand < hist ? far3 : 4u + (p - cand >= far1 ? 1u : 0u) + (p - cand >= far2 ? 1u : 0u)))
        {
            ok = false;
            p += 1u;
            ++nmiss;
        }
        else if (ok)
        {
            const uint32_t s = p - nbk, len = mlen + nbk;
            const uint32_t sl = s | (len << 16), of = p - cand;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (nrec == (uint32_t)k)
                {
                    rsl[k] = sl;
                    roff[k] = of;
                }
            ++nrec;
            if (nrec == LZ4_LANE_MAXREC)
                plim = 0u;
            p = s + len;
            anchor = p;
            last_end = p;
            nmiss = 0;
        }
        // (a waiting hit is either recorded or covered: nothing else to do)
        K5P(5);
    }

    // ---- what earlier sub-units' matches cover is dropped: exclusive prefix maximum of the match ends, in sub-unit order ----
    static_assert(rev, "the scans below run in lane order from the top");
    const uint32_t incl = lane_suffix_scan<true>(last_end, lane);
    const uint32_t cover = lane_from_sub_before(incl);
    uint32_t k0 = 0; // my first record that starts at or after the cover (starts ascend)
#pragma unroll
    for (int k = 0; k < 8; ++k)
        k0 += ((uint32_t)k < nrec && (rsl[k] & 0xFFFFu) < cover) ? 1u : 0u;
    const bool have = k0 < nrec;
    uint32_t first_start_v = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((uint32_t)k == k0)
            first_start_v = rsl[k] & 0xFFFFu;
    if (!have)
        first_start_v = 0u;
    // previous kept end = where the literals of my first kept sequence begin
    const uint32_t kincl = lane_suffix_scan<true>(have ? last_end : 0u, lane);
    uint32_t prev0 = lane_from_sub_before(kincl);
    prev0 = prev0 > my_start ? prev0 : my_start;
    // sizes
    const uint32_t cnt = have ? nrec - k0 : 0u;
    uint32_t bytes = 0, nlit = 0;
    {
        uint32_t prev = prev0;
#pragma unroll
        for (int k = 0; k < 8; ++k) // (a branch per slot on purpose: the last slots are rarely anybody's, and then the whole wave skips them)
        {
            const bool on = (uint32_t)k - k0 < cnt;
            const uint32_t s = rsl[k] & 0xFFFFu, len = rsl[k] >> 16;
            const uint32_t lit = s - prev;
            if (on)
            {
                bytes += 3u + lz4_len_bytes16(lit) + lit + lz4_len_bytes16(len - 4u);
                nlit += lit;
                prev = s + len;
            }
        }
    }
    const uint32_t a_incl = lane_suffix_scan<false>(FMT == 1 ? nlit : bytes, lane);
    const uint32_t c_incl = FMT == 1 ? lane_suffix_scan<false>(cnt, lane) : 0u;
    const uint32_t a_total = (uint32_t)__builtin_amdgcn_readlane((int)a_incl, 0); // (the last sub-unit is lane 0's)
    uint32_t o_pos = a_incl - (FMT == 1 ? nlit : bytes);
    uint32_t q_pos = FMT == 1 ? c_incl - cnt : 0u;
    const uint32_t last_kept_end = (uint32_t)__builtin_amdgcn_readlane((int)kincl, 0);

    K5P(6);
    // ---- emission: every lane writes its own sequences; literal runs above 16 bytes are copied by the whole wave ----
    if (!(xdbg & 1024u))
    {
        uint32_t prev = prev0;
        const uint64_t anyrec = __builtin_amdgcn_ballot_w64(cnt != 0u);
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
            const bool on = (uint32_t)k - k0 < cnt; // k0 <= k < nrec, one compare (cnt = nrec - k0, or 0)
            if (anyrec == 0ull || __builtin_amdgcn_ballot_w64(on) == 0ull)
                continue;
            uint32_t lit = 0, lit_src = 0, lit_dst = 0;
            if (on)
            {
                const uint32_t s = rsl[k] & 0xFFFFu, len = rsl[k] >> 16, off = roff[k];
                lit = s - prev;
                lit_src = prev;
                if constexpr (FMT == 1)
                {
                    lit_dst = o_pos;
                    zrecs[q_pos] = (uint64_t)lit | ((uint64_t)len << 16) | ((uint64_t)off << 32);
                    ++q_pos;
                    o_pos += lit;
                }
                else
                {
                    uint8_t* o = out + o_pos;
                    const uint32_t mcode = len - 4u;
                    o[0] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u));
                    uint32_t idx = 1u;
                    if (lit >= 15u)
                    {
                        uint32_t rem = lit - 15u;
                        for (; rem >= 255u; rem -= 255u)
                            o[idx++] = 255;
                        o[idx++] = (uint8_t)rem;
                    }
                    lit_dst = o_pos + idx;
                    idx += lit;
                    o[idx] = (uint8_t)off;
                    o[idx + 1u] = (uint8_t)(off >> 8);
                    idx += 2u;
                    if (mcode >= 15u)
                    {
                        uint32_t rem = mcode - 15u;
                        for (; rem >= 255u; rem -= 255u)
                            o[idx++] = 255;
                        o[idx++] = (uint8_t)rem;
                    }
                    o_pos += idx;
                }
                prev = s + len;
                if (lit <= 16u)
                {
                    // four bytes per store where four are left (the destination has any alignment: global memory takes unaligned
                    // dwords), single bytes for the rest -- never a byte beyond the run: the next byte is another lane's
                    typedef uint32_t u32_a1 __attribute__((aligned(1)));
                    uint8_t* o = out + lit_dst;
                    uint32_t j = 0;
                    for (; j + 4u <= lit; j += 4u)
                        *reinterpret_cast<u32_a1*>(o + j) = lds_read32x<PAD>(sdata, lit_src + j + head);
                    for (; j < lit; ++j)
                        o[j] = (uint8_t)lds_byte<PAD>(sdata, lit_src + j + head);
                }
            }
            uint64_t big = __builtin_amdgcn_ballot_w64(lit > 16u); // (lit is 0 where the lane has no k-th sequence)
            while (big)
            {
                const int f = __builtin_ctzll(big);
                big &= big - 1ull;
                wave_copy_lds_to_global<PAD>(out + __builtin_amdgcn_readlane(lit_dst, f), sdata, __builtin_amdgcn_readlane(lit_src, f) + head,
                                        __builtin_amdgcn_readlane(lit, f), lane);
            }
        }
    }
    K5P(7);
    // ---- the unit's result, as the batch parser leaves it ----
    const uint64_t hm = __builtin_amdgcn_ballot_w64(have);
    st.have_first = hm != 0ull;
    st.anchor = hm ? last_kept_end : my_start;
    if constexpr (FMT == 1)
    {
        st.op = a_total; // literal bytes so far
        st.nseq = (uint32_t)__builtin_amdgcn_readlane((int)c_incl, 0);
    }
    else
    {
        st.op = a_total;
        if (hm)
        {
            const int f = 63 - __builtin_clzll(hm); // the lane of the first sub-unit with a sequence
            const uint32_t first_start = __builtin_amdgcn_readlane(first_start_v, f);
            st.first_lit = first_start - my_start;
            st.first_hdr = 1u + lz4_len_bytes(st.first_lit);
        }
    }
}

// TAB = entries of a wave's private table (any multiple of 8: the index is mulhi(hash, TAB), not a mask); G = units per window
// group = waves per workgroup: the batch parser, 8 x 4 KiB units per 32 KiB window group, THREE workgroups per CU.
// CLS = classification pass of the two-pass scheme: groups without redundancy are skimmed here and now (the fast geometry: 24 waves
// per CU), groups with redundancy are only NOTED -- the 16-unit group they belong to goes onto `worklist` -- and left to the lane
// parser (k_lz4_lanes2), which then runs over that list.  CLS 0 = the batch parser for everything: unit sizes other than 4 KiB.
// (The lane parser used to be MODE 1 of this template: ablations/k_lz4_segments_modes.inc.)
template <int G, int TAB, int FMT, int CLS = 0>
__global__ __launch_bounds__(64 * G, 6) void k_lz4_segments(const uint8_t* __restrict__ src, const Lz4Block* __restrict__ blocks,
                                                             uint32_t nblocks, uint32_t grp0, uint32_t sub_bytes,
                                                             uint8_t* __restrict__ streams, Lz4Meta* __restrict__ meta,
                                                             uint64_t* __restrict__ zrecs, uint8_t* __restrict__ spec_dst, uint32_t dbg,
                                                             uint64_t* __restrict__ lane_recs, uint32_t ngroups, uint32_t* __restrict__ worklist)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr bool PAD = false; // (the padded window is the lane parser's)
    const uint32_t data_bytes = lz4_window_lds_bytes(G * sub_bytes + 64u, PAD);
    uint32_t* sdata = smem;
    const uint8_t* sbytes = reinterpret_cast<const uint8_t*>(sdata);
    uint32_t* flag = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + data_bytes); // 16 bytes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t* tab = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(smem) + data_bytes + 16u) + (size_t)wave * TAB;
    K5P_DECL
    // one window group per workgroup
    {
    const uint32_t grp = grp0 + blockIdx.x;
    uint32_t lo = 0, hi = nblocks;
    while (hi - lo > 1)
    {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((CLS ? blocks[mid].cgrp_base : blocks[mid].grp_base) <= grp)
            lo = mid;
        else
            hi = mid;
    }
    const Lz4Block blk = blocks[lo];
    const uint32_t gi = grp - (CLS ? blk.cgrp_base : blk.grp_base);
    const uint32_t group_start = gi * G * sub_bytes;                                   // block relative
    const uint32_t glen = blk.size - group_start < G * sub_bytes ? blk.size - group_start : G * sub_bytes;
    const uint8_t* g = src + blk.src_off + group_start;

    // ---- stage the whole group with every wave (16-byte loads from the aligned-down address), clear my table ----
    const uint32_t head_src = (uint32_t)((uintptr_t)g & 15u);
    // LDS byte address of position 0 of the group: the source's misalignment (the window is staged in aligned 16-byte lines) and, for
    // the second formulation of the lane parser, one line of padding in front (its extension reads 8 bytes below a position)
    const uint32_t head = head_src;
    {
        const uint4* gv = reinterpret_cast<const uint4*>(g - head_src);
        const uint32_t nvec = (head_src + glen + 15u) >> 4;
        uint4* sv = reinterpret_cast<uint4*>(sdata);
        for (uint32_t v0 = 0; v0 < nvec; v0 += 64 * G * 4)
        {
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const uint32_t v = v0 + u * 64 * G + tid;
                q[u] = v < nvec ? gv[v] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                const uint32_t v = v0 + u * 64 * G + tid;
                if (v < nvec)
                    sv[v] = q[u];
            }
        }
        uint4* tv = reinterpret_cast<uint4*>(tab);
        const uint32_t e1 = 0xFFFFFFFFu;
        const uint4 e = make_uint4(e1, e1, e1, e1);
#pragma unroll
        for (uint32_t v = 0; v < (TAB * 2 / 16 + 63) / 64; ++v) // TAB * 2 bytes, 64 lanes x 16 bytes per step
            if (v * 64 + lane < TAB * 2 / 16)
                tv[v * 64 + lane] = e;
        if (tid == 0)
            *flag = 0u;
    }
    __syncthreads();
    K5P(0);

    K5P(1);
    // ---- my unit, positions relative to the group start ----
    const uint32_t my_start = (uint32_t)wave * sub_bytes;
    const bool have_unit = my_start < glen;
    const bool emit_unit = have_unit;
    const uint32_t my_len = have_unit ? (glen - my_start < sub_bytes ? glen - my_start : sub_bytes) : 0u;
    const uint32_t unit = blk.seg_base + gi * G + (uint32_t)wave;
    // parsing limits (lz4.c:963-964: mflimit / matchlimit, applied at the BLOCK end)
    const int64_t blk_left = (int64_t)blk.size - (int64_t)group_start - (int64_t)my_start; // unit start .. block end
    int64_t sl = (int64_t)my_len - 4;
    if (sl > blk_left - 12)
        sl = blk_left - 12;
    const int32_t start_limit = have_unit ? (int32_t)((int64_t)my_start + sl) : -1; // < my_start when nothing may start
    const int64_t el = (int64_t)my_len < blk_left - 5 ? (int64_t)my_len : (blk_left - 5 > 0 ? blk_left - 5 : 0);
    const uint32_t end_limit = my_start + (uint32_t)el;

    uint8_t* out = streams + (uint64_t)unit * (FMT == 1 ? sub_bytes : lz4_stream_stride(sub_bytes));
    uint64_t* recs = FMT == 1 ? zrecs + (uint64_t)unit * (sub_bytes >> 2) : nullptr;
    Lz4Seq st;
    st.op = 0;
    st.nseq = 0;
    st.anchor = my_start;
    st.first_lit = st.first_hdr = 0;
    st.have_first = false;
    uint32_t pos = my_start, nfail = 0;
    bool met = false; // the group rendezvous happens exactly once per wave
    uint32_t batches = 0;
    uint32_t pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0, ps0 = 1, ps1 = 1, ps2 = 1, ps3 = 1; // the probe batches (SGPRs)
    static_assert(LZ4_PROBE_BATCHES == 4, "probe bookkeeping is unrolled by hand");

    for (;;)
    {
        const bool more = have_unit && (int32_t)pos <= start_limit;
        // ---- after PROBE batches (or at the end of a short unit): does anybody in the group see redundancy? ----
        if (!met && (batches == LZ4_PROBE_BATCHES || !more))
        {
            if (st.have_first && lane == 0)
                *flag = 1u; // benign race: every writer stores the same value
            __syncthreads();
            if constexpr (CLS != 0)
            {
                if (*flag != 0u)
                {
                    // redundancy: this group is the lane parser's.  Note the 16-unit group it belongs to (once) and leave.
                    if (tid == 0)
                    {
                        const uint32_t g16 = blk.grp_base + gi / (uint32_t)(LZ4_G_LANES / LZ4_G_BATCH);
                        // (bit h of the group's word: its half h holds redundancy; a half WITHOUT it is skimmed by this pass, and the
                        // lane kernel only takes its bytes as history)
#ifdef LTHIP_ABLATIONS
                        // (the whole-group experiments, LTHIP_LZ4_HALVES=0, run over a LIST of the noted groups)
                        if (atomicOr(&worklist[1u + ngroups + g16], 1u << (gi & 1u)) == 0u)
                            worklist[1u + atomicAdd(&worklist[0], 1u)] = g16;
#else
                        // one atomic nobody waits for: k_lz4_pair_halves reads the groups' words.  (Round 5: the list of noted groups
                        // that this thread used to build -- an atomic OR whose answer decides about an atomic add whose answer is the
                        // slot -- kept a workgroup that had nothing left to do on its CU for two round trips to memory.)
                        (void)__hip_atomic_fetch_or(&worklist[1u + ngroups + g16], 1u << (gi & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                    }
                    return;
                }
            }
            if (*flag != 0u && wave != 0 && have_unit && !(dbg & 1u))
            {
                // learn the history: insert every position before my unit, oldest first (plain stores, four
                // independent positions per lane in flight), then re-insert my own probed positions on top
                // every 4th position is enough: a match found one to three bytes late is recovered by the
                // backward extension, and the small table is polluted less (CPU model of this parse: ratio 1.755 -> 1.785)
                for (uint32_t q0 = 0; q0 < my_start; q0 += 1024) // my_start is a multiple of 1024
                {
                    uint32_t hv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        hv[u] = __umulhi(lds_read32(sdata, q0 + 4u * (u * 64 + (uint32_t)lane) + head) * 2654435761u, (uint32_t)TAB);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        tab[hv[u]] = (uint16_t)(q0 + 4u * (u * 64 + (uint32_t)lane));
                }
#define LT_REPLAY(r, PB, PS)                                                                            \
    if ((r) < batches)                                                                                  \
    {                                                                                                   \
        const uint32_t rp = (PB) + (uint32_t)lane * (PS);                                               \
        if ((int32_t)rp <= start_limit)                                                                 \
            tab[__umulhi(lds_read32(sdata, rp + head) * 2654435761u, (uint32_t)TAB)] = (uint16_t)rp;         \
    }
                LT_REPLAY(0u, pb0, ps0)
                LT_REPLAY(1u, pb1, ps1)
                LT_REPLAY(2u, pb2, ps2)
                LT_REPLAY(3u, pb3, ps3)
#undef LT_REPLAY
            }
            else if (*flag == 0u && more && !(dbg & 16u))
                nfail += 24u; // nobody in the 32 KiB group matched anything in its probe batches: one twin round skims the rest
            met = true;
        }
        if (!more)
            break;
        if (batches == 0) { pb0 = pos; ps0 = 1u + nfail; }
        else if (batches == 1) { pb1 = pos; ps1 = 1u + nfail; }
        else if (batches == 2) { pb2 = pos; ps2 = 1u + nfail; }
        else if (batches == 3) { pb3 = pos; ps3 = 1u + nfail; }
        ++batches;

        // ---- probe 64 positions (stride grows with consecutive misses, lz4.c:1044-1053) ----
        uint32_t stride = 1u + nfail;
        uint32_t p = pos + (uint32_t)lane * stride;
        bool valid = (int32_t)p <= start_limit;
        uint32_t v = 0, cand = LZ4_EMPTY, h = 0;
        bool ok = false;
        if (met && nfail != 0u && !(dbg & 8u))
        {
            // Miss mode: the NEXT batch is probed in the same breath (its LDS round trips overlap this batch's): a
            // unit of incompressible data is a chain of dependent probes and nothing else, so this halves its
            // latency.  The second batch reads the table before the first one's inserts and is only inserted if the
            // first one missed.
            const uint32_t pos2 = pos + 64u * stride;
            const uint32_t p2 = pos2 + (uint32_t)lane * (stride + 1u);
            const bool valid2 = (int32_t)p2 <= start_limit;
            uint32_t v2 = 0, cand2 = LZ4_EMPTY, h2 = 0;
            if (valid)
            {
                v = lds_read32(sdata, p + head);
                h = __umulhi(v * 2654435761u, (uint32_t)TAB);
                cand = tab[h];
            }
            if (valid2)
            {
                v2 = lds_read32(sdata, p2 + head);
                h2 = __umulhi(v2 * 2654435761u, (uint32_t)TAB);
                cand2 = tab[h2];
            }
            if (valid)
                tab[h] = (uint16_t)p;
            bool ok2 = false;
            if (valid && cand != LZ4_EMPTY && cand < p)
                ok = lds_read32(sdata, cand + head) == v;
            if (valid2 && cand2 != LZ4_EMPTY && cand2 < p2)
                ok2 = lds_read32(sdata, cand2 + head) == v2;
            if (__builtin_amdgcn_ballot_w64(ok) == 0ull)
            {
                if (valid2)
                    tab[h2] = (uint16_t)p2;
                if (__builtin_amdgcn_ballot_w64(ok2) == 0ull)
                {
                    pos = pos2 + 64u * (stride + 1u);
                    nfail += 2u;
                    continue;
                }
                pos = pos2; // the second batch becomes the current one
                ++nfail;
                stride += 1u;
                p = p2;
                cand = cand2;
                ok = ok2;
            }
        }
        else
        {
            if (valid)
            {
                v = lds_read32(sdata, p + head);
                h = __umulhi(v * 2654435761u, (uint32_t)TAB);
                cand = tab[h];
            }
            // every lane has read the table before anyone updates it (same wave: LDS operations execute in order)
            if (valid)
                tab[h] = (uint16_t)p;
            if (valid && cand != LZ4_EMPTY && cand < p)
                ok = lds_read32(sdata, cand + head) == v;
        }
        // matches whose source lies in the same batch are invisible to the table: look 1, 2, 4, 8 lanes back
        if (nfail == 0 && !(dbg & 2u))
        {
            uint32_t best = 0;
            bool found = false;
            // the four exchanges are issued together (one LDS round trip instead of four); a valid lane's lower neighbours
            // are valid too (their positions are smaller), so their validity needs no exchange
            uint32_t ovs[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                ovs[t] = __shfl_up(v, 8 >> t, 64);
#pragma unroll
            for (int t = 0; t < 4; ++t) // smallest distance wins (assigned last)
            {
                const int d = 8 >> t;
                if (valid && lane >= d && ovs[t] == v)
                {
                    best = p - (uint32_t)d * stride;
                    found = true;
                }
            }
            if (found && (!ok || best > cand))
            {
                cand = best;
                ok = true;
            }
        }
        uint64_t hits = __builtin_amdgcn_ballot_w64(ok);
        if (hits == 0ull)
        {
            pos += 64u * stride;
            ++nfail;
            continue;
        }
        if constexpr (CLS != 0)
        {
            if (!met)
            {
                // classification pass: one hit settles it -- the group is the lane parser's.  Raise the flag and go to the rendezvous
                // without parsing anything.  (The other waves are NOT polled out of their probe batches: a flag read per batch is one
                // more dependent LDS round trip in the chain that incompressible data consists of -- measured +8 % on random data.)
                st.have_first = true;
                if (lane == 0)
                    *flag = 1u;
                pos = (uint32_t)(start_limit + 1);
                continue;
            }
        }
        if (dbg & 4u)
        {
            // ablation: every hit at or after the previous match handled cooperatively with full extension
            do
            {
                const int f = __builtin_ctzll(hits);
                hits &= hits - 1ull;
                const uint32_t pf = pos + (uint32_t)f * stride;
                if (pf < st.anchor)
                    continue;
                lz4_coop_sequence<FMT>(sbytes, head, out, recs, lane, end_limit, pf, __builtin_amdgcn_readlane(cand, f), 4u, true, true, st);
            } while (hits);
            const uint32_t np = pos + 64u * stride;
            pos = np > st.anchor ? np : st.anchor;
            nfail = 0;
            continue;
        }
        if (stride != 1u)
        {
            // sparse mode (after misses): only the first hit, handled by the whole wave
            const int f = __builtin_ctzll(hits);
            lz4_coop_sequence<FMT>(sbytes, head, out, recs, lane, end_limit, pos + (uint32_t)f * stride, __builtin_amdgcn_readlane(cand, f), 4u,
                              true, true, st);
            pos = st.anchor;
            nfail = 0;
            continue;
        }
        // ---- dense mode: every lane measures its own match, in parallel: 16 bytes per LDS round trip (four dword pairs
        // in flight at once -- the batch is a latency chain, not a bandwidth problem), at most 36 bytes ----
        uint32_t mlen = ok ? 4u : 0u;
        bool act = ok;
        const uint32_t maxlen = ok ? end_limit - p : 0u;
#pragma unroll 1
        for (int t = 0; t < 2; ++t)
        {
            if (__builtin_amdgcn_ballot_w64(act) == 0ull)
                break;
            if (act)
            {
                const uint32_t a0 = p + mlen + head, b0 = cand + mlen + head;
                const uint32_t x0 = lds_read32(sdata, a0) ^ lds_read32(sdata, b0);
                const uint32_t x1 = lds_read32(sdata, a0 + 4u) ^ lds_read32(sdata, b0 + 4u);
                const uint32_t x2 = lds_read32(sdata, a0 + 8u) ^ lds_read32(sdata, b0 + 8u);
                const uint32_t x3 = lds_read32(sdata, a0 + 12u) ^ lds_read32(sdata, b0 + 12u);
                uint32_t add = 16u;
                if (x0)
                    add = (uint32_t)__builtin_ctz(x0) >> 3;
                else if (x1)
                    add = 4u + ((uint32_t)__builtin_ctz(x1) >> 3);
                else if (x2)
                    add = 8u + ((uint32_t)__builtin_ctz(x2) >> 3);
                else if (x3)
                    add = 12u + ((uint32_t)__builtin_ctz(x3) >> 3);
                mlen += add;
                if (add != 16u)
                    act = false;
                if (mlen >= maxlen)
                {
                    mlen = maxlen;
                    act = false;
                }
            }
        }
        const uint64_t longs = __builtin_amdgcn_ballot_w64(act); // still equal after 36 bytes: extend when selected
        // ... and how far it could grow backwards (at most 8 bytes, both dword pairs read at once; the anchor bounds it
        // at selection)
        uint32_t nbk = 0;
        if (ok && cand >= 8u)
        {
            const uint32_t x = lds_read32(sdata, p - 4u + head) ^ lds_read32(sdata, cand - 4u + head);
            const uint32_t y = lds_read32(sdata, p - 8u + head) ^ lds_read32(sdata, cand - 8u + head);
            nbk = x ? (uint32_t)__builtin_clz(x) >> 3 : (y ? 4u + ((uint32_t)__builtin_clz(y) >> 3) : 8u);
        }
        // ---- greedy selection in position order: a scalar walk over the SELECTED hits only ----
        uint64_t rem = hits, vecsel = 0ull;
        uint32_t sel_v = 0; // selected lanes: output offset << 8 | backward bytes << 4 | literal count
        while (rem)
        {
            if (st.anchor > pos)
            {
                const uint32_t sh = st.anchor - pos; // lanes whose position is already covered
                rem = sh >= 64u ? 0ull : rem & (~0ull << sh);
                if (!rem)
                    break;
            }
            const int f = __builtin_ctzll(rem);
            rem &= rem - 1ull;
            const uint32_t pf = pos + (uint32_t)f;
            const bool isl = (longs >> f) & 1ull;
            const uint32_t room = pf - st.anchor;
            if (isl || room > 20u)
            {
                lz4_coop_sequence<FMT>(sbytes, head, out, recs, lane, end_limit, pf, __builtin_amdgcn_readlane(cand, f),
                                  __builtin_amdgcn_readlane(mlen, f), isl, true, st);
                continue;
            }
            uint32_t nb = __builtin_amdgcn_readlane(nbk, f);
            nb = nb < room ? nb : room;
            const uint32_t lit = room - nb;
            const uint32_t len = __builtin_amdgcn_readlane(mlen, f) + nb;
            if (lit > 12u)
            {
                lz4_coop_sequence<FMT>(sbytes, head, out, recs, lane, end_limit, pf - nb, __builtin_amdgcn_readlane(cand, f) - nb, len, false,
                                  false, st);
                continue;
            }
            // short literals, match <= 44 bytes: emitted below by the lane itself
            sel_v = lane == f ? (FMT == 1 ? st.nseq << 20 : 0u) | (st.op << 8) | (nb << 4) | lit : sel_v;
            vecsel |= 1ull << f;
            if (!st.have_first)
            {
                st.have_first = true;
                st.first_lit = lit;
                st.first_hdr = 1u;
            }
            if constexpr (FMT == 1)
            {
                st.op += lit;
                st.nseq += 1u;
            }
            else
                st.op += lit + 3u + (len - 4u >= 15u ? 1u : 0u);
            st.anchor = pf + len - nb;
        }
        if ((vecsel >> lane) & 1ull)
        {
            const uint32_t lit = sel_v & 15u;
            const uint32_t nb = (sel_v >> 4) & 15u;
            const uint32_t lit_start_v = p - nb - lit;
            const uint32_t off = p - cand;
            if constexpr (FMT == 1)
            {
                uint8_t* o = out + ((sel_v >> 8) & 0xFFFu);
                for (uint32_t j = 0; j < lit; ++j)
                    o[j] = sbytes[lit_start_v + j + head];
                recs[sel_v >> 20] = (uint64_t)lit | ((uint64_t)(mlen + nb) << 16) | ((uint64_t)off << 32);
            }
            else
            {
                const uint32_t mcode = mlen + nb - 4u; // <= 40: at most one length byte
                uint8_t* o = out + (sel_v >> 8);
                o[0] = (uint8_t)((lit << 4) | (mcode < 15u ? mcode : 15u));
                for (uint32_t j = 0; j < lit; ++j)
                    o[1u + j] = sbytes[lit_start_v + j + head];
                o[1u + lit] = (uint8_t)off;
                o[2u + lit] = (uint8_t)(off >> 8);
                if (mcode >= 15u)
                    o[3u + lit] = (uint8_t)(mcode - 15u);
            }
        }
        const uint32_t np = pos + 64u;
        pos = np > st.anchor ? np : st.anchor;
        nfail = 0;
    }
    if constexpr (FMT == 1)
    {
        // the unit's trailing literals complete its literal buffer; meta = ZbUnitMeta {nseq, nlit, tail, 0}
        const uint32_t tail = emit_unit ? my_start + my_len - st.anchor : 0u;
        if (st.nseq != 0u)
        {
            for (uint32_t j = lane; j < tail; j += 64)
                out[st.op + j] = (uint8_t)lds_byte<PAD>(sdata, st.anchor + j + head);
        }
        else if (emit_unit && spec_dst)
        {
            // A unit without a sequence has no literal buffer (the entropy stage reads its bytes from the source,
            // ZbInput.src).  If its whole 128 KiB piece is like that the piece will most likely be a Raw_Block, and if the
            // pieces before it are raw too its bytes belong at  frame header + pieces * (3 + 128 KiB) + 3 + offset: put
            // them there now, from LDS; k_zstd_emit then only writes the 3-byte block header (and copies as usual when
            // the guess was wrong).
            const uint32_t pos = group_start + my_start;
            const uint64_t o = 13u + (uint64_t)(pos / Z_PIECE) * (Z_PIECE + 3u) + 3u + pos % Z_PIECE;
            if (o + my_len <= (uint64_t)blk.dst_cap)
                wave_copy_lds_to_global<PAD>(spec_dst + blk.dst_off + o, sdata, my_start + head, my_len, lane);
        }
        // is the unit one repeated byte?  (zstd stores a 128 KiB piece made of such units as an RLE_Block)
        uint32_t uniform = 0;
        if (emit_unit)
        {
            const uint32_t b0 = lds_byte<PAD>(sdata, my_start + head);
            const uint32_t rep = b0 * 0x01010101u;
            uint32_t diff = 0;
            // almost always settled by the first dword of every lane's 64 bytes
            if (64u * (uint32_t)lane + 4u <= my_len)
                diff = lds_read32x<PAD>(sdata, my_start + 64u * (uint32_t)lane + head) ^ rep;
            if (__builtin_amdgcn_ballot_w64(diff != 0u) == 0ull)
#pragma unroll 4
            for (uint32_t j = 0; j < 16u; ++j)
            {
                const uint32_t o = 64u * (uint32_t)lane + 4u * j;
                if (o + 4u <= my_len)
                    diff |= lds_read32x<PAD>(sdata, my_start + o + head) ^ rep;
                else if (o < my_len)
                    for (uint32_t k = o; k < my_len; ++k)
                        diff |= lds_byte<PAD>(sdata, my_start + k + head) ^ b0;
            }
            uniform = __builtin_amdgcn_ballot_w64(diff != 0u) == 0ull ? (0x100u | b0) : 0u;
        }
        if (emit_unit && lane == 0)
            reinterpret_cast<uint4*>(meta)[unit] = make_uint4(st.nseq, st.op + tail, tail, uniform);
    }
    else
    {
    // A unit without a single match is all literals.  If that turns out to be true of the WHOLE block (incompressible
    // data -- the common case for already-compressed assets), the payload is one literal run: header, then the source
    // bytes, and this unit's bytes belong at  header + its offset in the block.  They are still in LDS, so they are put
    // there now, speculatively; the stitch then only writes the header for such blocks instead of reading and writing
    // them again (a block that does have matches is laid out by the stitch as usual, overwriting these bytes).
    if (FMT == 0 && spec_dst && emit_unit && !st.have_first)
    {
        const uint64_t o = (uint64_t)(1u + lz4_len_bytes(blk.size)) + group_start + my_start;
        if (o + my_len <= (uint64_t)blk.dst_cap)
            wave_copy_lds_to_global<PAD>(spec_dst + blk.dst_off + o, sdata, my_start + head, my_len, lane);
    }
    if (emit_unit && lane == 0)
    {
        Lz4Meta m;
        m.seq_bytes = st.op;
        m.tail_lits = my_start + my_len - st.anchor;
        m.first_lit_len = st.first_lit;
        m.first_hdr_bytes = st.first_hdr;
        meta[unit] = m;
    }
    } // FMT 0
    K5P(8);
    K5P(9);
    } // the group
}

#ifdef LTHIP_ABLATIONS
#include "ablations/k_lz4_segments_modes.inc"
#endif

// ---------------------------------------------------------------------------------------------------
// K5 lane parser, round 4: HALF-GROUPS as the unit of work.  The classification pass flags redundancy per 32 KiB half of a 64 KiB
// group; a group with one redundant and one incompressible half (37 % of the listed groups of bench.py's compressible tree: regions
// of four kinds, a quarter of them random, at arbitrary offsets against the groups) kept eight of the workgroup's sixteen waves idle
// for the whole group.  Now the list is turned into ITEMS of two halves: the two halves of a group that is redundant throughout
// (linked: one contiguous 64 KiB window, as before), or two LONE halves of different groups, which share nothing but the workgroup --
// each has its own 32 KiB of the window (candidates below a half's start are refused), its own half of the shared table, its own
// source alignment.  What a half compresses to does not depend on its partner.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t LZ4_HALF_NONE = 0xFFFFFFFFu;
constexpr uint32_t LZ4_HALF_HIST = 0x40000000u; // on an item's first half: staged as HISTORY of the second, not parsed (k_lz4_pair_halves, hist)
// worklist layout behind the classification pass (ngroups = groups of the batch): [0] listed groups, [1 .. ngroups] their ids (ablation build only: the product notes the half bits and nothing else),
// [1 + ngroups ..) per-group half bits, [2 ngroups + 1] ticket, [2 ngroups + 2] items, [2 ngroups + 3] lone halves,
// [2 ngroups + 4 ..) items of four words {half a, half b, block of a, block of b} (up to two per group), [10 ngroups + 4 ..) lone halves
__host__ __device__ constexpr uint32_t lz4_items_off(uint32_t ngroups) { return (2u * ngroups + 4u + 3u) & ~3u; } // (16-byte aligned: items are read as uint4)
__device__ __forceinline__ uint32_t lz4_block_of_group(const Lz4Block* __restrict__ blocks, uint32_t nblocks, uint32_t grp)
{
    uint32_t lo = 0, hi = nblocks;
    while (hi - lo > 1)
    {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (blocks[mid].grp_base <= grp)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
// hist (the zstd flavour's "high" and "max" settings): every flagged half becomes an item of its own whose OTHER half is the 32 KiB in
// front of it, staged as history and not parsed -- a half's matches then reach 32 .. 64 KiB back wherever the half lies in its group
// (a lower half has no history otherwise, and an upper half only when the lower one is flagged too), at the price of eight idle waves
// per item.  The first half of every 128 KiB PIECE takes no history: the frame's pieces stay independent (k_zstd.hip decodes them
// with a wave each).
__global__ void k_lz4_pair_halves(uint32_t* __restrict__ wl, uint32_t ngroups, uint32_t phase, const Lz4Block* __restrict__ blocks, uint32_t nblocks,
                                  uint32_t hist)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t* items = wl + lz4_items_off(ngroups);
    uint32_t* lone = items + 8u * ngroups;
    if (phase == 0u)
    {
        // every group of the batch: which of its halves did the classification pass note?
        if (i >= ngroups)
            return;
        const uint32_t g = i, bits = wl[1u + ngroups + g] & 3u;
        if (bits == 0u)
            return;
        if (hist == 2u) // (LTHIP_LZ4_SPLITWG: every flagged half an item of its own)
        {
            for (uint32_t hh = 0; hh < 2u; ++hh)
                if ((bits >> hh) & 1u)
                    lone[atomicAdd(&wl[2u * ngroups + 3u], 1u)] = 2u * g + hh;
        }
        else if (hist)
        {
            const uint32_t b = lz4_block_of_group(blocks, nblocks, g);
            // (a piece = two groups; hist 3 -- the "max" setting -- lets a piece's first half see the piece before, except every
            // LZ4_Z_CHAIN-th piece of the block: the decoder runs the pieces in between as a chain, k_zstd.hip)
            const bool piece_first = hist == 3u ? ((g - blocks[b].grp_base) % (2u * LZ4_Z_CHAIN)) == 0u : ((g - blocks[b].grp_base) & 1u) == 0u;
            for (uint32_t hh = 0; hh < 2u; ++hh)
                if ((bits >> hh) & 1u)
                {
                    const uint32_t h = 2u * g + hh;
                    if (hh == 0u && piece_first)
                        lone[atomicAdd(&wl[2u * ngroups + 3u], 1u)] = h;
                    else
                    {
                        const uint32_t k = atomicAdd(&wl[2u * ngroups + 2u], 1u);
                        items[4u * k] = (h - 1u) | LZ4_HALF_HIST;
                        items[4u * k + 1u] = h;
                        items[4u * k + 2u] = b;
                        items[4u * k + 3u] = b;
                    }
                }
        }
        else if (bits == 3u)
        {
            const uint32_t k = atomicAdd(&wl[2u * ngroups + 2u], 1u);
            const uint32_t b = lz4_block_of_group(blocks, nblocks, g); // (the ten dependent loads of the search happen here, once, in parallel)
            items[4u * k] = 2u * g;
            items[4u * k + 1u] = 2u * g + 1u;
            items[4u * k + 2u] = b;
            items[4u * k + 3u] = b;
        }
        else if (bits)
            lone[atomicAdd(&wl[2u * ngroups + 3u], 1u)] = 2u * g + (bits >> 1);
    }
    else
    {
        const uint32_t nl = wl[2u * ngroups + 3u];
        if (hist == 2u)
        {
            if (i >= nl)
                return;
            const uint32_t k = atomicAdd(&wl[2u * ngroups + 2u], 1u);
            items[4u * k] = lone[i];
            items[4u * k + 1u] = LZ4_HALF_NONE;
            items[4u * k + 2u] = lz4_block_of_group(blocks, nblocks, lone[i] >> 1);
            items[4u * k + 3u] = 0u;
            return;
        }
        if (2u * i >= nl)
            return;
        const uint32_t k = atomicAdd(&wl[2u * ngroups + 2u], 1u);
        const uint32_t a = lone[2u * i], b = 2u * i + 1u < nl ? lone[2u * i + 1u] : LZ4_HALF_NONE;
        items[4u * k] = a;
        items[4u * k + 1u] = b;
        items[4u * k + 2u] = lz4_block_of_group(blocks, nblocks, a >> 1);
        items[4u * k + 3u] = b != LZ4_HALF_NONE ? lz4_block_of_group(blocks, nblocks, b >> 1) : 0u;
    }
}

// WG = waves per workgroup: 16 (the product: an item of two halves per workgroup, one workgroup per CU), or 8 (experiment of round 5,
// ablation build, LTHIP_LZ4_SPLITWG=1: every flagged half an item of its own, TWO independent workgroups of eight waves per CU with
// half the window and half the shared table each -- no half ever waits for another at a barrier, and one workgroup's staging runs beside
// the other's parse; the price: no half sees the half before it).
template <int FMT, int WG = 16>
__global__ __launch_bounds__(64 * WG, 4) void k_lz4_lanes2(const uint8_t* __restrict__ src, const Lz4Block* __restrict__ blocks, uint32_t nblocks,
                                                       uint32_t sub_bytes, uint8_t* __restrict__ streams, Lz4Meta* __restrict__ meta,
                                                       uint64_t* __restrict__ zrecs, uint8_t* __restrict__ spec_dst, uint32_t dbg,
                                                       uint32_t ngroups, uint32_t* __restrict__ worklist, uint32_t farlog)
{
    constexpr int G = LZ4_G_LANES, TAB = LZ4_TAB_SHARED, SH = LZ4_SH_LOG2;
    const uint32_t far1 = 1u << (farlog & 31u), far2 = 1u << ((farlog >> 8) & 31u), far3 = (farlog >> 16) & 255u; // (zstd flavour: lz4_lane_parse2; far3 = bytes a match into the history half must have)
    constexpr bool PAD = true;
    constexpr uint32_t GAP = 32u; // LDS bytes between the windows of two unlinked halves
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    static_assert(WG == 16 || WG == 8, "two halves per workgroup, or one");
    const uint32_t data_bytes = lz4_window_lds_bytes((uint32_t)WG * sub_bytes + 64u + LZ4_LPAD + GAP, true);
    uint32_t* sdata = smem;
    lds_window_must_start_at_zero(sdata);
    uint32_t* flag = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + data_bytes); // 16 bytes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t hsel = (uint32_t)wave >> 3, wih = (uint32_t)wave & 7u; // my half of the item, my unit in the half
    uint16_t* tab = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(smem) + data_bytes + 16u) + (size_t)wave * TAB;
    uint32_t* shr = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + data_bytes + 16u + (size_t)WG * TAB * 2);
    constexpr uint32_t SH_BYTES = WG == 16 ? (4u << SH) : (2u << SH); // (a lone half uses the lower half of the table)
    const uint32_t* items = worklist + lz4_items_off(ngroups);
    const uint32_t nitems = worklist[2u * ngroups + 2u];
    const uint32_t half_bytes = (uint32_t)(G / 2) * sub_bytes;
    K5P_DECL
    uint32_t sh_gen = 0;
    uint32_t next_idx = 0;
    // (Fetching the next item's lines into registers during the parse -- 16 VGPRs -- was built and measured in round 4 as in round 3:
    // no difference, 4.24 / 4.08 / 6.03 ms with it against 4.23 / 4.06 / 6.00 without on mixed / records / tokens.  An item's fixed
    // cost is LDS work -- 16 K ds_min_u32, the table clears and preloads -- and two barriers, not the latency of its loads.)
    for (uint32_t idx = blockIdx.x; idx < nitems; idx = next_idx)
    {
        if (sh_gen == 0u)
        {
            uint4* hv = reinterpret_cast<uint4*>(shr);
            const uint4 none = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            for (uint32_t v = tid; v < SH_BYTES / 16u; v += 64 * WG)
                hv[v] = none;
            __syncthreads();
            sh_gen = (dbg & (1u << 26)) ? 3u : 0xFFFFu;
        }
        --sh_gen;
        const uint32_t sh_base = sh_gen << 16;
        const uint4 item = reinterpret_cast<const uint4*>(items)[idx];
        const bool hist = (item.x & LZ4_HALF_HIST) != 0u; // (a first half is never LZ4_HALF_NONE)
        const uint32_t ha = item.x & ~LZ4_HALF_HIST, hb = item.y;
        const bool linked = hist || (hb == ha + 1u && !(ha & 1u)); // one contiguous 64 KiB window
        const uint32_t hid = hsel ? hb : ha;
        const bool valid = hid != LZ4_HALF_NONE;
        const uint32_t grp = valid ? hid >> 1 : ha >> 1, hh = valid ? hid & 1u : 0u;
        const Lz4Block blk = blocks[valid && hsel ? item.w : item.z];
        const uint32_t gi = grp - blk.grp_base;
        const uint32_t half_start = (gi * (uint32_t)G + hh * (uint32_t)(G / 2)) * sub_bytes; // block relative
        const uint32_t hlen = valid && half_start < blk.size ? (blk.size - half_start < half_bytes ? blk.size - half_start : half_bytes) : 0u;
        const uint8_t* g = src + blk.src_off + half_start;
        const uint32_t head_src = (uint32_t)((uintptr_t)g & 15u);
        const uint32_t wbase = hsel * half_bytes;                       // window position of my half's first byte
        const uint32_t extra = hsel && !linked ? GAP : 0u;              // its LDS displacement
        const uint32_t head = head_src + LZ4_LPAD + extra;              // LDS byte address of window position x: x + head
        const uint32_t lo_bound = hsel && !linked ? half_bytes : 0u;
        const bool split = !linked || hist; // every half has its own half of the shared table
        const uint32_t sh_shift = split ? 33u - (uint32_t)SH : 32u - (uint32_t)SH;
        const uint32_t sh_off = split ? hsel << (SH - 1) : 0u;
        const uint32_t sh_inv = hist && hsel == 0u ? 0xFFFFu : 0u; // a history half keeps its LATEST occurrences: positions stored inverted

        // ---- every half is staged by its own eight waves (16-byte loads from the aligned-down address); the aligned dwords enter the
        // half's part of the shared table from the registers that stage them ----
        {
            const uint4* gv = reinterpret_cast<const uint4*>(g - head_src);
            const uint32_t nvec = hlen ? (head_src + hlen + 15u) >> 4 : 0u;
            const uint32_t line0 = (wbase + extra + LZ4_LPAD) >> 4; // LDS line (unpadded numbering) of my half's source line 0
            const uint32_t th = (uint32_t)tid & 511u;
            for (uint32_t v0 = 0; v0 < nvec; v0 += 512u * 4u)
            {
                uint4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const uint32_t v = v0 + (uint32_t)u * 512u + th;
                    q[u] = v < nvec ? gv[v] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const uint32_t v = v0 + (uint32_t)u * 512u + th;
                    if (v < nvec)
                    {
                        const uint32_t D = 4u * (line0 + v);
                        uint32_t* w = sdata + lds_pidx<true>(D);
                        w[0] = q[u].x;
                        w[1] = q[u].y;
                        w[2] = q[u].z;
                        w[3] = q[u].w;
                        if ((D & 31u) == 0u && D != 0u) // the first line of a row: repeated behind the row before
                        {
                            w[-3] = q[u].x;
                            w[-2] = q[u].y;
                            w[-1] = q[u].z;
                        }
                        const uint32_t pq = wbase + 16u * v - head_src; // window position of the line's first byte
                        const uint32_t g4[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                        // a dword that an EARLIER dword of the same line repeats cannot lower its slot (the table keeps the minimum):
                        // data of a short period (runs, 8-byte patterns) would otherwise send every lane's four updates to one or two
                        // slots, which the LDS serialises (21 % of the kernel on "lines")
                        // ... and the same for the dword at the same place of the line before (the lane before holds it): zero pages
                        // and other runs leave one update per wave
                        // (all of it unconditional, the truth values combined with | and &: as a chain of || and && the compiler wrapped every
                        // neighbour test into its own exec-mask region -- 28 scalar instructions a line in a phase where all sixteen waves
                        // of the CU queue at its one scalar unit)
                        const bool nb = (lane != 0) & (pq >= wbase + 16u);
                        bool dup[4] = {false, g4[1] == g4[0], (bool)((g4[2] == g4[0]) | (g4[2] == g4[1])), (bool)((g4[3] == g4[1]) | (g4[3] == g4[2]) | (g4[3] == g4[0]))};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            dup[k] = (bool)(dup[k] | (nb & ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)g4[k], 0x138 /* wave_shr:1 */, 0xf, 0xf, true) == g4[k])));
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                        {
                            const uint32_t pk = pq + 4u * (uint32_t)k;
                            // (line 0 may begin before the half: a wrapped value is above it -- and then dword k may be the half's first
                            // occurrence although an earlier dword of the line equals it: line 0 takes no shortcut)
                            if ((pk - wbase < hlen) & (!dup[k] | (v == 0u) | (sh_inv != 0u))) // (inside the half; a history half keeps the LATEST: no shortcut)
                                (void)__hip_atomic_fetch_min(&shr[((g4[k] * 2654435761u) >> sh_shift) + sh_off], sh_base | (pk ^ sh_inv), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            uint4* tv = reinterpret_cast<uint4*>(tab);
            const uint4 e = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (uint32_t v = 0; v < (TAB * 2 / 16 + 63) / 64; ++v)
                if (v * 64 + lane < TAB * 2 / 16)
                    tv[v * 64 + lane] = e;
            if (tid == 0)
                flag[1] = gridDim.x + atomicAdd(worklist + 2u * ngroups + 1u, 1u);
        }
        __syncthreads();
        K5P(0);
        next_idx = flag[1];
        K5P(1);
        // ---- my unit, positions relative to the window ----
        const uint32_t my_start = (uint32_t)wave * sub_bytes;
        const bool have_unit = wih * sub_bytes < hlen && !(hist && hsel == 0u); // (a history half is staged and seeded, not parsed)
        const uint32_t my_len = have_unit ? (hlen - wih * sub_bytes < sub_bytes ? hlen - wih * sub_bytes : sub_bytes) : 0u;
        const uint32_t unit = blk.seg_base + gi * (uint32_t)G + hh * (uint32_t)(G / 2) + wih;
        const int64_t blk_left = (int64_t)blk.size - (int64_t)half_start - (int64_t)(wih * sub_bytes); // unit start .. block end
        int64_t sl = (int64_t)my_len - 4;
        if (sl > blk_left - 12)
            sl = blk_left - 12;
        const int32_t start_limit = have_unit ? (int32_t)((int64_t)my_start + sl) : -1;
        const int64_t el = (int64_t)my_len < blk_left - 5 ? (int64_t)my_len : (blk_left - 5 > 0 ? blk_left - 5 : 0);
        const uint32_t end_limit = my_start + (uint32_t)el;
        uint8_t* out = streams + (uint64_t)unit * (FMT == 1 ? sub_bytes : lz4_stream_stride(sub_bytes));
        uint64_t* recs = FMT == 1 ? zrecs + (uint64_t)unit * (sub_bytes >> 2) : nullptr;
        Lz4Seq st;
        st.op = 0;
        st.nseq = 0;
        st.anchor = my_start;
        st.first_lit = st.first_hdr = 0;
        st.have_first = false;
        // the private table starts with the aligned dwords of the unit before mine (my half's, or -- linked -- the other half's last)
        if (have_unit && (wih > 0u || (hsel && linked)) && !(dbg & 1u))
        {
            const uint32_t p0 = my_start - sub_bytes;
            const uint32_t l0 = (p0 + head) >> 4, l1 = (my_start + head + 15u) >> 4;
            for (uint32_t j0 = l0; j0 < l1; j0 += 256)
            {
                uint4 w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const uint32_t j = j0 + (uint32_t)u * 64u + (uint32_t)lane;
                    const uint32_t* q = sdata + lds_pidx<true>(4u * j);
                    w[u] = j < l1 ? make_uint4(q[0], q[1], q[2], q[3]) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const uint32_t j = j0 + (uint32_t)u * 64u + (uint32_t)lane;
                    const uint32_t q = 16u * j - head;
                    const uint32_t g4[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        const uint32_t pk = q + 4u * (uint32_t)k;
                        if (j < l1 && pk >= p0 && pk < my_start)
                            tab[lz4_tab_slot<TAB, 2>(g4[k] * 2654435761u)] = (uint16_t)pk;
                    }
                }
            }
        }
        if (have_unit)
        {
            K5P(2);
            lz4_lane_parse2<TAB, FMT, SH>(sdata, head, tab, shr, sh_base, lane, my_start, my_len, start_limit, end_limit, sub_bytes >> 6, out, recs, st,
                                          dbg K5P_PASS, lo_bound, sh_shift, sh_off, linked ? 0xFFFFFFFFu : wbase, far1, far2, hist ? half_bytes : 0u, far3);
        }
        if constexpr (FMT == 1)
        {
            const uint32_t tail = have_unit ? my_start + my_len - st.anchor : 0u;
            if (st.nseq != 0u)
            {
                for (uint32_t j = lane; j < tail; j += 64)
                    out[st.op + j] = (uint8_t)lds_byte<PAD>(sdata, st.anchor + j + head);
            }
            else if (have_unit && spec_dst)
            {
                const uint32_t pos = half_start + wih * sub_bytes;
                const uint64_t o = 13u + (uint64_t)(pos / Z_PIECE) * (Z_PIECE + 3u) + 3u + pos % Z_PIECE;
                if (o + my_len <= (uint64_t)blk.dst_cap)
                    wave_copy_lds_to_global<PAD>(spec_dst + blk.dst_off + o, sdata, my_start + head, my_len, lane);
            }
            uint32_t uniform = 0;
            if (have_unit)
            {
                const uint32_t b0 = lds_byte<PAD>(sdata, my_start + head);
                const uint32_t rep = b0 * 0x01010101u;
                uint32_t diff = 0;
                if (64u * (uint32_t)lane + 4u <= my_len)
                    diff = lds_read32x<PAD>(sdata, my_start + 64u * (uint32_t)lane + head) ^ rep;
                if (__builtin_amdgcn_ballot_w64(diff != 0u) == 0ull)
#pragma unroll 4
                    for (uint32_t j = 0; j < 16u; ++j)
                    {
                        const uint32_t o = 64u * (uint32_t)lane + 4u * j;
                        if (o + 4u <= my_len)
                            diff |= lds_read32x<PAD>(sdata, my_start + o + head) ^ rep;
                        else if (o < my_len)
                            for (uint32_t k = o; k < my_len; ++k)
                                diff |= lds_byte<PAD>(sdata, my_start + k + head) ^ b0;
                    }
                uniform = __builtin_amdgcn_ballot_w64(diff != 0u) == 0ull ? (0x100u | b0) : 0u;
            }
            if (have_unit && lane == 0)
                reinterpret_cast<uint4*>(meta)[unit] = make_uint4(st.nseq, st.op + tail, tail, uniform);
        }
        else
        {
            if (spec_dst && have_unit && !st.have_first)
            {
                const uint64_t o = (uint64_t)(1u + lz4_len_bytes(blk.size)) + half_start + wih * sub_bytes;
                if (o + my_len <= (uint64_t)blk.dst_cap)
                    wave_copy_lds_to_global<PAD>(spec_dst + blk.dst_off + o, sdata, my_start + head, my_len, lane);
            }
            if (have_unit && lane == 0)
            {
                Lz4Meta m;
                m.seq_bytes = st.op;
                m.tail_lits = my_start + my_len - st.anchor;
                m.first_lit_len = st.first_lit;
                m.first_hdr_bytes = st.first_hdr;
                meta[unit] = m;
            }
        }
        K5P(8);
        __syncthreads(); // every wave is done with the window before the next item overwrites it
        K5P(9);
    }
    K5P_FLUSH;
}

__device__ __forceinline__ void wg_emit_header(uint8_t* dst, uint32_t lits, uint32_t match_nibble, int tid, int nthreads)
{
    if (tid == 0)
        dst[0] = (uint8_t)(((lits < 15u ? lits : 15u) << 4) | match_nibble);
    if (lits >= 15u)
    {
        const uint32_t len = lits - 15u;
        const uint32_t n = len / 255u + 1u;
        for (uint32_t j = tid; j < n; j += nthreads)
            dst[1 + j] = j + 1 == n ? (uint8_t)(len % 255u) : (uint8_t)255;
    }
}

// ---------------------------------------------------------------------------------------------------
// K6a: per-block serial walk over segment results
// ---------------------------------------------------------------------------------------------------
// One wave per block.  The walk over the units is inherently serial (a unit's placement depends on the literal
// carry of its predecessors) but its inputs are not: 64 result records are loaded per step with one coalesced
// load, the serial logic then runs wave-uniformly on readlane'd values, and the placement records are stored
// coalesced again.  Trailing literals belong to a literal RUN that is closed by the next unit with a match (or by
// the end of the block); the start of a run's literal area is only known when it closes, so units remember
// (run index, offset inside the run) and k_lz4_stitch_copy resolves it through the run table.
__global__ __launch_bounds__(64) void k_lz4_stitch_scan(const Lz4Block* __restrict__ blocks, uint32_t b0, uint32_t nblocks, uint32_t SEG,
                                                        const Lz4Meta* __restrict__ meta, Lz4Plan* __restrict__ plan,
                                                        uint32_t* __restrict__ runs, Lz4BlockOut* __restrict__ bout,
                                                        uint32_t* __restrict__ out_sizes, uint8_t* __restrict__ dst, uint32_t spec,
                                                        uint32_t* __restrict__ worklist /* [0] = count, then group ids, then (from 1 + wl_cap) their blocks */,
                                                        uint32_t wl_cap)
{
    const uint32_t b = blockIdx.x + b0;
    if (b >= nblocks)
        return;
    const int lane = threadIdx.x;
    const Lz4Block blk = blocks[b];
    const uint32_t run_base = blk.seg_base + b; // this block's slice of the run table (<= nseg + 1 runs)
    uint64_t out_pos = 0; // 64-bit: pathological inputs are caught against dst_cap at the end
    uint32_t carry = 0;
    uint32_t run = 0;
    // The walk is a recurrence over (literal carry, output position, run count), restated as three wave scans per 64
    // units so that no step is serial:  a unit WITH a match resets the carry to its tail literals, one without adds its
    // length (segmented inclusive scan);  only units with a match advance the output (prefix sum of token + length
    // bytes + literals + body, all functions of the carry that reaches them) and close a run (prefix count).
    Lz4Meta m;
    m.seq_bytes = m.tail_lits = m.first_lit_len = m.first_hdr_bytes = 0;
    if ((uint32_t)lane < blk.nseg)
        m = meta[blk.seg_base + (uint32_t)lane];
    for (uint32_t i0 = 0; i0 < blk.nseg; i0 += 64)
    {
        const uint32_t i = i0 + (uint32_t)lane;
        const Lz4Meta cur = m;
        if (i + 64u < blk.nseg) // next step's records are in flight while this one is scanned
            m = meta[blk.seg_base + i + 64u];
        const bool valid = i < blk.nseg;
        const bool has = valid && cur.seq_bytes != 0u;
        const uint32_t seg_len = valid ? (blk.size - i * SEG < SEG ? blk.size - i * SEG : SEG) : 0u;
        // inclusive segmented scan of the carry each unit leaves behind
        uint32_t sv = has ? cur.tail_lits : seg_len;
        uint32_t sf = has ? 1u : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            const uint32_t pv = __shfl_up(sv, d, 64);
            const uint32_t pf = __shfl_up(sf, d, 64);
            if (lane >= d)
            {
                if (!sf)
                    sv += pv;
                sf |= pf;
            }
        }
        if (!sf)
            sv += carry; // nobody before me in this step had a match: the carry of the previous steps reaches me
        uint32_t cin = __shfl_up(sv, 1, 64); // the carry that reaches me
        if (lane == 0)
            cin = carry;
        // output advance of the units with a match
        const uint32_t L = cin + cur.first_lit_len;
        const uint32_t hdr = 1u + lz4_len_bytes(L);
        const uint64_t adv = has ? (uint64_t)hdr + L + (cur.seq_bytes - cur.first_hdr_bytes - cur.first_lit_len) : 0ull;
        uint64_t pos = adv; // inclusive prefix sum
#pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            const uint64_t pv = __shfl_up(pos, d, 64);
            if (lane >= d)
                pos += pv;
        }
        const uint64_t my_pos = out_pos + pos - adv;
        const uint64_t hm = __builtin_amdgcn_ballot_w64(has);
        const uint32_t before = (uint32_t)__builtin_popcountll(hm & ((1ull << lane) - 1ull));
        Lz4Plan pl;
        pl.hdr_pos = 0xFFFFFFFFu;
        pl.hdr_lits = pl.first_lit_dst = pl.body_dst = 0;
        pl.tail_rel = cin;
        pl.run = run_base + run + before;
        if (has)
        {
            const uint64_t lit_dst = my_pos + hdr;
            runs[run_base + run + before] = (uint32_t)lit_dst; // closes the open run
            pl.hdr_pos = (uint32_t)my_pos;
            pl.hdr_lits = L;
            pl.first_lit_dst = (uint32_t)(lit_dst + cin);
            pl.body_dst = pl.first_lit_dst + cur.first_lit_len;
            pl.tail_rel = 0; // this unit's tail opens the next run
            pl.run = run_base + run + before + 1u;
        }
        if (valid)
            plan[blk.seg_base + i] = pl;
        out_pos += __shfl(pos, 63, 64);
        carry = __shfl(sv, 63, 64); // lanes past the end add nothing
        run += (uint32_t)__builtin_popcountll(hm);
    }
    // final literal-only sequence (lz4.c:1302-1329)
    const uint32_t hdr = 1u + lz4_len_bytes(carry);
    const uint64_t lit_dst = out_pos + hdr;
    const uint64_t total = lit_dst + carry;
    // Which groups does the copy kernel have to visit?  A block without a single match was laid out by the match finder
    // (spec): its one header is written here and the copy kernel never hears of it.
    if (total <= (uint64_t)blk.dst_cap)
    {
        if (run == 0u && spec)
        {
            if (blk.nseg)
                wg_emit_header(dst + blk.dst_off + out_pos, carry, 0u, lane, 64);
        }
        else
        {
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(worklist, blk.ngrp);
            base = __builtin_amdgcn_readfirstlane(base);
            for (uint32_t k = lane; k < blk.ngrp; k += 64)
            {
                worklist[1u + base + k] = blk.grp_base + k;
                worklist[1u + wl_cap + base + k] = b; // (the copy kernel used to search the block table for it: ten dependent loads per group)
            }
        }
    }
    if (lane == 0)
    {
        runs[run_base + run] = (uint32_t)lit_dst;
        Lz4BlockOut bo;
        bo.final_hdr_pos = (uint32_t)out_pos;
        bo.final_lits = carry;
        bo.total = total <= (uint64_t)blk.dst_cap ? (uint32_t)total : 0u;
        bo.pad = run == 0u ? 1u : 0u; // no unit found a match: the literals were placed by the match finder already
        bout[b] = bo;
        out_sizes[b] = bo.total;
    }
}

// ---------------------------------------------------------------------------------------------------
// K6b: one workgroup per segment moves its pieces into place
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wg_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, int tid,
                                        int nthreads)
{
    if (n == 0)
        return;
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
    for (uint32_t v = tid; v < nvec; v += nthreads)
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


constexpr int K6_THREADS = 256;

// one workgroup per window group: its (up to gunits) units are moved into place one after the other
#ifdef LTHIP_K6_WAVES
__attribute__((amdgpu_waves_per_eu(LTHIP_K6_WAVES, LTHIP_K6_WAVES)))
#endif
__global__ __launch_bounds__(K6_THREADS) void k_lz4_stitch_copy(const uint8_t* __restrict__ src,
                                                                 const Lz4Block* __restrict__ blocks, uint32_t nblocks,
                                                                 uint32_t SEG, const uint8_t* __restrict__ streams,
                                                                 const Lz4Meta* __restrict__ meta,
                                                                 const Lz4Plan* __restrict__ plan,
                                                                 const uint32_t* __restrict__ runs,
                                                                 const Lz4BlockOut* __restrict__ bout,
                                                                 uint8_t* __restrict__ dst, const uint32_t* __restrict__ worklist,
                                                                 uint32_t gunits /* units per window group */, uint32_t wl_cap)
{
    const int tid = threadIdx.x;
    const uint32_t nwork = worklist[0];
    (void)nblocks;
  for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x)
  {
    const uint32_t grp = worklist[1u + wi];
    const uint32_t lo = worklist[1u + wl_cap + wi]; // the group's block (k_lz4_stitch_scan)
    const Lz4Block blk = blocks[lo];
    const Lz4BlockOut bo = bout[lo];
    uint8_t* d = dst + blk.dst_off;
    const uint32_t i0 = (grp - blk.grp_base) * gunits;
    const uint32_t i1 = i0 + gunits < blk.nseg ? i0 + gunits : blk.nseg;
    // each wave moves whole units on its own (4 units in flight per workgroup: their table loads overlap)
    const int lane = tid & 63;
    for (uint32_t i = i0 + (uint32_t)(tid >> 6); i < i1; i += K6_THREADS / 64)
    {
        const uint32_t seg = blk.seg_base + i;
        const uint32_t seg_start = i * SEG;
        const uint32_t len = blk.size - seg_start < SEG ? blk.size - seg_start : SEG;
        const Lz4Meta m = meta[seg];
        const Lz4Plan pl = plan[seg];
        const uint8_t* s = src + blk.src_off + seg_start;
        const uint8_t* stream = streams + (uint64_t)seg * lz4_stream_stride(SEG);
        uint32_t tail = len;
        if (m.seq_bytes)
        {
            wg_emit_header(d + pl.hdr_pos, pl.hdr_lits, stream[0] & 15u, lane, 64);
            wg_copy(d + pl.first_lit_dst, s, m.first_lit_len, lane, 64);
            const uint32_t skip = m.first_hdr_bytes + m.first_lit_len;
            wg_copy(d + pl.body_dst, stream + skip, m.seq_bytes - skip, lane, 64);
            tail = m.tail_lits;
        }
        wg_copy(d + runs[pl.run] + pl.tail_rel, s + (len - tail), tail, lane, 64);
        if (i + 1 == blk.nseg)
            wg_emit_header(d + bo.final_hdr_pos, bo.final_lits, 0u, lane, 64);
    }
  }
}

// empty blocks have no segment: their single 0x00 token is written here
__global__ void k_lz4_empty_blocks(const Lz4Block* __restrict__ blocks, uint32_t nblocks,
                                   const Lz4BlockOut* __restrict__ bout, uint8_t* __restrict__ dst)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks)
        return;
    if (blocks[b].nseg == 0 && bout[b].total == 1)
        dst[blocks[b].dst_off] = 0;
}

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
