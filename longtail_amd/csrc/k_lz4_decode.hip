// k_lz4_decode.hip -- LZ4 block decoders on gfx950 (LZ4_decompress_safe rules, lib/lz4/ext/lz4.c:2215-2435).
#include "lthip_internal.h"
#include "origin_exec.h"

#include <stdlib.h>
#include <time.h>

#include <algorithm>

namespace
{

struct Lz4Block // one payload and where its bytes go
{
    uint64_t src_off;
    uint64_t dst_off;
    uint32_t size;
    uint32_t dst_cap;
    uint32_t out_index; // where the result goes in the caller's array (the host splits a call between two decoders)
    uint32_t pad;
};

#ifdef LTHIP_ABLATIONS
#include "ablations/k_lz4_decode_plain.inc"
#endif

// The same rules, restructured around what made the wave-per-block decoder above slow: every
// token, length byte and offset was a dependent GLOBAL load, every literal and match a wave-wide byte store, and every match a
// wait for all of those stores.
//   * the payload streams through a 4 KiB LDS window (16-byte coalesced refills); 64 payload bytes at a time sit in a register
//     window (one byte per lane) with the next one prefetched, so a short sequence (no length bytes: most of them) is parsed
//     with `readlane`s alone and its literals are stored from the registers they are already in;
//   * the OUTPUT is produced into an 8 KiB LDS ring and leaves for global memory 2 KiB at a time with aligned 16-byte stores;
//     matches with offset <= 8192 (the bulk) never touch global memory, farther ones read bytes that were flushed long ago.
// Measured (8 MiB blocks of the "mixed" workload, 100-160 K sequences each): 106 ms per block against 173 ms; what remains is the
// serial parse itself (77 ms with every copy switched off: ~280 instructions per sequence issued by a single wave), so the
// throughput of a batch comes from the number of blocks in flight: 41 GB/s at 512 blocks, 140 GB/s at 2048.
constexpr uint32_t DEC_IN = 2048u, DEC_RING = 8192u, DEC_FLUSH = 2048u; // 10 KiB of LDS per decoding wave: 16 per CU
constexpr int DEC_IN_VECS = DEC_IN / 1024; // 16-byte vectors per lane and refill
#ifdef LTHIP_DEC_PROF /* debug build only (make prof): where a decoding wave spends its cycles */
__device__ unsigned long long g_dec_prof[16];
#define DEC_T0() const unsigned long long t0__ = clock64()
#define DEC_ACC(i) do { if (lane == 0) atomicAdd(&g_dec_prof[i], clock64() - t0__); } while (0)
#define DEC_CNT(i) do { if (lane == 0) atomicAdd(&g_dec_prof[i], 1ull); } while (0)
#else
#define DEC_T0() ((void)0)
#define DEC_ACC(i) ((void)0)
#define DEC_CNT(i) ((void)0)
#endif

constexpr uint32_t PD_UNIT = 65536u, PD_TILE = 8192u, PD_RUNIN = 512u, PD_NONE = 0xFFFFFFFFu;
constexpr uint32_t PD_FINAL = 1u, PD_INVALID = 2u;

struct PdBlock
{
    uint64_t src_off;
    uint64_t dst_off;
    uint32_t size;
    uint32_t dst_cap;
    uint32_t out_index;
    uint32_t tile_base; // first PdTile / tile output position of the block
    uint32_t ntiles;
    uint32_t unit_base; // first unit record / flag of the block
    uint32_t nunits_cap; // units the capacity allows
    uint32_t pad;
};
struct PdTile
{
    uint32_t entry; // first token of the chain at or past the tile's first byte (PD_NONE: the chain jumped over the tile)
    uint32_t exit;  // first token of the chain at or past the tile's end (the payload size after the last sequence)
    uint32_t out;   // bytes the sequences from `entry` up to `exit` produce
    uint32_t flags; // PD_FINAL: the chain ended properly inside; PD_INVALID: it hit something no decoder accepts
};
// the payload through a 4 KiB LDS window and a 64-byte register window (the serial decoder's scheme, positions only)
struct PdReader
{
    const uint8_t* in_al;
    uint8_t* s_in;
    uint32_t head, n;
    int lane;
    int64_t wa, w0;
    uint32_t w;
    __device__ __forceinline__ void init(const uint8_t* in, uint32_t n_, uint8_t* s, int l)
    {
        head = (uint32_t)((uintptr_t)in & 15u);
        in_al = in - head;
        s_in = s;
        n = n_;
        lane = l;
        wa = -(int64_t)DEC_IN;
        w0 = -1000;
        w = 0;
    }
    __device__ __forceinline__ uint32_t byte_at(int64_t p)
    {
        if (p < w0 || p >= w0 + 64)
        {
            const int64_t a = p + head;
            if (a < wa || a + 64 > wa + (int64_t)DEC_IN)
            {
                wa = a & ~(int64_t)15;
                const int64_t end = (int64_t)n + head;
                __syncthreads();
                uint4 q[DEC_IN_VECS];
#pragma unroll
                for (int u = 0; u < DEC_IN_VECS; ++u)
                {
                    const int64_t o = wa + 16 * (int64_t)(u * 64 + lane);
                    q[u] = *reinterpret_cast<const uint4*>(in_al + (o < end ? o : wa));
                }
#pragma unroll
                for (int u = 0; u < DEC_IN_VECS; ++u)
                    reinterpret_cast<uint4*>(s_in)[u * 64 + lane] = q[u];
                __syncthreads();
            }
            w = s_in[(uint32_t)(a - wa) + (uint32_t)lane];
            w0 = p;
        }
        return __builtin_amdgcn_readlane(w, (int)(p - w0));
    }
    // length bytes from position p on (a run of 255s and its terminator), 64 at a time; the byte at q may be read while q < limit
    __device__ __forceinline__ bool more_len(int64_t& p, uint64_t& len, const int64_t limit, const uint64_t cap)
    {
        for (;;)
        {
            if (p >= limit)
                return false;
            (void)byte_at(p);
            const int d = (int)(p - w0);
            const uint64_t not255 = __builtin_amdgcn_ballot_w64(w != 255u) >> d;
            const int run = not255 ? __builtin_ctzll(not255) : 64 - d; // 255s in front of the terminator (or to the window's end)
            if (p + run >= limit)
                return false; // one of them, or the terminator, lies at or past the limit
            len += 255ull * (uint64_t)run;
            p += run;
            if (len > cap)
                return false;
            if (not255)
            {
                len += __builtin_amdgcn_readlane(w, d + run);
                ++p;
                return len <= cap;
            }
        }
    }
};

// One sequence at position ip, lengths only.  0: a sequence, `next` = the token after it, `out` = bytes it produces; 1: the last
// sequence of the payload (literals up to its very end); 2: nothing a decoder accepts at any output position.
__device__ __forceinline__ int pd_hop(PdReader& r, const int64_t ip, const uint64_t cap, int64_t& next, uint64_t& out,
                                      uint32_t* lit_pos = nullptr, uint32_t* lit_len = nullptr)
{
    const int64_t n = r.n;
    if (ip >= n)
        return 2;
    int64_t p = ip;
    const uint32_t token = r.byte_at(p++);
    uint64_t len = token >> 4;
    if (len == 15 && !r.more_len(p, len, n - 15, cap))
        return 2;
    if (lit_pos)
    {
        *lit_pos = (uint32_t)p;
        *lit_len = (uint32_t)len;
    }
    if (p + (int64_t)len > n - 8)
    {
        out = len;
        next = n;
        return p + (int64_t)len == n ? 1 : 2;
    }
    p += (int64_t)len + 2;
    uint64_t ml = token & 15u;
    if (ml == 15 && !r.more_len(p, ml, n - 4, cap))
        return 2;
    out = len + ml + 4;
    next = p;
    return 0;
}

struct PdPos
{
    int64_t ip;  // a token position (the payload size after the last sequence)
    uint64_t op; // output position of that token's first literal
    int kind;    // 0: stopped at a bound, 1: the payload ended properly, 2: damage at ip
};

// Follow the chain of tokens from (ip, op) while ip < ip_stop and the sequence at ip ends at or before op_stop.  64 positions at a
// time: every lane reads its byte of the window as a token and computes where the next token would be and what the sequence
// would produce (no length bytes, or one match-length byte below 255 inside the window; anything else is left to pd_hop), then
// the chain is a walk over two registers: two readlanes and a dozen scalar instructions per sequence (pd_hop: ~80).
template <bool OPSTOP> // OPSTOP = false: op_stop is "never" (the test per sequence is compiled out)
__device__ __forceinline__ PdPos pd_walk(PdReader& r, int64_t ip, uint64_t op, const int64_t ip_stop, const uint64_t op_stop, const uint64_t cap)
{
    const int32_t n = (int32_t)r.n; // below 2^31 (LZ4_MAX_INPUT_SIZE)
    const uint64_t op_limit = cap + 65536ull; // more than any block may hold: the chain is damaged (or not the true one)
    for (;;)
    {
        if (ip >= ip_stop)
            return PdPos{ip, op, 0};
        if (ip >= n)
            return PdPos{ip, op, 2};
        if (r.w0 != ip)
        {
            r.w0 = ip - 64; // force a reseed at exactly ip
            (void)r.byte_at(ip);
        }
        const uint32_t w = r.w;
        const uint32_t litl = w >> 4, mlcl = w & 15u;
        const int32_t pl = (int32_t)ip + r.lane;
        // literal length: the nibble, or 15 + ONE length byte (the byte after the token, when it is in the window and below 255)
        const bool big = litl == 15u;
        const uint32_t after = (uint32_t)__shfl((int)w, (r.lane + 1) & 63, 64);
        const uint32_t lit = big ? 15u + after : litl, hdr = big ? 2u : 1u;
        const uint32_t e = (uint32_t)r.lane + hdr + lit + 2u; // where a match-length byte would be, relative to the window
        const uint32_t ext = (uint32_t)__shfl((int)w, (int)(e & 63u), 64);
        const bool one = mlcl == 15u;
        bool simple = pl + (int32_t)(hdr + lit) <= n - 8; // "ip + len > n - 8" ends the payload
        if (big) // its length byte: "ip >= n - 15" before it, "ip > n - 15" after it
            simple = simple && r.lane < 63 && after != 255u && pl + 2 <= n - 15;
        if (one)
            simple = simple && e < 64u && pl + (int32_t)(hdr + lit) + 2 < n - 4 && ext != 255u;
        const uint32_t outv = lit + mlcl + 4u + (one ? ext : 0u);
        const uint32_t nxtv = e + (one ? 1u : 0u);
        // 32-bit running values: op stays below op_limit + 64 x 300 < 2^32
        uint32_t o32 = (uint32_t)op;
        const uint32_t stop32 = op_stop < 0xFFFFFFFFull ? (uint32_t)op_stop : 0xFFFFFFFFu;
        const uint32_t limit32 = (uint32_t)op_limit;
        const uint32_t left = ip_stop - ip < 0x7FFFFFFF ? (uint32_t)(ip_stop - ip) : 0x7FFFFFFFu; // the walk stops at relative position >= left
        uint64_t ok = __builtin_amdgcn_ballot_w64(simple);
        if (left < 64u)
            ok &= (1ull << left) - 1ull; // no token at or past the bound is followed
        uint32_t cur = 0;
        int why = 0; // 1: bound reached, 2: damage, 0: window exhausted or a token for pd_hop
        if constexpr (!OPSTOP)
        {
            // No test per sequence is needed: the chain through the window is resolved for all 64 possible starts at once by pointer
            // doubling on the vector unit (the scalar unit is what these kernels are short of).  Per lane one word {bytes produced
            // << 10 | position reached}, position >= 512 = final (>= 64 + 512: left the window; 512 + i: stands on token i, which
            // the walk does not follow); five rounds of "take over what the position I reached has reached" cover 32 sequences,
            // a window holds at most 21.
            const bool follow = (ok >> r.lane) & 1ull;
            uint32_t st = follow ? (outv << 10) | (nxtv >= 64u ? 512u + nxtv : nxtv) : 512u + (uint32_t)r.lane;
#pragma unroll
            for (int round = 0; round < 5; ++round)
            {
                const uint32_t there = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((st & 63u) << 2), (int)st);
                if (!(st & 512u))
                    st = ((st & ~1023u) + (there & ~1023u)) | (there & 1023u);
            }
            const uint32_t s0 = __builtin_amdgcn_readlane(st, 0);
            o32 += s0 >> 10;
            cur = (s0 & 1023u) - 512u; // always final after five rounds: >= 64 = left the window there, < 64 = stands on that token
        }
        while (OPSTOP && cur < 64u && ((ok >> cur) & 1ull)) // per sequence: a bit test, two readlanes, an add
        {
            const uint32_t o = __builtin_amdgcn_readlane(outv, (int)cur);
            if (OPSTOP && o32 + o > stop32)
            {
                why = 1;
                break;
            }
            o32 += o;
            cur = __builtin_amdgcn_readlane(nxtv, (int)cur);
        }
        if (!why)
            why = o32 > limit32 ? 2 : cur >= left ? 1 : 0;
        ip += cur;
        op = (op & ~0xFFFFFFFFull) | o32;
        if (why)
            return PdPos{ip, op, why == 1 ? 0 : 2};
        if (cur < 64u)
        {
            int64_t next = 0;
            uint64_t out = 0;
            const int kind = pd_hop(r, ip, cap, next, out);
            if (kind == 2)
                return PdPos{ip, op, 2};
            if (op + out > op_stop)
                return PdPos{ip, op, 0};
            op += out;
            if (op > op_limit)
                return PdPos{ip, op, 2};
            ip = next;
            if (kind == 1)
                return PdPos{ip, op, 1};
        }
    }
}

// the chain from `start` through the tile [t0, t1)
__device__ __forceinline__ PdTile pd_walk_tile(PdReader& r, int64_t start, const int64_t t0, const int64_t t1, const uint64_t cap)
{
    PdTile rec{PD_NONE, 0u, 0u, 0u};
    PdPos a = pd_walk<false>(r, start, 0, t0, ~0ull, cap); // to the first token at or past t0
    if (a.kind == 0 && a.ip < t1)
    {
        rec.entry = (uint32_t)a.ip;
        a = pd_walk<false>(r, a.ip, 0, t1, ~0ull, cap);
    }
    else if (a.kind == 0)
        a.op = 0; // jumped over the tile
    rec.exit = (uint32_t)a.ip;
    rec.out = a.op < 0xFFFFFFFFull ? (uint32_t)a.op : 0xFFFFFFFFu; // saturated: anything this large is beyond every capacity
    rec.flags = a.kind == 1 ? PD_FINAL : a.kind == 2 ? PD_INVALID : 0u;
    return rec;
}

// What a UNIT decoder (the block-parallel path further down) knows beyond the block: it owns the output range [lo, hi) of the block
// and starts at a sequence boundary (ip0, op0) at or before lo that k_lz4_pd_link found; all values are wave-uniform.
struct PdUnit
{
    int64_t lo, hi;
    int64_t ip0, op0;
    bool last;                // the unit that runs to the end of the payload (hi is not a limit for it)
    const uint32_t* prev_done; // completion flag of the unit before this one (sources below lo live there)
    bool defer;                // a source below lo: do not wait for that unit, give the unit up (it is then executed on origins)
    uint32_t* timeout;         // set when a wait gave up (the block is reported as damaged)
    volatile uint32_t* dbg;    // LTHIP_LZ4_PD_TRACE: host-visible progress words of this workgroup (else nullptr)
};
#define PD_DBG(i, v) do { if (UNIT && un.dbg && lane == 0) un.dbg[i] = (uint32_t)(v); } while (0)
enum : uint32_t { DEC_ERROR = 0xFFFFFFFFu, DEC_UNIT_OK = 0u, DEC_UNIT_END = 1u, DEC_UNIT_DEFER = 2u };

// One payload (UNIT = false; returns the decoded size or DEC_ERROR) or one unit of it (UNIT = true; returns DEC_UNIT_OK, DEC_UNIT_END
// when it reached the valid end of the payload, or DEC_ERROR).
// I = index type: int32_t when every payload and capacity of the batch is below 1 GiB (half the scalar work), else int64_t
template <typename I, bool UNIT>
__device__ __forceinline__ uint32_t lz4_decode_one(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const I n, const I cap,
                                                   const uint32_t dec_nobatch, uint8_t* __restrict__ s_in, uint8_t* __restrict__ s_ring,
                                                   const int lane, const PdUnit& un)
{
    const I lo = UNIT ? (I)un.lo : (I)0;
    const I hi = UNIT && !un.last ? (I)un.hi : cap; // nothing at or past hi is stored by this call
    const uint32_t head = (uint32_t)((uintptr_t)in & 15u);
    const uint8_t* in_al = in - head; // 16-byte aligned; payload byte p sits at aligned offset p + head
    const uint32_t g = (uint32_t)((uintptr_t)out & 15u);
    uint8_t* out_al = out - g; // output byte q sits at aligned offset q + g, and in the ring at (q + g) mod 8 KiB
    uint32_t result = 0xFFFFFFFFu;
    I wa = -(I)DEC_IN; // aligned offset of s_in[0]; nothing loaded yet
    // make payload bytes [p, p + k) (k <= 128; bytes at or past n are never used) resident; returns the index of p in s_in
    auto need = [&](I p, uint32_t k) -> uint32_t {
        const I a = p + (I)head;
        if (a < wa || a + (I)k > wa + (I)DEC_IN)
        {
            DEC_CNT(8);
            wa = a & ~(I)15;
            const I end = n + (I)head; // first aligned offset past the payload
            __syncthreads();
            uint4 q[DEC_IN_VECS]; // loads in flight; vectors past the payload re-read the window's first one (their bytes are never used)
#pragma unroll
            for (int u = 0; u < DEC_IN_VECS; ++u)
            {
                const I o = wa + 16 * (I)(u * 64 + lane);
                q[u] = *reinterpret_cast<const uint4*>(in_al + (int64_t)(o < end ? o : wa));
            }
#pragma unroll
            for (int u = 0; u < DEC_IN_VECS; ++u)
                reinterpret_cast<uint4*>(s_in)[u * 64 + lane] = q[u];
            __syncthreads();
        }
        return (uint32_t)(a - wa);
    };
    if (cap == 0)
    {
        if (n == 1 && in[0] == 0)
            result = 0;
    }
    else if (n > 0)
    {
        I ip = 0, op = 0;
        I flushed = UNIT ? (lo + (I)g) & ~(I)(DEC_FLUSH - 1u) : (I)0; // aligned output offset (multiple of DEC_FLUSH) up to which the ring has been written out
        I drained = flushed; // aligned output offset up to which the flush stores are known to have landed
        // UNIT: everything that crosses to another workgroup is stored write-through and loaded past the L1 (sc1 both sides,
        // MI355X_MICROARCH.md "inter-workgroup visibility"); a buffer descriptor of the block's output carries the cache bits
        [[maybe_unused]] __amdgpu_buffer_rsrc_t orsrc;
        if constexpr (UNIT)
            orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out_al, 0, (int)((uint32_t)cap + g), 0x00020000);
        [[maybe_unused]] bool prev_ready = false, gave_up = false, deferred = false;
        // sources below lo were written by the unit before this one (an offset is below 64 KiB = one unit): wait for its flag once
        auto wait_prev = [&]() {
            if constexpr (UNIT)
            {
                if (prev_ready || !un.prev_done)
                    return;
                uint32_t spins = 0;
                while (__hip_atomic_load(un.prev_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                {
                    __builtin_amdgcn_s_sleep(32);
                    if (++spins > (1u << 22)) // seconds: something is broken; report the block as damaged instead of hanging
                    {
                        gave_up = true;
                        if (lane == 0)
                            __hip_atomic_store(un.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                prev_ready = true;
            }
        };
        auto out_byte = [&](int64_t q) -> uint8_t { // output byte q of the block from global memory
            if constexpr (UNIT)
                return __builtin_amdgcn_raw_buffer_load_b8(orsrc, (int)((uint32_t)q + g), 0, 16);
            else
                return out[q];
        };
        I w0 = -1000;  // payload position of lane 0 of the register window
        uint32_t w = 0, wn = 0; // the window and the one 40 bytes further on (fetched while this one is parsed)
#define RING(q) (((uint32_t)(q) + g) & (DEC_RING - 1u))
        // ring -> global for aligned offsets [flushed, upto): whole 16-byte vectors, bytes at the two ragged ends of the block
        auto flush = [&](I upto) {
            DEC_T0();
            while (flushed < upto)
            {
                const I stop = upto - flushed < (I)DEC_FLUSH ? upto : flushed + (I)DEC_FLUSH;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                    const I P = flushed + 16 * (I)(u * 64 + lane);
                    if (P >= stop)
                        continue;
                    const uint8_t* r = s_ring + ((uint32_t)P & (DEC_RING - 1u));
                    if constexpr (UNIT)
                    {
                        const I first = lo + (I)g, lim = stop < hi + (I)g ? stop : hi + (I)g; // only [lo, hi) is this unit's to store
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        if (P >= first && P + 16 <= lim)
                            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(r), orsrc, (int)(uint32_t)P, 0, 16);
                        else
                            for (int k = 0; k < 16; ++k)
                                if (P + k >= first && P + k < lim)
                                    __builtin_amdgcn_raw_buffer_store_b8(r[k], orsrc, (int)(uint32_t)(P + k), 0, 16);
                    }
                    else if (P >= (I)g && P + 16 <= stop)
                        *reinterpret_cast<uint4*>(out_al + (int64_t)P) = *reinterpret_cast<const uint4*>(r);
                    else
                        for (int k = 0; k < 16; ++k)
                            if (P + k >= (I)g && P + k < stop)
                                out_al[(int64_t)P + k] = r[k];
                }
                flushed = stop;
            }
            DEC_ACC(3);
        };
        auto seed = [&](I p) {
            const uint32_t i = need(p, 128);
            w = s_in[i + (uint32_t)lane];
            wn = s_in[i + 40u + (uint32_t)lane];
            w0 = p;
        };
        auto byte_at = [&](I p) -> uint32_t {
            if (p < w0 || p >= w0 + 64)
                seed(p);
            return __builtin_amdgcn_readlane(w, (int)(p - w0));
        };
        // literals: payload [p, p + len) -> the ring at output position o
        auto copy_lits = [&](I p, I o, I len) {
            // A long run (incompressible stretches; a whole block of random data is ONE run) does not have to pass through the ring:
            // bring the ring to a flush boundary, copy the bulk payload -> output directly with 16-byte stores, and leave the last
            // DEC_RING bytes of the run to the loop below, so that the ring holds what later matches may reach.
            if (len >= (I)(3u * DEC_RING))
            {
                const uint32_t mis = ((uint32_t)o + g) & (DEC_FLUSH - 1u);
                I pre = mis ? (I)(DEC_FLUSH - mis) : (I)0;
                while (pre > 0) // (at most one flush granule, the loop below in small)
                {
                    const uint32_t i = need(p, 1);
                    I avail = (I)DEC_IN - (I)i;
                    const uint32_t c = (uint32_t)(pre < avail ? pre : avail);
                    for (uint32_t j = lane; j < c; j += 64)
                        s_ring[RING((uint32_t)o + j)] = s_in[i + j];
                    p += (I)c;
                    o += (I)c;
                    len -= (I)c;
                    pre -= (I)c;
                }
                flush(o + (I)g); // now flushed == o + g, a multiple of the flush granule
                const I bulk = ((len - (I)DEC_RING) / (I)DEC_FLUSH) * (I)DEC_FLUSH;
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef u32x4 u32x4_a1 __attribute__((aligned(1)));
                for (I q = 16 * (I)lane; q < bulk; q += 1024)
                {
                    const u32x4 v = *reinterpret_cast<const u32x4_a1*>(in + (int64_t)(p + q)); // (the payload has any alignment)
                    if constexpr (UNIT)
                        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (int)(uint32_t)(o + (I)g + q), 0, 16);
                    else
                        *reinterpret_cast<u32x4*>(out_al + (int64_t)(o + (I)g + q)) = v;
                }
                p += bulk;
                o += bulk;
                len -= bulk;
                flushed += bulk; // (drained stays behind: a far match waits for these stores like for any flush)
            }
            while (len > 0)
            {
                const uint32_t i = need(p, 1);
                I avail = (I)DEC_IN - (I)i;
                avail = avail < (I)DEC_FLUSH ? avail : (I)DEC_FLUSH;
                const uint32_t c = (uint32_t)(len < avail ? len : avail);
                for (uint32_t j = lane; j < c; j += 64)
                    s_ring[RING((uint32_t)o + j)] = s_in[i + j];
                p += (I)c;
                o += (I)c;
                len -= (I)c;
                if (o + (I)g - flushed >= (I)DEC_FLUSH)
                    flush((o + (I)g) & ~(I)(DEC_FLUSH - 1u));
            }
        };
        // match of `ml` bytes at distance `off` (1 <= off <= op) appended at op; returns nothing, advances op
        auto copy_match = [&](uint32_t off, I ml) {
            DEC_T0();
            if (off > DEC_RING)
                DEC_CNT(9);
            while (ml > 0)
            {
                // segments of <= 2 KiB (out[q] = out[q - off] holds for every q of a match, so a segment is a match of its own;
                // with that bound no ring slot is overwritten before its last read and before it has been flushed)
                uint32_t seg = ml < (I)DEC_FLUSH ? (uint32_t)ml : DEC_FLUSH;
                if (UNIT && op - (I)off < lo)
                {
                    // the source starts in the unit before this one: that part (it ends at lo at the latest, so it cannot
                    // overlap what is written here) comes from global memory once that unit has published its bytes
                    const I gap = lo - (op - (I)off);
                    seg = (I)seg < gap ? seg : (uint32_t)gap;
                    if (un.defer)
                    {
                        deferred = true; // (whatever else the caller does with this unit is thrown away)
                        op += ml;
                        return;
                    }
                    wait_prev();
                    for (uint32_t j = lane; j < seg; j += 64)
                        s_ring[RING((uint32_t)op + j)] = out_byte((int64_t)op - off + j);
                }
                else if (off <= DEC_RING)
                {
                    const uint32_t base = (uint32_t)op - off;
                    if (off >= 64u) // a 64-byte step never reads what it writes
                        for (uint32_t j = lane; j < seg; j += 64)
                            s_ring[RING((uint32_t)op + j)] = s_ring[RING(base + j)];
                    else if (seg <= 64u)
                    {
                        uint32_t r = (uint32_t)lane; // lane mod off
                        for (uint32_t t = off; t < seg; t += off)
                            r = (uint32_t)lane >= t ? (uint32_t)lane - t : r;
                        if ((uint32_t)lane < seg)
                            s_ring[RING((uint32_t)op + (uint32_t)lane)] = s_ring[RING(base + r)];
                    }
                    else // overlapping copy: byte j of the match equals byte (j mod off) of the seed
                        for (uint32_t j = lane; j < seg; j += 64)
                            s_ring[RING((uint32_t)op + j)] = s_ring[RING(base + j % off)];
                }
                else
                {
                    // the source left the ring long ago (off > 8 KiB, flushes every 2 KiB) -- but its stores must have landed
                    if (op - (I)off + (I)seg + (I)g > drained)
                    {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_s_waitcnt(0);
                        drained = flushed;
                    }
                    for (uint32_t j = lane; j < seg; j += 64) // off > 8192 > seg: no overlap
                        s_ring[RING((uint32_t)op + j)] = out_byte((int64_t)op - off + j);
                }
                op += (I)seg;
                ml -= (I)seg;
                if (op + (I)g - flushed >= (I)DEC_FLUSH)
                    flush((op + (I)g) & ~(I)(DEC_FLUSH - 1u));
            }
            DEC_ACC(2);
        };
        bool entered = true;
        if constexpr (UNIT)
        {
            if (lo > 0)
            {
                // Walk (positions only) from the sequence boundary the link pass handed over to the sequence that covers lo;
                // sequences that end at or before lo belong to earlier units, which also check them.  The covering sequence is
                // checked exactly as the general path below checks it and executed from lo on.
                entered = false;
                {
                    PdReader rd;
                    rd.init(in, (uint32_t)n, s_in, lane);
                    const PdPos at = pd_walk<true>(rd, un.ip0, (uint64_t)un.op0, INT64_MAX, (uint64_t)lo, (uint64_t)cap);
                    ip = (I)at.ip; // the covering sequence (or where the chain broke: the loop below reports it)
                    op = (I)at.op;
                    wa = -(I)DEC_IN; // the reader went through s_in: nothing the decoder knows about is in it any more
                    w0 = -1000;
                }
                // length bytes after a token nibble of 15; `limit`: the byte at position p may be read while p < limit
                auto more_len = [&](I& p, I& len, const I limit) -> bool {
                    for (;;) // 64 length bytes at a time: a block of incompressible data is ONE sequence with 32 K of them
                    {
                        if (p >= limit)
                            return false;
                        (void)byte_at(p);
                        const int d = (int)(p - w0);
                        const uint64_t not255 = __builtin_amdgcn_ballot_w64(w != 255u) >> d;
                        const int run = not255 ? __builtin_ctzll(not255) : 64 - d; // 255s in front of the terminator (or to the window's end)
                        if (p + run >= limit)
                            return false; // one of them, or the terminator, lies at or past the limit
                        len += (I)(255 * run);
                        p += run;
                        if (len > cap)
                            return false;
                        if (not255)
                        {
                            len += (I)__builtin_amdgcn_readlane(w, d + run);
                            ++p;
                            return len <= cap;
                        }
                    }
                };
                while (ip < n)
                {
                    PD_DBG(4, ip);
                    PD_DBG(5, op);
                    I p = ip;
                    const uint32_t token = byte_at(p++);
                    I len = (I)(token >> 4);
                    // general path: "ip >= n - 15" before the first length byte, "ip > n - 15" after each = may read while p < n - 15
                    if (len == 15 && !more_len(p, len, n - 15))
                        break;
                    const I lits = p; // payload position of the literals
                    if (op + len > cap - 12 || p + len > n - 8)
                    {
                        // the last sequence of the payload (or damage)
                        if (p + len != n || op + len > cap || op + len < lo || (!un.last && op + len < hi))
                            break;
                        const I a = op > lo ? op : lo;
                        I c = op + len - a;
                        if (!un.last && a + c > hi)
                            c = hi - a;
                        if (c > 0)
                            copy_lits(lits + (a - op), a, c);
                        op += len;
                        if (un.last)
                        {
                            flush(op + (I)g);
                            result = DEC_UNIT_END;
                        }
                        else
                        {
                            flush(hi + (I)g);
                            result = DEC_UNIT_OK;
                        }
                        break;
                    }
                    p += len;
                    const I offpos = p;
                    p += 2;
                    I ml = (I)(token & 15);
                    if (ml == 15 && !more_len(p, ml, n - 4)) // general path: "ip >= n - 5 + 1" before every length byte
                        break;
                    ml += 4;
                    if (op + len + ml <= lo)
                    {
                        op += len + ml;
                        ip = p;
                        continue;
                    }
                    const uint32_t off = byte_at(offpos) | (byte_at(offpos + 1) << 8);
                    if (off == 0 || (I)off > op + len || op + len + ml > cap - 5)
                        break;
                    const I lit_end = op + len;
                    if (lit_end > lo)
                    {
                        const I a = op > lo ? op : lo;
                        I c = lit_end - a;
                        if (!un.last && a + c > hi)
                            c = hi - a;
                        if (c > 0)
                            copy_lits(lits + (a - op), a, c);
                    }
                    op = lit_end > lo ? lit_end : lo;
                    I rest = ml - (op - lit_end);
                    if (!un.last && op + rest > hi)
                        rest = hi - op;
                    if (rest > 0)
                        copy_match(off, rest);
                    else if (rest < 0)
                        op = hi; // the literals alone reached the end of the unit
                    ip = p;
                    entered = true;
                    break;
                }
            }
        }
        DEC_T0();
        while (entered)
        {
            if (UNIT && deferred)
                break;
            if (UNIT && !un.last && op >= hi)
            {
                flush(hi + (I)g);
                result = DEC_UNIT_OK;
                break;
            }
            if (ip >= n)
                break;
            // ---- short sequences (no length bytes, <= 14 literals, match <= 18): everything is in the register window ----
            {
                I d = ip - w0;
                if constexpr (UNIT)
                {
                    // many waves per CU hide the latency of an LDS read: start every step with the token in lane 0, so that a
                    // batch sees the whole window (the sliding scheme below is for a lone wave per CU)
                    if (d != 0)
                    {
                        const uint32_t i = need(ip, 64);
                        w = s_in[i + (uint32_t)lane];
                        w0 = ip;
                        d = 0;
                    }
                }
                else if (d > 47 && d < 88) // slide: the prefetched window becomes the current one
                {
                    w = wn;
                    w0 += 40;
                    d -= 40;
                    const uint32_t i = need(w0 + 40, 64);
                    wn = s_in[i + (uint32_t)lane];
                }
                if (d < 0 || d > 47)
                {
                    seed(ip);
                    d = 0;
                }
                // ---- batch path: SEVERAL short sequences per step.  Every lane reads "its" byte of the window as if it were a
                // token (<= 14 literals, match <= 18, everything it needs inside the window); a scalar walk follows the chain of
                // real tokens from the current position (a few SALU instructions per token instead of the ~120 of the
                // one-sequence path); all their literals go into the ring with ONE store (a lane's byte belongs to the nearest
                // token before it), the matches are copied by the tokens' own lanes, all at once when their sources lie before
                // the batch, otherwise in dependency order.  The conditions are the one-sequence path's, token by token: the first
                // token that fails any of them ends the batch and is left to the code below (which also decides about errors).
                // ---- a literal run of 15..269 bytes (ONE length byte) and its match, the wave together: what the general code below does
                // for such a sequence, without its window re-seeds and loops (it is 8-10 % of the sequences of this library's encoder
                // and cost half the time).  Same conditions, same verdicts; everything else is left to the general code. ----
                if constexpr (UNIT)
                {
                    const uint32_t tk0 = __builtin_amdgcn_readlane(w, 0); // d == 0 here
                    const uint32_t e0 = __builtin_amdgcn_readlane(w, 1);
                    if ((tk0 >> 4) == 15u && e0 != 255u)
                    {
                        const I lit = (I)(15u + e0), lp = ip + 2;
                        I ml = (I)(tk0 & 15u) + 4;
                        if (ip + 2 <= n - 15 && !(op + lit > cap - 12 || lp + lit > n - 8))
                        {
                            const uint32_t i = need(lp, (uint32_t)lit + 3u);
                            const uint32_t b0 = __builtin_amdgcn_readfirstlane((uint32_t)s_in[i + (uint32_t)lit]),
                                           b1 = __builtin_amdgcn_readfirstlane((uint32_t)s_in[i + (uint32_t)lit + 1u]),
                                           b2 = __builtin_amdgcn_readfirstlane((uint32_t)s_in[i + (uint32_t)lit + 2u]);
                            I adv = 2 + lit + 2;
                            bool take = true;
                            if ((tk0 & 15u) == 15u)
                            {
                                take = b2 != 255u; // longer matches: general code
                                ml += (I)b2;
                                adv += 1;
                            }
                            if (take && (un.last || op + lit + ml <= hi))
                            {
                                const uint32_t off = b0 | (b1 << 8);
                                if (off == 0 || (I)off > op + lit || op + lit + ml > cap - 5)
                                    break;
                                for (uint32_t j = lane; j < (uint32_t)lit; j += 64)
                                    s_ring[RING((uint32_t)op + j)] = s_in[i + j];
                                op += lit;
                                copy_match(off, ml);
                                ip += adv;
                                DEC_CNT(11);
                                continue;
                            }
                        }
                    }
                }
                // (a first token with 15 or more literals is for the general code: do not set a batch up for it)
                if (d <= 40 && !dec_nobatch && (__builtin_amdgcn_readlane(w, (int)d) >> 4) != 15u)
                {
                    constexpr uint32_t INLANE_MAX = 64u;                    // longest match a lane copies on its own
                    constexpr uint32_t RING_SAFE = DEC_RING - 1792u;        // a batch writes up to 21 x (14 + 64) bytes ahead of `op`
                    const uint32_t litl = w >> 4, mlcl = w & 15u;
                    const int e1 = lane + 1 + (int)litl;                    // where my offset would be
                    const bool one = mlcl == 15u;                           // one match-length byte behind the offset (19..273 bytes)
                    const uint32_t offl = (uint32_t)__shfl((int)w, e1 & 63, 64) | ((uint32_t)__shfl((int)w, (e1 + 1) & 63, 64) << 8);
                    const uint32_t ext = (uint32_t)__shfl((int)w, (e1 + 2) & 63, 64);
                    const bool candidate = litl < 15u && e1 + (one ? 2 : 1) < 64 && (!one || ext != 255u);
                    const uint32_t seqlen = 3u + litl + (one ? 1u : 0u), mll = mlcl + 4u + (one ? ext : 0u);
                    // long matches, and matches of more than 18 bytes whose source has left the ring, are copied by the whole wave:
                    // such a token is taken as the LAST of a batch
                    const bool coop = candidate && (mll > INLANE_MAX || (mll > 18u && offl > RING_SAFE));
                    const uint64_t okm = __builtin_amdgcn_ballot_w64(candidate && !coop), cpm = __builtin_amdgcn_ballot_w64(coop);
                    // the chain of real tokens from the current position: a bit walks up the mask of in-lane candidates (shifted out
                    // of the word when the chain leaves the window); a wave-copied candidate where it stops is taken as the last one
                    uint64_t vis = 0ull, bit = 1ull << d;
                    uint32_t cur = (uint32_t)d;
                    while (okm & bit)
                    {
                        vis |= bit;
                        const uint32_t step = __builtin_amdgcn_readlane(seqlen, (int)cur);
                        cur += step;
                        bit <<= step;
                    }
                    vis |= cpm & bit;
                    const int ntok = __builtin_popcountll(vis);
                    if (ntok >= 2)
                    {
                        const bool tv = (vis >> lane) & 1ull;
                        const uint32_t adv = tv ? litl + (coop ? 0u : mll) : 0u; // what the lanes themselves append
                        uint32_t incl = adv; // inclusive prefix sum over the lanes
#pragma unroll
                        for (int sh = 1; sh < 64; sh <<= 1)
                        {
                            const uint32_t o = (uint32_t)__shfl_up((int)incl, sh, 64);
                            if (lane >= sh)
                                incl += o;
                        }
                        const I opl = op + (I)(incl - adv);   // where my literals go (if I am a token)
                        const I ipl = w0 + (I)lane;           // my payload position
                        const I opm = opl + (I)litl;          // where my match goes
                        // the one-sequence path's conditions (a length byte at ipl + 3 + litl <= n - 6 passes the general path's
                        // "ip < n - 4" by construction)
                        const bool bad = tv && (opl + (I)litl > cap - 12 || ipl + 1 + (I)litl > n - 8 || opm + (I)mll > cap - 5 || offl == 0u ||
                                                (I)offl > opm || (UNIT && opm - (I)offl < lo));
                        const uint64_t badm = __builtin_amdgcn_ballot_w64(bad);
                        if (badm)
                            vis &= (1ull << __builtin_ctzll(badm)) - 1ull;
                        if (__builtin_popcountll(vis) >= 2)
                        {
                            const bool tk2 = (vis >> lane) & 1ull;
                            const int lastl = 63 - __builtin_clzll(vis);
                            // literals, all tokens at once
                            {
                                const uint64_t below = vis & ((1ull << lane) - 1ull);
                                const int pi = below ? 63 - __builtin_clzll(below) : 0;
                                const uint32_t plit = (uint32_t)__shfl((int)litl, pi, 64);
                                const uint32_t pop = (uint32_t)__shfl((int)(uint32_t)opl, pi, 64);
                                const uint32_t rel = (uint32_t)(lane - pi - 1);
                                if (below && rel < plit)
                                    s_ring[RING(pop + rel)] = (uint8_t)w;
                            }
                            // matches: sources further back than the ring safely holds come from global memory -- landed?
                            const bool mine = tk2 && !coop;
                            const bool glob = mine && offl > RING_SAFE; // at most 18 bytes (longer ones are `coop`)
                            if (__builtin_amdgcn_ballot_w64(glob && opm - (I)offl + (I)mll + (I)g > drained))
                            {
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                __builtin_amdgcn_s_waitcnt(0);
                                drained = flushed;
                            }
                            uint64_t pend = vis & ~cpm;
                            // tokens whose source has left the ring (rare): from global memory, first -- that data is final, nothing in
                            // the batch feeds it -- and in a branch of its own, so that its 18 addresses are not computed for every batch
                            const uint64_t globm = __builtin_amdgcn_ballot_w64(glob);
                            if (globm)
                            {
                                if (glob)
                                {
                                    const int64_t sp = (int64_t)opm - (int64_t)offl; // > RING_SAFE bytes back: cannot overlap the target
                                    uint32_t bytes[18];
#pragma unroll
                                    for (uint32_t k = 0; k < 18u; ++k)
                                        bytes[k] = k < mll ? out_byte(sp + k) : 0u;
#pragma unroll
                                    for (uint32_t k = 0; k < 18u; ++k)
                                        if (k < mll)
                                            s_ring[RING((uint32_t)opm + k)] = (uint8_t)bytes[k];
                                }
                                pend &= ~globm;
                            }
                            while (pend)
                            {
                                const int first = __builtin_ctzll(pend);
                                // positions relative to `op` (a batch appends at most 1792 bytes): everything before the first pending
                                // token's match is final
                                const int32_t rel_m = (int32_t)(opm - op);
                                const int32_t frontier = (int32_t)__builtin_amdgcn_readlane((uint32_t)rel_m, first);
                                const bool ready = ((pend >> lane) & 1ull) && (lane == first || rel_m - (int32_t)offl + (int32_t)mll <= frontier);
                                if (ready)
                                {
                                    // four bytes per step where the offset allows it (unaligned LDS dwords; a step that would cross the
                                    // end of the ring goes byte by byte): the loop is lane-divergent, every trip costs the wave its mask
                                    // bookkeeping.  Overlapping matches (offset < length) replicate their seed exactly because every
                                    // byte is stored before the next is read.
                                    typedef uint32_t u32_a1 __attribute__((aligned(1)));
                                    const uint32_t so = (uint32_t)opm - offl;
                                    uint32_t k = 0;
                                    if (offl >= 4u)
                                        for (; k + 4u <= mll; k += 4u)
                                        {
                                            const uint32_t a = RING(so + k), b = RING((uint32_t)opm + k);
                                            if (a <= DEC_RING - 4u && b <= DEC_RING - 4u)
                                                *reinterpret_cast<u32_a1*>(s_ring + b) = *reinterpret_cast<const u32_a1*>(s_ring + a);
                                            else
                                                for (uint32_t j = 0; j < 4u; ++j)
                                                    s_ring[RING((uint32_t)opm + k + j)] = s_ring[RING(so + k + j)];
                                        }
                                    for (; k < mll; ++k)
                                        s_ring[RING((uint32_t)opm + k)] = s_ring[RING(so + k)];
                                }
                                pend &= ~__builtin_amdgcn_ballot_w64(ready);
                            }
                            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, lastl);
                            op += (I)total;
                            ip = w0 + (I)lastl + (I)__builtin_amdgcn_readlane(seqlen, lastl);
                            DEC_CNT(12);
#ifdef LTHIP_DEC_PROF
                            if (lane == 0)
                                atomicAdd(&g_dec_prof[13], (unsigned long long)__builtin_popcountll(vis));
#endif
                            if (vis & cpm)
                            {
                                // the last token's match, by the whole wave (op stands at its first byte)
                                const uint32_t offc = __builtin_amdgcn_readlane(offl, lastl);
                                I mlc2 = (I)__builtin_amdgcn_readlane(mll, lastl);
                                if (UNIT && !un.last && op + mlc2 > hi)
                                    mlc2 = hi - op;
                                if (mlc2 > 0)
                                    copy_match(offc, mlc2);
                            }
                            else if (op + (I)g - flushed >= (I)DEC_FLUSH)
                                flush((op + (I)g) & ~(I)(DEC_FLUSH - 1u));
                            continue;
                        }
                    }
                }
                const uint32_t tk = __builtin_amdgcn_readlane(w, (int)d);
                const I lit = (I)(tk >> 4), mlc = (I)(tk & 15u);
                // exactly the conditions under which the general code below takes its plain path for this token; a match
                // length with ONE extension byte (19..272 bytes) is still read from the window
                if (lit < 15 && !(op + lit > cap - 12 || ip + 1 + lit > n - 8))
                {
                    I ml = mlc + 4, adv = 3 + lit;
                    bool fast = true;
                    if (mlc == 15)
                    {
                        // its position ip + 3 + lit <= n - 6 satisfies the general path's "ip < n - 4" test by construction
                        const I e = d + 3 + lit;
                        const uint32_t v = e <= 63 ? (uint32_t)__builtin_amdgcn_readlane(w, (int)(e <= 63 ? e : 0)) : 255u;
                        fast = v != 255u;
                        ml += (I)v;
                        adv += 1;
                    }
                    if (fast && op + lit + ml <= cap - 5)
                    {
                        const uint32_t off = __builtin_amdgcn_readlane(w, (int)(d + 1 + lit)) | (__builtin_amdgcn_readlane(w, (int)(d + 2 + lit)) << 8);
                        if (off == 0 || (I)off > op + lit)
                            break;
                        const I rel = (I)lane - d - 1; // my byte is literal `rel` of this sequence
                        if (rel >= 0 && rel < lit)
                            s_ring[RING((uint32_t)op + (uint32_t)rel)] = (uint8_t)w;
                        op += lit;
                        copy_match(off, ml);
                        ip += adv;
                        DEC_CNT(10);
                        continue;
                    }
                }
            }
            const uint32_t token = byte_at(ip++);
            I len = (I)(token >> 4);
            bool bad = false;
            if (len == 15)
            { // read_variable_length(&ip, iend-RUN_MASK, 1), lz4.c:1979-2013
                uint32_t v;
                if (ip >= n - 15)
                    bad = true;
                else
                    do
                    {
                        v = byte_at(ip++);
                        len += (I)v;
                        if (ip > n - 15 || len > cap) // a literal run longer than the capacity is rejected below anyway
                        {
                            bad = true;
                            break;
                        }
                    } while (v == 255);
            }
            if (bad)
                break;
            if (op + len > cap - 12 || ip + len > n - 8)
            {
                if (ip + len != n || op + len > cap)
                    break;
                if (UNIT && !un.last)
                {
                    // the payload ends in a later unit: this one stores its part of the literals (and must be filled by them)
                    if (op + len < hi)
                        break;
                    copy_lits(ip, op, hi - op);
                    op = hi;
                    continue;
                }
                copy_lits(ip, op, len);
                op += len;
                flush(op + (I)g);
                result = UNIT ? (uint32_t)DEC_UNIT_END : (uint32_t)op;
                break;
            }
            if (UNIT && !un.last && op + len >= hi)
            {
                copy_lits(ip, op, hi - op); // the rest of this sequence belongs to (and is checked by) the next unit
                op = hi;
                continue;
            }
            copy_lits(ip, op, len);
            ip += len;
            op += len;
            const uint32_t off = byte_at(ip) | (byte_at(ip + 1) << 8);
            ip += 2;
            if (off == 0 || (I)off > op)
                break;
            I ml = (I)(token & 15);
            if (ml == 15)
            {
                uint32_t v;
                do
                {
                    if (ip >= n - 5 + 1)
                    {
                        bad = true;
                        break;
                    }
                    v = byte_at(ip++);
                    ml += (I)v;
                    if (ml > cap) // rejected below anyway
                    {
                        bad = true;
                        break;
                    }
                } while (v == 255);
            }
            if (bad)
                break;
            ml += 4;
            if (op + ml > cap - 5)
                break;
            if (UNIT && !un.last && op + ml > hi)
                ml = hi - op;
            copy_match(off, ml);
            DEC_CNT(11);
        }
        DEC_ACC(0);
        if (gave_up)
            result = DEC_ERROR;
        if (UNIT && deferred)
            result = DEC_UNIT_DEFER;
#undef RING
    }
    return result;
}

template <typename I>
__global__ __launch_bounds__(64) void k_lz4_decode_lds(const uint8_t* __restrict__ src, const Lz4Block* __restrict__ blocks,
                                                       uint32_t nblocks, uint8_t* __restrict__ dst,
                                                       uint32_t* __restrict__ out_sizes, uint32_t dec_nobatch)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN];
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[DEC_RING];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks)
        return;
    const int lane = threadIdx.x;
    const Lz4Block blk = blocks[b];
    const uint32_t result = lz4_decode_one<I, false>(src + blk.src_off, dst + blk.dst_off, (I)blk.size, (I)blk.dst_cap, dec_nobatch, s_in,
                                                     s_ring, lane, PdUnit{});
    if (lane == 0)
        out_sizes[blk.out_index] = result;
}

// ---------------------------------------------------------------------------------------------------
// Block-parallel decoding.  A block is a serial object only at the level of WHERE things are: the position of a token follows from the
// token before it and the output position from all the lengths so far.  Those positions are found first, by many waves, and then
// the block is decoded in UNITS of 64 KiB of output, one wave each:
//   1. k_lz4_pd_tiles   one wave per 8 KiB TILE of the payload follows the chain of tokens (lengths only, nothing is copied) from 512
//                       bytes before its tile -- a guess -- and records where the chain enters the tile, where it leaves it and how
//                       many bytes the sequences in between produce.  Chains that start at different positions merge as soon as
//                       they hit a common token, so for the true entry the guess is almost always what was recorded.
//   2. k_lz4_pd_link    one wave per block hops from tile to tile along the true chain (payload position and output position
//                       known): a tile whose recorded entry is the true one costs a table look-up, any other tile is walked again
//                       from the true position.  It writes the output position of every tile entry and, for every unit, the tile
//                       in which the sequence covering the unit's first byte lives.
//   3. k_lz4_pd_units   one wave per unit (persistent workgroups drawing tickets): walks from its tile's entry to the sequence
//                       that covers its first output byte and decodes with lz4_decode_one<UNIT> what falls into [lo, hi).  A match
//                       that reaches below lo reads what the unit before wrote (offsets are below 64 KiB): it waits for that
//                       unit's flag -- tickets are drawn unit-major, so whatever a unit waits for was drawn before it and is
//                       running or done.  Payloads of this library's encoder never wait (matches stay inside 64 KiB groups);
//                       a payload with a sliding window decodes as a chain of units, about as fast as the serial decoder.
//   4. k_lz4_pd_finish  result per block: the size the link pass found if every unit agreed, else the error value.
// Every check of the serial decoder is made by the unit that executes the sequence (with the true positions), the positional ones
// also by the link pass; the results (size or error) are the serial decoder's.
// ---------------------------------------------------------------------------------------------------
struct PdState
{
    uint32_t total; // decoded size of the block
    uint32_t nunits;
    uint32_t err;    // link pass or a unit found damage
    uint32_t end_ok; // the last unit reached the proper end
};

__device__ __forceinline__ uint32_t pd_block_of(const PdBlock* __restrict__ blocks, uint32_t nblocks, uint32_t x, bool by_tile)
{
    uint32_t lo = 0, hi = nblocks; // last block whose base <= x
    while (hi - lo > 1)
    {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((by_tile ? blocks[mid].tile_base : blocks[mid].unit_base) <= x)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Where the first sequence of every block ends: a block of incompressible data is ONE sequence (a token, 32 K length bytes, the
// literals); tiles that lie inside it need no walk.
__global__ __launch_bounds__(64) void k_lz4_pd_first(const uint8_t* __restrict__ src, const PdBlock* __restrict__ blocks, uint32_t nblocks,
                                                     uint32_t* __restrict__ first_end)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks)
        return;
    const PdBlock blk = blocks[b];
    PdReader r;
    r.init(src + blk.src_off, blk.size, s_in, threadIdx.x);
    int64_t next = 0;
    uint64_t out = 0;
    uint32_t lp = 0, ll = 0;
    const int kind = pd_hop(r, 0, blk.dst_cap, next, out, &lp, &ll);
    if (threadIdx.x == 0)
    {
        first_end[3 * b] = kind == 2 ? 0u : (uint32_t)next;
        first_end[3 * b + 1] = kind == 2 ? 0u : lp; // where its literals are in the payload ...
        first_end[3 * b + 2] = kind == 2 ? 0u : ll; // ... and how many: a unit that lies inside them is a plain copy
    }
}

__global__ __launch_bounds__(64) void k_lz4_pd_tiles(const uint8_t* __restrict__ src, const PdBlock* __restrict__ blocks, uint32_t nblocks,
                                                     uint32_t ntiles, PdTile* __restrict__ tiles, const uint32_t* __restrict__ first_end)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN];
    const uint32_t t = blockIdx.x;
    if (t >= ntiles)
        return;
    const int lane = threadIdx.x;
    const uint32_t b = pd_block_of(blocks, nblocks, t, true);
    const PdBlock blk = blocks[b];
    const uint32_t j = t - blk.tile_base;
    if (j > 0 && ((uint64_t)j + 1) * PD_TILE <= first_end[3 * b]) // inside the block's first sequence: no token of the chain is here
    {
        if (threadIdx.x == 0)
            tiles[t] = PdTile{PD_NONE, 0u, 0u, 0u};
        return;
    }
    PdReader r;
    r.init(src + blk.src_off, blk.size, s_in, lane);
    const int64_t t0 = (int64_t)j * PD_TILE;
    const int64_t t1 = t0 + PD_TILE < (int64_t)blk.size ? t0 + PD_TILE : (int64_t)blk.size;
    const PdTile rec = pd_walk_tile(r, j == 0 ? 0 : t0 - PD_RUNIN, t0, t1, blk.dst_cap);
    if (lane == 0)
        tiles[t] = rec;
}

// Second guess per tile.  Where a tile's chain leaves it (rec.exit) is, if that chain was the true one by then, where the true chain
// enters the tile it lands in; when that tile's own guess recorded another entry -- it began inside a long literal run, or the
// data kept two chains apart for more than the run-in -- the tile is walked again from there, here, by one wave per case and all
// cases at once, instead of one after the other by the link pass (a lone wave needs ~100 us per tile).  The result goes to the
// tile's ALTERNATIVE record (first come, first served).
__global__ __launch_bounds__(64) void k_lz4_pd_patch(const uint8_t* __restrict__ src, const PdBlock* __restrict__ blocks, uint32_t nblocks,
                                                     uint32_t ntiles, const PdTile* __restrict__ tiles, PdTile* __restrict__ alt,
                                                     uint32_t* __restrict__ claim)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN];
    const uint32_t t = blockIdx.x;
    if (t >= ntiles)
        return;
    const PdTile mine = tiles[t];
    if (mine.entry == PD_NONE || mine.flags)
        return;
    const PdBlock blk = blocks[pd_block_of(blocks, nblocks, t, true)];
    const uint32_t j = mine.exit / PD_TILE;
    if (j >= blk.ntiles || tiles[blk.tile_base + j].entry == mine.exit)
        return;
    uint32_t won = 0;
    if (threadIdx.x == 0)
        won = atomicCAS(&claim[blk.tile_base + j], 0u, 1u) == 0u ? 1u : 0u;
    if (!__builtin_amdgcn_readfirstlane(won))
        return;
    PdReader r;
    r.init(src + blk.src_off, blk.size, s_in, threadIdx.x);
    const int64_t t0 = (int64_t)j * PD_TILE;
    const int64_t t1 = t0 + PD_TILE < (int64_t)blk.size ? t0 + PD_TILE : (int64_t)blk.size;
    const PdTile rec = pd_walk_tile(r, mine.exit, t0, t1, blk.dst_cap);
    if (threadIdx.x == 0)
        alt[blk.tile_base + j] = rec;
}

__global__ __launch_bounds__(64) void k_lz4_pd_link(const uint8_t* __restrict__ src, const PdBlock* __restrict__ blocks, uint32_t nblocks,
                                                    PdTile* __restrict__ tiles, const PdTile* __restrict__ alt,
                                                    uint32_t* __restrict__ tile_op, uint32_t* __restrict__ unit_tile,
                                                    PdState* __restrict__ state, uint32_t* __restrict__ stats)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks)
        return;
    const int lane = threadIdx.x;
    const PdBlock blk = blocks[b];
    PdReader r;
    r.init(src + blk.src_off, blk.size, s_in, lane);
    PdTile* const bt = tiles + blk.tile_base;
    const uint64_t cap = blk.dst_cap;
    int64_t p = 0;
    uint64_t op = 0;
    uint32_t next_unit = 0, err = 0, rewalks = 0;
    long long walk_cycles = 0;
    const long long k0 = stats ? clock64() : 0;
    // 64 tile records at a time, one per lane
    uint32_t cj = PD_NONE;
    PdTile mine{PD_NONE, 0u, 0u, 0u};
    for (;;)
    {
        const uint32_t j = (uint32_t)(p / PD_TILE);
        if (j >= blk.ntiles) // a chain may only end through a PD_FINAL record
        {
            err = 1;
            break;
        }
        if (cj == PD_NONE || j < cj || j >= cj + 64u)
        {
            cj = j;
            mine = cj + (uint32_t)lane < blk.ntiles ? bt[cj + (uint32_t)lane] : PdTile{PD_NONE, 0u, 0u, 0u};
        }
        const int src_lane = (int)(j - cj);
        PdTile rec;
        rec.entry = __builtin_amdgcn_readlane(mine.entry, src_lane);
        rec.exit = __builtin_amdgcn_readlane(mine.exit, src_lane);
        rec.out = __builtin_amdgcn_readlane(mine.out, src_lane);
        rec.flags = __builtin_amdgcn_readlane(mine.flags, src_lane);
        if (rec.entry != (uint32_t)p)
        {
            const PdTile second = alt[blk.tile_base + j]; // (all lanes read the same record)
            if (second.entry == (uint32_t)p)
            {
                rec = second;
                if (lane == 0)
                    bt[j] = rec; // the units start from this record
            }
        }
        if (rec.entry != (uint32_t)p)
        {
            // neither guess was the true entry: walk the tile from where the chain really enters it
            const int64_t t0 = (int64_t)j * PD_TILE;
            const int64_t t1 = t0 + PD_TILE < (int64_t)blk.size ? t0 + PD_TILE : (int64_t)blk.size;
            const long long c0 = stats ? clock64() : 0;
            rec = pd_walk_tile(r, p, t0, t1, cap);
            if (lane == 0)
                bt[j] = rec; // the units start from this record
            ++rewalks;
            if (stats)
                walk_cycles += clock64() - c0;
        }
        if (lane == 0)
            tile_op[blk.tile_base + j] = (uint32_t)op;
        // units whose first byte is produced inside this tile
        const uint64_t op_end = op + rec.out;
        while ((uint64_t)next_unit * PD_UNIT < op_end && next_unit < blk.nunits_cap)
        {
            if (lane == 0)
                unit_tile[blk.unit_base + next_unit] = j;
            ++next_unit;
        }
        op = op_end;
        if ((rec.flags & PD_INVALID) || op > cap)
        {
            err = 1;
            break;
        }
        if (rec.flags & PD_FINAL)
            break;
        if ((int64_t)rec.exit <= p) // cannot happen for a walked record; never loop on a damaged table
        {
            err = 1;
            break;
        }
        p = rec.exit;
    }
    if (lane == 0)
    {
        if (next_unit == 0 && blk.nunits_cap) // an empty result still has one unit: it checks the end of the payload
        {
            unit_tile[blk.unit_base] = 0;
            next_unit = 1;
        }
        PdState st;
        st.total = (uint32_t)op;
        st.nunits = err ? 0u : next_unit;
        st.err = err;
        st.end_ok = 0;
        state[b] = st;
        if (stats)
        {
            atomicAdd(&stats[0], rewalks);
            atomicAdd(&stats[1], blk.ntiles);
            atomicAdd(&stats[2], (uint32_t)(walk_cycles >> 10));
            atomicAdd(&stats[3], (uint32_t)((clock64() - k0) >> 10));
        }
    }
}

// ticket -> (unit row, block): blocks sorted by descending unit capacity, row k holds the blocks with more than k units
struct PdTickets
{
    const uint32_t* order;    // block indices, most units first
    const uint32_t* row_base; // first ticket of row k, k = 0 .. rows (row_base[rows] = number of tickets)
    uint32_t rows;
    uint32_t total;
    uint32_t uniform; // != 0: every block has `rows` units (ticket = row * uniform + block)
};

#ifdef LTHIP_ABLATIONS
#include "ablations/k_lz4_pd_units.inc"
#endif


// ---------------------------------------------------------------------------------------------------
// Units given up by k_lz4_pd_units because a match reaches into the unit before (a payload with a sliding window: the reference's
// parse), executed on ORIGINS (origin_exec.h): k_lz4_po_trace, one wave per such unit, all of them at once -- the unit's sequences
// from the one that covers its first byte, clipped to [lo, hi), 64 at a time: a literal's origin is its position in the payload, a
// source byte below lo becomes its output position -- then k_lz4_po_gather, launch k for unit k of every block.  The checks are
// the serial decoder's (lz4.c:1979-2250), made with the true positions by the unit that holds the sequence's end, as in
// lz4_decode_one<UNIT>.
// ---------------------------------------------------------------------------------------------------
// BYTES (round 3, last part): the same walk for EVERY unit, executing bytes (zo_batch_bytes) straight into the block's output -- the
// unit decoder of the block-parallel path since then (k_lz4_pd_units / lz4_decode_one<UNIT> stays as LTHIP_LZ4_PX=0): a unit that meets
// a source below its start stops and is marked for the origin pass, which is this kernel with BYTES = false.
template <bool BYTES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lz4_po_trace(const uint8_t* __restrict__ src, const PdBlock* __restrict__ blocks, uint32_t nblocks,
                                                    uint32_t unit0, const PdTile* __restrict__ tiles, const uint32_t* __restrict__ tile_op,
                                                    const uint32_t* __restrict__ unit_tile, PdState* __restrict__ state,
                                                    uint32_t* __restrict__ unit_mode, uint32_t* __restrict__ org_arena, uint8_t* __restrict__ dst,
                                                    const uint32_t* __restrict__ first, uint32_t* __restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN];
    __shared__ uint32_t s_ia[64], s_om[64];
    __shared__ uint32_t s_ll[96], s_lp[96], s_ml[96], s_off[96], s_op[96]; // the batch being collected
    const uint32_t u = unit0 + blockIdx.x;
    if (!BYTES && unit_mode[u] != 1u)
        return;
    const int lane = threadIdx.x;
    const uint32_t b = pd_block_of(blocks, nblocks, u, false);
    const PdBlock blk = blocks[b];
    const PdState st = state[b];
    const uint32_t k = u - blk.unit_base;
    if (BYTES && k >= st.nunits)
        return;
    if constexpr (BYTES)
    {
        const uint32_t f_pos = first[3 * b + 1], f_len = first[3 * b + 2];
        if (k + 1u < st.nunits && (uint64_t)(k + 1u) * PD_UNIT <= f_len)
        {
            // the whole unit lies inside the literals of the block's first sequence (incompressible data is ONE sequence): a plain copy;
            // the sequence itself is checked by the unit in which it ends
            typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
            const uint8_t* from = src + blk.src_off + f_pos + (uint64_t)k * PD_UNIT;
            uint8_t* to = dst + blk.dst_off + (uint64_t)k * PD_UNIT;
            for (uint32_t v0 = 0; v0 < PD_UNIT / 16u; v0 += 256u)
            {
                u32x4_a1 q[4];
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    q[w] = *reinterpret_cast<const u32x4_a1*>(from + 16u * (v0 + (uint32_t)(w * 64 + lane)));
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    *reinterpret_cast<u32x4_a1*>(to + 16u * (v0 + (uint32_t)(w * 64 + lane))) = q[w];
            }
            return;
        }
    }
    const bool last = k + 1u == st.nunits;
    const int32_t n = (int32_t)blk.size;
    const int64_t cap = blk.dst_cap;
    const int64_t lo = (int64_t)k * PD_UNIT, hi = last ? cap : lo + (int64_t)PD_UNIT;
    uint32_t* org = BYTES ? nullptr : org_arena + (uint64_t)blockIdx.x * PD_UNIT; // origin of output byte lo + q: org[q]
    uint8_t* out = BYTES ? dst + blk.dst_off + lo : nullptr;                      // ... or the byte itself
    bool given_up = false; // BYTES: a source below lo
    const uint32_t j0 = unit_tile[u];
    int64_t ip = k == 0u ? 0 : (int64_t)tiles[blk.tile_base + j0].entry;
    int64_t op = k == 0u ? 0 : (int64_t)tile_op[blk.tile_base + j0];
    PdReader r;
    r.init(src + blk.src_off, blk.size, s_in, lane);
    bool bad = false, ended = false;
    if (lo > 0)
    {
        // positions only, up to the sequence that covers lo (what ends at or before lo is an earlier unit's, which also checks it)
        const PdPos at = pd_walk<true>(r, ip, (uint64_t)op, INT64_MAX, (uint64_t)lo, (uint64_t)cap);
        ip = at.ip;
        op = (int64_t)at.op;
    }
    uint32_t cnt = 0; // sequences collected
    // the batch: every lane clips its sequence to [lo, hi); the origins of all of them in one go
    auto run_batch = [&]() {
        const bool act = (uint32_t)lane < cnt;
        const int64_t sop = act ? (int64_t)s_op[lane] + (lo & ~0xFFFFFFFFll) : hi; // (positions are kept as their low 32 bits)
        const uint32_t lit = act ? s_ll[lane] : 0u, mlen = act ? s_ml[lane] : 0u;
        const int64_t lit_end = sop + (int64_t)lit;
        const int64_t a0 = sop > lo ? sop : lo, a1 = lit_end < hi ? lit_end : hi;
        const int64_t m0 = lit_end > lo ? lit_end : lo, m1 = lit_end + (int64_t)mlen < hi ? lit_end + (int64_t)mlen : hi;
        const uint32_t c_ll = act && a1 > a0 ? (uint32_t)(a1 - a0) : 0u, c_ml = act && m1 > m0 ? (uint32_t)(m1 - m0) : 0u;
        const uint32_t c_li = (act ? s_lp[lane] : 0u) + (uint32_t)(a0 - sop);
        const uint32_t offv = act ? s_off[lane] : 1u;
        const int64_t first = (int64_t)s_op[0] + (lo & ~0xFFFFFFFFll);
        const uint32_t pos = (uint32_t)((first > lo ? first : lo) - lo);
        const uint32_t i_a = zo_scan_incl(c_ll + c_ml);
        __builtin_amdgcn_wave_barrier();
        if constexpr (BYTES)
        {
            if (!zo_batch_bytes(out, src + blk.src_off, lane, act, c_ll, c_li, c_ml, offv, i_a, pos, s_ia, s_om))
                given_up = true;
        }
        else
            zo_batch(org, (uint32_t)lo, lane, act, c_ll, c_li, c_ml, offv, i_a, pos, s_ia, s_om);
        cnt = 0;
    };
    while (!bad && !ended && op < hi && !given_up)
    {
        if (cnt > 42u) // (a window holds at most 21 sequences, the slow path adds one)
            run_batch();
        if (given_up)
            break;
        if (ip >= n)
        {
            bad = true;
            break;
        }
        // ---- 64 positions at a time (pd_walk's scheme, with offsets): every lane reads its byte of the window as a token; a sequence
        // without length bytes (or with ONE match-length byte below 255) that lies inside the window is "simple" ----
        if (r.w0 != ip)
        {
            r.w0 = ip - 64; // force a reseed at exactly ip
            (void)r.byte_at(ip);
        }
        const uint32_t w = r.w;
        const uint32_t litl = w >> 4, mlcl = w & 15u;
        const int32_t pl = (int32_t)ip + lane;
        const bool big = litl == 15u;
        const uint32_t after = (uint32_t)__shfl((int)w, (lane + 1) & 63, 64);
        const uint32_t lit = big ? 15u + after : litl, hdr = big ? 2u : 1u;
        const uint32_t e = (uint32_t)lane + hdr + lit + 2u; // where a match-length byte would be, relative to the window
        const uint32_t ext = (uint32_t)__shfl((int)w, (int)(e & 63u), 64);
        const uint32_t offv = (uint32_t)__shfl((int)w, (int)((e - 2u) & 63u), 64) | ((uint32_t)__shfl((int)w, (int)((e - 1u) & 63u), 64) << 8);
        const bool one = mlcl == 15u;
        bool simple = pl + (int32_t)(hdr + lit) <= n - 8 && e <= 64u; // "ip + len > n - 8" ends the payload; the offset inside the window
        if (big) // its length byte: "ip >= n - 15" before it, "ip > n - 15" after it
            simple = simple && lane < 63 && after != 255u && pl + 2 <= n - 15;
        if (one)
            simple = simple && e < 64u && pl + (int32_t)(hdr + lit) + 2 < n - 4 && ext != 255u;
        const uint32_t mlen = mlcl + 4u + (one ? ext : 0u);
        const uint32_t outv = lit + mlen, nxtv = e + (one ? 1u : 0u);
        const uint64_t ok = __builtin_amdgcn_ballot_w64(simple);
        uint64_t chain = 0;
        uint32_t cur = 0;
        while (cur < 64u && ((ok >> cur) & 1ull))
        {
            chain |= 1ull << cur;
            cur = __builtin_amdgcn_readlane(nxtv, (int)cur);
        }
        if (chain)
        {
            // where each of them starts, and the checks that need it ("op + len > cap - 12", the offset, "op + ml > cap - 5"); the first
            // one that fails them (or starts at or past hi) ends the chain: the slow path below says what it is
            const bool inch = (chain >> lane) & 1ull;
            const uint32_t pre = zo_scan_incl(inch ? outv : 0u);
            const int64_t sop = op + (int64_t)(pre - (inch ? outv : 0u)), lit_end = sop + (int64_t)lit;
            const bool good = inch && sop < hi && lit_end <= cap - 12 && lit_end + (int64_t)mlen <= cap - 5 && offv != 0u && (int64_t)offv <= lit_end;
            const uint64_t failed = chain & ~__builtin_amdgcn_ballot_w64(good);
            const uint64_t kept = failed ? chain & ((1ull << __builtin_ctzll(failed)) - 1ull) : chain;
            if ((kept >> lane) & 1ull)
            {
                const uint32_t at = cnt + (uint32_t)__builtin_popcountll(kept & ((1ull << lane) - 1ull));
                s_ll[at] = lit;
                s_lp[at] = (uint32_t)pl + hdr;
                s_ml[at] = mlen;
                s_off[at] = offv;
                s_op[at] = (uint32_t)sop;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cnt += (uint32_t)__builtin_popcountll(kept);
            if (failed)
            {
                const int fb = __builtin_ctzll(failed);
                op = op + (int64_t)((uint32_t)__builtin_amdgcn_readlane((int)(pre - outv), fb)); // where that one starts
                ip += fb;
                if (op >= hi)
                    break;
            }
            else
            {
                op += (int64_t)((uint32_t)__builtin_amdgcn_readlane((int)pre, 63));
                ip += cur;
                continue;
            }
        }
        else if (cur == 0u && (ok & 1ull) == 0ull)
            ; // the token at ip is not a simple one
        // ---- one sequence the careful way (length bytes, the end of the payload, damage) ----
        {
            int64_t next = 0;
            uint64_t out = 0;
            uint32_t lp = 0, len = 0;
            const int kind = pd_hop(r, ip, (uint64_t)cap, next, out, &lp, &len);
            if (kind == 2)
            {
                bad = true;
                break;
            }
            const int64_t lit_end = op + (int64_t)len;
            if (kind == 1 || lit_end > cap - 12)
            {
                // the last sequence of the payload: literals up to its very end (or damage)
                if (kind != 1 || lit_end > cap || lit_end < lo || (!last && lit_end < hi))
                {
                    bad = true;
                    break;
                }
                ended = true;
            }
            const int64_t m = ended ? 0 : (int64_t)out - (int64_t)len; // match length
            uint32_t o = 1;
            if (!ended)
            {
                o = r.byte_at((int64_t)lp + len) | (r.byte_at((int64_t)lp + len + 1) << 8);
                if (o == 0u || (int64_t)o > lit_end || lit_end + m > cap - 5)
                {
                    bad = true;
                    break;
                }
            }
            if (lane == 0)
            {
                s_ll[cnt] = len;
                s_lp[cnt] = lp;
                s_ml[cnt] = (uint32_t)m;
                s_off[cnt] = o;
                s_op[cnt] = (uint32_t)op;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            ++cnt;
            op = lit_end + m;
            ip = next;
        }
    }
    if (!bad && cnt && !given_up)
        run_batch();
    if (lane == 0)
    {
        if (BYTES && given_up)
        {
            unit_mode[u] = 1u; // a match reaches into the unit before: executed on origins (this kernel again, BYTES = false)
            atomicAdd(&counters[2], 1u);
        }
        else if (bad)
        {
            atomicOr(&state[b].err, 1u);
            unit_mode[u] = 2u; // (nothing to gather)
        }
        else if (ended && last)
            atomicOr(&state[b].end_ok, 1u);
    }
}

// unit k of the blocks [g0, g0 + gridDim.y): 256 threads x 16 bytes per workgroup
__global__ __launch_bounds__(256) void k_lz4_po_gather(const uint8_t* __restrict__ src, const PdBlock* __restrict__ blocks, uint32_t g0, uint32_t k,
                                                      uint32_t unit0, const PdState* __restrict__ state, const uint32_t* __restrict__ unit_mode,
                                                      const uint32_t* __restrict__ org_arena, uint8_t* __restrict__ dst)
{
    const uint32_t b = g0 + blockIdx.y;
    const PdBlock blk = blocks[b];
    const PdState st = state[b];
    if (k >= st.nunits || k >= blk.nunits_cap)
        return;
    const uint32_t u = blk.unit_base + k;
    if (unit_mode[u] != 1u)
        return;
    const uint32_t lo = k * PD_UNIT;
    const uint32_t len = st.total > lo ? (st.total - lo < PD_UNIT ? st.total - lo : PD_UNIT) : 0u;
    const uint32_t q = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (q >= len)
        return;
    const uint32_t nq = len - q < 16u ? len - q : 16u;
    const uint32_t* org = org_arena + (uint64_t)(u - unit0) * PD_UNIT + q;
    const uint8_t* lits = src + blk.src_off;
    uint8_t* out = dst + blk.dst_off;
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    uint32_t o[16], w[4] = {0, 0, 0, 0};
    if (nq == 16u)
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
            o[i] = i < nq ? org[i] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i)
    {
        const uint32_t x = (o[i] & ZO_FLAG) ? out[o[i] & ~ZO_FLAG] : lits[o[i]];
        w[i >> 2] |= x << (8u * (i & 3u));
    }
    uint8_t* to = out + lo + q;
    if (nq == 16u)
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
        for (uint32_t i = 0; i < nq; ++i)
            to[i] = (uint8_t)(w[i >> 2] >> (8u * (i & 3u)));
}

__global__ void k_lz4_pd_finish(const PdBlock* __restrict__ blocks, uint32_t nblocks, const PdState* __restrict__ state,
                                uint32_t* __restrict__ out_sizes)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks)
        return;
    const PdState st = state[b];
    out_sizes[blocks[b].out_index] = (!st.err && st.end_ok) ? st.total : DEC_ERROR;
}

} // namespace

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
#ifdef LTHIP_DEC_PROF
extern "C" __attribute__((visibility("default"))) int lthip_dec_prof_dump(void)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dec_prof), sizeof(h)) != hipSuccess)
        return -1;
    fprintf(stderr, "lz4 decoder: total %.1f Mcycles (wave-summed), copy_match %.1f (incl. flush), flush %.1f, refills %llu, far matches %llu, "
                    "short sequences %llu, general sequences %llu, batches %llu with %llu sequences\n",
            h[0] / 1e6, h[2] / 1e6, h[3] / 1e6, h[8], h[9], h[10], h[11], h[12], h[13]);
    memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dec_prof), h, sizeof(h));
    return 0;
}
#endif

// The block-parallel path for the blocks listed in `idx` (see the comment above k_lz4_pd_tiles).
static int lz4_decompress_parallel(lthip_ctx* ctx, const void* d_src, const std::vector<uint32_t>& idx, const uint64_t* src_offsets,
                                   const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets, const uint32_t* dst_caps,
                                   uint32_t* d_out_sizes, uint32_t nobatch)
{
    const uint32_t nb = (uint32_t)idx.size();
    std::vector<PdBlock> hb(nb);
    uint64_t ntiles = 0, nunits = 0;
    uint32_t rows = 0;
    bool small = true, uniform = true;
    for (uint32_t i = 0; i < nb; ++i)
    {
        const uint32_t b = idx[i];
        PdBlock& pb = hb[i];
        pb.src_off = src_offsets[b];
        pb.dst_off = dst_offsets[b];
        pb.size = src_sizes[b];
        pb.dst_cap = dst_caps[b];
        pb.out_index = b;
        pb.tile_base = (uint32_t)ntiles;
        pb.ntiles = (src_sizes[b] + PD_TILE - 1) / PD_TILE;
        pb.unit_base = (uint32_t)nunits;
        pb.nunits_cap = (uint32_t)(((uint64_t)dst_caps[b] + PD_UNIT - 1) / PD_UNIT);
        pb.pad = 0;
        ntiles += pb.ntiles;
        nunits += pb.nunits_cap;
        rows = pb.nunits_cap > rows ? pb.nunits_cap : rows;
        uniform = uniform && pb.nunits_cap == hb[0].nunits_cap;
        small = small && src_sizes[b] < (1u << 30) && dst_caps[b] < (1u << 30);
    }
    if (ntiles > 0x7FFFFFF0ull || nunits > 0x7FFFFFF0ull)
        return lthip_fail(ctx, EINVAL, "lz4", "too many blocks in one decode call");
    // tickets: unit-major (every block's unit k before any block's unit k + 1), so that a unit only ever waits for lower tickets
    std::vector<uint32_t> order(nb), row_base((size_t)rows + 1, 0u);
    if (!uniform)
    {
        for (uint32_t i = 0; i < nb; ++i)
            order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return hb[x].nunits_cap > hb[y].nunits_cap; });
        std::vector<uint32_t> more((size_t)rows + 1, 0u); // more[k] = blocks with exactly k units, then suffix sums
        for (uint32_t i = 0; i < nb; ++i)
            ++more[hb[i].nunits_cap];
        uint32_t with_more = 0; // blocks with more than k units, for k = rows - 1 .. 0
        std::vector<uint32_t> count(rows, 0u);
        for (uint32_t k = rows; k-- > 0;)
        {
            with_more += more[(size_t)k + 1];
            count[k] = with_more;
        }
        for (uint32_t k = 0; k < rows; ++k)
            row_base[(size_t)k + 1] = row_base[k] + count[k];
    }
    // device tables: [PdTile x ntiles][tile_op x ntiles][unit_tile x nunits][state x nb][order x nb][row_base x rows+1][done x nunits][counters x 8][first_end, first literals' position and count x nb][alt PdTile x ntiles][claim x ntiles]
    const size_t o_tiles = 0, o_top = o_tiles + sizeof(PdTile) * ntiles, o_ut = o_top + 4 * ntiles, o_state = o_ut + 4 * nunits,
                 o_order = o_state + sizeof(PdState) * nb, o_rows = o_order + 4 * (size_t)nb, o_done = o_rows + 4 * ((size_t)rows + 1),
                 o_cnt = o_done + 4 * nunits, o_first = o_cnt + 32, o_alt = (o_first + 12 * (size_t)nb + 15) & ~(size_t)15, o_claim = o_alt + sizeof(PdTile) * ntiles,
                 o_mode = o_claim + 4 * ntiles, o_end = o_mode + 4 * nunits;
    void *tab, *blk;
    int err = lthip_scratch(ctx, S_LZ4_STREAM, o_end, &tab);
    if (!err)
        err = lthip_scratch(ctx, S_LZ4_META, sizeof(PdBlock) * (size_t)nb, &blk);
    if (!err)
        err = lthip_stage_upload(ctx, blk, hb.data(), sizeof(PdBlock) * (size_t)nb, ctx->stream);
    if (!err && !uniform)
    {
        err = lthip_stage_upload(ctx, (uint8_t*)tab + o_order, order.data(), 4 * (size_t)nb, ctx->stream);
        if (!err)
            err = lthip_stage_upload(ctx, (uint8_t*)tab + o_rows, row_base.data(), 4 * ((size_t)rows + 1), ctx->stream);
    }
    if (err)
        return err;
    uint8_t* t8 = (uint8_t*)tab;
    LTHIP_CHECK(ctx, hipMemsetAsync(t8 + o_done, 0, o_first - o_done, ctx->stream));
    LTHIP_CHECK(ctx, hipMemsetAsync(t8 + o_alt, 0xFF, sizeof(PdTile) * ntiles, ctx->stream)); // entry = PD_NONE
    LTHIP_CHECK(ctx, hipMemsetAsync(t8 + o_claim, 0, 4 * ntiles + 4 * nunits, ctx->stream)); // flags and counters: zero before every launch
    const PdBlock* d_blocks = (const PdBlock*)blk;
    PdTile* d_tiles = (PdTile*)(t8 + o_tiles);
    uint32_t* d_top = (uint32_t*)(t8 + o_top);
    uint32_t* d_ut = (uint32_t*)(t8 + o_ut);
    PdState* d_state = (PdState*)(t8 + o_state);
    uint32_t* d_done = (uint32_t*)(t8 + o_done);
    uint32_t* d_cnt = (uint32_t*)(t8 + o_cnt);
    uint32_t* d_first = (uint32_t*)(t8 + o_first);
    LTHIP_ABLATION_ENV(env_stats, "LTHIP_LZ4_PD_STATS");
    LTHIP_ABLATION_ENV(env_trace, "LTHIP_LZ4_PD_TRACE"); // debugging: synchronize and report after every launch
    const bool stats = env_stats.get() >= 0, trace = env_trace.get() >= 0;
#define PD_TRACE(what)                                                                 \
    do                                                                                 \
    {                                                                                  \
        if (trace)                                                                     \
        {                                                                              \
            const hipError_t e__ = lthip_stream_wait(ctx);                  \
            fprintf(stderr, "lz4 parallel decode: %s -> %s\n", what, hipGetErrorString(e__)); \
        }                                                                              \
    } while (0)
    hipLaunchKernelGGL(k_lz4_pd_first, dim3(nb), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, nb, d_first);
    LTHIP_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_lz4_pd_tiles, dim3((uint32_t)ntiles), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, nb, (uint32_t)ntiles, d_tiles,
                       d_first);
    LTHIP_LAUNCH_CHECK(ctx);
    PD_TRACE("tiles");
    hipLaunchKernelGGL(k_lz4_pd_patch, dim3((uint32_t)ntiles), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, nb, (uint32_t)ntiles,
                       (const PdTile*)d_tiles, (PdTile*)(t8 + o_alt), (uint32_t*)(t8 + o_claim));
    LTHIP_LAUNCH_CHECK(ctx);
    PD_TRACE("patch");
    hipLaunchKernelGGL(k_lz4_pd_link, dim3(nb), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, nb, d_tiles, (const PdTile*)(t8 + o_alt), d_top, d_ut,
                       d_state, stats ? d_cnt + 4 : nullptr);
    LTHIP_LAUNCH_CHECK(ctx);
    PD_TRACE("link");
    PdTickets tk;
    tk.order = (const uint32_t*)(t8 + o_order);
    tk.row_base = (const uint32_t*)(t8 + o_rows);
    tk.rows = rows;
    tk.total = uniform ? rows * nb : row_base[rows];
    tk.uniform = uniform ? nb : 0u;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    LTHIP_ABLATION_ENV(env_per_cu, "LTHIP_LZ4_PD_WG_PER_CU");
    const uint32_t per_cu = (uint32_t)(env_per_cu.get() > 0 ? env_per_cu.get() : 16); // 10 KiB of LDS each
    const uint64_t resident = (uint64_t)ncu * per_cu;
    const uint32_t grid = (uint32_t)(tk.total < resident ? tk.total : resident);
    volatile uint32_t* dbg = nullptr;
    if (trace)
    {
        void* hp = nullptr;
        LTHIP_CHECK(ctx, lthip_hip_host_malloc(&hp, 32 * (size_t)grid, hipHostMallocMapped));
        memset(hp, 0, 32 * (size_t)grid);
        dbg = (volatile uint32_t*)hp; // host-visible progress words, freed below once the kernel is through
    }
    // LTHIP_LZ4_PD_WAIT=1: round 2's way -- a unit whose matches reach into the unit before WAITS for it (a chain of units)
    LTHIP_ABLATION_ENV(env_wait, "LTHIP_LZ4_PD_WAIT");
    const bool wait_mode = env_wait.get() > 0;
    uint32_t* d_mode = wait_mode ? nullptr : (uint32_t*)(t8 + o_mode);
    LTHIP_ABLATION_ENV(env_px, "LTHIP_LZ4_PX"); // 0: the unit decoder of round 2 (lz4_decode_one<UNIT> through an LDS ring: ablations/k_lz4_pd_units.inc)
    const bool px = d_mode && env_px.get() != 0;
#ifndef LTHIP_ABLATIONS
    (void)d_done;
    (void)grid;
    (void)dbg;
    (void)small;
    (void)nobatch;
#endif
    if (px)
        hipLaunchKernelGGL(k_lz4_po_trace<true>, dim3((uint32_t)nunits), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, nb, 0u,
                           (const PdTile*)d_tiles, (const uint32_t*)d_top, (const uint32_t*)d_ut, d_state, d_mode, (uint32_t*)nullptr,
                           (uint8_t*)d_dst, (const uint32_t*)d_first, d_cnt);
#ifdef LTHIP_ABLATIONS
    else if (small)
        hipLaunchKernelGGL(k_lz4_pd_units<int32_t>, dim3(grid), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, (uint8_t*)d_dst, d_tiles,
                           d_top, d_ut, d_state, d_done, d_cnt, tk, nobatch, dbg, (const uint32_t*)d_first, d_mode);
    else
        hipLaunchKernelGGL(k_lz4_pd_units<int64_t>, dim3(grid), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, (uint8_t*)d_dst, d_tiles,
                           d_top, d_ut, d_state, d_done, d_cnt, tk, nobatch, dbg, (const uint32_t*)d_first, d_mode);
#endif
    LTHIP_LAUNCH_CHECK(ctx);
    if (d_mode)
    {
        // units that were given up (a payload with a sliding window): on origins, as many blocks at a time as the arena's budget
        // allows (4 bytes per byte of output; LTHIP_ORIGIN_MIB, default: lthip_origin_budget_mib).  The one place where this call waits for the device.
        uint32_t given_up = 0;
        LTHIP_CHECK(ctx, hipMemcpyAsync(&given_up, d_cnt + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        if (given_up)
        {
            const uint64_t budget_units = (lthip_origin_budget_mib() << 20) / ((uint64_t)PD_UNIT * 4u);
            uint64_t most = 0;
            for (uint32_t g0 = 0; g0 < nb;)
            {
                uint64_t units = hb[g0].nunits_cap;
                uint32_t g1 = g0 + 1;
                while (g1 < nb && units + hb[g1].nunits_cap <= budget_units)
                    units += hb[g1++].nunits_cap;
                most = units > most ? units : most;
                g0 = g1;
            }
            void* d_org;
            if ((err = lthip_scratch(ctx, S_Z_ORG, (size_t)most * PD_UNIT * 4u + 256, &d_org)))
                return err;
            for (uint32_t g0 = 0; g0 < nb;)
            {
                uint64_t units = hb[g0].nunits_cap;
                uint32_t g1 = g0 + 1, rows_g = hb[g0].nunits_cap;
                while (g1 < nb && units + hb[g1].nunits_cap <= budget_units)
                {
                    rows_g = hb[g1].nunits_cap > rows_g ? hb[g1].nunits_cap : rows_g;
                    units += hb[g1++].nunits_cap;
                }
                hipLaunchKernelGGL(k_lz4_po_trace<false>, dim3((uint32_t)units), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, nb, hb[g0].unit_base,
                                   (const PdTile*)d_tiles, (const uint32_t*)d_top, (const uint32_t*)d_ut, d_state, d_mode, (uint32_t*)d_org,
                                   (uint8_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
                LTHIP_LAUNCH_CHECK(ctx);
                for (uint32_t k = 1; k < rows_g; ++k) // (unit 0 has nothing before it)
                    hipLaunchKernelGGL(k_lz4_po_gather, dim3(PD_UNIT / 4096u, g1 - g0), dim3(256), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, g0, k,
                                       hb[g0].unit_base, (const PdState*)d_state, (const uint32_t*)d_mode, (const uint32_t*)d_org, (uint8_t*)d_dst);
                LTHIP_LAUNCH_CHECK(ctx);
                g0 = g1;
            }
            PD_TRACE("origins");
        }
    }
    if (trace)
    {
        for (int sec = 0; sec < 5 && hipStreamQuery(ctx->stream) == hipErrorNotReady; ++sec)
        {
            struct timespec ts = {1, 0};
            nanosleep(&ts, nullptr);
            fprintf(stderr, "units after %d s:", sec + 1);
            for (uint32_t w = 0; w < grid && w < 8; ++w)
                fprintf(stderr, " [t%u k%u ip0 %u op0 %u | pre %x %x | main %x %u]", dbg[w * 8], dbg[w * 8 + 1], dbg[w * 8 + 2], dbg[w * 8 + 3],
                        dbg[w * 8 + 4], dbg[w * 8 + 5], dbg[w * 8 + 6], dbg[w * 8 + 7]);
            fprintf(stderr, "\n");
        }
    }
    PD_TRACE("units");
    if (dbg)
        (void)hipHostFree((void*)dbg); // (PD_TRACE synchronised the stream)
    hipLaunchKernelGGL(k_lz4_pd_finish, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, d_blocks, nb, d_state, d_out_sizes);
    LTHIP_LAUNCH_CHECK(ctx);
    if (stats)
    {
        uint32_t h[8] = {0};
        LTHIP_CHECK(ctx, hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
        fprintf(stderr, "lz4 parallel decode: %u blocks, %llu tiles (%u walked again by the link pass), %llu units, %u tickets, timeouts %u\n", nb,
                (unsigned long long)ntiles, h[4], (unsigned long long)nunits, tk.total, h[1]);
        fprintf(stderr, "   link pass: %u Kcycles in all, %u Kcycles walking tiles again (summed over the blocks)\n", h[7], h[6]);
    }
    return 0;
}

extern "C" int lthip_lz4_decompress_blocks(lthip_ctx* ctx, const void* d_src, uint32_t block_count, const uint64_t* src_offsets,
                                           const uint32_t* src_sizes, void* d_dst, const uint64_t* dst_offsets,
                                           const uint32_t* dst_caps, uint32_t* d_out_sizes)
{
    if (!ctx || !d_out_sizes || (block_count && (!src_offsets || !src_sizes || !dst_offsets || !dst_caps)))
        return EINVAL;
    if (block_count == 0)
        return 0;
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    LTHIP_ABLATION_ENV(env_plain, "LTHIP_LZ4_PLAIN_DECODER");     // every byte through global memory (ablations/k_lz4_decode_plain.inc)
    LTHIP_ABLATION_ENV(env_serial, "LTHIP_LZ4_SERIAL_DECODER");   // one wave per block for every block
    LTHIP_ABLATION_ENV(env_nobatch, "LTHIP_LZ4_NO_BATCH_DECODE"); // one sequence per step only
    const bool plain = env_plain.get() >= 0, serial_only = env_serial.get() >= 0;
    const uint32_t nobatch = env_nobatch.get() >= 0 ? 1u : 0u;
    // blocks of at least two units go to the block-parallel path, the rest to the wave-per-block decoder
    std::vector<uint32_t> par;
    std::vector<Lz4Block> hb;
    hb.reserve(block_count);
    bool small = true;
    for (uint32_t b = 0; b < block_count; ++b)
    {
        if (src_sizes[b] > 0x7E000000u)
            return lthip_fail(ctx, EINVAL, "lz4", "payload larger than LZ4_MAX_INPUT_SIZE");
        if (!plain && !serial_only && dst_caps[b] >= 2u * PD_UNIT && dst_caps[b] <= 0x7E000000u && src_sizes[b] >= 64u)
        {
            par.push_back(b);
            continue;
        }
        Lz4Block x;
        x.src_off = src_offsets[b];
        x.dst_off = dst_offsets[b];
        x.size = src_sizes[b];
        x.dst_cap = dst_caps[b];
        x.out_index = b;
        x.pad = 0;
        hb.push_back(x);
        small = small && src_sizes[b] < (1u << 30) && dst_caps[b] < (1u << 30);
    }
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    int err = 0;
    if (!par.empty() && (err = lz4_decompress_parallel(ctx, d_src, par, src_offsets, src_sizes, d_dst, dst_offsets, dst_caps, d_out_sizes, nobatch)))
        return err;
    if (hb.empty())
        return 0;
    const uint32_t ns = (uint32_t)hb.size();
    void* p;
    if ((err = lthip_scratch(ctx, S_LZ4_BLOCKS, sizeof(Lz4Block) * (size_t)ns, &p)))
        return err;
    if ((err = lthip_stage_upload(ctx, p, hb.data(), sizeof(Lz4Block) * (size_t)ns, ctx->stream))) // no host stall
        return err;
    Lz4Block* d_blocks = (Lz4Block*)p;
#ifdef LTHIP_ABLATIONS
    if (plain)
        hipLaunchKernelGGL(k_lz4_decode, dim3(ns), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, ns, (uint8_t*)d_dst, d_out_sizes);
    else
#endif
    if (small)
        hipLaunchKernelGGL(k_lz4_decode_lds<int32_t>, dim3(ns), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, ns, (uint8_t*)d_dst,
                           d_out_sizes, nobatch);
    else
        hipLaunchKernelGGL(k_lz4_decode_lds<int64_t>, dim3(ns), dim3(64), 0, ctx->stream, (const uint8_t*)d_src, d_blocks, ns, (uint8_t*)d_dst,
                           d_out_sizes, nobatch);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
