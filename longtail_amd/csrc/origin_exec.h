// origin_exec.h -- LZ sequences executed on ORIGINS instead of bytes (round 3; shared by the zstd and the LZ4 decoder).
//
// A payload whose matches reach across the pieces it is decoded in (a zstd frame of the reference encoder: the window is the
// frame; an LZ4 block with a sliding window) is ONE chain of dependent copies: executed on bytes, every piece would wait for the end
// of the piece before it.  Executed on 32-bit origins -- "literal number i" or "byte p of an earlier piece" -- the pieces are
// independent: a match copies the origins of its source, a source byte below the piece's start is recorded as its position, chains
// inside the piece collapse as the trace goes.  A gather pass then fills piece k of every payload in launch k:
// out[q] = literal or out[p], with everything below piece k final by then.
#ifndef LTHIP_ORIGIN_EXEC_H
#define LTHIP_ORIGIN_EXEC_H

#include <stdint.h>

constexpr uint32_t ZO_FLAG = 0x80000000u; // origin: a byte of an earlier piece (its position below) -- else: the literal's index

// inclusive prefix sum over the 64 lanes with DPP moves (row shifts inside the rows of 16, then the two row broadcasts)
__device__ __forceinline__ uint32_t zo_scan_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return v;
}

// stores of this wave's lanes -> loads of this wave's lanes, through memory: the stores have to have left the wave (vmcnt), the
// CU's vector cache is written through and shared by whoever runs on the CU (agent scope would write the L2 back, per round)
__device__ __forceinline__ void zo_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One batch of up to 64 sequences, one per lane (`act`): `ll` bytes whose origins are li, li + 1, ... (literals), then `ml` bytes that
// repeat what lies `off` bytes before them (ml may be 0).  org[q] = origin of the piece's byte q; `pos` = q of the batch's first byte
// (the sequences follow each other: i_a = inclusive prefix sum of ll + ml); `start` = position of org[0] in the payload's output: a
// source byte below 0 becomes ZO_FLAG | (start + q).  The caller has checked the sequences (off >= 1, off <= start + match position).
// Literals first (they depend on nothing), then the matches in rounds: a match goes when none of the batch's matches it reads from
// is still pending -- which ones those are follows from the sequences that hold its first and its last source byte (binary search in
// the prefix sums).  Short matches are copied by their own lanes, a long one by the whole wave when it is the lowest pending one.
__device__ __forceinline__ void zo_batch(uint32_t* __restrict__ org, const uint32_t start, const int lane, const bool act, const uint32_t ll,
                                         const uint32_t li, const uint32_t ml, const uint32_t off, const uint32_t i_a, const uint32_t pos,
                                         uint32_t* s_ia, uint32_t* s_om)
{
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    const uint32_t o_l = pos + (i_a - ll - ml), o_m = o_l + ll; // where my literals / my match go
    // ---- literals: their origin is their index ----
    if (act && ll <= 32u)
    {
        uint32_t j = 0;
        for (; j + 4u <= ll; j += 4u)
        {
            u32x4_a4 v;
            v.x = li + j;
            v.y = li + j + 1u;
            v.z = li + j + 2u;
            v.w = li + j + 3u;
            *reinterpret_cast<u32x4_a4*>(org + o_l + j) = v;
        }
        for (; j < ll; ++j)
            org[o_l + j] = li + j;
    }
    for (uint64_t big = __builtin_amdgcn_ballot_w64(act && ll > 32u); big; big &= big - 1ull)
    {
        const int u = __builtin_ctzll(big);
        const uint32_t nn = (uint32_t)__builtin_amdgcn_readlane((int)ll, u), from = (uint32_t)__builtin_amdgcn_readlane((int)li, u),
                       to = (uint32_t)__builtin_amdgcn_readlane((int)o_l, u);
        for (uint32_t j = lane; j < nn; j += 64)
            org[to + j] = from + j;
    }
    // ---- matches ----
    const bool has = act && ml != 0u;
    const int32_t a = (int32_t)o_m - (int32_t)off; // first source byte (below 0: an earlier piece)
    const uint32_t span = ml < off ? ml : off;     // distinct source bytes (off < ml: byte j = source byte j mod off)
    uint64_t dep = 0;
    s_ia[lane] = pos + i_a;
    s_om[lane] = o_m;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (has && a + (int32_t)span > (int32_t)pos)
    {
        const uint32_t xa = a > (int32_t)pos ? (uint32_t)a : pos, xb = (uint32_t)(a + (int32_t)span) - 1u;
        uint32_t ja = 0, jb = 0; // smallest j with s_ia[j] > x
#pragma unroll
        for (int st = 32; st; st >>= 1)
        {
            if (s_ia[ja + st - 1] <= xa)
                ja += st;
            if (s_ia[jb + st - 1] <= xb)
                jb += st;
        }
        if (jb > (uint32_t)lane)
            jb = (uint32_t)lane; // (cannot be: a source ends where its match begins)
        int32_t hi = (int32_t)jb;
        if (jb == (uint32_t)lane || xb < s_om[jb])
            hi -= 1; // my own sequence / only the literals of that one
        if (hi >= (int32_t)ja)
            dep = ((hi >= 63 ? 0ull : (1ull << (hi + 1))) - 1ull) & ~((1ull << ja) - 1ull);
    }
    __builtin_amdgcn_wave_barrier();
    const bool own = has && ml <= 64u;
    uint64_t pend = __builtin_amdgcn_ballot_w64(has);
    // Round 0 needs no wait: it takes the matches whose source lies entirely before the batch (final since the last batch's wait) next
    // to the literal stores; the wait behind it covers both.  A batch without a match that reads from the batch itself -- most of
    // them -- is one memory round trip, not two.
    const bool early = has && a + (int32_t)span <= (int32_t)pos;
    bool round0 = true;
    do
    {
        const bool free_now = round0 ? early : !(pend & dep);
        const bool ready = own && ((pend >> lane) & 1ull) && free_now;
        if (ready)
        {
            if (off >= ml)
            {
                uint32_t j = 0;
                for (; j + 4u <= ml; j += 4u)
                {
                    u32x4_a4 v;
                    const int32_t sp = a + (int32_t)j;
                    if (sp >= 0)
                        v = *reinterpret_cast<const u32x4_a4*>(org + sp);
                    else
                    {
                        v.x = ZO_FLAG | (uint32_t)((int32_t)start + sp);
                        v.y = sp + 1 < 0 ? ZO_FLAG | (uint32_t)((int32_t)start + sp + 1) : org[sp + 1];
                        v.z = sp + 2 < 0 ? ZO_FLAG | (uint32_t)((int32_t)start + sp + 2) : org[sp + 2];
                        v.w = sp + 3 < 0 ? ZO_FLAG | (uint32_t)((int32_t)start + sp + 3) : org[sp + 3];
                    }
                    *reinterpret_cast<u32x4_a4*>(org + o_m + j) = v;
                }
                for (; j < ml; ++j)
                {
                    const int32_t sp = a + (int32_t)j;
                    org[o_m + j] = sp < 0 ? ZO_FLAG | (uint32_t)((int32_t)start + sp) : org[sp];
                }
            }
            else
            {
                uint32_t m = 0;
                for (uint32_t j = 0; j < ml; ++j)
                {
                    const int32_t sp = a + (int32_t)m;
                    org[o_m + j] = sp < 0 ? ZO_FLAG | (uint32_t)((int32_t)start + sp) : org[sp];
                    m = m + 1u == off ? 0u : m + 1u;
                }
            }
        }
        const uint64_t readym = __builtin_amdgcn_ballot_w64(ready);
        // long matches that read from nothing pending: the whole wave, one after the other (the lowest pending one always qualifies)
        const uint64_t longm = __builtin_amdgcn_ballot_w64(has && !own && ((pend >> lane) & 1ull) && free_now);
        for (uint64_t todo = longm; todo; todo &= todo - 1ull)
        {
            const int f = __builtin_ctzll(todo);
            const uint32_t gm = (uint32_t)__builtin_amdgcn_readlane((int)ml, f), go = (uint32_t)__builtin_amdgcn_readlane((int)off, f),
                           gd = (uint32_t)__builtin_amdgcn_readlane((int)o_m, f);
            const int32_t ga = (int32_t)gd - (int32_t)go;
            for (uint32_t j = lane; j < gm; j += 64)
            {
                const int32_t sp = ga + (int32_t)(go < gm ? j % go : j);
                org[gd + j] = sp < 0 ? ZO_FLAG | (uint32_t)((int32_t)start + sp) : org[sp];
            }
        }
        pend &= ~(readym | longm);
        zo_sync();
        round0 = false;
    } while (pend);
}

// The same batch executed on BYTES (a piece none of whose matches reaches below its start: payloads of this library's LZ4 encoder, whose
// matches stay inside 64 KiB groups; units of other payloads that happen not to): out[q] = byte q of the piece, literals copied from
// `lits` (literal number i = lits[i]), matches from out[] itself -- through memory, like the origins.  Returns false, having written
// nothing but literals, when a match of the batch reads below 0: the caller gives the piece up (it is then executed on origins).
__device__ __forceinline__ bool zo_batch_bytes(uint8_t* __restrict__ out, const uint8_t* __restrict__ lits, const int lane, const bool act,
                                               const uint32_t ll, const uint32_t li, const uint32_t ml, const uint32_t off, const uint32_t i_a,
                                               const uint32_t pos, uint32_t* s_ia, uint32_t* s_om)
{
    typedef uint32_t u32_a1 __attribute__((aligned(1)));
    typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    const uint32_t o_l = pos + (i_a - ll - ml), o_m = o_l + ll; // where my literals / my match go
    const bool has = act && ml != 0u;
    const int32_t a = (int32_t)o_m - (int32_t)off; // first source byte
    if (__builtin_amdgcn_ballot_w64(has && a < 0))
        return false;
    // ---- literals ----
    if (act && ll <= 32u)
    {
        uint32_t j = 0;
        for (; j + 4u <= ll; j += 4u)
            *reinterpret_cast<u32_a1*>(out + o_l + j) = *reinterpret_cast<const u32_a1*>(lits + li + j);
        for (; j < ll; ++j)
            out[o_l + j] = lits[li + j];
    }
    for (uint64_t big = __builtin_amdgcn_ballot_w64(act && ll > 32u); big; big &= big - 1ull)
    {
        const int u = __builtin_ctzll(big);
        const uint32_t nn = (uint32_t)__builtin_amdgcn_readlane((int)ll, u), from = (uint32_t)__builtin_amdgcn_readlane((int)li, u),
                       to = (uint32_t)__builtin_amdgcn_readlane((int)o_l, u);
        uint32_t j = 16u * (uint32_t)lane;
        for (; j + 16u <= nn; j += 1024u)
            *reinterpret_cast<u32x4_a1*>(out + to + j) = *reinterpret_cast<const u32x4_a1*>(lits + from + j);
        if (j < nn) // (the run's last vector: byte by byte)
            for (uint32_t k = j; k < nn && k < j + 16u; ++k)
                out[to + k] = lits[from + k];
    }
    // ---- matches: the rounds of zo_batch ----
    const uint32_t span = ml < off ? ml : off;
    uint64_t dep = 0;
    s_ia[lane] = pos + i_a;
    s_om[lane] = o_m;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (has && a + (int32_t)span > (int32_t)pos)
    {
        const uint32_t xa = a > (int32_t)pos ? (uint32_t)a : pos, xb = (uint32_t)(a + (int32_t)span) - 1u;
        uint32_t ja = 0, jb = 0; // smallest j with s_ia[j] > x
#pragma unroll
        for (int st = 32; st; st >>= 1)
        {
            if (s_ia[ja + st - 1] <= xa)
                ja += st;
            if (s_ia[jb + st - 1] <= xb)
                jb += st;
        }
        if (jb > (uint32_t)lane)
            jb = (uint32_t)lane;
        int32_t hi = (int32_t)jb;
        if (jb == (uint32_t)lane || xb < s_om[jb])
            hi -= 1; // my own sequence / only the literals of that one
        if (hi >= (int32_t)ja)
            dep = ((hi >= 63 ? 0ull : (1ull << (hi + 1))) - 1ull) & ~((1ull << ja) - 1ull);
    }
    __builtin_amdgcn_wave_barrier();
    const bool own = has && ml <= 64u;
    uint64_t pend = __builtin_amdgcn_ballot_w64(has);
    // Round 0 needs no wait: it takes the matches whose source lies entirely before the batch (final since the last batch's wait) next
    // to the literal stores; the wait behind it covers both.  A batch without a match that reads from the batch itself -- most of
    // them -- is one memory round trip, not two.
    const bool early = has && a + (int32_t)span <= (int32_t)pos;
    bool round0 = true;
    do
    {
        const bool free_now = round0 ? early : !(pend & dep);
        const bool ready = own && ((pend >> lane) & 1ull) && free_now;
        if (ready)
        {
            const uint8_t* sp = out + a;
            uint8_t* dp = out + o_m;
            uint32_t j = 0;
            if (off >= ml)
            {
                for (; j + 16u <= ml; j += 16u)
                    *reinterpret_cast<u32x4_a1*>(dp + j) = *reinterpret_cast<const u32x4_a1*>(sp + j);
                for (; j + 4u <= ml; j += 4u)
                    *reinterpret_cast<u32_a1*>(dp + j) = *reinterpret_cast<const u32_a1*>(sp + j);
                for (; j < ml; ++j)
                    dp[j] = sp[j];
            }
            else
            {
                // overlapping: byte j = seed byte j mod off (the seed lies before the match: final)
                uint32_t m = 0;
                for (; j < ml; ++j)
                {
                    dp[j] = sp[m];
                    m = m + 1u == off ? 0u : m + 1u;
                }
            }
        }
        const uint64_t readym = __builtin_amdgcn_ballot_w64(ready);
        // long matches that read from nothing pending: the whole wave, one after the other (the lowest pending one always qualifies)
        const uint64_t longm = __builtin_amdgcn_ballot_w64(has && !own && ((pend >> lane) & 1ull) && free_now);
        for (uint64_t todo = longm; todo; todo &= todo - 1ull)
        {
            const int f = __builtin_ctzll(todo);
            const uint32_t gm = (uint32_t)__builtin_amdgcn_readlane((int)ml, f), go = (uint32_t)__builtin_amdgcn_readlane((int)off, f),
                           gd = (uint32_t)__builtin_amdgcn_readlane((int)o_m, f);
            const uint8_t* sp = out + (gd - go);
            if (go >= gm)
            {
                uint32_t j = 16u * (uint32_t)lane;
                for (; j + 16u <= gm; j += 1024u)
                    *reinterpret_cast<u32x4_a1*>(out + gd + j) = *reinterpret_cast<const u32x4_a1*>(sp + j);
                if (j < gm)
                    for (uint32_t k = j; k < gm && k < j + 16u; ++k)
                        out[gd + k] = sp[k];
            }
            else
                for (uint32_t j = lane; j < gm; j += 64)
                    out[gd + j] = sp[j % go];
        }
        pend &= ~(readym | longm);
        zo_sync();
        round0 = false;
    } while (pend);
    return true;
}

#endif
