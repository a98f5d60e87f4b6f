// k_buzhash.hip -- content-defined chunking on gfx950.
//
// Reference behaviour: lib/hpcdcchunker/longtail_hpcdcchunker.c (Longtail_HPCDCNextChunk :225-310).
// MI355X formulation:
//   K1 k_buzhash_prefix_dma: H(p) = XOR_{j<48} rotl(T[byte[p-1-j]], j) is a pure function of the 48 bytes before p, so every
//                           position of every part is evaluated independently.  ONE persistent workgroup of 16 waves per CU;
//                           each WAVE owns 4 KiB wave-tiles end to end -- no workgroup barrier in the loop.  The tile comes
//                           straight from global memory into the wave's LDS buffer (global_load_lds_dwordx4), every lane
//                           keeps the SUFFIX XORs of its own 64-byte run in registers (prefix-XOR formulation: one table
//                           look-up per byte, no 48-byte warm-up; the neighbour's suffixes through one v_xor_b32_dpp).
//                           T[256] is replicated 64x in LDS with a 256-byte stride (one copy per lane): the LDS address of
//                           T[b] is {lane*4, b} = ONE v_perm_b32 of the data dword, and lookups never conflict.
//                           `H % d == d-1` is a necessary test on the odd part of d accumulated over eight steps, the exact
//                           multiply-add + rotate + compare only in the rare wave-uniform branch.  Output is a two level
//                           bitmap: level 0 one bit per byte (only non-zero words are stored), level 1 one bit per 64-byte
//                           run (wave ballot, always stored).  (Earlier formulations -- the rolling-window kernel of
//                           rounds 1-2, register staging -- live in ablations/ and are compiled by `make ablations` only.)
//   K2 select_cuts        : one wave per part walks chunk by chunk: wave-wide load of the level-1 words that
//                           cover (start+min, start+max], first flagged run, one level-0 word, ffs.
//   K3 compact            : scan of per-part counts, gather into dense (offset,len) arrays.
#include "lthip_internal.h"

#include <stdlib.h>
#include <type_traits>

namespace
{

__constant__ uint32_t c_buztab[256] = {
#include "buzhash_table.inc"
};

constexpr int RUN = 64;                      // bytes per thread
constexpr int TILE = 256 * RUN;              // 16 KiB: the plan's tile (four wave-tiles)
constexpr int ROW_DW = 17;                   // 16 data dwords + 1 pad: thread-strided ds_read_b32 hits 32 distinct banks
constexpr int TAB_REP = 64;                  // T[v] replicated once per lane, 256 bytes apart: a lookup address is
                                             // {byte, lane * 4} = ONE v_perm_b32 of the data dword, and never conflicts

// 32-bit load from a raw LDS byte address (the device pass only: LDS pointers are 32 bits wide there)
__device__ __forceinline__ uint32_t lds_load_u32(uint32_t byte_addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(3))) uint32_t*)byte_addr;
#else
    (void)byte_addr;
    return 0u;
#endif
}

__device__ __forceinline__ uint32_t lds_load_u8(uint32_t byte_addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(3))) uint8_t*)byte_addr;
#else
    (void)byte_addr;
    return 0u;
#endif
}

__device__ __forceinline__ uint4 lds_load_u128(uint32_t byte_addr) // 16-byte aligned: ONE ds_read_b128
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4 q = *(const __attribute__((address_space(3))) v4*)byte_addr;
    return make_uint4(q.x, q.y, q.z, q.w);
#else
    (void)byte_addr;
    return make_uint4(0, 0, 0, 0);
#endif
}

__device__ __forceinline__ uint32_t rotl32(uint32_t x, uint32_t r) { return __builtin_amdgcn_alignbit(x, x, (32u - r) & 31u); }
__device__ __forceinline__ uint32_t rotr32(uint32_t x, uint32_t r) { return __builtin_amdgcn_alignbit(x, x, r & 31u); }

// ---------------------------------------------------------------------------------------------------
// tile -> part table (built once per plan)
// ---------------------------------------------------------------------------------------------------
__global__ void k_tile_table(const PartDev* __restrict__ parts, uint32_t nparts, uint32_t* __restrict__ tile_part,
                             uint64_t ntiles)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles)
        return;
    // last part whose tile_base <= t and that owns at least one tile
    uint32_t lo = 0, hi = nparts; // invariant: parts[lo].tile_base <= t
    while (hi - lo > 1)
    {
        uint32_t mid = lo + (hi - lo) / 2;
        if ((uint64_t)parts[mid].tile_base <= t)
            lo = mid;
        else
            hi = mid;
    }
    tile_part[t] = lo; // empty parts share their tile_base with the next part; the search lands on the last one (non-empty)
}

// ---------------------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------------------
// The unit of work is a WAVE-tile: 4 KiB of one part (+64 B halo in front) staged into the wave's private LDS
// rows; waves of a workgroup share nothing but the read-only table, so there is no barrier in the loop.  The loads
// of the next wave-tile are issued before the current one is hashed (register double buffering).
constexpr int WROWS = 64 + 1;            // 64 data rows + 1 halo row
constexpr int WTILE = 64 * RUN;          // 4 KiB

#ifdef LTHIP_ABLATIONS
#include "ablations/k_buzhash_roll.inc"
#endif

// ---------------------------------------------------------------------------------------------------
// K1, prefix-XOR formulation (round 3; the default).
//
// In the de-rotated frame U[q] = rotr(T[b_q], q mod 32) the window hash of hpcdcchunker.c:266-306 is
//     H(p) = rotl( XOR_{q=p-48}^{p-1} U[q], (p-1) mod 32 )
// (rotl(U[p-1-j], p-1) = rotl(T[b_{p-1-j}], j)), so a lane that owns the 64-byte aligned run [q0, q0+64) needs ONE table
// look-up per byte and no 48-byte warm-up: with the suffix XORs S[k] = XOR_{i>=k} U[q0+i] (S[64] = 0) of its own run,
//     k >= 47:  W = S[k-47] ^ S[k+1]                                  (window inside the run)
//     k <  47:  W = S[0] ^ S[k+1] ^ S'[k+17]                          (S' = the suffix XORs of the run before = lane-1:
//                                                                      one v_xor_b32_dpp wave_ror:1)
//     H(q0+k+1) = rotl(W, k mod 32)                                   (compile-time rotate: q0 is a multiple of 64)
// Lane 0's predecessor is the halo row: its 47 suffix XORs are made by the whole wave (one byte per lane, wave scan),
// parked in LDS, and loaded into LANE 63's registers S[17..63] once that lane has used them itself -- the steps run from
// k = 63 down, so a register's own use (step j-1) precedes its use by the neighbour (step j-17) -- where wave_ror:1
// hands them to lane 0.  The cut test is accumulated: y = H*inv + inv (<= qodd iff the odd part of d divides H+1) of
// eight steps goes through v_min3_u32, one compare and one scalar branch per eight bytes; the exact test only in the
// rare branch.  ~8.4 VALU instructions per byte and lane instead of 11.5, 64 + 16 live values instead of 112 + 28.
// ---------------------------------------------------------------------------------------------------
#ifndef K1_MAD64
#define K1_MAD64 0 // measured: 5.88 against 5.76 ms per 16 GiB -- not kept
#endif
constexpr int HALO_DW = 64; // per wave: suffix XORs of the halo row, dword j = S'[j]

struct PrefixConsts
{
    uint32_t lane4;     // lane * 4: low byte of a look-up address
    uint32_t hj;        // the halo byte this lane looks up (63 - lane)
    uint32_t halo_byte; // LDS byte address of the wave's halo[] array
    uint32_t thr;       // y <= thr is necessary for a cut
    uint32_t exact_add; // H*inv + addc = y + exact_add
};

// One wave-tile: win[] = the lane's own 64-byte run, hb = the lane's halo byte (byte 63 - lane of the 64 bytes before the
// tile; lanes >= 47 are ignored).  Returns the 64 candidate bits of the run, bit k <=> cut position q0 + k + 1.
template <int MODE>
__device__ __forceinline__ uint64_t prefix_tile(const uint32_t (&win)[16], uint32_t hb, const PrefixConsts& pc, const DivTest& dv,
                                                int lane, uint32_t* __restrict__ halo)
{
    // ---- halo: S'[j] for j = 17..63 by a wave scan, to LDS ----
    {
        uint32_t x = lds_load_u32((hb << 8) | pc.lane4);
        x = rotr32(x, pc.hj);
        if (lane >= 47)
            x = 0u;
        // inclusive XOR scan over the lanes: lane i gets the bytes 63-i .. 63 = S'[63 - i]
        x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); // row_shr:1
        x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); // row_shr:2
        x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); // row_shr:4
        x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); // row_shr:8
        x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15
        x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31
        halo[pc.hj] = x;
    }

    // ---- pass A: suffix XORs of the lane's own run ----
    uint32_t S[65];
#define LT_LOOKUP(j) S[j] = lds_load_u32(__builtin_amdgcn_perm(win[(j) >> 2], pc.lane4, 0x0c0c0400u | ((uint32_t)((j) & 3) << 8)))
#pragma unroll
    for (int j = 63; j >= 0; --j)
        LT_LOOKUP(j);
#undef LT_LOOKUP
    __builtin_amdgcn_sched_barrier(0);
    S[64] = 0u;
#pragma unroll
    for (int j = 63; j >= 0; --j)
        S[j] = S[j + 1] ^ ((j & 31) ? rotr32(S[j], (uint32_t)(j & 31)) : S[j]);
    __builtin_amdgcn_wave_barrier(); // halo[] written by all lanes, read by lane 63 below

    // ---- pass B: eight steps per group, from the top ----
    // Lane 63 hands the halo's suffixes to lane 0 (wave_ror:1): once the steps down to k = 4n are done, lane 63 has used
    // S[4n+4 .. 4n+7] for the last time (own use of S[j]: step j - 1), and the neighbour steps that read them (k = j - 17 =
    // 4n-13 .. 4n-10) lie two to three half-groups ahead: lane 63 takes the halo's values there with one exec-masked
    // ds_read_b128 per half-group, written by hand -- the compiler's version of `if (lane == 63) S[j] = halo[j]` waits for
    // the LDS right behind the branch; here the wait (lgkmcnt counts: LDS operations complete in order) sits in front of the
    // half-group that reads them, with the two younger loads still in flight.
    uint32_t mlo = 0, mhi = 0;
    uint32_t inv_v; // the addend in a VGPR: v_add_u32 with two VGPR sources issues at the double rate, with an SGPR source it does not
    asm volatile("v_mov_b32 %0, %1" : "=v"(inv_v) : "s"(dv.inv));
#if K1_MAD64
    const uint64_t inv64 = (uint64_t)inv_v;
#endif
    // A register that an asm load is still writing must not be touched by anything the compiler generates (a copy made
    // between the load and its wait reads the OLD value whenever the data has not landed: a rare, timing-dependent wrong
    // hand-off -- seen with the tuple taken apart right behind the load).  So a quad stays ONE 128-bit value from the load
    // statement to the wait statement, both tie it in and out, and it is taken apart only behind the wait.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 Q[16];
    auto halo_quad = [&](auto QI) __attribute__((always_inline)) // lane 63: quad q = halo[4q .. 4q+3], issued
    {
        constexpr int q = decltype(QI)::value;
        if constexpr (q >= 4 && q <= 15)
        {
            u32x4 v = {S[4 * q], S[4 * q + 1], S[4 * q + 2], S[4 * q + 3]};
            uint64_t save;
            asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[msk]\n\t"
                         "ds_read_b128 %[r], %[ad] offset:%[o]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [sv] "=&s"(save), [r] "+v"(v)
                         : [ad] "v"(pc.halo_byte), [msk] "s"(0x8000000000000000ull), [o] "i"(16 * q)
                         : "memory"); // reads halo[]: the store above may not sink below it
            Q[q] = v;
        }
    };
    auto halo_wait = [&](auto QI) __attribute__((always_inline)) // in front of the first half-group that reads quad q through the DPP
    {
        constexpr int q = decltype(QI)::value;
        if constexpr (q >= 4 && q <= 15)
        {
            // younger loads still in flight: those of the (at most two) half-groups executed since quad q was issued
            u32x4 v = Q[q];
            if constexpr (q >= 6)
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v));
            else if constexpr (q == 5)
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
            S[4 * q] = v.x;
            S[4 * q + 1] = v.y;
            S[4 * q + 2] = v.z;
            S[4 * q + 3] = v.w;
        }
    };
    auto group = [&](auto G) __attribute__((always_inline))
    {
        constexpr int g = decltype(G)::value;
        uint32_t y[8];
#pragma unroll
        for (int half = 1; half >= 0; --half)
        {
            // steps 8g + 4 half + 3 .. 8g + 4 half =: 4n + 3 .. 4n read, through the DPP, S[4n+17 .. 4n+20]: quads n+4 and n+5
            // (S[4n+20], the first register of quad n+5, was waited for by half-group n+1, which ran before this one)
            if (half == 1)
                halo_wait(std::integral_constant<int, 2 * g + 1 + 4>{});
            else
                halo_wait(std::integral_constant<int, 2 * g + 4>{});
#pragma unroll
            for (int ii = 3; ii >= 0; --ii)
            {
                const int i = 4 * half + ii;
                const int k = 8 * g + i;
                uint32_t w;
                if (k >= 47)
                    w = S[k - 47] ^ S[k + 1];
                else
                    w = (S[0] ^ S[k + 1]) ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S[(k + 17) & 63], 0x13C, 0xf, 0xf, false); // wave_ror:1
                const uint32_t h = (k & 31) ? rotl32(w, (uint32_t)(k & 31)) : w;
                if (MODE == 1)
                    y[i] = ~h & (dv.d - 1u);
                else
                {
#if K1_MAD64
                    // h * inv + inv in ONE instruction (measured 5.3 cycles against 4.2 + 2.8 for v_mul_lo_u32 + v_add_u32); the
                    // upper half of the 64-bit result and the carry are not used
                    uint64_t t, carry;
                    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t), "=s"(carry) : "v"(h), "s"(dv.inv), "v"(inv64));
                    y[i] = (uint32_t)t;
#else
                    y[i] = h * dv.inv + inv_v;
#endif
                }
            }
            // steps down to 4n are done: quad n + 1
            if (half == 1)
                halo_quad(std::integral_constant<int, 2 * g + 2>{});
            else
                halo_quad(std::integral_constant<int, 2 * g + 1>{});
        }
        const uint32_t m = min(min(min(y[0], y[1]), y[2]), min(min(min(y[3], y[4]), y[5]), min(y[6], y[7])));
        if (__builtin_amdgcn_ballot_w64(m <= pc.thr) != 0ull) // wave-uniform, one group in ~6
        {
            asm volatile("");
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                const int k = 8 * g + i;
                if (__builtin_amdgcn_ballot_w64(y[i] <= pc.thr) != 0ull)
                {
                    asm volatile("");
                    const bool hit = MODE == 1 ? y[i] == 0u : rotr32(y[i] + pc.exact_add, dv.k2) <= dv.qlim; // h % d == d-1
                    uint32_t hbit;
                    asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(hbit) : "s"(__builtin_amdgcn_ballot_w64(hit)));
                    if (k < 32)
                        mlo |= hbit << k;
                    else
                        mhi |= hbit << (k - 32);
                }
            }
        }
    };
    group(std::integral_constant<int, 7>{});
    group(std::integral_constant<int, 6>{});
    group(std::integral_constant<int, 5>{});
    group(std::integral_constant<int, 4>{});
    group(std::integral_constant<int, 3>{});
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 0>{});
    return ((uint64_t)mhi << 32) | mlo;
}

// candidate bits of a run -> the legal ones (cuts p with 48 <= p <= size), 0 for runs beyond the part
__device__ __forceinline__ uint64_t prefix_legal(uint64_t m, uint64_t q0, uint64_t size)
{
    if (q0 >= size)
        return 0ull;
    if (q0 < 47)
        m &= ~0ull << (47 - q0);
    const uint64_t remain = size - q0; // >= 1
    if (remain < 64)
        m &= (1ull << remain) - 1ull;
    return m;
}

__device__ __forceinline__ void prefix_table_init(uint32_t* tab, int tid, int nthreads)
{
    for (int v = tid; v < 256; v += nthreads)
    {
        const uint32_t tv = c_buztab[v];
        uint4 q = make_uint4(tv, tv, tv, tv);
        uint4* dst = reinterpret_cast<uint4*>(tab + v * TAB_REP);
#pragma unroll
        for (int j = 0; j < TAB_REP / 4; ++j)
            dst[j] = q;
    }
    __syncthreads(); // the only barrier: table visible to every wave
    if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)tab != 0u)
        __builtin_trap(); // a look-up address is the v_perm result itself (see k_buzhash_candidates)
}

#ifdef LTHIP_ABLATIONS
#include "ablations/k_buzhash_prefix_regs.inc"
#endif

// Flavour 2 (default): the wave-tile comes straight from global memory into LDS (global_load_lds_dwordx4, 1 KiB per
// instruction, no staging registers and no ds_write pass).  The LDS image of an LDS-DMA is lane-linear (base + instruction
// offset + lane * 16), so the rows cannot be padded; the bank spread comes from the SOURCE side instead: the wave's buffer holds
// 65 rows x 4 vectors of 16 bytes (row 0 = the 64 bytes before the tile), and slot (r, c') receives data vector
// c = c' ^ ((r >> 2) & 3) of row r -- lane L then reads its row r = L + 1 with four ds_read_b128 at slots c ^ ((r >> 2) & 3),
// which puts the sixteen lanes of every ds_read_b128 service group on sixteen different 16-byte slots (checked for all four
// groups).  ONE buffer per wave: the run is in registers after the four reads, so the next tile's DMA is issued right behind
// them and has the whole hashing of this tile to land; the results of a tile are stored at the start of the NEXT iteration,
// in front of that issue, so that the vmcnt(0) which retires the DMA never waits for a store just issued.
constexpr int DMA_BUF_B = WROWS * 64; // 4160 bytes, linear

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_buzhash_prefix_dma(const uint8_t* __restrict__ data,
                                                                       const PartDev* __restrict__ parts,
                                                                       const uint32_t* __restrict__ tile_part,
                                                                       uint32_t ntiles, DivTest dv,
                                                                       uint64_t* __restrict__ bm0,
                                                                       uint64_t* __restrict__ bm1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* buf = smem + 256 * TAB_REP + wave * ((DMA_BUF_B >> 2) + HALO_DW);
    uint32_t* halo = buf + (DMA_BUF_B >> 2);
    prefix_table_init(smem, tid, 64 * WAVES);

    const uint32_t buf_byte = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)buf);
    PrefixConsts pc;
    pc.lane4 = (uint32_t)lane * 4u;
    pc.hj = 63u - (uint32_t)lane;
    pc.halo_byte = buf_byte + DMA_BUF_B;
    pc.thr = MODE == 1 ? 0u : 0xFFFFFFFFu / (dv.d >> dv.k2);
    pc.exact_add = dv.addc - dv.inv;

    // DMA instruction u moves vector V = 64 u + lane = slot (r = V >> 2, c' = V & 3); (r >> 2) & 3 = (lane >> 4) & 3 for every u
    const uint32_t src_vec = 4u * ((uint32_t)lane >> 2) + (((uint32_t)lane & 3u) ^ (((uint32_t)lane >> 4) & 3u));
    const uint32_t src_off = 16u * src_vec; // byte offset of the lane's source vector inside a 1 KiB piece
    // own row r = lane + 1: data vector c sits at slot c ^ f, f = (r >> 2) & 3
    const uint32_t own_f = (((uint32_t)lane + 1u) >> 2) & 3u;
    const uint32_t own_row = buf_byte + 64u * ((uint32_t)lane + 1u);

    const uint64_t nwt = (uint64_t)ntiles * 4u;
    const uint64_t wstride = (uint64_t)gridDim.x * (uint64_t)WAVES;
    uint64_t wt = (uint64_t)blockIdx.x * (uint64_t)WAVES + (uint64_t)wave;
    if (wt >= nwt)
        return;

    // issue the DMA of the wave-tile at part-relative `sp` of part `p` into the wave's buffer.  M0 (the LDS base of an LDS-DMA)
    // points at the MIDDLE of the buffer for the whole kernel, the five pieces are told apart by the instruction offset
    // (13 bits, signed: -2048 .. +2048), which the hardware adds to the LDS address and to the global address alike.
    const uint32_t buf_mid = buf_byte + 2048u;
    auto piece = [&](auto U, const uint8_t* mid_src) __attribute__((always_inline))
    {
        constexpr int u = decltype(U)::value;
        const uint32_t so = src_off, bm = buf_mid; // (named here: asm operands alone do not capture in a generic lambda)
        asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[v], %[b] offset:%[o]"
                     :
                     : [v] "v"(so), [l] "s"(bm), [b] "s"(mid_src), [o] "i"(1024 * u - 2048)
                     : "memory");
    };
    auto issue = [&](const PartDev& p, uint64_t sp) __attribute__((always_inline))
    {
        const uint8_t* mid_src = data + p.off + sp - 64 + 2048; // the source of the buffer's middle (sp == 0: row 0 lies before the part and is not loaded)
        if (sp >= 64 && sp + (uint64_t)WTILE <= p.size) // every vector inside the part: the common case
        {
            asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %[v], %[b] offset:-2048\n\t"
                         "global_load_lds_dwordx4 %[v], %[b] offset:-1024\n\t"
                         "global_load_lds_dwordx4 %[v], %[b] offset:0\n\t"
                         "global_load_lds_dwordx4 %[v], %[b] offset:1024"
                         :
                         : [v] "v"(src_off), [l] "s"(buf_mid), [b] "s"(mid_src)
                         : "memory");
            if (lane < 4) // vectors 256..259: row 64 (src_off = 16 * lane for these lanes)
                piece(std::integral_constant<int, 4>{}, mid_src);
        }
        else
        {
            // a tile at an end of its part: vectors outside the part are not loaded (the slot keeps stale bytes; every
            // position they could influence is masked by prefix_legal), the vector that straddles the end is assembled
            // from byte loads and stored by its lane
            auto edge = [&](auto U) __attribute__((always_inline))
            {
                constexpr int u = decltype(U)::value;
                const int64_t g = (int64_t)sp - 64 + 1024 * u + (int64_t)src_off;
                const bool lane_on = u < 4 || lane < 4;
                const bool whole = lane_on && g >= 0 && (uint64_t)g + 16 <= p.size;
                const bool part = lane_on && g >= 0 && (uint64_t)g < p.size && (uint64_t)g + 16 > p.size;
                if (whole)
                    piece(U, mid_src);
                if (part)
                {
                    const uint8_t* src = data + p.off;
                    uint32_t w[4] = {0, 0, 0, 0};
                    const uint32_t n = (uint32_t)(p.size - (uint64_t)g);
                    for (uint32_t bb = 0; bb < n; ++bb)
                        w[bb >> 2] |= (uint32_t)src[g + bb] << (8 * (bb & 3));
                    uint32_t* d = buf + 256 * u + 4 * lane;
                    d[0] = w[0];
                    d[1] = w[1];
                    d[2] = w[2];
                    d[3] = w[3];
                }
            };
            edge(std::integral_constant<int, 0>{});
            edge(std::integral_constant<int, 1>{});
            edge(std::integral_constant<int, 2>{});
            edge(std::integral_constant<int, 3>{});
            edge(std::integral_constant<int, 4>{});
        }
    };

    PartDev pd = parts[tile_part[wt >> 2]];
    uint64_t span = ((wt >> 2) - pd.tile_base) * (uint64_t)TILE + (wt & 3u) * (uint64_t)WTILE; // part-relative
    issue(pd, span);

    uint64_t pend_m = 0, pend_summary = 0, pend_i0 = 0, pend_i1 = 0; // results of the tile before, stored one iteration late
    bool pend = false;
    for (;;)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tile has landed (and the stores of two tiles ago are done)
        uint32_t win[16];
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            const uint4 q = lds_load_u128(own_row + 16u * ((uint32_t)c ^ own_f));
            win[4 * c + 0] = q.x;
            win[4 * c + 1] = q.y;
            win[4 * c + 2] = q.z;
            win[4 * c + 3] = q.w;
        }
        const uint32_t hb = lds_load_u8(buf_byte + pc.hj); // row 0 = the halo row, f(0) = 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the buffer is free

        if (pend)
        {
            if (pend_m != 0ull)
                bm0[pend_i0 + (uint64_t)lane] = pend_m; // one word per 64-byte run
            if (lane == 0)
                bm1[pend_i1] = pend_summary;            // one word per 4 KiB
        }

        const uint64_t next = wt + wstride;
        PartDev npd = pd;
        uint64_t next_span = 0;
        if (next < nwt)
        {
            npd = parts[tile_part[next >> 2]];
            next_span = ((next >> 2) - npd.tile_base) * (uint64_t)TILE + (next & 3u) * (uint64_t)WTILE;
            issue(npd, next_span);
        }

        pend_m = prefix_legal(prefix_tile<MODE>(win, hb, pc, dv, lane, halo), span + (uint64_t)lane * RUN, pd.size);
        pend_summary = __builtin_amdgcn_ballot_w64(pend_m != 0ull);
        pend_i0 = pd.bm0_base + (span >> 6);
        pend_i1 = pd.bm1_base + (span >> 12);
        pend = true;

        if (next >= nwt)
            break;
        wt = next;
        pd = npd;
        span = next_span;
    }
    if (pend_m != 0ull)
        bm0[pend_i0 + (uint64_t)lane] = pend_m;
    if (lane == 0)
        bm1[pend_i1] = pend_summary;
}

// ---------------------------------------------------------------------------------------------------
// K2: one wave per part
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src_lane)
{
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_lane);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void k_select_cuts(const PartDev* __restrict__ parts, uint32_t nparts,
                                                    const uint64_t* __restrict__ bm0, const uint64_t* __restrict__ bm1,
                                                    uint32_t min_chunk, uint32_t max_chunk, uint2* __restrict__ region,
                                                    uint32_t* __restrict__ part_count)
{
    const uint32_t p = blockIdx.x;
    if (p >= nparts)
        return;
    const int lane = threadIdx.x;
    const PartDev pd = parts[p];
    const uint64_t size = pd.size;
    const uint64_t* b0 = bm0 + pd.bm0_base;
    const uint64_t* b1 = bm1 + pd.bm1_base;
    uint2* out = region + pd.region_base;

    uint64_t s = 0;
    uint32_t n = 0;
    while (s < size)
    {
        const uint64_t left = size - s;
        uint64_t len;
        if (left <= min_chunk)
            len = left; // hpcdcchunker.c:257-264
        else
        {
            const uint64_t end = left > max_chunk ? max_chunk : left; // :284
            len = end;
            // a cut of length L needs candidate bit q = s+L-1, L in [min+1, end]
            const uint64_t qlo = s + min_chunk;
            const uint64_t qhi = s + end - 1;
            if (qlo <= qhi)
            {
                const uint64_t rlo = qlo >> 6, rhi = qhi >> 6; // 64-byte runs
                const uint64_t wlo = rlo >> 6, whi = rhi >> 6; // level-1 words
                bool found = false;
                for (uint64_t w0 = wlo; w0 <= whi && !found; w0 += 64)
                {
                    const uint64_t w = w0 + (uint64_t)lane;
                    uint64_t word = 0;
                    if (w <= whi)
                    {
                        word = b1[w];
                        if (w == wlo)
                            word &= ~0ull << (rlo & 63);
                        if (w == whi && (rhi & 63) != 63)
                            word &= (1ull << ((rhi & 63) + 1)) - 1ull;
                    }
                    for (;;)
                    {
                        const uint64_t any = __builtin_amdgcn_ballot_w64(word != 0ull);
                        if (any == 0ull)
                            break;
                        const int f = __builtin_ctzll(any);
                        const uint64_t fw = bcast64(word, f);
                        const int bit = __builtin_ctzll(fw);
                        const uint64_t r = ((w0 + (uint64_t)f) << 6) + (uint64_t)bit;
                        uint64_t m = b0[r]; // wave-uniform address
                        if (r == rlo)
                            m &= ~0ull << (qlo & 63);
                        if (r == rhi && (qhi & 63) != 63)
                            m &= (1ull << ((qhi & 63) + 1)) - 1ull;
                        if (m != 0ull)
                        {
                            const uint64_t q = (r << 6) + (uint64_t)__builtin_ctzll(m);
                            len = q - s + 1;
                            found = true;
                            break;
                        }
                        if (lane == f)
                            word &= word - 1ull; // drop that run, try the next flagged one
                    }
                }
            }
        }
        if (lane == 0)
            out[n] = make_uint2((uint32_t)s, (uint32_t)len);
        ++n;
        s += len;
    }
    if (lane == 0)
        part_count[p] = n;
}

// ---------------------------------------------------------------------------------------------------
// scans + compaction
// ---------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total, uint32_t* sh /*>= 8*/)
{
    // wave scan (DPP-free, shuffles) + cross-wave combine
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o)
            x += y;
    }
    if (lane == 63)
        sh[wave] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w)
    {
        uint32_t t = sh[w];
        if (w < wave)
            base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// pass 1: per-block sums
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(const uint32_t* __restrict__ in, uint64_t n,
                                                            const uint32_t* __restrict__ n_dev,
                                                            uint32_t* __restrict__ sums)
{
    __shared__ uint32_t sh[8];
    if (n_dev)
        n = *n_dev < n ? *n_dev : n;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
    {
        uint64_t idx = base + (uint64_t)i * SCAN_THREADS + threadIdx.x;
        if (idx < n)
            acc += in[idx];
    }
    uint32_t tot;
    block_exclusive_scan(acc, &tot, sh);
    if (threadIdx.x == 0)
        sums[blockIdx.x] = tot;
}

// pass 2: single block scans the block sums in place, writes grand total to sums[nblocks]
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_spine(uint32_t* __restrict__ sums, uint32_t nblocks)
{
    __shared__ uint32_t sh[8];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += SCAN_THREADS)
    {
        uint32_t i = b0 + threadIdx.x;
        uint32_t v = i < nblocks ? sums[i] : 0;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan(v, &tot, sh);
        if (i < nblocks)
            sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0)
        sums[nblocks] = carry;
}

// pass 3: out[i] = exclusive prefix ; out[n] = total (out has n+1 entries)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint32_t* __restrict__ in, uint64_t n,
                                                             const uint32_t* __restrict__ n_dev,
                                                             const uint32_t* __restrict__ sums, uint32_t nblocks,
                                                             uint32_t* __restrict__ out)
{
    __shared__ uint32_t sh[8];
    if (n_dev)
        n = *n_dev < n ? *n_dev : n;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
    {
        uint64_t idx = base + i;
        v[i] = idx < n ? in[idx] : 0;
        acc += v[i];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(acc, &tot, sh) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
    {
        uint64_t idx = base + i;
        if (idx < n)
            out[idx] = ex;
        ex += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        out[n] = sums[nblocks];
}

// dense gather: one wave per part
__global__ __launch_bounds__(64) void k_compact_chunks(const PartDev* __restrict__ parts, uint32_t nparts,
                                                       const uint2* __restrict__ region,
                                                       const uint32_t* __restrict__ part_first,
                                                       uint64_t* __restrict__ chunk_offsets,
                                                       uint32_t* __restrict__ chunk_lens)
{
    const uint32_t p = blockIdx.x;
    if (p >= nparts)
        return;
    const PartDev pd = parts[p];
    const uint32_t first = part_first[p];
    const uint32_t n = part_first[p + 1] - first;
    const uint2* in = region + pd.region_base;
    for (uint32_t i = threadIdx.x; i < n; i += 64)
    {
        uint2 e = in[i];
        chunk_offsets[first + i] = pd.off + (uint64_t)e.x;
        chunk_lens[first + i] = e.y;
    }
}

// ---------------------------------------------------------------------------------------------------
// NextChunkFromBuffer (hpcdcchunker.c:452-523): one wave, every lane evaluates the hash of one candidate
// length directly.  For the first 47 lengths after `min` the reference's rolling window still holds bytes
// buf[0..48) (it is seeded from the FRONT of the buffer, :488-494), so
//   h(k) = rotl(h0,k) ^ XOR_{i=1..k} rotl( rotl(T[buf[i-1]],16) ^ T[buf[min+i-1]], k-i ),  h0 = XOR_i rotl(T[buf[i]],47-i)
// and from k = 48 on it is the ordinary 48-byte window hash.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_chunk_from_buffer(const uint8_t* __restrict__ buf, uint32_t n, uint32_t min_chunk,
                                                          DivTest dv, uint64_t* __restrict__ out_len)
{
    const int lane = threadIdx.x;
    uint32_t h0 = 0;
    for (uint32_t i = 0; i < 48; ++i)
        h0 ^= rotl32(c_buztab[buf[i]], (47u - i) & 31u);
    uint32_t result = n;
    const uint32_t kmax = n - min_chunk; // candidate lengths min+1 .. n
    for (uint32_t k0 = 1; k0 <= kmax; k0 += 64)
    {
        const uint32_t k = k0 + (uint32_t)lane;
        bool hit = false;
        if (k <= kmax)
        {
            uint32_t h;
            if (k < 48)
            {
                h = rotl32(h0, k & 31u);
                for (uint32_t i = 1; i <= k; ++i)
                    h ^= rotl32(rotl32(c_buztab[buf[i - 1]], 16) ^ c_buztab[buf[min_chunk + i - 1]], (k - i) & 31u);
            }
            else
            {
                const uint32_t p = min_chunk + k; // window = the 48 bytes before p
                h = 0;
                for (uint32_t j = 0; j < 48; ++j)
                    h ^= rotl32(c_buztab[buf[p - 1 - j]], j & 31u);
            }
            hit = dv.pow2 ? (h & (dv.d - 1u)) == dv.d - 1u : rotr32(h * dv.inv + dv.addc, dv.k2) <= dv.qlim;
        }
        const uint64_t any = __builtin_amdgcn_ballot_w64(hit);
        if (any)
        {
            result = min_chunk + k0 + (uint32_t)__builtin_ctzll(any);
            break;
        }
    }
    if (lane == 0)
        *out_len = result;
}

} // namespace

int lthip_launch_from_buffer(lthip_ctx* ctx, const uint8_t* d_data, uint32_t n, uint32_t min_chunk, const DivTest& dv,
                             uint64_t* d_out)
{
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    hipLaunchKernelGGL(k_chunk_from_buffer, dim3(1), dim3(64), 0, ctx->stream, d_data, n, min_chunk, dv, d_out);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
int lthip_launch_tile_table(lthip_ctx* ctx, lthip_plan* plan)
{
    if (plan->ntiles == 0)
        return 0;
    LaunchTimer t(ctx, LTHIP_K_OTHER);
    const uint32_t blocks = (uint32_t)div_up_u64(plan->ntiles, 256);
    hipLaunchKernelGGL(k_tile_table, dim3(blocks), dim3(256), 0, ctx->stream, plan->d_parts, plan->nparts,
                       plan->d_tile_part, plan->ntiles);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

#ifdef LTHIP_ABLATIONS
constexpr bool K1_HAS_REG_FLAVOUR = true;
#else
constexpr bool K1_HAS_REG_FLAVOUR = false;
#endif

template <int WAVES, bool DMA>
static int launch_buzhash_prefix(lthip_ctx* ctx, const lthip_plan* plan, const uint8_t* d_data, uint64_t* bm0, uint64_t* bm1)
{
    static_assert(DMA || K1_HAS_REG_FLAVOUR, "the register-staged flavour is compiled by the ablation build only");
    constexpr size_t per_wave = DMA ? (DMA_BUF_B >> 2) + HALO_DW : WROWS * ROW_DW + HALO_DW;
    static_assert(sizeof(uint32_t) * (256 * TAB_REP + WAVES * per_wave) <= 160 * 1024, "LDS budget: one workgroup per CU");
    const size_t lds = sizeof(uint32_t) * (256 * TAB_REP + WAVES * per_wave);
#ifdef LTHIP_ABLATIONS
    auto k0 = DMA ? &k_buzhash_prefix_dma<0, WAVES> : &k_buzhash_prefix<0, WAVES>;
    auto k1 = DMA ? &k_buzhash_prefix_dma<1, WAVES> : &k_buzhash_prefix<1, WAVES>;
#else
    auto k0 = &k_buzhash_prefix_dma<0, WAVES>;
    auto k1 = &k_buzhash_prefix_dma<1, WAVES>;
#endif
    static bool granted[64] = {}; // per device: more than 64 KiB of dynamic LDS has to be granted explicitly
    if (ctx->device < 0 || ctx->device >= 64 || !granted[ctx->device])
    {
        LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (ctx->device >= 0 && ctx->device < 64)
            granted[ctx->device] = true;
    }
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    uint32_t grid = (uint32_t)ncu;
    if ((uint64_t)grid * WAVES > plan->ntiles * 4u)
        grid = (uint32_t)div_up_u64(plan->ntiles * 4u, WAVES);
    LaunchTimer t(ctx, LTHIP_K_BUZHASH);
    hipLaunchKernelGGL(plan->div.pow2 ? k1 : k0, dim3(grid), dim3(64 * WAVES), lds, ctx->stream, d_data, plan->d_parts,
                       plan->d_tile_part, (uint32_t)plan->ntiles, plan->div, bm0, bm1);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

#ifdef LTHIP_ABLATIONS
// LTHIP_K1 (ablation build): "dma16" (the product's kernel) / "dma12" = the prefix-XOR kernel fed by LDS-DMA, 16 / 12 waves per CU;
// "prefix16" / "prefix12" = the same with register staging; "roll" = the rolling-window kernel of rounds 1-2
static int k1_flavour()
{
    // (read again after lthip_debug_reload_env: tools/ablations/k1_stress_tib.py runs two flavours against each other in one process)
    static std::atomic<int> f{-1};
    static std::atomic<uint32_t> seen{0};
    const uint32_t gen = g_lthip_env_gen;
    if (seen.load(std::memory_order_acquire) != gen)
    {
        const char* e = getenv("LTHIP_K1");
        f.store(!e ? 0 : !strcmp(e, "roll") ? 4 : !strcmp(e, "prefix12") ? 3 : !strcmp(e, "prefix16") ? 2 : !strcmp(e, "dma12") ? 1 : 0,
                std::memory_order_relaxed);
        seen.store(gen, std::memory_order_release);
    }
    return f.load(std::memory_order_relaxed);
}

static int launch_buzhash_roll(lthip_ctx* ctx, const lthip_plan* plan, const uint8_t* d_data, uint64_t* bm0, uint64_t* bm1)
{
    static_assert(sizeof(uint32_t) * (256 * TAB_REP + K1_WAVES * WROWS * ROW_DW) <= 160 * 1024, "LDS budget: one workgroup per CU");
    const size_t lds = sizeof(uint32_t) * (256 * TAB_REP + K1_WAVES * WROWS * ROW_DW);
    if (!ctx->k1_lds_enabled) // per context = per device: more than 64 KiB of dynamic LDS has to be granted explicitly
    {
        LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_buzhash_candidates<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LTHIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_buzhash_candidates<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->k1_lds_enabled = true;
    }
    // one persistent workgroup of 12 waves per CU (64 KiB table + 12 wave row buffers = 117 KiB of LDS)
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    uint32_t grid = (uint32_t)ncu;
    if ((uint64_t)grid * K1_WAVES > plan->ntiles * 4u)
        grid = (uint32_t)div_up_u64(plan->ntiles * 4u, K1_WAVES);
    LaunchTimer t(ctx, LTHIP_K_BUZHASH);
    if (plan->div.pow2)
        hipLaunchKernelGGL(k_buzhash_candidates<1>, dim3(grid), dim3(K1_THREADS), lds, ctx->stream, d_data, plan->d_parts,
                           plan->d_tile_part, (uint32_t)plan->ntiles, plan->div, bm0, bm1);
    else
        hipLaunchKernelGGL(k_buzhash_candidates<0>, dim3(grid), dim3(K1_THREADS), lds, ctx->stream, d_data, plan->d_parts,
                           plan->d_tile_part, (uint32_t)plan->ntiles, plan->div, bm0, bm1);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
#endif

int lthip_launch_buzhash(lthip_ctx* ctx, const lthip_plan* plan, const uint8_t* d_data, uint64_t* bm0, uint64_t* bm1)
{
    if (plan->ntiles == 0)
        return 0;
#ifdef LTHIP_ABLATIONS
    const int flavour = k1_flavour();
    if (flavour == 1)
        return launch_buzhash_prefix<12, true>(ctx, plan, d_data, bm0, bm1);
    if (flavour == 2)
        return launch_buzhash_prefix<16, false>(ctx, plan, d_data, bm0, bm1);
    if (flavour == 3)
        return launch_buzhash_prefix<12, false>(ctx, plan, d_data, bm0, bm1);
    if (flavour == 4)
        return launch_buzhash_roll(ctx, plan, d_data, bm0, bm1);
#endif
    return launch_buzhash_prefix<16, true>(ctx, plan, d_data, bm0, bm1);
}

int lthip_launch_select(lthip_ctx* ctx, const lthip_plan* plan, const uint64_t* bm0, const uint64_t* bm1, uint2* region,
                        uint32_t* part_count)
{
    LaunchTimer t(ctx, LTHIP_K_SELECT);
    hipLaunchKernelGGL(k_select_cuts, dim3(plan->nparts), dim3(64), 0, ctx->stream, plan->d_parts, plan->nparts, bm0, bm1,
                       plan->min_chunk, plan->max_chunk, region, part_count);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

// d_out[i] = sum of d_in[0..i) for i in [0, n] where n = min(n_bound, *d_n) (d_n may be null); d_out holds n+1 entries
int lthip_exclusive_scan_u32(lthip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n_bound,
                             const uint32_t* d_n, int kid)
{
    const uint64_t nblocks64 = n_bound ? div_up_u64(n_bound, SCAN_BLOCK) : 1;
    if (nblocks64 > 0x7FFFFFFFull)
        return lthip_fail(ctx, EINVAL, "scan", "too many elements");
    const uint32_t nblocks = (uint32_t)nblocks64;
    void* tmp;
    int err = lthip_scratch(ctx, S_SCAN_TMP, ((size_t)nblocks + 1) * 4, &tmp);
    if (err)
        return err;
    uint32_t* sums = (uint32_t*)tmp;
    LaunchTimer t(ctx, kid);
    hipLaunchKernelGGL(k_scan_sums, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, d_in, n_bound, d_n, sums);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, sums, nblocks);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, d_in, n_bound, d_n, sums, nblocks,
                       d_out);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}

int lthip_launch_compact(lthip_ctx* ctx, const lthip_plan* plan, const uint2* region, const uint32_t* part_count,
                         uint32_t* d_part_first, uint64_t* d_chunk_offsets, uint32_t* d_chunk_lens)
{
    int err = lthip_exclusive_scan_u32(ctx, part_count, d_part_first, plan->nparts, nullptr, LTHIP_K_COMPACT);
    if (err || plan->nparts == 0)
        return err;
    LaunchTimer t(ctx, LTHIP_K_COMPACT);
    hipLaunchKernelGGL(k_compact_chunks, dim3(plan->nparts), dim3(64), 0, ctx->stream, plan->d_parts, plan->nparts, region,
                       (const uint32_t*)d_part_first, d_chunk_offsets, d_chunk_lens);
    LTHIP_LAUNCH_CHECK(ctx);
    return 0;
}
