// k_buzhash.hip -- content-defined chunking on gfx950.
//
// Reference behaviour: lib/hpcdcchunker/longtail_hpcdcchunker.c (Longtail_HPCDCNextChunk :225-310).
// MI355X formulation:
//   K1 buzhash_candidates : H(p) = XOR_{j<48} rotl(T[byte[p-1-j]], j) is a pure function of the 48 bytes before
//                           p, so every position of every part is evaluated independently.  ONE persistent
//                           768-thread workgroup per CU (12 waves, 156 VGPRs, no scratch); each WAVE owns 4 KiB
//                           wave-tiles end to end -- no workgroup barrier in the loop: the next tile is
//                           prefetched into registers (coalesced 16-byte loads) while the current one is hashed
//                           from the wave's private LDS rows (17-dword pitch, conflict-free); every lane rolls
//                           the 48-byte window over its own 64-byte run.  T[256] is replicated 64x in LDS with
//                           a 256-byte stride (one copy per lane): the LDS address of T[b] is {lane*4, b} =
//                           ONE v_perm_b32 of the data dword, and lookups never conflict.  `H % d == d-1` is a
//                           3-instruction necessary test on the odd part of d, the exact multiply-add + rotate
//                           + compare only in the rare wave-uniform branch.  Output is a two level bitmap:
//                           level 0 one bit per byte (only non-zero words are stored), level 1 one bit per
//                           64-byte run (wave ballot, always stored).
//   K2 select_cuts        : one wave per part walks chunk by chunk: wave-wide load of the level-1 words that
//                           cover (start+min, start+max], first flagged run, one level-0 word, ffs.
//   K3 compact            : scan of per-part counts, gather into dense (offset,len) arrays.
#include "lthip_internal.h"

namespace
{

__constant__ uint32_t c_buztab[256] = {
#include "buzhash_table.inc"
};

constexpr int K1_WAVES = 12;                 // one workgroup per CU: 3 waves per SIMD, <= 168 VGPRs each (16 waves at 128 VGPRs spill and gain nothing: measured)
constexpr int K1_THREADS = 64 * K1_WAVES;
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
constexpr int WVECS = WROWS * 4;         // 260 16-byte vectors per wave-tile
constexpr int WTILE = 64 * RUN;          // 4 KiB

struct TileRegs
{
    uint4 q[5];
};

__device__ __forceinline__ void tile_load(TileRegs& r, const uint8_t* __restrict__ data, const PartDev& pd,
                                          uint64_t span_start, int lane)
{
    const uint8_t* src = data + pd.off;
#pragma unroll
    for (int u = 0; u < 5; ++u)
    {
        const int v = lane + u * 64;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (v < WVECS)
        {
            const int64_t g = (int64_t)span_start - 64 + 16 * (int64_t)v;
            if (g >= 0)
            {
                if ((uint64_t)g + 16 <= pd.size)
                    q = *reinterpret_cast<const uint4*>(src + g);
                else if ((uint64_t)g < pd.size)
                {
                    uint32_t w[4] = {0, 0, 0, 0};
                    const uint32_t n = (uint32_t)(pd.size - (uint64_t)g);
                    for (uint32_t b = 0; b < n; ++b)
                        w[b >> 2] |= (uint32_t)src[g + b] << (8 * (b & 3));
                    q = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
        r.q[u] = q;
    }
}

__device__ __forceinline__ void tile_store(const TileRegs& r, uint32_t* __restrict__ rows, int lane)
{
#pragma unroll
    for (int u = 0; u < 5; ++u)
    {
        const int v = lane + u * 64;
        if (v < WVECS)
        {
            uint32_t* d = rows + (v >> 2) * ROW_DW + (v & 3) * 4;
            d[0] = r.q[u].x;
            d[1] = r.q[u].y;
            d[2] = r.q[u].z;
            d[3] = r.q[u].w;
        }
    }
}

template <int MODE> // 0 = general d (multiply test), 1 = power-of-two d (mask test)
__global__ __launch_bounds__(K1_THREADS, 1) void k_buzhash_candidates(const uint8_t* __restrict__ data,
                                                                       const PartDev* __restrict__ parts,
                                                                       const uint32_t* __restrict__ tile_part,
                                                                       uint32_t ntiles, DivTest dv,
                                                                       uint64_t* __restrict__ bm0,
                                                                       uint64_t* __restrict__ bm1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* tab = smem; // [256][TAB_REP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // provably wave-uniform: tile bookkeeping stays in SGPRs
    uint32_t* rows = smem + 256 * TAB_REP + wave * (WROWS * ROW_DW); // this wave's [WROWS][ROW_DW]

    // replicated substitution table: tab[v*TAB_REP + r] = T[v]
    if (tid < 256)
    {
        const uint32_t tv = c_buztab[tid];
        uint4 q = make_uint4(tv, tv, tv, tv);
        uint4* dst = reinterpret_cast<uint4*>(tab + tid * TAB_REP);
#pragma unroll
        for (int j = 0; j < TAB_REP / 4; ++j)
            dst[j] = q;
    }
    __syncthreads(); // the only barrier: table visible to every wave
    // The table is the first thing in LDS and the kernel has no static LDS, so its LDS address is 0 and a lookup address
    // is the v_perm result itself (no base add); checked, not assumed.
    if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)smem != 0u)
        __builtin_trap();
    const uint32_t lane4 = (uint32_t)lane * 4u;
    // x divisible by odd d' <=> x * d'^-1 mod 2^32 <= (2^32-1)/d'
    const uint32_t qodd = MODE == 1 ? 0u : 0xFFFFFFFFu / (dv.d >> dv.k2);

    const uint64_t nwt = (uint64_t)ntiles * 4u; // wave-tiles; wave-tile wt is quarter (wt & 3) of 16 KiB tile (wt >> 2)
    const uint64_t wstride = (uint64_t)gridDim.x * (uint64_t)K1_WAVES;
    uint64_t wt = (uint64_t)blockIdx.x * (uint64_t)K1_WAVES + (uint64_t)wave;
    if (wt >= nwt)
        return;
    PartDev pd = parts[tile_part[wt >> 2]];
    uint64_t span = ((wt >> 2) - pd.tile_base) * (uint64_t)TILE + (wt & 3u) * (uint64_t)WTILE; // part-relative
    TileRegs regs;
    tile_load(regs, data, pd, span, lane);

    for (;;)
    {
        __builtin_amdgcn_wave_barrier();
        tile_store(regs, rows, lane);
        __builtin_amdgcn_wave_barrier();

        // issue the next wave-tile's global loads now; they land while this one is being hashed
        const uint64_t next = wt + wstride;
        PartDev npd = pd;
        uint64_t next_span = 0;
        if (next < nwt)
        {
            npd = parts[tile_part[next >> 2]];
            next_span = ((next >> 2) - npd.tile_base) * (uint64_t)TILE + (next & 3u) * (uint64_t)WTILE;
            tile_load(regs, data, npd, next_span, lane);
        }

        const uint64_t q0 = span + (uint64_t)lane * RUN; // first byte of my run
        uint32_t mlo = 0, mhi = 0;
        if (q0 < pd.size)
        {
            // window bytes: j in [0,112): j<48 = the 48 bytes before the run, j>=48 = the run
            uint32_t win[28];
            const uint32_t* prev = rows + lane * ROW_DW;
#pragma unroll
            for (int i = 0; i < 12; ++i)
                win[i] = prev[4 + i];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                win[12 + i] = prev[ROW_DW + i];

            // T[byte] of the window bytes is fetched in two batches of independent LDS reads (80 + 32) that stream
            // through the LDS pipe back to back, each followed by pure register arithmetic.  The value that enters
            // the hash at step k leaves it at step k+48 (hpcdcchunker.c:294-296; rotl(T[out], 48 & 31)).
            uint32_t tv[48 + RUN];
            uint32_t h = 0;
    // byte offset of T[b] for this lane = b * 256 + lane * 4 = bytes {lane4, b, 0, 0}
#define LT_LOOKUP(j)                                                                                                 \
    tv[j] = lds_load_u32(__builtin_amdgcn_perm(win[(j) >> 2], lane4, 0x0c0c0400u | ((uint32_t)((j) & 3) << 8)))
#define LT_STEP(k)                                                                                                   \
    {                                                                                                                \
        h = rotl32(h, 1) ^ rotl32(tv[k], 16) ^ tv[48 + (k)];                                                         \
        bool pre;                                                                                                    \
        if (MODE == 1)                                                                                               \
            pre = (h & (dv.d - 1u)) == dv.d - 1u;                                                                    \
        else /* necessary: the odd part of d divides h+1 (3 ops); the exact test waits in the rare branch */        \
            pre = h * dv.inv + dv.inv <= qodd;                                                                       \
        if (__builtin_amdgcn_ballot_w64(pre) != 0ull) /* wave-uniform and rare (1 position in d's odd part) */       \
        {                                                                                                            \
            asm volatile(""); /* keep this a real scalar branch (no if-conversion of the bit-set) */                 \
            const bool hit = MODE == 1 ? pre : rotr32(h * dv.inv + dv.addc, dv.k2) <= dv.qlim; /* h % d == d-1 */   \
            /* 0/1 shifted by an inline constant: a select between 0 and 1 << k would park 64 constants in VGPRs */ \
            uint32_t hb;                                                                                             \
            asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(hb) : "s"(__builtin_amdgcn_ballot_w64(hit)));       \
            if ((k) < 32)                                                                                            \
                mlo |= hb << (k);                                                                                    \
            else                                                                                                     \
                mhi |= hb << ((k)-32);                                                                               \
        }                                                                                                            \
    }
#pragma unroll
            for (int j = 0; j < 80; ++j)
                LT_LOOKUP(j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 48; ++j)
                h = rotl32(h, 1) ^ tv[j];
#pragma unroll
            for (int k = 0; k < 32; ++k)
                LT_STEP(k)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 80; j < 48 + RUN; ++j)
                LT_LOOKUP(j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 32; k < RUN; ++k)
                LT_STEP(k)
#undef LT_LOOKUP
#undef LT_STEP
            // bit k <=> cut position p = q0+k+1 ; legal cuts are 48 <= p <= size
            uint64_t m = ((uint64_t)mhi << 32) | mlo;
            if (q0 < 47)
                m &= ~0ull << (47 - q0);
            const uint64_t remain = pd.size - q0; // >= 1
            if (remain < 64)
                m &= (1ull << remain) - 1ull;
            mlo = (uint32_t)m;
            mhi = (uint32_t)(m >> 32);
        }
        const uint64_t m = ((uint64_t)mhi << 32) | mlo;
        const uint64_t summary = __builtin_amdgcn_ballot_w64(m != 0ull);
        if (m != 0ull)
            bm0[pd.bm0_base + (span >> 6) + (uint64_t)lane] = m; // one word per 64-byte run
        if (lane == 0)
            bm1[pd.bm1_base + (span >> 12)] = summary;            // one word per 4 KiB

        if (next >= nwt)
            break;
        wt = next;
        pd = npd;
        span = next_span;
    }
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

int lthip_launch_buzhash(lthip_ctx* ctx, const lthip_plan* plan, const uint8_t* d_data, uint64_t* bm0, uint64_t* bm1)
{
    if (plan->ntiles == 0)
        return 0;
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
