/* longtail_synth.h -- deterministic synthetic asset bytes, shared by the HIP generator kernel
 * (longtail_amd/csrc/synth.hip), the oracle (oracle/synth_oracle.c) and bench.py.
 *
 * NOT derived from the reference (longtail ships no data generator); this only defines the
 * workload of SURVEY.md §8(d) in a COUNTER-BASED form so that a GPU can fill 64 GiB in parallel
 * and a CPU can reproduce any byte range: byte b of asset `seed` is a pure function of
 * (seed, b / 8, kind).
 *
 *   kind 0  "random"        incompressible: every 8-byte word = splitmix64 finaliser of (seed, word index).
 *   kind 1  "mixed"         per 64 KiB region one of four classes (chosen by hash of seed+region):
 *                           0 random | 1 32-byte records (24 fixed bytes from a 64-entry dictionary + 8
 *                           varying) | 2 8-byte tokens from a 1024-entry vocabulary | 3 256-byte constant
 *                           lines.  Gives LZ4/ZStd real matches at several offsets and match lengths.
 *   kind 2  "zero"          all zero bytes (no Buzhash candidates: every chunk is `max` long).
 *   kind 10+c               "mixed" with every region forced to class c (codec tuning: one match structure at a time).
 */
#ifndef LONGTAIL_SYNTH_H
#define LONGTAIL_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LT_SYNTH_FN static __host__ __device__ __forceinline__
#else
#define LT_SYNTH_FN static inline
#endif

#define LT_SYNTH_RANDOM 0
#define LT_SYNTH_MIXED 1
#define LT_SYNTH_ZERO 2
#define LT_SYNTH_CLASS0 10

LT_SYNTH_FN uint64_t lt_synth_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Seed of asset number `index` in a tree (bench.py and the tests use the same rule). */
LT_SYNTH_FN uint64_t lt_synth_asset_seed(uint64_t tree_seed, uint64_t index)
{
    return lt_synth_mix(tree_seed + 0x9E3779B97F4A7C15ull * (index + 1));
}

/* The 8-byte little-endian word number `w` (byte offset 8*w) of the asset with seed `seed`. */
LT_SYNTH_FN uint64_t lt_synth_word(uint64_t seed, uint64_t w, int kind)
{
    const uint64_t r = lt_synth_mix(seed + 0x9E3779B97F4A7C15ull * (w + 1));
    if (kind == LT_SYNTH_RANDOM)
        return r;
    if (kind == LT_SYNTH_ZERO)
        return 0;
    /* LT_SYNTH_MIXED */
    {
        const uint64_t region = w >> 13; /* 64 KiB = 8192 words */
        const uint64_t rr = lt_synth_mix(seed ^ (0xD1B54A32D192ED03ull * (region + 1)));
        const unsigned cls = kind >= LT_SYNTH_CLASS0 ? (unsigned)(kind - LT_SYNTH_CLASS0) & 3u : (unsigned)(rr & 3u);
        if (cls == 0)
            return r;
        if (cls == 1)
        {
            /* 32-byte records: words 0..2 from dictionary entry id, word 3 varies */
            const uint64_t rec = w >> 2;
            const unsigned k = (unsigned)(w & 3u);
            const uint64_t id = lt_synth_mix(rr + rec * 0x2545F4914F6CDD1Dull) & 63u;
            if (k == 3)
                return r;
            return lt_synth_mix(0x5EED0000ull + id * 4u + k);
        }
        if (cls == 2)
            return lt_synth_mix(0x70CAB000ull + (r & 1023u)) & 0x7F7F7F7F7F7F7F7Full;
        /* cls 3: one value per 256-byte line, only 4 distinct byte values in it */
        return lt_synth_mix(rr + (w >> 5)) & 0x0303030303030303ull;
    }
}

#endif /* LONGTAIL_SYNTH_H */
