/* oracle.h -- TEST INFRASTRUCTURE. CPU restatement of longtail's chunk -> hash -> compress hot path.
 *
 * Plain C99, no dependency on the reference tree, so it travels to the GPU box.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this library, and only as
 * the checker; the product library (longtail_amd/csrc) never includes or links anything from here.
 *
 * PARITY PINNED: every function below is checked (tests/test_oracle_*.py) against
 *   - the reference's own golden vectors (test/test.cpp:465-474 BLAKE3 KAT, :3423-3445 the 20 chunk
 *     ranges of testdata/chunker.input, :2092-2192 the 38-byte LZ4 payload), committed as fixtures
 *     under tests/golden/, and
 *   - oracle/_ref/liblongtail_ref.so = the reference compiled from /root/reference (oracle/Makefile).
 */
#ifndef LONGTAIL_ORACLE_H
#define LONGTAIL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hpcdc chunker (reference: lib/hpcdcchunker/longtail_hpcdcchunker.c) ---- */
#define LTO_WINDOW 48u /* ChunkerWindowSize, hpcdcchunker.c:12 */

extern const uint32_t lto_buzhash_table[256]; /* hpcdcchunker.c:23-88 */

/* hpcdcchunker.c:126-129 */
uint32_t lto_hpcdc_discriminator(uint32_t avg);

/* Streaming restatement of Longtail_HPCDCNextChunk + FeedChunker (hpcdcchunker.c:182-310) driven by
 * a feeder with StorageChunkFeederFunc semantics (src/longtail.c:1923-1960) over data[0..size).
 * Emulates the 4*max ring buffer, refill rule and rolling window literally.
 * Writes up to cap chunk lengths; returns the chunk count (may exceed cap: nothing past cap is written). */
uint64_t lto_hpcdc_chunk_stream(const uint8_t* data, uint64_t size, uint32_t min, uint32_t avg, uint32_t max,
                                uint32_t* out_lens, uint64_t cap);

/* Restatement of HPCDCChunker_NextChunkFromBuffer (hpcdcchunker.c:452-523), window-initialisation
 * quirk preserved (window seeded from buf[0..48) instead of buf[min-48..min)).  Returns the chunk
 * length taken from the front of buf[0..size). size must be > 0. */
uint64_t lto_hpcdc_next_from_buffer(const uint8_t* buf, uint64_t size, uint32_t min, uint32_t avg, uint32_t max);

/* "GPU-shaped" formulation: H(p) is a pure function of the 48 bytes before p. */
uint32_t lto_buzhash_at(const uint8_t* data, uint64_t p); /* requires p >= 48 */
/* candidate(p) <=> H(p) % d == d-1 ; sets bit p of bitmap (size/8+1 bytes, caller-zeroed) for p in [48,size] */
void lto_hpcdc_candidates(const uint8_t* data, uint64_t size, uint32_t d, uint8_t* bitmap);
/* selection walk over a candidate bitmap; same output contract as lto_hpcdc_chunk_stream */
uint64_t lto_hpcdc_select(const uint8_t* bitmap, uint64_t size, uint32_t min, uint32_t max, uint32_t* out_lens,
                          uint64_t cap);
/* convenience = candidates + select */
uint64_t lto_hpcdc_chunk_pure(const uint8_t* data, uint64_t size, uint32_t min, uint32_t avg, uint32_t max,
                              uint32_t* out_lens, uint64_t cap);

/* ---- BLAKE3 (reference: lib/blake3/ext/blake3.c, blake3_portable.c, blake3_impl.h; wrapper
 *      lib/blake3/longtail_blake3.c:81-102) ---- */
void lto_blake3(const void* data, size_t len, uint8_t out32[32]);
/* first 8 output bytes as little-endian u64 == Blake3Hash_HashBuffer */
uint64_t lto_blake3_u64(const void* data, size_t len);
/* batch: hashes[i] = blake3_u64(data + offsets[i], lens[i]) */
void lto_blake3_u64_many(const uint8_t* data, const uint64_t* offsets, const uint32_t* lens, uint64_t count,
                         uint64_t* hashes);

/* ---- LZ4 block format (reference: lib/lz4/ext/lz4.c; wrapper lib/lz4/longtail_lz4.c:47-102) ---- */
size_t lto_lz4_bound(size_t n); /* LZ4_COMPRESSBOUND, lz4.h:215 */
/* Bit-exact restatement of LZ4_compress_fast(src,dst,n,cap,1) (lz4.c:930-1338,1382-1403) for the
 * noDict case on a 64-bit little-endian host. Returns compressed size, 0 on failure. */
int lto_lz4_compress(const uint8_t* src, int n, uint8_t* dst, int cap);
/* Restatement of LZ4_decompress_safe semantics (lz4.c:2451): returns decoded size or <0 on malformed input. */
int lto_lz4_decompress(const uint8_t* src, int n, uint8_t* dst, int cap);

/* ---- synthetic data (include/longtail_synth.h; not reference-derived) ---- */
void lto_synth_fill(uint8_t* dst, uint64_t nbytes, uint64_t seed, uint64_t byte_offset, int kind);
uint64_t lto_synth_asset_seed(uint64_t tree_seed, uint64_t index);
/* SURVEY.md §8(c,d)'s xorshift64 stream: s ^= s << 13; s ^= s >> 7; s ^= s << 17, the state after each step as 8 LE bytes */
void lto_xorshift_fill(uint8_t* dst, uint64_t nbytes, uint64_t seed);

/* ---- whole-path CPU timing leg for bench.py (single thread): chunk + hash + lz4 over parts ---- */
struct lto_ingest_result
{
    uint64_t chunk_count;
    uint64_t hash_xor;       /* xor of all chunk hashes */
    uint64_t hash_sum;       /* sum of all chunk hashes (mod 2^64) */
    uint64_t compressed_bytes;
    double seconds_chunk;
    double seconds_hash;
    double seconds_compress;
};
int lto_ingest(const uint8_t* data, uint64_t size, uint64_t part_size, uint32_t target_chunk_size,
               uint32_t block_size, int do_compress, struct lto_ingest_result* out);

/* ---- zstd: host instantiation of the GPU's block encoder (zstd_model.c) ---- */
size_t ltz_model_bound(size_t n);
int ltz_model_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n);
int ltz_model_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n);
/* 0: the one-lane model; 1 / 2: the sub-block encoder runs with SIXTY-FOUR lanes on the host (zstd_model_lanes.c: the code the kernel
 * compiles, as 64 fibers), the lanes in ascending / descending order between two meeting points.  ltz_model_compress returns -2 when the
 * lanes do not keep step; ltz_lanes_bad_site(0 / 1) then give the two source lines of zstd_block_core.h they stood at. */
void ltz_model_lanes64(int mode);
uint32_t ltz_lanes_bad_site(int k);
uint32_t ltz_model_encode_block_src(const void* unit_meta, const uint8_t* unit_lits, const uint64_t* unit_recs, uint32_t nunits,
                                    uint32_t raw_size, const uint8_t* src, uint8_t* out); /* units without a sequence: literals = src */
uint32_t ltz_model_encode_block(const void* unit_meta, const uint8_t* unit_lits, const uint64_t* unit_recs, uint32_t nunits,
                                uint32_t raw_size, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
