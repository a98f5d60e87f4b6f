/* plugin_codec.c -- Longtail_CompressionAPI objects (LZ4, ZStd) on the GPU; C99 host code over lthip_*.
 *
 * Mirrors the reference wrappers
 *   lib/lz4/longtail_lz4.c   : type id 'lz42' (:10), CreateForLZ4 (:12-23), bound (:47-50), Compress (:52-77,
 *                              0 bytes -> ENOMEM), Decompress (:79-102, malformed -> EBADF)
 *   lib/zstd/longtail_zstd.c : type ids 'ztd1'..'ztd5' (:17-22), CreateForZstd (:30-41), bound (:72-75),
 *                              Compress (:107-141), Decompress (:144-177)
 * Called by CompressBlock / DecompressBlock only (lib/compressblockstore/longtail_compressblockstore.c:106,121,321);
 * `compressed` there is &header[2], i.e. only 4-byte aligned host memory -- nothing here assumes more.
 */
#include "plugin_common.h"

#define LTP_LZ4_TYPE ((((uint32_t)'l') << 24) + (((uint32_t)'z') << 16) + (((uint32_t)'4') << 8) + ((uint32_t)'2'))
#define LTP_ZSTD_TYPE ((((uint32_t)'z') << 24) + (((uint32_t)'t') << 16) + (((uint32_t)'d') << 8))

/* How a block crosses the link.  The copy engines of the MI355X boxes measured serve ONE direction at a time (h2d and d2h queued on two
 * streams take the sum of their times, profiles/r05_pcie_duplex.json) while a copy made by the compute units runs beside an engine
 * copy in the other direction: payloads come back through lthip_link_copy (k_link_copy), blocks go up through the engines, so that
 * one worker's download overlaps another worker's upload. */
#ifdef LTP_D2H_ENGINE
#define LTP_FETCH lthip_copy_d2h
#else
#define LTP_FETCH lthip_link_copy
#endif
#ifdef LTP_H2D_KERNEL
#define LTP_SEND lthip_link_copy
#else
#define LTP_SEND lthip_copy_h2d
#endif

enum
{
    CODEC_LZ4 = 0,
    CODEC_ZSTD = 1
};

struct HipCodecAPI
{
    struct Longtail_CompressionAPI api;
    int codec;
};

static pthread_mutex_t g_codec_lock = PTHREAD_MUTEX_INITIALIZER;
static int g_codec_live;

static void HipCodec_Dispose(struct Longtail_API* api)
{
    ltp_free(api);
    pthread_mutex_lock(&g_codec_lock);
    const int last = --g_codec_live == 0;
    pthread_mutex_unlock(&g_codec_lock);
    if (last)
        ltp_codec_batch_shutdown(); /* the dispatcher thread and its context go with the last CompressionAPI */
}

static size_t HipCodec_GetMaxCompressedSize(struct Longtail_CompressionAPI* compression_api, uint32_t settings_id, size_t size)
{
    (void)settings_id;
    struct HipCodecAPI* a = (struct HipCodecAPI*)compression_api;
    return a->codec == CODEC_LZ4 ? lthip_lz4_bound(size) : lthip_zstd_bound(size);
}

static int codec_batching(void)
{
    static int v = -1;
    if (v < 0)
    {
        const char* e = getenv("LONGTAIL_HIP_CODEC_BATCH");
        v = !(e && e[0] == '0');
    }
    return v;
}

/* what the codec reported for my block -> the reference wrappers' error codes, the payload into the caller's buffer */
static int fetch_payload(struct ltp_thread_state* ts, int codec, int decompress, uint32_t produced, char* dst, size_t cap, size_t* out_n)
{
    if (decompress)
    {
        if (produced == 0xFFFFFFFFu)
            return codec == CODEC_ZSTD ? EINVAL : EBADF; /* longtail_lz4.c:95-99; longtail_zstd.c:168-172 */
    }
    else if (produced == 0)
        return ENOMEM; /* longtail_lz4.c:70-74 */
    if (produced > cap)
        return EIO;
    int err = 0;
    if (produced)
    {
        uint8_t* stage = (uint8_t*)ts->h_pin.p + 64;
        err = LTP_FETCH(ts->ctx, stage, ts->d_out.p, produced);
        if (!err)
            err = lthip_ctx_sync(ts->ctx);
        if (!err)
            memcpy(dst, stage, produced);
    }
    if (!err)
        *out_n = produced;
    return err;
}

/* One block through the bulk API: host -> pinned staging -> device -> kernels -> pinned staging -> host.  The caller's buffers
 * are pageable (`compressed` is &header[2] of a Longtail_Alloc block, compressblockstore.c:117-125); copying them through the
 * calling thread's own pinned buffers keeps the DMA engine at full rate and lets the 32..256 bikeshed workers that call Compress
 * concurrently (one context + stream each) overlap their copies and kernels.  Two synchronisations per call: the payload size
 * has to be known before the payload is fetched. */
static int run_block(int codec, int decompress, int quality, const char* src, char* dst, size_t n, size_t cap, size_t* out_n)
{
    struct ltp_thread_state* ts = ltp_thread_state_get();
    if (!ts)
        return ENODEV;
    if (n > 0x7E000000u || cap > 0xFFFFFFF0u)
        return EINVAL;
    lthip_ctx* ctx = ts->ctx;
    int err = ltp_dev_reserve(ctx, &ts->d_in, n + 64);
    if (!err)
        err = ltp_dev_reserve(ctx, &ts->d_out, cap + 64);
    if (!err)
        err = ltp_dev_reserve(ctx, &ts->d_aux, 64);
    if (!err)
        err = ltp_pin_reserve(ctx, &ts->h_pin, 64 + (n > cap ? n : cap));
    if (err)
        return err;
    uint8_t* stage = (uint8_t*)ts->h_pin.p + 64;
    if (n)
    {
        /* (the runtime's own pageable path -- hipMemcpyAsync straight from the caller's buffer, no staging copy -- was measured in round 6:
         * WriteContent 49 -> 23 GB/s on compressible data, 39 -> 33 on random: it pins on the fly, page by page, on the calling thread) */
        memcpy(stage, src, n);
        err = LTP_SEND(ctx, ts->d_in.p, stage, n);
        if (err)
            return err;
    }
    const uint64_t zero = 0;
    const uint32_t sz = (uint32_t)n, dcap = (uint32_t)cap;
    if (codec_batching())
    {
        /* my block is on the device (my stream, synchronised): through the codec together with the other threads' blocks */
        uint32_t produced = 0;
        if (n)
            err = lthip_ctx_sync(ctx);
        if (!err)
            err = ltp_codec_batch(codec, decompress, quality, ts->d_in.p, sz, ts->d_out.p, dcap, &produced);
        if (err)
            return err;
        return fetch_payload(ts, codec, decompress, produced, dst, cap, out_n);
    }
    if (decompress && codec == CODEC_ZSTD)
        err = lthip_zstd_decompress_blocks(ctx, ts->d_in.p, 1, &zero, &sz, ts->d_out.p, &zero, &dcap, (uint32_t*)ts->d_aux.p);
    else if (decompress)
        err = lthip_lz4_decompress_blocks(ctx, ts->d_in.p, 1, &zero, &sz, ts->d_out.p, &zero, &dcap, (uint32_t*)ts->d_aux.p);
    else if (codec == CODEC_LZ4)
        err = lthip_lz4_compress_blocks(ctx, ts->d_in.p, 1, &zero, &sz, ts->d_out.p, &zero, &dcap, (uint32_t*)ts->d_aux.p, 0);
    else
        err = lthip_zstd_compress_blocks_q(ctx, ts->d_in.p, 1, &zero, &sz, ts->d_out.p, &zero, &dcap, (uint32_t*)ts->d_aux.p, quality);
    uint32_t* h_size = (uint32_t*)ts->h_pin.p;
    if (!err)
        err = lthip_copy_d2h(ctx, h_size, ts->d_aux.p, 4);
    if (!err)
        err = lthip_ctx_sync(ctx); /* the input staging is free again from here on */
    if (err)
        return err;
    return fetch_payload(ts, codec, decompress, *h_size, dst, cap, out_n);
}

static int HipCodec_Compress(struct Longtail_CompressionAPI* compression_api, uint32_t settings_id, const char* uncompressed,
                             char* compressed, size_t uncompressed_size, size_t max_compressed_size,
                             size_t* out_compressed_size)
{
    if (!compression_api || !compressed || !out_compressed_size || (uncompressed_size && !uncompressed))
        return EINVAL;
    struct HipCodecAPI* a = (struct HipCodecAPI*)compression_api;
    if (a->codec == CODEC_LZ4 && settings_id != LTP_LZ4_TYPE)
        return EINVAL; /* longtail_lz4.c:33 */
    if (a->codec == CODEC_ZSTD && (settings_id & 0xffffff00u) != LTP_ZSTD_TYPE)
        return EINVAL;
    /* SettingsIDToCompressionSetting (lib/zstd/longtail_zstd.c:43-60): 'ztd1' -> level 0 (= zstd's default, 3), 'ztd2' -> 3, 'ztd3' ->
     * 22, 'ztd4' -> 8, 'ztd5' -> the LOW type id itself (the macro shadows the constant, :12 vs :22), which zstd clamps to 22; any
     * other id -> 0, the default -- the reference does not reject it, neither does this.  Levels 3 / 8 / 22 become the three parses
     * of lthip_zstd_compress_blocks_q. */
    int quality = LTHIP_ZSTD_Q_DEFAULT;
    if (a->codec == CODEC_ZSTD)
        quality = lthip_zstd_quality_of_settings(settings_id);
    return run_block(a->codec, 0, quality, uncompressed, compressed, uncompressed_size, max_compressed_size, out_compressed_size);
}

static int HipCodec_Decompress(struct Longtail_CompressionAPI* compression_api, const char* compressed, char* uncompressed,
                               size_t compressed_size, size_t max_uncompressed_size, size_t* out_uncompressed_size)
{
    if (!compression_api || !compressed || !out_uncompressed_size || (max_uncompressed_size && !uncompressed))
        return EINVAL;
    struct HipCodecAPI* a = (struct HipCodecAPI*)compression_api;
    return run_block(a->codec, 1, 0, compressed, uncompressed, compressed_size, max_uncompressed_size, out_uncompressed_size);
}

static struct Longtail_CompressionAPI* make_codec(int codec)
{
    if (lthip_device_count() <= 0)
        return 0; /* no GPU: fail loudly */
    struct HipCodecAPI* a = (struct HipCodecAPI*)ltp_alloc("HipCompressionAPI", sizeof *a);
    if (!a)
        return 0;
    a->api.m_API.Dispose = HipCodec_Dispose;
    a->api.GetMaxCompressedSize = HipCodec_GetMaxCompressedSize;
    a->api.Compress = HipCodec_Compress;
    a->api.Decompress = HipCodec_Decompress;
    a->codec = codec;
    pthread_mutex_lock(&g_codec_lock);
    ++g_codec_live;
    pthread_mutex_unlock(&g_codec_lock);
    return &a->api;
}

uint32_t Longtail_GetHipLZ4DefaultQuality(void) { return LTP_LZ4_TYPE; }

struct Longtail_CompressionAPI* Longtail_CreateHipLZ4CompressionAPI(void) { return make_codec(CODEC_LZ4); }

struct Longtail_CompressionAPI* Longtail_CompressionRegistry_CreateForHipLZ4(uint32_t compression_type, uint32_t* out_settings)
{
    if (compression_type != LTP_LZ4_TYPE)
        return 0;
    if (out_settings)
        *out_settings = compression_type;
    return Longtail_CreateHipLZ4CompressionAPI();
}

struct Longtail_CompressionAPI* Longtail_CreateHipZStdCompressionAPI(void) { return make_codec(CODEC_ZSTD); }

struct Longtail_CompressionAPI* Longtail_CompressionRegistry_CreateForHipZstd(uint32_t compression_type, uint32_t* out_settings)
{
    if ((compression_type & 0xffffff00u) != LTP_ZSTD_TYPE)
        return 0;
    if (out_settings)
        *out_settings = compression_type;
    return Longtail_CreateHipZStdCompressionAPI();
}
