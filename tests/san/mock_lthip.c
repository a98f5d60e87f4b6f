/* mock_lthip.c -- TEST INFRASTRUCTURE: a CPU stand-in for the lthip_* device API, built ONLY into tests/san/plugin_san (the
 * AddressSanitizer / UBSan run of the plain-C plugin layer, which cannot run in a sanitized process on a GPU-less container
 * otherwise).  "Device memory" is host memory, the kernels are the oracle's C restatement.  Nothing here is part of the
 * product: liblongtail_hip.so never links it (tests/test_abi.py checks the product for oracle references). */
#include "../../include/longtail_hip.h"
#include "../../oracle/oracle.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>

int ltz_model_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n);
int ltz_model_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n);

struct lthip_ctx
{
    int dummy;
};
#define MOCK_MAX_PARTS 64
struct lthip_plan
{
    uint64_t capacity, size; /* single-part view (part 0) */
    uint32_t min, avg, max;
    uint32_t nparts, cap_parts;
    uint64_t offs[MOCK_MAX_PARTS], sizes[MOCK_MAX_PARTS], cap_bytes;
};

static int g_fail_alloc_after = -1; /* fault injection: the n-th allocation from now fails */
void mock_fail_alloc_after(int n) { g_fail_alloc_after = n; }
static int alloc_fails(void)
{
    if (g_fail_alloc_after < 0)
        return 0;
    if (g_fail_alloc_after-- == 0)
        return 1;
    return 0;
}

int lthip_device_count(void) { return 1; }
int lthip_ctx_create(int device, void* s, lthip_ctx** out)
{
    (void)device;
    (void)s;
    *out = (lthip_ctx*)calloc(1, sizeof(lthip_ctx));
    return *out ? 0 : ENOMEM;
}
void lthip_ctx_destroy(lthip_ctx* c) { free(c); }
int lthip_ctx_sync(lthip_ctx* c) { return c ? 0 : EINVAL; }
const char* lthip_ctx_error(const lthip_ctx* c)
{
    (void)c;
    return "";
}
int lthip_malloc_device(lthip_ctx* c, size_t n, void** out)
{
    (void)c;
    if (alloc_fails())
        return ENOMEM;
    *out = malloc(n ? n : 16);
    return *out ? 0 : ENOMEM;
}
void lthip_free_device(lthip_ctx* c, void* p)
{
    (void)c;
    free(p);
}
int lthip_malloc_pinned(lthip_ctx* c, size_t n, void** out) { return lthip_malloc_device(c, n, out); }
void lthip_free_pinned(lthip_ctx* c, void* p)
{
    (void)c;
    free(p);
}
int lthip_copy_h2d(lthip_ctx* c, void* d, const void* h, size_t n)
{
    (void)c;
    memcpy(d, h, n);
    return 0;
}
int lthip_copy_d2h(lthip_ctx* c, void* h, const void* d, size_t n)
{
    (void)c;
    memcpy(h, d, n);
    return 0;
}
int lthip_set_blocking_waits(int device, int on)
{
    (void)device;
    (void)on;
    return 0;
}
int lthip_link_copy(lthip_ctx* c, void* dst, const void* src, size_t n)
{
    (void)c;
    memcpy(dst, src, n);
    return 0;
}
int lthip_plan_create(lthip_ctx* c, uint32_t parts, const uint64_t* offs, const uint64_t* sizes, uint32_t mn, uint32_t av, uint32_t mx,
                      lthip_plan** out)
{
    (void)c;
    if (parts < 1 || parts > MOCK_MAX_PARTS || offs[0] != 0 || mn < 48 || mn > av || av > mx)
        return EINVAL;
    if (alloc_fails())
        return ENOMEM;
    lthip_plan* p = (lthip_plan*)calloc(1, sizeof *p);
    if (!p)
        return ENOMEM;
    p->capacity = p->size = sizes[0];
    p->nparts = p->cap_parts = parts;
    for (uint32_t i = 0; i < parts; ++i)
    {
        p->offs[i] = offs[i];
        p->sizes[i] = sizes[i];
        p->cap_bytes += sizes[i];
    }
    p->min = mn;
    p->avg = av;
    p->max = mx;
    *out = p;
    return 0;
}
int lthip_plan_resize_single(lthip_ctx* c, lthip_plan* p, uint64_t size)
{
    (void)c;
    if (size > p->capacity)
        return EINVAL;
    p->size = size;
    p->sizes[0] = size;
    return 0;
}
int lthip_plan_reaim(lthip_ctx* c, lthip_plan* p, uint32_t parts, const uint64_t* offs, const uint64_t* sizes)
{
    (void)c;
    uint64_t bytes = 0;
    if (parts < 1 || parts > p->cap_parts)
        return EINVAL;
    for (uint32_t i = 0; i < parts; ++i)
    {
        if (offs[i] & 15u)
            return EINVAL;
        p->offs[i] = offs[i];
        p->sizes[i] = sizes[i];
        bytes += sizes[i];
    }
    if (bytes > p->cap_bytes)
        return EINVAL;
    p->nparts = parts;
    p->size = sizes[0];
    return 0;
}
void lthip_plan_destroy(lthip_ctx* c, lthip_plan* p)
{
    (void)c;
    free(p);
}
uint64_t lthip_plan_chunk_capacity(const lthip_plan* p)
{
    uint64_t cap = 0;
    for (uint32_t i = 0; i < p->nparts; ++i)
        cap += p->sizes[i] ? p->sizes[i] / p->min + 1 : 0;
    return cap ? cap : 1;
}
int lthip_chunk_hash(lthip_ctx* c, const lthip_plan* p, const void* d, uint64_t* offs, uint32_t* lens, uint64_t* hashes, uint32_t* first,
                     uint64_t* out_total)
{
    (void)c;
    uint64_t n_all = 0;
    for (uint32_t part = 0; part < p->nparts; ++part)
    {
        const uint64_t size = p->sizes[part];
        const uint8_t* base = (const uint8_t*)d + p->offs[part];
        const uint64_t cap = size / p->min + 2;
        uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * cap);
        if (!tmp)
            return ENOMEM;
        const uint64_t n = size ? lto_hpcdc_chunk_stream(base, size, p->min, p->avg, p->max, tmp, cap) : 0;
        uint64_t o = (uint64_t)(base - (const uint8_t*)d);
        first[part] = (uint32_t)n_all;
        for (uint64_t i = 0; i < n; ++i)
        {
            offs[n_all + i] = o;
            lens[n_all + i] = tmp[i];
            o += tmp[i];
        }
        free(tmp);
        if (hashes)
            lto_blake3_u64_many((const uint8_t*)d, offs + n_all, lens + n_all, n, hashes + n_all);
        n_all += n;
    }
    first[p->nparts] = (uint32_t)n_all;
    if (out_total)
        *out_total = n_all;
    return 0;
}
int lthip_chunk_from_buffer(lthip_ctx* c, const void* d, uint64_t size, uint32_t mn, uint32_t av, uint32_t mx, uint64_t* out_len)
{
    (void)c;
    *out_len = lto_hpcdc_next_from_buffer((const uint8_t*)d, size, mn, av, mx);
    return 0;
}
/* streaming: the "device" stack block holds a pointer to everything the stream has sent so far; the accumulators stay reachable from a
 * global list so that an abandoned stream (an injected failure) is not a leak report */
struct mock_stream
{
    uint64_t magic;
    uint8_t* acc;
    size_t len, cap;
};
static uint8_t* g_stream_accs[256];
static int g_stream_acc_count;
int lthip_b3_stream_batch(lthip_ctx* c, const void* d, uint64_t batch_index, void* d_stack)
{
    (void)c;
    struct mock_stream* st = (struct mock_stream*)d_stack;
    if (batch_index == 0)
    {
        memset(st, 0, sizeof *st);
        st->magic = 0x53545245414dull;
    }
    if (st->magic != 0x53545245414dull || st->len != (size_t)batch_index << 20)
        return EINVAL;
    if (st->len + (1u << 20) > st->cap)
    {
        const size_t cap = st->cap ? st->cap * 2 : (4u << 20);
        uint8_t* n = (uint8_t*)realloc(st->acc, cap);
        if (!n)
            return ENOMEM;
        for (int i = 0; i < g_stream_acc_count; ++i)
            if (g_stream_accs[i] == st->acc)
                g_stream_accs[i] = n;
        if (!st->acc && g_stream_acc_count < 256)
            g_stream_accs[g_stream_acc_count++] = n;
        st->acc = n;
        st->cap = cap;
    }
    memcpy(st->acc + st->len, d, 1u << 20);
    st->len += 1u << 20;
    return 0;
}
int lthip_b3_stream_final(lthip_ctx* c, const void* d_tail, uint32_t tail_len, uint64_t batch_count, const void* d_stack, uint64_t* out)
{
    (void)c;
    if (batch_count == 0)
    {
        const uint64_t off = 0;
        lto_blake3_u64_many((const uint8_t*)(tail_len ? d_tail : (const void*)""), &off, &tail_len, 1, out);
        return 0;
    }
    struct mock_stream* st = (struct mock_stream*)d_stack;
    if (st->magic != 0x53545245414dull || st->len != (size_t)batch_count << 20 || !tail_len)
        return EINVAL;
    uint8_t* all = (uint8_t*)malloc(st->len + tail_len);
    if (!all)
        return ENOMEM;
    memcpy(all, st->acc, st->len);
    memcpy(all + st->len, d_tail, tail_len);
    const uint64_t off = 0;
    const uint32_t len = (uint32_t)(st->len + tail_len);
    lto_blake3_u64_many(all, &off, &len, 1, out);
    free(all);
    for (int i = 0; i < g_stream_acc_count; ++i)
        if (g_stream_accs[i] == st->acc)
            g_stream_accs[i] = g_stream_accs[--g_stream_acc_count];
    free(st->acc);
    st->acc = 0;
    return 0;
}
int lthip_hash_runs_u64(lthip_ctx* c, const uint64_t* v, const uint32_t* first, uint32_t n, uint64_t* out)
{
    (void)c;
    for (uint32_t i = 0; i < n; ++i)
    {
        const uint64_t off = 8ull * first[i];
        const uint32_t len = 8u * (first[i + 1] - first[i]);
        lto_blake3_u64_many((const uint8_t*)v, &off, &len, 1, out + i);
    }
    return 0;
}
int lthip_hash_runs_u64_bounded(lthip_ctx* c, const uint64_t* v, const uint32_t* first, uint32_t n, uint64_t total_bound, uint64_t run_bound,
                                uint64_t* out)
{
    /* the bounds the caller states must hold (the device launch is sized by them) */
    if (n && total_bound && (first[n] - first[0] > total_bound))
        return EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (run_bound && first[i + 1] - first[i] > run_bound)
            return EINVAL;
    return lthip_hash_runs_u64(c, v, first, n, out);
}
int lthip_hash_one(lthip_ctx* c, const void* in, uint32_t len, uint64_t* out)
{
    (void)c;
    const uint64_t off = 0;
    if (len > 65536u)
        return EINVAL;
    lto_blake3_u64_many((const uint8_t*)in, &off, &len, 1, out);
    return 0;
}
int lthip_hash_ranges(lthip_ctx* c, const void* d, uint64_t n, const uint64_t* offs, const uint32_t* lens, uint32_t max_len, uint64_t* out)
{
    (void)c;
    (void)max_len;
    lto_blake3_u64_many((const uint8_t*)d, offs, lens, n, out);
    return 0;
}
size_t lthip_lz4_bound(size_t n) { return n > 0x7E000000u ? 0 : n + n / 255 + 16; }
size_t lthip_zstd_bound(size_t n) { return n + (n >> 8) + (n < (128u << 10) ? ((128u << 10) - n) >> 11 : 0); }
int lthip_lz4_compress_blocks(lthip_ctx* c, const void* s, uint32_t nb, const uint64_t* so, const uint32_t* ss, void* d, const uint64_t* dof,
                              const uint32_t* dc, uint32_t* out, int seg)
{
    (void)c;
    (void)seg;
    for (uint32_t b = 0; b < nb; ++b)
    {
        const int k = lto_lz4_compress((const uint8_t*)s + so[b], (int)ss[b], (uint8_t*)d + dof[b], (int)dc[b]);
        out[b] = k > 0 ? (uint32_t)k : 0u;
    }
    return 0;
}
int lthip_lz4_decompress_blocks(lthip_ctx* c, const void* s, uint32_t nb, const uint64_t* so, const uint32_t* ss, void* d, const uint64_t* dof,
                                const uint32_t* dc, uint32_t* out)
{
    (void)c;
    for (uint32_t b = 0; b < nb; ++b)
    {
        const int k = lto_lz4_decompress((const uint8_t*)s + so[b], (int)ss[b], (uint8_t*)d + dof[b], (int)dc[b]);
        out[b] = k >= 0 ? (uint32_t)k : 0xFFFFFFFFu;
    }
    return 0;
}
int lthip_zstd_compress_blocks(lthip_ctx* c, const void* s, uint32_t nb, const uint64_t* so, const uint32_t* ss, void* d, const uint64_t* dof,
                               const uint32_t* dc, uint32_t* out);
int lthip_zstd_compress_blocks_q(lthip_ctx* c, const void* s, uint32_t nb, const uint64_t* so, const uint32_t* ss, void* d, const uint64_t* dof,
                                 const uint32_t* dc, uint32_t* out, int quality)
{
    if (quality < 0 || quality > 2)
        return EINVAL;
    return lthip_zstd_compress_blocks(c, s, nb, so, ss, d, dof, dc, out); /* (the model has one parse) */
}
int lthip_zstd_compress_blocks(lthip_ctx* c, const void* s, uint32_t nb, const uint64_t* so, const uint32_t* ss, void* d, const uint64_t* dof,
                               const uint32_t* dc, uint32_t* out)
{
    (void)c;
    for (uint32_t b = 0; b < nb; ++b)
    {
        /* the model wants its own (larger) bound: encode into a scratch buffer and copy what fits the caller's capacity */
        const size_t bound = ss[b] + (ss[b] >> 8) + 64 + 3 * ((size_t)ss[b] / (128u << 10) + 1);
        uint8_t* tmp = (uint8_t*)malloc(bound);
        size_t k = 0;
        out[b] = 0u;
        if (tmp && ltz_model_compress((const uint8_t*)s + so[b], ss[b], tmp, bound, &k) == 0 && k <= dc[b])
        {
            memcpy((uint8_t*)d + dof[b], tmp, k);
            out[b] = (uint32_t)k;
        }
        free(tmp);
    }
    return 0;
}
int lthip_zstd_decompress_blocks(lthip_ctx* c, const void* s, uint32_t nb, const uint64_t* so, const uint32_t* ss, void* d, const uint64_t* dof,
                                 const uint32_t* dc, uint32_t* out)
{
    (void)c;
    for (uint32_t b = 0; b < nb; ++b)
    {
        size_t k = 0;
        out[b] = ltz_model_decompress((const uint8_t*)s + so[b], ss[b], (uint8_t*)d + dof[b], dc[b], &k) == 0 ? (uint32_t)k : 0xFFFFFFFFu;
    }
    return 0;
}

int lthip_zstd_quality_of_settings(uint32_t settings_id)
{
    const uint32_t low = settings_id & 0xFFu;
    return low == '4' ? 1 : (low == '3' || low == '5') ? 2 : 0;
}
