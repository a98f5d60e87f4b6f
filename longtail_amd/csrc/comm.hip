// comm.hip -- the collectives of the multi-GPU exchange behind the C ABI, one process per GPU:
//   all-gather   per-job chunk counts, chunk hashes, chunk lengths, the first-seen answers (lthip_exchange_layout)
//   all-to-all   the sharded first-seen table: every (hash, position) to the rank that owns the hash, the answers back
// Transport 1 = RCCL on the context's stream (ncclAllGather; grouped ncclSend / ncclRecv for the all-to-all).  RCCL is bound at
// run time (dlopen of librccl.so.1): a single-GPU process never loads it, and the library has no link-time dependency on it.
// Transport 2 = a shared-memory segment on the host ("shm", chosen by whoever makes the id: LTHIP_COMM_TRANSPORT=shm), for boxes
// that do not have N GPUs: the same entry points and the same call sequence with N processes on one GPU (or, with ctx == NULL and
// host pointers, on none) -- a functional stand-in for the launch, the handshake and the exchange, never a measured path.
// The 128-byte id travels between the processes by whatever means the embedder has (bench.py: a file, or the torch.distributed
// store it was launched with).
#include "lthip_internal.h"

#include <atomic>
#include <dlfcn.h>
#include <fcntl.h>
#include <link.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace
{
struct Rccl
{
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr; // optional: the transport's own text of what went wrong
    bool ok = false;
    char path[512] = {0};  // where the library was found (lthip_comm_library), or what was tried
    char how[64] = {0};    // which rule found it
    // "<ncclGetErrorString>: <ncclGetLastError>" of a failed call (a diagnosis for the handshake's JSON line, not just a code)
    const char* text(ncclResult_t e, ncclComm_t c, char* buf, size_t cap) const
    {
        const char* last = GetLastError ? GetLastError(c) : nullptr;
        snprintf(buf, cap, "%s%s%s", GetErrorString ? GetErrorString(e) : "?", last && last[0] ? ": " : "", last && last[0] ? last : "");
        return buf;
    }
};

// An RCCL that is ALREADY mapped into this process (torch's bundled torch/lib/librccl.so on a box without /opt/rocm/lib/librccl.so.1,
// or one the embedder linked) is the one to bind to: two copies of the runtime in one process do not share communicators.
struct FindLoaded
{
    char rccl[512];
    char torch_dir[512];
};
int find_loaded_cb(struct dl_phdr_info* info, size_t, void* data)
{
    FindLoaded* f = (FindLoaded*)data;
    const char* n = info->dlpi_name;
    if (!n || !n[0])
        return 0;
    const char* base = strrchr(n, '/');
    base = base ? base + 1 : n;
    if (!f->rccl[0] && !strncmp(base, "librccl.so", 10) && strlen(n) < sizeof f->rccl)
        strcpy(f->rccl, n);
    if (!f->torch_dir[0] && (!strncmp(base, "libtorch_hip.so", 15) || !strncmp(base, "libtorch_cpu.so", 15)) && (size_t)(base - n) < sizeof f->torch_dir)
    {
        memcpy(f->torch_dir, n, (size_t)(base - n));
        f->torch_dir[base - n] = 0;
    }
    return 0;
}

Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        // 1. $LTHIP_RCCL_PATH (the embedder knows best)  2. a copy already mapped into the process  3. the loader's search path and
        // ROCm's directory  4. beside a mapped libtorch (torch/lib/librccl.so: where the GPU boxes of this project keep theirs)
        auto open_as = [&](const char* name, const char* how) {
            if (x.so || !name || !name[0])
                return;
            if ((x.so = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
            {
                snprintf(x.path, sizeof x.path, "%s", name);
                snprintf(x.how, sizeof x.how, "%s", how);
            }
        };
        open_as(getenv("LTHIP_RCCL_PATH"), "LTHIP_RCCL_PATH");
        FindLoaded f;
        memset(&f, 0, sizeof f);
        dl_iterate_phdr(find_loaded_cb, &f);
        open_as(f.rccl, "already loaded in this process");
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
            open_as(name, "loader search path");
        if (f.torch_dir[0])
        {
            char beside[600];
            for (const char* leaf : {"librccl.so", "librccl.so.1"})
            {
                snprintf(beside, sizeof beside, "%s%s", f.torch_dir, leaf);
                open_as(beside, "beside the loaded libtorch");
            }
        }
        if (!x.so)
        {
            snprintf(x.path, sizeof x.path, "not found (tried $LTHIP_RCCL_PATH, a loaded librccl, librccl.so.1, librccl.so, /opt/rocm/lib, %s)",
                     f.torch_dir[0] ? f.torch_dir : "no libtorch loaded");
            return x;
        }
        {
            // the file behind the handle (the loader may have resolved a bare name)
            void* sym = dlsym(x.so, "ncclGetUniqueId");
            Dl_info di;
            if (sym && dladdr(sym, &di) && di.dli_fname && di.dli_fname[0])
                snprintf(x.path, sizeof x.path, "%s", di.dli_fname);
        }
        x.GetLastError = (decltype(x.GetLastError))dlsym(x.so, "ncclGetLastError");
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.so, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.so, "ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.so, "ncclCommDestroy");
        x.CommCount = (decltype(x.CommCount))dlsym(x.so, "ncclCommCount");
        x.AllGather = (decltype(x.AllGather))dlsym(x.so, "ncclAllGather");
        x.Send = (decltype(x.Send))dlsym(x.so, "ncclSend");
        x.Recv = (decltype(x.Recv))dlsym(x.so, "ncclRecv");
        x.GroupStart = (decltype(x.GroupStart))dlsym(x.so, "ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))dlsym(x.so, "ncclGroupEnd");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.so, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.CommCount && x.AllGather && x.Send && x.Recv && x.GroupStart &&
               x.GroupEnd && x.GetErrorString;
        return x;
    }();
    return r;
}

// ---- the shared-memory transport --------------------------------------------------------------------------------------------
const char SHM_MAGIC[8] = {'L', 'T', 'H', 'I', 'P', 'S', 'H', 'M'};

struct ShmHeader
{
    std::atomic<uint32_t> ready;      // set by rank 0 once the header is initialised
    uint32_t nranks;
    uint64_t slot_bytes;
    std::atomic<uint32_t> arrived;    // barrier: ranks that reached it
    std::atomic<uint32_t> generation; // barrier: incremented by the last one to arrive
    std::atomic<uint32_t> attached;   // ranks that mapped the segment (the last one to leave unlinks it)
    std::atomic<uint32_t> broken;     // sticky: the errno of the first rank that timed out or failed locally; every later collective fails
    uint32_t pad[8];
};
static_assert(sizeof(ShmHeader) == 64, "header is one cache line");
// per rank: a table of 2 * nranks u64 (what an all-to-all sender tells its receivers) followed by the data slot
inline size_t shm_table_bytes(uint32_t n) { return ((size_t)2 * n * 8 + 63) / 64 * 64; }

struct Shm
{
    ShmHeader* h = nullptr;
    uint8_t* base = nullptr;
    size_t map_bytes = 0;
    char path[96] = {0};
    uint64_t* table(uint32_t r) const { return (uint64_t*)(base + 64 + (size_t)r * (shm_table_bytes(h->nranks) + h->slot_bytes)); }
    uint8_t* slot(uint32_t r) const { return (uint8_t*)table(r) + shm_table_bytes(h->nranks); }
};

double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int shm_timeout_s()
{
    const char* e = getenv("LTHIP_COMM_TIMEOUT_S");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 300;
}

// A rank that cannot go on (a local copy failed, a peer never came) marks the communicator broken BEFORE it would have arrived at the
// next barrier: the peers waiting there leave with the error in the same round instead of waiting out the timeout, and nobody is ever
// released by fewer than nranks real arrivals (a timed-out rank's increment used to stay behind).  Sticky: the communicator is done.
void shm_break(Shm& s, int err)
{
    uint32_t none = 0;
    s.h->broken.compare_exchange_strong(none, (uint32_t)(err ? err : EIO), std::memory_order_acq_rel);
}
int shm_broken(const Shm& s) { return (int)s.h->broken.load(std::memory_order_acquire); }

// every rank of the communicator calls this the same number of times
int shm_barrier(Shm& s)
{
    if (shm_broken(s))
        return EPIPE;
    const uint32_t gen = s.h->generation.load(std::memory_order_acquire);
    if (s.h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == s.h->nranks)
    {
        s.h->arrived.store(0, std::memory_order_relaxed);
        s.h->generation.store(gen + 1, std::memory_order_release);
        return 0;
    }
    const double t0 = now_s();
    const int limit = shm_timeout_s();
    for (uint32_t spins = 0; s.h->generation.load(std::memory_order_acquire) == gen; ++spins)
    {
        if (shm_broken(s))
            return EPIPE; // a peer gave up (its errno is in the header)
        if (spins < 2000)
            sched_yield();
        else
        {
            usleep(50);
            if ((spins & 1023) == 0 && now_s() - t0 > limit)
            {
                shm_break(s, ETIMEDOUT); // a peer died or never came: nobody may be released by this rank's stale arrival later
                return ETIMEDOUT;
            }
        }
    }
    return 0;
}
} // namespace

struct lthip_comm
{
    int transport; // LTHIP_COMM_RCCL / LTHIP_COMM_SHM
    ncclComm_t comm;
    int nranks, rank;
    Shm shm;
};

namespace
{
// host <-> "wherever the caller's pointer lives": with a context the pointers are device pointers (copies on the context's stream,
// waited for -- the host transport is synchronous by nature), without one they are host pointers
int xfer(lthip_ctx* ctx, void* dst, const void* src, size_t n, bool to_host)
{
    if (!n)
        return 0;
    if (!ctx)
    {
        memcpy(dst, src, n);
        return 0;
    }
    LTHIP_CHECK(ctx, hipMemcpyAsync(dst, src, n, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, ctx->stream));
    return 0;
}
int xfer_wait(lthip_ctx* ctx)
{
    if (ctx)
        LTHIP_CHECK(ctx, lthip_stream_wait(ctx));
    return 0;
}

int shm_allgather(lthip_ctx* ctx, lthip_comm* c, const void* send, void* recv, size_t bytes)
{
    Shm& s = c->shm;
    const size_t S = s.h->slot_bytes;
    for (size_t off = 0; off < bytes; off += S)
    {
        const size_t n = bytes - off < S ? bytes - off : S;
        int e = xfer(ctx, s.slot(c->rank), (const uint8_t*)send + off, n, true);
        if (!e)
            e = xfer_wait(ctx);
        if (e)
            shm_break(s, e); // (published before the barrier: the peers leave in this round, not after the timeout)
        const int b = shm_barrier(s);
        if (e || b)
            return e ? e : b;
        for (int r = 0; r < c->nranks && !e; ++r)
            e = xfer(ctx, (uint8_t*)recv + (size_t)r * bytes + off, s.slot(r), n, false);
        if (!e)
            e = xfer_wait(ctx);
        if (e)
            shm_break(s, e);
        const int b2 = shm_barrier(s);
        if (e || b2)
            return e ? e : b2;
    }
    return 0;
}

int shm_alltoallv(lthip_ctx* ctx, lthip_comm* c, const void* send, const uint64_t* scnt, const uint64_t* sdis, void* recv,
                  const uint64_t* rcnt, const uint64_t* rdis, uint32_t eb)
{
    Shm& s = c->shm;
    const size_t S = s.h->slot_bytes;
    const int n = c->nranks;
    // what every sender holds for every receiver (bytes), and the extent of its send buffer
    uint64_t* t = s.table(c->rank);
    uint64_t extent = 0;
    for (int r = 0; r < n; ++r)
    {
        t[2 * r] = sdis[r] * eb;
        t[2 * r + 1] = scnt[r] * eb;
        if (scnt[r] && (sdis[r] + scnt[r]) * eb > extent)
            extent = (sdis[r] + scnt[r]) * eb;
    }
    int b = shm_barrier(s);
    if (b)
        return b;
    uint64_t rounds = 0;
    for (int r = 0; r < n; ++r)
    {
        const uint64_t* tr = s.table(r);
        uint64_t ext = 0;
        for (int q = 0; q < n; ++q)
            if (tr[2 * q + 1] && tr[2 * q] + tr[2 * q + 1] > ext)
                ext = tr[2 * q] + tr[2 * q + 1];
        if ((ext + S - 1) / S > rounds)
            rounds = (ext + S - 1) / S;
        if (tr[2 * c->rank + 1] != rcnt[r] * eb)
            rounds = UINT64_MAX; // the sender's count for me is not what I expect: every rank sees some such pair or none ...
    }
    // ... but not necessarily every rank, so the ranks agree on it before anyone leaves: an all-gather of one byte through the
    // slots (host pointers whatever the caller's are; the tables stay as they are until the next call)
    uint8_t bad = rounds == UINT64_MAX, all_bad[256];
    if (n > 256)
        return EINVAL;
    const int ea = shm_allgather(nullptr, c, &bad, all_bad, 1);
    if (ea)
        return ea;
    for (int r = 0; r < n; ++r)
        if (all_bad[r])
            return EINVAL;
    for (uint64_t p = 0; p < rounds; ++p)
    {
        const uint64_t lo = p * S, hi = lo + S;
        int e = 0;
        if (lo < extent)
        {
            e = xfer(ctx, s.slot(c->rank), (const uint8_t*)send + lo, (size_t)((extent < hi ? extent : hi) - lo), true);
            if (!e)
                e = xfer_wait(ctx);
        }
        if (e)
            shm_break(s, e);
        b = shm_barrier(s);
        if (e || b)
            return e ? e : b;
        for (int r = 0; r < n && !e; ++r)
        {
            const uint64_t* tr = s.table(r);
            const uint64_t a = tr[2 * c->rank], z = a + tr[2 * c->rank + 1]; // sender r's bytes for me: [a, z) of its send buffer
            const uint64_t x = a > lo ? a : lo, y = z < hi ? z : hi;
            if (x < y)
                e = xfer(ctx, (uint8_t*)recv + rdis[r] * eb + (x - a), s.slot(r) + (x - lo), (size_t)(y - x), false);
        }
        if (!e)
            e = xfer_wait(ctx);
        if (e)
            shm_break(s, e);
        b = shm_barrier(s);
        if (e || b)
            return e ? e : b;
    }
    return 0;
}
} // namespace

extern "C" int lthip_comm_unique_id(void* id128)
{
    if (!id128)
        return EINVAL;
    const char* tr = getenv("LTHIP_COMM_TRANSPORT");
    if (tr && !strcmp(tr, "shm"))
    {
        uint8_t* id = (uint8_t*)id128;
        memset(id, 0, LTHIP_COMM_ID_BYTES);
        memcpy(id, SHM_MAGIC, 8);
        uint8_t rnd[16];
        int fd = open("/dev/urandom", O_RDONLY);
        if (fd < 0 || read(fd, rnd, sizeof(rnd)) != (ssize_t)sizeof(rnd))
        {
            if (fd >= 0)
                close(fd);
            return EIO;
        }
        close(fd);
        for (int i = 0; i < 16; ++i)
            snprintf((char*)id + 8 + 2 * i, 3, "%02x", rnd[i]);
        return 0;
    }
    Rccl& r = rccl();
    if (!r.ok)
        return ENOSYS; // no RCCL on this machine
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess)
        return EIO;
    static_assert(sizeof(id) == LTHIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

static int shm_create(lthip_ctx* ctx, int nranks, int rank, const uint8_t* id, lthip_comm** out)
{
    char name[40] = {0};
    memcpy(name, id + 8, 32);
    for (int i = 0; i < 32; ++i)
        if (!((name[i] >= '0' && name[i] <= '9') || (name[i] >= 'a' && name[i] <= 'f')))
            return EINVAL;
    lthip_comm* k = new lthip_comm;
    k->transport = LTHIP_COMM_SHM;
    k->comm = nullptr;
    k->nranks = nranks;
    k->rank = rank;
    snprintf(k->shm.path, sizeof(k->shm.path), "/dev/shm/lthip_comm_%s", name);
    const char* se = getenv("LTHIP_COMM_SHM_SLOT");
    uint64_t slot = se ? strtoull(se, nullptr, 10) : 0;
    if (slot < 64)
        slot = 16u << 20;
    slot = (slot + 63) / 64 * 64;
    const double t0 = now_s();
    const int limit = shm_timeout_s();
    int fd = -1;
    if (rank == 0)
    {
        fd = open(k->shm.path, O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)(64 + (size_t)nranks * (shm_table_bytes(nranks) + slot))) != 0)
        {
            const int e = errno;
            if (fd >= 0)
                close(fd);
            delete k;
            return ctx ? lthip_fail(ctx, e ? e : EIO, "lthip_comm_create", "cannot create the shared-memory segment") : (e ? e : EIO);
        }
    }
    else
    {
        while ((fd = open(k->shm.path, O_RDWR)) < 0)
        {
            if (now_s() - t0 > limit)
            {
                delete k;
                return ETIMEDOUT;
            }
            usleep(1000);
        }
    }
    // (a peer maps what rank 0 has sized: wait for the size, then for the header)
    // (rank 0 made the name: whatever goes wrong on rank 0 from here on, the name goes away with it -- the peers that did attach wait
    // out their barrier and fail, and nothing stays behind in /dev/shm)
    auto fail = [&](int e, void* mapped) {
        if (mapped)
            munmap(mapped, k->shm.map_bytes);
        if (rank == 0)
            unlink(k->shm.path);
        delete k;
        return e;
    };
    struct stat st;
    memset(&st, 0, sizeof st);
    for (;;)
    {
        if (fstat(fd, &st) != 0)
        {
            const int e = errno ? errno : EIO;
            close(fd);
            return fail(e, nullptr);
        }
        if (st.st_size >= 64)
            break;
        if (now_s() - t0 > limit)
        {
            close(fd);
            return fail(ETIMEDOUT, nullptr);
        }
        usleep(1000);
    }
    k->shm.map_bytes = (size_t)st.st_size;
    void* p = mmap(nullptr, k->shm.map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED)
        return fail(ENOMEM, nullptr);
    k->shm.base = (uint8_t*)p;
    k->shm.h = (ShmHeader*)p;
    if (rank == 0)
    {
        k->shm.h->nranks = (uint32_t)nranks;
        k->shm.h->slot_bytes = slot;
        k->shm.h->arrived.store(0);
        k->shm.h->generation.store(0);
        k->shm.h->attached.store(0);
        k->shm.h->broken.store(0);
        k->shm.h->ready.store(1, std::memory_order_release);
    }
    else
        while (!k->shm.h->ready.load(std::memory_order_acquire))
        {
            if (now_s() - t0 > limit)
                return fail(ETIMEDOUT, p);
            usleep(1000);
        }
    if (k->shm.h->nranks != (uint32_t)nranks || k->shm.map_bytes < 64 + (size_t)nranks * (shm_table_bytes(nranks) + k->shm.h->slot_bytes))
    {
        shm_break(k->shm, EINVAL); // (the ranks that agree with each other leave their create barrier now, not after the timeout)
        return fail(EINVAL, p);    // the ranks disagree about the size of the communicator
    }
    k->shm.h->attached.fetch_add(1);
    const int b = shm_barrier(k->shm); // like ncclCommInitRank: returns once every rank is there
    if (b)
    {
        // this rank counted itself in and leaves without a communicator: count it out again, and the last one out takes the name
        // with it (a partial failure -- some ranks through the barrier, this one timed out -- would otherwise leave the survivors'
        // "last one to leave unlinks" at one forever and the segment in /dev/shm)
        if (k->shm.h->attached.fetch_sub(1) == 1 && rank != 0)
            unlink(k->shm.path);
        return fail(b, p);
    }
    *out = k;
    return 0;
}

extern "C" int lthip_comm_create(lthip_ctx* ctx, int nranks, int rank, const void* id128, lthip_comm** out)
{
    if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks)
        return EINVAL;
    if (!memcmp(id128, SHM_MAGIC, 8))
        return shm_create(ctx, nranks, rank, (const uint8_t*)id128, out);
    if (!ctx)
        return EINVAL; // RCCL moves device memory on a context's stream
    Rccl& r = rccl();
    if (!r.ok)
        return lthip_fail(ctx, ENOSYS, "lthip_comm_create: librccl", r.path);
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t e = r.CommInitRank(&c, nranks, id, rank);
    if (e != ncclSuccess)
    {
        char buf[256];
        return lthip_fail(ctx, EIO, "ncclCommInitRank", r.text(e, nullptr, buf, sizeof buf));
    }
    lthip_comm* k = new lthip_comm;
    k->transport = LTHIP_COMM_RCCL;
    k->comm = c;
    k->nranks = nranks;
    k->rank = rank;
    *out = k;
    return 0;
}

extern "C" int lthip_comm_destroy(lthip_comm* comm)
{
    if (!comm)
        return 0;
    if (comm->transport == LTHIP_COMM_SHM)
    {
        if (comm->shm.base)
        {
            const bool last = comm->shm.h->attached.fetch_sub(1) == 1;
            munmap(comm->shm.base, comm->shm.map_bytes);
            if (last)
                unlink(comm->shm.path);
        }
        delete comm;
        return 0;
    }
    Rccl& r = rccl();
    if (r.ok && comm->comm)
        (void)r.CommDestroy(comm->comm);
    delete comm;
    return 0;
}

// Where the RCCL this library binds to was found (and by which rule), or what was tried: first-contact diagnosis of an N > 1 launch.
extern "C" const char* lthip_comm_library(const char** out_how)
{
    Rccl& r = rccl();
    if (out_how)
        *out_how = r.how;
    return r.path;
}

extern "C" int lthip_comm_info(const lthip_comm* comm, int* out_nranks, int* out_rank, int* out_transport)
{
    if (!comm)
        return EINVAL;
    int n = comm->nranks;
    if (comm->transport == LTHIP_COMM_RCCL)
    {
        Rccl& r = rccl();
        if (!r.ok || r.CommCount(comm->comm, &n) != ncclSuccess) // what RCCL itself says, not what it was asked for
            return EIO;
    }
    else
        n = (int)comm->shm.h->nranks;
    if (out_nranks)
        *out_nranks = n;
    if (out_rank)
        *out_rank = comm->rank;
    if (out_transport)
        *out_transport = comm->transport;
    return 0;
}

extern "C" int lthip_comm_allgather(lthip_ctx* ctx, lthip_comm* comm, const void* d_send, void* d_recv, uint64_t count, uint32_t elem_bytes)
{
    if (!comm || !d_send || !d_recv || !elem_bytes || (!ctx && comm->transport != LTHIP_COMM_SHM))
        return EINVAL;
    if (count == 0)
        return 0;
    if (comm->transport == LTHIP_COMM_SHM)
    {
        const int e = shm_allgather(ctx, comm, d_send, d_recv, (size_t)count * elem_bytes);
        return e && ctx && !ctx->err[0] ? lthip_fail(ctx, e, "lthip_comm_allgather", "shared-memory transport") : e;
    }
    Rccl& r = rccl();
    if (!r.ok)
        return lthip_fail(ctx, ENOSYS, "lthip_comm_allgather", r.path);
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    const ncclResult_t e = r.AllGather(d_send, d_recv, (size_t)count * elem_bytes, ncclUint8, comm->comm, ctx->stream);
    if (e != ncclSuccess)
    {
        char buf[256];
        return lthip_fail(ctx, EIO, "ncclAllGather", r.text(e, comm->comm, buf, sizeof buf));
    }
    return 0;
}

extern "C" int lthip_comm_alltoallv(lthip_ctx* ctx, lthip_comm* comm, const void* d_send, const uint64_t* send_counts,
                                    const uint64_t* send_displs, void* d_recv, const uint64_t* recv_counts, const uint64_t* recv_displs,
                                    uint32_t elem_bytes)
{
    if (!comm || !send_counts || !send_displs || !recv_counts || !recv_displs || !elem_bytes || (!ctx && comm->transport != LTHIP_COMM_SHM))
        return EINVAL;
    uint64_t any_send = 0, any_recv = 0;
    for (int r = 0; r < comm->nranks; ++r)
    {
        any_send += send_counts[r];
        any_recv += recv_counts[r];
    }
    if ((any_send && !d_send) || (any_recv && !d_recv))
        return EINVAL;
    if (comm->transport == LTHIP_COMM_SHM)
    {
        const int e = shm_alltoallv(ctx, comm, d_send, send_counts, send_displs, d_recv, recv_counts, recv_displs, elem_bytes);
        return e && ctx && !ctx->err[0] ? lthip_fail(ctx, e, "lthip_comm_alltoallv", "shared-memory transport") : e;
    }
    Rccl& r = rccl();
    if (!r.ok)
        return lthip_fail(ctx, ENOSYS, "lthip_comm_alltoallv", r.path);
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    // one group: every pair's send and receive are posted together, RCCL runs them concurrently over the xGMI links (point to
    // point: the natural shape of an all-to-all on this fabric); a rank's own share is a send to itself
    ncclResult_t e = r.GroupStart();
    for (int p = 0; p < comm->nranks && e == ncclSuccess; ++p)
    {
        if (send_counts[p])
            e = r.Send((const uint8_t*)d_send + send_displs[p] * elem_bytes, (size_t)send_counts[p] * elem_bytes, ncclUint8, p, comm->comm, ctx->stream);
        if (recv_counts[p] && e == ncclSuccess)
            e = r.Recv((uint8_t*)d_recv + recv_displs[p] * elem_bytes, (size_t)recv_counts[p] * elem_bytes, ncclUint8, p, comm->comm, ctx->stream);
    }
    const ncclResult_t g = r.GroupEnd();
    if (e == ncclSuccess)
        e = g;
    if (e != ncclSuccess)
    {
        char buf[256];
        return lthip_fail(ctx, EIO, "ncclSend/ncclRecv", r.text(e, comm->comm, buf, sizeof buf));
    }
    return 0;
}
