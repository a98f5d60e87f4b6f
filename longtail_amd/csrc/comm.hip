// comm.hip -- the collective of the multi-GPU exchange behind the C ABI: RCCL's all-gather on the context's
// stream.  RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU process never loads it, and the library has no
// link-time dependency on it.  One process per GPU; the 128-byte unique id travels between the processes by whatever means the
// embedder has (bench.py: the torch.distributed store it is launched with).
#include "lthip_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

struct lthip_comm
{
    ncclComm_t comm;
    int nranks, rank;
};

namespace
{
struct Rccl
{
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((x.so = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!x.so)
            return x;
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.so, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.so, "ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.so, "ncclCommDestroy");
        x.AllGather = (decltype(x.AllGather))dlsym(x.so, "ncclAllGather");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.so, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.GetErrorString;
        return x;
    }();
    return r;
}
} // namespace

extern "C" int lthip_comm_unique_id(void* id128)
{
    if (!id128)
        return EINVAL;
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

extern "C" int lthip_comm_create(lthip_ctx* ctx, int nranks, int rank, const void* id128, lthip_comm** out)
{
    if (!ctx || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks)
        return EINVAL;
    Rccl& r = rccl();
    if (!r.ok)
        return lthip_fail(ctx, ENOSYS, "lthip_comm_create", "librccl.so.1 not found");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t e = r.CommInitRank(&c, nranks, id, rank);
    if (e != ncclSuccess)
        return lthip_fail(ctx, EIO, "ncclCommInitRank", r.GetErrorString(e));
    lthip_comm* k = new lthip_comm;
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
    Rccl& r = rccl();
    if (r.ok && comm->comm)
        (void)r.CommDestroy(comm->comm);
    delete comm;
    return 0;
}

extern "C" int lthip_comm_allgather(lthip_ctx* ctx, lthip_comm* comm, const void* d_send, void* d_recv, uint64_t count, uint32_t elem_bytes)
{
    if (!ctx || !comm || !d_send || !d_recv || !elem_bytes)
        return EINVAL;
    if (count == 0)
        return 0;
    Rccl& r = rccl();
    if (!r.ok)
        return lthip_fail(ctx, ENOSYS, "lthip_comm_allgather", "librccl.so.1 not found");
    LTHIP_CHECK(ctx, hipSetDevice(ctx->device));
    const ncclResult_t e = r.AllGather(d_send, d_recv, (size_t)count * elem_bytes, ncclUint8, comm->comm, ctx->stream);
    if (e != ncclSuccess)
        return lthip_fail(ctx, EIO, "ncclAllGather", r.GetErrorString(e));
    return 0;
}
