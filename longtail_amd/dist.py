"""Multi-GPU exchange step of the ingest path.

The unit of independence is the reference's own job = one (asset, 64 MiB part) of ChunkAssets (src/longtail.c:2396-2458).
`JobPartition` lists the jobs of a GIVEN tree and assigns them to ranks (C host code: lthip_make_jobs /
lthip_partition_jobs -- contiguous byte-balanced ranges, LPT or job mod R; deterministic, so no communication); every rank
chunks + hashes its own jobs in ascending job order; `exchange_chunks` all-gathers the per-job chunk counts and the chunk
hash / length arrays and puts the runs back into JOB order (lthip_exchange_layout), which is the order the serial
first-seen pass (:2951-2970) and the VersionIndex layout depend on.  backend "nccl" is RCCL on ROCm; the same code runs on
CPU tensors with "gloo" (tests/test_dist_gloo.py).  `sharded_first_seen` adds the all-to-all of the first-seen table sharded by
hash.  Every collective runs either through torch.distributed or, with `comm=` (longtail_amd.lib.Comm), through the C ABI's
lthip_comm_allgather / lthip_comm_alltoallv (comm.hip) -- the torch-free launch of tools/run8.sh and bench.py's self-launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

POLICIES = {"range": 0, "lpt": 1, "mod": 2}
REORDER_PIECE = 1 << 15  # elements per workgroup of lthip_exchange_reorder (256 KiB of hashes)


class StepProfile:
    """Where an N-rank step's exchange time goes (tools/exchange_cost.py): wall time between marks, booked to "host" (numpy / C tables,
    Python), "device" (kernels and torch ops, waited for) or "transport" (collectives, waited for).  Marks synchronise the device, so a
    profiled step is slower than a plain one: it is a breakdown, not a measurement of the step."""

    def __init__(self, ctx=None):
        import time

        self.ctx, self.ms, self._now = ctx, {"host": 0.0, "device": 0.0, "transport": 0.0}, time.perf_counter
        self.detail = {}
        self.t = self._now()

    def start(self):
        self._wait()
        self.t = self._now()

    def _wait(self):
        if self.ctx is not None:
            self.ctx.sync()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    def mark(self, kind: str, what: str):
        self._wait()
        t = self._now()
        d = (t - self.t) * 1e3
        self.ms[kind] += d
        self.detail[what] = self.detail.get(what, 0.0) + d
        self.t = t


def _mark(prof, kind, what):
    if prof is not None:
        prof.mark(kind, what)


def shard_range(n_items: int, world: int, rank: int):
    """Contiguous, balanced [lo, hi) of equal items for `rank` (weak-scaling trees of equal files)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class JobPartition:
    """The (asset, part) jobs of a tree and their assignment to `world` ranks."""

    def __init__(self, asset_sizes, target_chunk_size: int, world: int, policy: str = "range", lib=None):
        from .lib import load

        self.lib = lib or load()
        d = self.lib.dll
        sizes = np.ascontiguousarray(asset_sizes, dtype=np.uint64)
        self.asset_sizes = sizes
        self.world = world
        self.policy = policy
        n = int(d.lthip_job_count(len(sizes), sizes.ctypes.data, target_chunk_size))
        self.job_asset = np.zeros(n, np.uint32)
        self.job_offset = np.zeros(n, np.uint64)
        self.job_size = np.zeros(n, np.uint64)
        err = d.lthip_make_jobs(len(sizes), sizes.ctypes.data, target_chunk_size, n, self.job_asset.ctypes.data,
                                self.job_offset.ctypes.data, self.job_size.ctypes.data)
        if err:
            raise RuntimeError(f"lthip_make_jobs: errno {err}")
        self.job_rank = np.zeros(n, np.uint32)
        self.rank_bytes = np.zeros(world, np.uint64)
        err = d.lthip_partition_jobs(n, self.job_size.ctypes.data, world, POLICIES[policy], self.job_rank.ctypes.data,
                                     self.rank_bytes.ctypes.data)
        if err:
            raise RuntimeError(f"lthip_partition_jobs: errno {err}")
        self.job_count = n
        self.jobs_per_rank = np.bincount(self.job_rank, minlength=world).astype(np.int64)
        self._jobs_of = {}

    def jobs_of(self, rank: int) -> np.ndarray:
        """Indices of the rank's jobs, ascending (= the order it must process them in)."""
        got = self._jobs_of.get(rank)
        if got is None:
            got = self._jobs_of[rank] = np.flatnonzero(self.job_rank == rank)
        return got

    def is_rank_major(self) -> bool:
        """True when the rank-major concatenation of the ranks' job lists is already job order."""
        return bool((np.diff(self.job_rank.astype(np.int64)) >= 0).all())

    def layout(self, gathered_counts: np.ndarray, count_stride: int, chunk_stride: int):
        n = self.job_count
        src, dst, cnt = np.zeros(n, np.uint64), np.zeros(n + 1, np.uint64), np.zeros(n, np.uint32)
        g = np.ascontiguousarray(gathered_counts, dtype=np.uint32)
        err = self.lib.dll.lthip_exchange_layout(n, self.job_rank.ctypes.data, self.world, g.ctypes.data, count_stride, chunk_stride,
                                                 src.ctypes.data, dst.ctypes.data, cnt.ctypes.data)
        if err:
            raise RuntimeError(f"lthip_exchange_layout: errno {err} (per-job counts do not match the assignment)")
        return src, dst, cnt

    def ranges(self, src, dst, cnt, max_piece: int = REORDER_PIECE):
        """The layout merged into maximal runs and cut into pieces of at most max_piece elements (lthip_exchange_ranges): what
        Context.exchange_reorder takes.  Range policy: world runs; O(jobs) in C."""
        f = self.lib.dll.lthip_exchange_ranges
        n = self.job_count
        k = int(f(n, src.ctypes.data, dst.ctypes.data, cnt.ctypes.data, max_piece, 0, None, None, None))
        r_src, r_dst, r_cnt = np.zeros(k, np.uint64), np.zeros(k, np.uint64), np.zeros(k, np.uint32)
        got = int(f(n, src.ctypes.data, dst.ctypes.data, cnt.ctypes.data, max_piece, k, r_src.ctypes.data, r_dst.ctypes.data, r_cnt.ctypes.data))
        assert got == k
        return r_src, r_dst, r_cnt


def _allgather(t: torch.Tensor, world: int, group=None, comm=None) -> torch.Tensor:
    if comm is not None:
        # the collective through the C ABI (lthip_comm_allgather = ncclAllGather on the context's stream); the host reads what it
        # gathered only after the context has been synchronised.  (A communicator without a context -- the shared-memory stand-in in
        # the CPU tests -- takes CPU tensors.)
        assert (t.device.type == "cuda") == (comm.ctx is not None), "tensors must live where the communicator's pointers do"
        out = comm.allgather(t)
        comm.sync()
        return out
    out = torch.empty(t.numel() * world, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def exchange_chunks(part: JobPartition, job_counts: torch.Tensor, hashes: torch.Tensor, lens: torch.Tensor | None, total: int,
                    ctx=None, group=None, comm=None, rank: int | None = None, prof: StepProfile | None = None):
    """job_counts: int32 tensor, chunk count of each of THIS rank's jobs (ascending job order); hashes (int64) / lens (int32):
    tensors whose first `total` entries are this rank's chunks in that order.  Returns a dict:
      hashes, lens   all ranks' chunks in JOB order (lens None when not given)
      job_first      int64 numpy [job_count + 1]: index of each job's first chunk (last entry = total chunks)
      mine           (job indices of this rank, ascending)
    On CUDA tensors with a context the reorder is one lthip_exchange_reorder per array over the merged runs of the layout (nothing
    per chunk happens on the host: the host sees the count matrix and O(jobs) tables only); CPU tensors (gloo tests) are indexed.
    `comm` (longtail_amd.lib.Comm): run the three all-gathers through the C ABI's RCCL entry instead of torch.distributed.
    """
    world = part.world
    if rank is None:  # (given by a launcher that does not use torch.distributed: tools/run8.sh, `comm` carries the collectives)
        rank = dist.get_rank(group) if (dist.is_initialized() and world > 1) else 0
    mine = part.jobs_of(rank)
    assert job_counts.numel() == len(mine), "one chunk count per own job"
    if world == 1:
        first = np.zeros(part.job_count + 1, np.int64)
        np.cumsum(job_counts.cpu().numpy().astype(np.int64), out=first[1:])
        return dict(hashes=hashes[:total], lens=None if lens is None else lens[:total], job_first=first, mine=mine)
    out_dev = hashes.device
    staged = comm is None and out_dev.type == "cuda" and dist.is_initialized() and dist.get_backend(group) != "nccl"  # functional path for CPU-only backends on a GPU box
    if staged:
        job_counts, hashes, lens = job_counts.cpu(), hashes[:total].cpu(), None if lens is None else lens[:total].cpu()
    dev = hashes.device
    # (1) per-job counts, padded to the largest job list
    count_stride = max(int(part.jobs_per_rank.max()), 1)
    send = torch.zeros(count_stride, dtype=torch.int32, device=dev)
    send[: len(mine)] = job_counts.to(torch.int32)
    _mark(prof, "device", "counts: pad")
    gathered_counts = _allgather(send, world, group, comm).cpu().numpy().view(np.uint32)
    _mark(prof, "transport", "counts: all-gather + D2H")
    per_rank_total = gathered_counts.reshape(world, count_stride).astype(np.int64).sum(axis=1)
    assert int(per_rank_total[rank]) == total
    # (2) chunk arrays, padded to the largest rank
    chunk_stride = max(int(per_rank_total.max()), 1)

    def padded(t):
        if t.numel() >= chunk_stride:
            return t[:chunk_stride]
        return torch.nn.functional.pad(t[:total], (0, chunk_stride - total))

    p_hash = padded(hashes)
    p_lens = padded(lens) if lens is not None else None
    _mark(prof, "device", "chunks: pad")
    g_hash = _allgather(p_hash, world, group, comm)
    g_lens = _allgather(p_lens, world, group, comm) if lens is not None else None
    _mark(prof, "transport", "chunks: all-gather")
    src, dst, cnt = part.layout(gathered_counts, count_stride, chunk_stride)
    n_all = int(dst[-1])
    ranges = None
    if out_dev.type == "cuda" and ctx is not None and not staged:
        ranges = part.ranges(src, dst, cnt)
        _mark(prof, "host", "layout + ranges (O(jobs), C)")
        o_hash = torch.empty(n_all, dtype=torch.int64, device=out_dev)
        ctx.exchange_reorder(g_hash, o_hash, ranges)
        o_lens = None
        if g_lens is not None:
            o_lens = torch.empty(n_all, dtype=torch.int32, device=out_dev)
            ctx.exchange_reorder(g_lens, o_lens, ranges)
        _mark(prof, "device", "chunks: reorder to job order")
    else:
        c64 = cnt.astype(np.int64)
        perm = np.repeat(src.astype(np.int64) - dst[:-1].astype(np.int64), c64) + np.arange(n_all, dtype=np.int64)
        idx = torch.from_numpy(perm).to(dev)
        o_hash = g_hash[idx].to(out_dev)
        o_lens = g_lens[idx].to(out_dev) if g_lens is not None else None
    return dict(hashes=o_hash, lens=o_lens, job_first=dst.astype(np.int64), mine=mine,
                layout=dict(src=src, dst=dst, cnt=cnt, chunk_stride=chunk_stride, staged=staged, ranges=ranges))


def _min_ordinal(h: torch.Tensor, o: torch.Tensor, ctx=None):
    """first[j] = smallest ordinal among the items with the hash of item j; number of distinct hashes.  Device tensors go through
    lthip_dedup_min_ordinal (open addressing, atomicMin); CPU tensors (the gloo tests) through numpy."""
    if h.device.type == "cuda" and ctx is not None:
        return ctx.dedup_min_ordinal(h, o)
    hn, on = h.cpu().numpy().view(np.uint64), o.cpu().numpy().astype(np.int64)
    if len(hn) == 0:
        return torch.zeros(0, dtype=torch.int32, device=h.device), 0
    uniq, inv = np.unique(hn, return_inverse=True)
    m = np.full(len(uniq), np.iinfo(np.int64).max, np.int64)
    np.minimum.at(m, inv, on)
    return torch.from_numpy(m[inv].astype(np.int32)).to(h.device), int(len(uniq))


def _a2a_counts(send_counts: torch.Tensor, world: int, group=None, comm=None) -> list:
    """recv_counts[p] = what rank p holds for me: row `me` of the all-gathered send-count matrix (comm) or an all-to-all of it."""
    if comm is not None:
        g = comm.allgather(send_counts.to(torch.int64))
        comm.sync()
        return [int(x) for x in g.view(world, world)[:, comm.rank].cpu().tolist()]
    recv_counts = torch.empty(world, dtype=torch.int64, device=send_counts.device)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    return [int(x) for x in recv_counts.cpu().tolist()]


def _a2a(send: torch.Tensor, sc: list, rc: list, group=None, comm=None) -> torch.Tensor:
    if comm is not None:
        # lthip_comm_alltoallv: grouped ncclSend / ncclRecv on the context's stream; the caller synchronises the communicator after its
        # group of exchanges, before torch ops (which may run on another stream) read what was received
        return comm.alltoallv(send, sc, rc)
    recv = torch.empty(sum(rc), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send, rc, sc, group=group)
    return recv


def sharded_first_seen(part: JobPartition, ex: dict, my_hashes: torch.Tensor, total: int, ctx=None, group=None, comm=None,
                       rank: int | None = None, prof: StepProfile | None = None):
    """The first-seen pass (src/longtail.c:2951-2970) with the table SHARDED by hash instead of replicated: every rank routes each of
    its chunks (hash, global position in job order) to the owner of the hash (all-to-all), the owner keeps the minimum position per
    hash (lthip_dedup_min_ordinal) and answers, and one all-gather spreads the answers: a rank inserts ~1/N of the tree's chunks instead
    of all N shares (round 2 rebuilt the whole table on every rank).  `ex` = the result of exchange_chunks (job layout).
    Returns (first: int32 tensor [all chunks] in job order -- the position of the first chunk with the same hash --, unique count).
    The result does not depend on the number of ranks: it is the minimum position per hash.

    Device tensors + a context: nothing per chunk happens on the host (round 5; round 4 built two permutations of all chunks in numpy
    and uploaded them from pageable memory every step).  The positions come from lthip_job_ordinals (O(own jobs) host tables), the
    answers go back into job order through the merged runs exchange_chunks already made (lthip_exchange_reorder); the host reads the
    N x N count matrix between the collectives and the distinct count at the end, nothing else."""
    world = part.world
    job_first, mine = ex["job_first"], ex["mine"]
    n_all = int(job_first[-1])
    if n_all >= 1 << 31:
        raise ValueError("more than 2^31 chunks")
    dev = my_hashes.device
    if world == 1:
        first, uniq = _min_ordinal(my_hashes[:total], torch.arange(total, dtype=torch.int32, device=dev), ctx)
        return first, uniq
    if rank is None:
        rank = comm.rank if comm is not None else dist.get_rank(group)
    lay = ex["layout"]
    staged = lay["staged"]
    if staged:
        my_hashes = my_hashes[:total].cpu()
    wdev = my_hashes.device
    on_device = wdev.type == "cuda" and ctx is not None and lay.get("ranges") is not None
    # global positions of my chunks: job j's run starts at job_first[j]
    cnt = (job_first[mine + 1] - job_first[mine]).astype(np.int64)
    assert int(cnt.sum()) == total
    local_first = np.concatenate([[0], np.cumsum(cnt)[:-1]]) if len(mine) else np.zeros(0, np.int64)
    if on_device:
        ordinals = ctx.job_ordinals(local_first.astype(np.uint32), job_first[mine].astype(np.uint32), total)
        _mark(prof, "host", "positions: job tables (O(own jobs))")
    else:
        starts = np.repeat(job_first[mine] - local_first, cnt)
        ordinals = torch.from_numpy((starts + np.arange(total, dtype=np.int64)).astype(np.int32)).to(wdev)
    h = my_hashes[:total]
    owner = ((h >> 40) & 0xFFFFFF) % world  # 24 bits from the middle of the digest
    order = torch.argsort(owner, stable=True)
    send_h, send_o = h[order].contiguous(), ordinals[order].contiguous()
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    _mark(prof, "device", "route: positions, owner, sort, counts")
    rc = _a2a_counts(send_counts, world, group, comm)
    sc = [int(x) for x in send_counts.cpu().tolist()]
    _mark(prof, "transport", "route: count matrix")
    recv_h = _a2a(send_h, sc, rc, group, comm)
    recv_o = _a2a(send_o, sc, rc, group, comm)
    if comm is not None:
        comm.sync()  # (the receives were queued on the context's stream: torch's ops below may run on another one)
    _mark(prof, "transport", "route: all-to-all (hash, position)")
    if staged and ctx is not None:
        d = torch.device("cuda", torch.cuda.current_device())
        f, uniq = _min_ordinal(recv_h.to(d), recv_o.to(d), ctx)
        f = f.cpu()
    elif on_device:
        f, uniq = ctx.dedup_min_ordinal(recv_h, recv_o, sync=False)  # (the count stays on the device until the end)
    else:
        f, uniq = _min_ordinal(recv_h, recv_o, ctx if wdev.type == "cuda" else None)
    _mark(prof, "device", "owner: minimum position per hash")
    back = _a2a(f.contiguous(), rc, sc, group, comm)
    if comm is not None:
        comm.sync()
    _mark(prof, "transport", "answer: all-to-all")
    my_first = torch.empty(total, dtype=torch.int32, device=wdev)
    my_first[order] = back
    u = uniq.to(torch.int64).reshape(1) if torch.is_tensor(uniq) else torch.tensor([uniq], dtype=torch.int64, device=wdev)
    # every rank's answers, in job order (the padded all-gather + reorder of exchange_chunks)
    chunk_stride = lay["chunk_stride"]
    send = my_first if total >= chunk_stride else torch.nn.functional.pad(my_first, (0, chunk_stride - total))
    send = send[:chunk_stride].contiguous()
    _mark(prof, "device", "answer: scatter + pad")
    if comm is not None:
        u = comm.allgather(u)
        g = comm.allgather(send)
        comm.sync()
        u = u.sum()
    else:
        dist.all_reduce(u, group=group)
        g = _allgather(send, world, group, None)
    _mark(prof, "transport", "answer: all-gather")
    if on_device:
        first = torch.empty(n_all, dtype=torch.int32, device=wdev)
        ctx.exchange_reorder(g, first, lay["ranges"])
        _mark(prof, "device", "answer: reorder to job order")
    else:
        c64 = lay["cnt"].astype(np.int64)
        perm = np.repeat(lay["src"].astype(np.int64) - lay["dst"][:-1].astype(np.int64), c64) + np.arange(n_all, dtype=np.int64)
        first = g[torch.from_numpy(perm).to(wdev)]
    uniq_all = int(u.item())
    _mark(prof, "transport", "distinct count: D2H")
    return first.to(dev), uniq_all


def allgather_hashes(local_hashes: torch.Tensor, total: int, group=None):
    """Weak-scaling form for trees where rank r owns a contiguous file range: the rank-major concatenation of the ranks' hash
    arrays IS tree order.  Returns (all_hashes, my_base, counts)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_hashes[:total], 0, [total]
    rank = dist.get_rank(group)
    out_dev = local_hashes.device
    if out_dev.type == "cuda" and dist.get_backend(group) != "nccl":
        local_hashes = local_hashes[:total].cpu()  # functional path for CPU-only backends (tests on a 1-GPU box)
    dev = local_hashes.device
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([total], dtype=torch.int64, device=dev), group=group)
    counts_h = [int(c) for c in counts.cpu().tolist()]
    pad = max(max(counts_h), 1)
    send = local_hashes[:pad]
    if send.numel() < pad:  # capacity smaller than the largest peer's count
        send = torch.nn.functional.pad(local_hashes[:total], (0, pad - total))
    recv = torch.empty(pad * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    pieces = [recv[r * pad : r * pad + counts_h[r]] for r in range(world)]
    return torch.cat(pieces).to(out_dev), sum(counts_h[:rank]), counts_h
