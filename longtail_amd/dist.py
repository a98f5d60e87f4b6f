"""Multi-GPU exchange step of the ingest path: ranks own disjoint file ranges, chunk + hash them
locally, then all-gather their chunk-hash arrays so that every rank can run the same first-seen dedup
(src/longtail.c:2951-2970) over the tree-ordered concatenation.  backend "nccl" is RCCL on ROCm; the same code runs
on CPU tensors with "gloo" (tests/test_dist_gloo.py).  This is the only collective on the path."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int):
    """Contiguous, balanced [lo, hi) of items (files / parts, already in the tree's strcmp order) for `rank`."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allgather_hashes(local_hashes: torch.Tensor, total: int, group=None):
    """local_hashes: int64 tensor whose first `total` entries are this rank's chunk hashes in (asset, part, chunk) order.
    Returns (all_hashes, my_base, counts): the rank-major concatenation (= tree order when ranks own contiguous file
    ranges), the index of this rank's first chunk in it, and the per-rank counts (host list)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_hashes[:total], 0, [total]
    rank = dist.get_rank(group)
    out_dev = local_hashes.device
    if out_dev.type == "cuda" and dist.get_backend(group) != "nccl":
        local_hashes = local_hashes[:total].cpu()  # functional fallback for CPU-only backends (tests on a 1-GPU box)
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
