"""world_size-2 gloo test (CPU) of the N>1 path: file sharding, the chunk-hash all-gather and the first-seen dedup over
the gathered array.  Chunk hashes come from the oracle; the GPU kernels are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from longtail_amd.dist import allgather_hashes, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _first_seen(h):
    seen, first = {}, np.zeros(len(h), np.int64)
    for i, x in enumerate(h.tolist()):
        first[i] = seen.setdefault(x, i)
    return first, len(seen)


def _tree_hashes():
    from tests._libs import oracle

    o = oracle()
    files = [o.synth(200000 + 1000 * i, o.asset_seed(5, i % 7), i % 3) for i in range(11)]  # i%7: duplicate files across ranks
    per_file = [o.chunk_and_hash(f, 8192, 32768, 131072)[2].view(np.int64) for f in files]
    return per_file


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per_file = _tree_hashes()
    lo, hi = shard_range(len(per_file), world, rank)
    mine = np.concatenate(per_file[lo:hi]) if hi > lo else np.zeros(0, np.int64)
    cap = len(mine) + 5  # arena larger than the count, like the real output arrays
    buf = torch.zeros(cap, dtype=torch.int64)
    buf[: len(mine)] = torch.from_numpy(mine)
    allh, base, counts = allgather_hashes(buf, len(mine))
    first, uniq = _first_seen(allh.numpy())
    torch.save({"all": allh, "base": base, "counts": counts, "first": first, "uniq": uniq, "lo": lo, "hi": hi}, f"{out}/r{rank}.pt")
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 65536):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_allgather_and_dedup_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(world)]
    per_file = _tree_hashes()
    serial = np.concatenate(per_file)
    exp_first, exp_uniq = _first_seen(serial)
    for r in range(world):
        assert (res[r]["all"].numpy() == serial).all()  # rank-major concat == tree order
        assert (res[r]["first"] == exp_first).all() and res[r]["uniq"] == exp_uniq
        assert res[r]["base"] == sum(len(x) for x in per_file[: res[r]["lo"]])
    assert exp_uniq < len(serial)  # the duplicate files really dedup across ranks
    # every chunk is owned (compressed) by exactly one rank: the one that saw it first
    owned = [np.nonzero(res[r]["first"][res[r]["base"] : res[r]["base"] + res[r]["counts"][r]] ==
                        np.arange(res[r]["base"], res[r]["base"] + res[r]["counts"][r]))[0] for r in range(world)]
    assert sum(len(o) for o in owned) == exp_uniq
