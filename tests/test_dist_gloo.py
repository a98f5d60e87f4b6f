"""gloo tests (CPU, world sizes 2 and 4) of the N>1 path: the job partitioner for a GIVEN tree (the reference's (asset, part)
jobs, src/longtail.c:2396-2458, assigned by contiguous byte ranges / LPT / job mod R), the all-gather of per-job chunk lists, the
reorder into job order and the first-seen dedup over it.  The tree has an asset whose parts straddle ranks (intra-file segment
sharding, BASELINE.json configs[4]).  Chunk hashes come from the oracle; the GPU kernels are covered by the -m gpu tests
(tests/test_gpu_dist.py runs the same flow with the kernels and compares the VersionIndex with the reference's)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from longtail_amd.dist import JobPartition, allgather_hashes, exchange_chunks, shard_range, sharded_first_seen

TARGET = 256  # part = 256 KiB, chunks 48 / 128 / 512 bytes: multi-part assets stay small enough for the CPU
PARAMS = (48, 128, 512)


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


def tree_files():
    """Asset bytes in asset order: a 6-part asset, duplicates, an empty file, an exact multiple of the part size."""
    from tests._libs import oracle

    o = oracle()
    part = TARGET * 1024
    sizes = [70000, part * 5 + 12345, 0, 300, part * 2, 99999, 70000, part + 1, 5000, 180000, 64]
    seeds = [1, 2, 3, 4, 5, 6, 1, 8, 9, 2, 11]  # 0/6 identical files; 9 repeats the head of 1
    return [o.synth(n, o.asset_seed(5, s), i % 3 if i != 9 else 1 % 3) for i, (n, s) in enumerate(zip(sizes, seeds))]


def job_chunks(files, part: JobPartition, j: int):
    from tests._libs import oracle

    a, off, size = int(part.job_asset[j]), int(part.job_offset[j]), int(part.job_size[j])
    _, lens, hashes = oracle().chunk_and_hash(files[a][off : off + size], *PARAMS)
    return hashes.view(np.int64), lens.view(np.int32)


def serial_lists(files):
    part = JobPartition([len(f) for f in files], TARGET, 1)
    per_job = [job_chunks(files, part, j) for j in range(part.job_count)]
    return part, per_job


def _worker(rank, world, port, out, policy):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    files = tree_files()
    part = JobPartition([len(f) for f in files], TARGET, world, policy)
    mine = part.jobs_of(rank)
    lists = [job_chunks(files, part, int(j)) for j in mine]
    counts = torch.tensor([len(h) for h, _ in lists], dtype=torch.int32)
    total = int(counts.sum())
    cap = total + 5  # arenas larger than the count, like the real output arrays
    hashes, lens = torch.zeros(cap, dtype=torch.int64), torch.zeros(cap, dtype=torch.int32)
    if total:
        hashes[:total] = torch.from_numpy(np.concatenate([h for h, _ in lists]))
        lens[:total] = torch.from_numpy(np.concatenate([l for _, l in lists]))
    res = exchange_chunks(part, counts, hashes, lens, total)
    first, uniq = _first_seen(res["hashes"].numpy())
    # the hash-range-sharded table (round 3): all-to-all of (hash, position) to the hash's owner, minimum per hash, answers spread
    s_first, s_uniq = sharded_first_seen(part, res, hashes, total)
    torch.save({"hashes": res["hashes"], "lens": res["lens"], "job_first": res["job_first"], "mine": res["mine"], "first": first,
                "uniq": uniq, "job_rank": part.job_rank, "rank_bytes": part.rank_bytes, "sharded_first": s_first.numpy().astype(np.int64),
                "sharded_uniq": s_uniq}, f"{out}/r{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,policy", [(2, "range"), (2, "lpt"), (4, "range"), (4, "lpt"), (4, "mod")])
def test_exchange_reproduces_job_order_and_first_seen(tmp_path, world, policy):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), policy), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(world)]
    files = tree_files()
    part1, per_job = serial_lists(files)
    s_hash = np.concatenate([h for h, _ in per_job])
    s_lens = np.concatenate([l for _, l in per_job])
    s_first = np.concatenate([[0], np.cumsum([len(h) for h, _ in per_job])])
    exp_first, exp_uniq = _first_seen(s_hash)
    assert exp_uniq < len(s_hash)  # the duplicate files really dedup, across ranks
    job_rank = res[0]["job_rank"]
    # the 6-part asset (asset 1) is spread over more than one rank: intra-file segment sharding
    assert len(set(job_rank[part1.job_asset == 1].tolist())) > 1
    for r in range(world):
        # the inputs of Longtail_BuildVersionIndex (chunk hashes + sizes in (asset, part, chunk) order, :2499-2517) are identical
        # to the single-rank ones on every rank => so is the VersionIndex built from them
        assert (res[r]["hashes"].numpy() == s_hash).all() and (res[r]["lens"].numpy() == s_lens).all()
        assert (res[r]["job_first"] == s_first).all()
        assert (res[r]["first"] == exp_first).all() and res[r]["uniq"] == exp_uniq
        # ... and the sharded table gives every rank the same first-seen index as the serial pass (:2951-2970): the VersionIndex
        # built from it is the single-rank one
        assert (res[r]["sharded_first"] == exp_first).all() and res[r]["sharded_uniq"] == exp_uniq
        assert (res[r]["job_rank"] == job_rank).all()  # every rank computed the same assignment
    # every job has exactly one owner, every unique chunk exactly one first-seen owner
    owners = np.zeros(part1.job_count, np.int64)
    for r in range(world):
        owners[res[r]["mine"]] += 1
    assert (owners == 1).all()
    chunk_job = np.repeat(np.arange(part1.job_count), np.diff(s_first))
    owned = sum(int(((exp_first == np.arange(len(s_hash))) & (job_rank[chunk_job] == r)).sum()) for r in range(world))
    assert owned == exp_uniq


def test_job_list_is_the_references():
    """1 + size / (target * 1024) jobs per asset (src/longtail.c:2402), ranges as :2439-2440."""
    part = TARGET * 1024
    sizes = [0, 1, part - 1, part, part + 1, 5 * part, 5 * part + 7]
    p = JobPartition(sizes, TARGET, 1)
    assert p.job_count == sum(1 + s // part for s in sizes) == 1 + 1 + 1 + 2 + 2 + 6 + 6
    j = 0
    for a, s in enumerate(sizes):
        for k in range(1 + s // part):
            assert (int(p.job_asset[j]), int(p.job_offset[j]), int(p.job_size[j])) == (a, k * part, min(part, s - k * part))
            j += 1
    assert int(p.job_size[3 + 1]) == 0  # the exact multiple ends with an empty job


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_partition_policies_balance(world):
    rng = np.random.default_rng(world)
    part = 65536 * 1024
    trees = {
        "configs3": np.full(4096, 1 << 20, np.uint64),                                     # equal 1 MiB files
        "configs4": np.full(4, 16 << 30, np.uint64),                                       # 4 x 16 GiB: 257 jobs each
        "mixed": np.exp(rng.uniform(np.log(4096), np.log(4 << 30), 300)).astype(np.uint64),  # north-star tree
        "tiny": np.array([5, 0, 0, 7], np.uint64),
    }
    for name, sizes in trees.items():
        biggest = min(int(sizes.max()), part)
        for policy in ("range", "lpt", "mod"):
            p = JobPartition(sizes, 65536, world, policy)
            assert int(p.rank_bytes.sum()) == int(sizes.sum())
            assert (p.job_rank < world).all()
            q = JobPartition(sizes, 65536, world, policy)
            assert (p.job_rank == q.job_rank).all()  # deterministic
            if policy == "range":
                assert p.is_rank_major()
            if policy in ("range", "lpt") and name != "tiny":
                share = int(sizes.sum()) / world
                assert int(p.rank_bytes.max()) <= share + biggest + 4096 * 64, (name, policy, world)
        if name == "configs4" and world == 8:
            p = JobPartition(sizes, 65536, 8, "range")
            # each 16 GiB asset is split over two ranks: intra-file segment sharding
            assert [len(set(p.job_rank[p.job_asset == a].tolist())) for a in range(4)] == [2, 2, 2, 2]


def _worker_weak(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    files = tree_files()
    p1, per_job = serial_lists(files)
    per_file = [np.concatenate([per_job[j][0] for j in np.flatnonzero(p1.job_asset == a)]) for a in range(len(files))]
    lo, hi = shard_range(len(per_file), world, rank)
    mine = np.concatenate(per_file[lo:hi]) if hi > lo else np.zeros(0, np.int64)
    buf = torch.zeros(len(mine) + 5, dtype=torch.int64)
    buf[: len(mine)] = torch.from_numpy(mine)
    allh, base, counts = allgather_hashes(buf, len(mine))
    torch.save({"all": allh, "base": base, "counts": counts, "lo": lo}, f"{out}/w{rank}.pt")
    dist.destroy_process_group()


def test_weak_scaling_allgather_world2(tmp_path):
    """The weak-scaling form bench.py uses by default: contiguous file ranges, rank-major concatenation == tree order."""
    world = 2
    mp.spawn(_worker_weak, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    files = tree_files()
    p1, per_job = serial_lists(files)
    serial = np.concatenate([h for h, _ in per_job])
    for r in range(world):
        res = torch.load(tmp_path / f"w{r}.pt", weights_only=False)
        assert (res["all"].numpy() == serial).all()
        lo_job = int(np.flatnonzero(p1.job_asset >= res["lo"])[0])
        assert res["base"] == sum(len(per_job[j][0]) for j in range(lo_job))


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 65536):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_dry_run_reports_the_partition_without_a_gpu():
    """bench.py --gpus 8 --dry-run: partition balance and exchange volume for BASELINE.json configs[3] / configs[4], host only."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    for extra, jobs in (([], 65536), (["--file-mib", "16384", "--codec", "zstd"], 1028)):
        out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--dry-run", "--scaling", "strong"] + extra,
                             capture_output=True, text=True, timeout=300, cwd=root)
        assert out.returncode == 0, out.stderr[-800:]
        j = json.loads(out.stdout.strip().splitlines()[-1])
        assert j["dry_run"] and j["jobs"] == jobs and sum(j["bytes_per_rank"]) == 64 << 30
        assert j["imbalance_max_over_mean"] < 1.01 and len(j["jobs_per_rank"]) == 8
        assert j["first_seen_table_inserts_per_rank"]["sharded"] * 8 <= j["first_seen_table_inserts_per_rank"]["replicated"] + 8


@pytest.mark.parametrize("world,policy", [(1, "range"), (3, "range"), (4, "lpt"), (8, "mod")])
def test_exchange_ranges_is_the_layout_merged_and_cut(world, policy):
    """lthip_exchange_ranges (what the device reorder takes): applying its pieces is the permutation the per-job layout describes; the
    range policy collapses to one run per rank; no piece is longer than asked."""
    rng = np.random.default_rng(7)
    sizes = rng.integers(0, 300 << 20, size=40).astype(np.uint64)
    sizes[5] = 0
    part = JobPartition(sizes, 65536, world, policy)
    counts = rng.integers(0, 50, size=part.job_count).astype(np.uint32)
    counts[rng.integers(0, part.job_count, size=5)] = 0
    count_stride = int(part.jobs_per_rank.max())
    g = np.zeros((world, count_stride), np.uint32)
    totals = np.zeros(world, np.int64)
    for r in range(world):
        mine = part.jobs_of(r)
        g[r, : len(mine)] = counts[mine]
        totals[r] = counts[mine].sum()
    chunk_stride = max(int(totals.max()), 1)
    src, dst, cnt = part.layout(g.reshape(-1), count_stride, chunk_stride)
    n_all = int(dst[-1])
    gathered = rng.integers(0, 1 << 62, size=world * chunk_stride).astype(np.int64)
    want = np.concatenate([gathered[int(s) : int(s) + int(c)] for s, c in zip(src, cnt)]) if n_all else np.zeros(0, np.int64)
    for piece in (7, 1 << 15):
        r_src, r_dst, r_cnt = part.ranges(src, dst, cnt, max_piece=piece)
        assert (r_cnt <= piece).all() and (r_cnt > 0).all()
        got = np.full(n_all, -1, np.int64)
        for s, d, c in zip(r_src, r_dst, r_cnt):
            got[int(d) : int(d) + int(c)] = gathered[int(s) : int(s) + int(c)]
        assert np.array_equal(got, want)
        if policy == "range" and piece == 1 << 15:
            assert len(r_src) <= world  # (rank-major order is job order: one run per rank that has chunks)
