"""-m gpu: the RCCL collective behind the C ABI (comm.hip).  One GPU here, so the communicator has ONE rank: the
entry points, the run-time binding of librccl and the stream semantics are what this checks; the multi-rank data path is the same
three padded all-gathers tests/test_dist_gloo.py checks with gloo (world 2 and 4)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_allgather(gpu):
    from longtail_amd.lib import Comm, LongtailHipError

    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # the bootstrap needs an interface; a GPU box may have no other
    try:
        uid = Comm.unique_id()
    except LongtailHipError as e:  # ENOSYS: no librccl on this machine
        pytest.skip(str(e))
    assert len(uid) == 128 and any(uid)
    try:
        comm = Comm(gpu, 1, 0, uid)
    except LongtailHipError as e:
        pytest.skip(f"RCCL could not bootstrap here: {e}")
    try:
        for n, dt in ((1, torch.int32), (1000, torch.int64), (123457, torch.uint8)):
            send = torch.arange(n, device="cuda").to(dt)
            recv = comm.allgather(send)
            gpu.sync()
            assert recv.numel() == n and torch.equal(recv, send)
        # exchange_chunks with a communicator and ONE rank is the identity in job order
        from longtail_amd.dist import JobPartition, exchange_chunks

        part = JobPartition(np.array([300 << 10, 0, 70 << 10], np.uint64), 65536, 1)
        counts = torch.tensor([3, 1, 0, 2, 1, 0][: part.job_count], dtype=torch.int32, device="cuda")
        total = int(counts.sum())
        hashes = torch.arange(100, 100 + total, dtype=torch.int64, device="cuda")
        ex = exchange_chunks(part, counts, hashes, None, total, gpu, comm=comm)
        assert torch.equal(ex["hashes"], hashes) and int(ex["job_first"][-1]) == total
    finally:
        comm.close()


def test_exchange_reorder_and_job_ordinals_against_numpy(gpu):
    """The device side of the N-rank exchange (round 5: nothing per chunk on the host): lthip_exchange_reorder applies the merged runs
    of the layout, lthip_job_ordinals gives every local chunk its position in job order -- both equal to the numpy permutations the
    CPU path of longtail_amd.dist builds."""
    from longtail_amd.dist import JobPartition

    rng = np.random.default_rng(11)
    sizes = rng.integers(0, 200 << 20, size=300).astype(np.uint64)
    for world, policy in ((4, "lpt"), (8, "range"), (3, "mod")):
        part = JobPartition(sizes, 65536, world, policy)
        counts = rng.integers(0, 40, size=part.job_count).astype(np.uint32)
        count_stride = int(part.jobs_per_rank.max())
        g = np.zeros((world, count_stride), np.uint32)
        totals = np.zeros(world, np.int64)
        for r in range(world):
            mine = part.jobs_of(r)
            g[r, : len(mine)] = counts[mine]
            totals[r] = counts[mine].sum()
        chunk_stride = int(totals.max())
        src, dst, cnt = part.layout(g.reshape(-1), count_stride, chunk_stride)
        n_all = int(dst[-1])
        perm = np.repeat(src.astype(np.int64) - dst[:-1].astype(np.int64), cnt.astype(np.int64)) + np.arange(n_all, dtype=np.int64)
        for piece in (5, 1 << 15):
            ranges = part.ranges(src, dst, cnt, max_piece=piece)
            for dt in (torch.int64, torch.int32, torch.uint8):
                gathered = torch.randint(0, 127, (world * chunk_stride,), device="cuda").to(dt)
                out = torch.zeros(n_all, dtype=dt, device="cuda")
                gpu.exchange_reorder(gathered, out, ranges)
                gpu.sync()
                assert torch.equal(out.cpu(), gathered.cpu()[torch.from_numpy(perm)]), (world, policy, piece, dt)
        job_first = dst.astype(np.int64)
        for r in range(world):
            mine = part.jobs_of(r)
            c = counts[mine].astype(np.int64)
            total = int(c.sum())
            local_first = np.concatenate([[0], np.cumsum(c)[:-1]]) if len(mine) else np.zeros(0, np.int64)
            got = gpu.job_ordinals(local_first.astype(np.uint32), job_first[mine].astype(np.uint32), total)
            gpu.sync()
            want = np.repeat(job_first[mine] - local_first, c) + np.arange(total, dtype=np.int64)
            assert np.array_equal(got.cpu().numpy().astype(np.int64), want), (world, policy, r)


def test_abi_version_and_result_struct_size(gpu):
    """include/longtail_hip.h LTHIP_ABI_VERSION: the library says which binary interface it implements, and lthip_ingest_finish never
    writes past the struct size its caller states (the struct grew without a version)."""
    import ctypes as C
    import re
    from pathlib import Path

    from longtail_amd.lib import ABI_VERSION, load

    text = (Path(__file__).resolve().parents[1] / "include" / "longtail_hip.h").read_text()
    assert int(re.search(r"#define LTHIP_ABI_VERSION (\d+)", text).group(1)) == ABI_VERSION == load().dll.lthip_abi_version()
