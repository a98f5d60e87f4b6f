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
