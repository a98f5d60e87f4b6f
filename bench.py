#!/usr/bin/env python3
"""bench.py -- ingest throughput of the HIP hot path on MI355X: SURVEY.md §8(d)'s metric.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = the three reference calls the metric is defined over, on a synthetic tree whose bytes are already resident in HBM:

  Longtail_CreateVersionIndex   (src/longtail.c:2808)  plan -> buzhash candidate scan -> cut selection -> compaction -> BLAKE3
                                                       leaves / parents over the rank's own (asset, 64 MiB part) jobs;
                                                       [N > 1: RCCL all-gather of per-job chunk counts, hashes, lengths];
                                                       first-seen dedup, content + path hashes, SERIALIZED VersionIndex on the host
  Longtail_CreateMissingContent (src/longtail.c:6882)  the chunks this rank writes, greedy block packing (:6801-6860), block hashes
  Longtail_WriteContent         (src/longtail.c:4760)  device block assembly where needed, per-block LZ4 / ZStd into stored-block
                                                       images (BlockIndex + [raw][compressed] + payload) in a bounded arena (null
                                                       sink), SERIALIZED StoreIndex on the host
  (lthip_chunk_hash + lthip_ingest_index / _write / _finish, include/longtail_hip.h)

Workload (default = BASELINE.json configs[2]): a tree of 1 MiB random files, 64 GiB PER GPU (--scaling weak: the tree has
N x 64 GiB) or in total (--scaling strong, configs[3]); the tree's jobs are assigned to ranks by lthip_partition_jobs
(byte-balanced contiguous ranges by default: an asset's parts may straddle ranks, configs[4]); every rank synthesizes only its own
parts.  value = bytes of the whole tree / max-over-ranks wall time of K steps (barrier + synchronize on both sides).
The default run also measures, after the headline, the compressible variant, the north-star mixed-size tree, the zstd codec and
BASELINE.json configs[4]'s shape (4 x 16 GiB files, ZStd) on compressible bytes ("secondary"; at N=1 also "restore": the device
decoders' GB/s on 512 stored blocks of the compressible workload, round trip verified, and "host_fed": the session fed from pinned
host memory), and times the reference's bikeshed-threaded CPU path on a bounded sample of EVERY one of those trees ("cpu_baseline",
with the reference codec's ratio and the container's CPU quota beside it; class CpuReference) -- plus, in a process of its own
(tools/drop_in_child.py), the unmodified reference core with this library's plugin objects on the same sample ("drop_in").
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
KINDS = {"random": 0, "mixed": 1, "zero": 2, "records": 11, "tokens": 12, "lines": 13}
# wave-instructions per 4 KiB of input (one wave: 64 lanes x 64 bytes), from the ISA dumps / SQ_INSTS_VALU, priced at
# one VALU instruction per SIMD per 4 cycles at the nominal 2.4 GHz on 256 CUs x 4 SIMDs: the integer-issue roof of K1 / K3
VALU_INSTR_PER_4KIB = {"buzhash": 537, "blake3_leaf": 695}  # K1: SQ_INSTS_VALU / wave-tiles, profiles/r03a_pmc_k1.txt (round 2: 735)
VALU_ISSUE_PER_S = 256 * 4 * 2.4e9 / 4.0


def synth_mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def asset_seeds(tree_seed: int, first: int, count: int) -> np.ndarray:
    """lt_synth_asset_seed (include/longtail_synth.h) vectorised."""
    with np.errstate(over="ignore"):
        idx = np.arange(first, first + count, dtype=np.uint64) + np.uint64(1)
        return synth_mix(np.uint64(tree_seed) + np.uint64(0x9E3779B97F4A7C15) * idx)


def make_tree(kind_of_tree: str, total_bytes: int, file_bytes: int, dups: bool = False):
    """The tree as a struct Longtail_FileInfos taken apart (src/longtail.h:1684-1692): sizes, path data, offsets, permissions.
    dups: a quarter of the files repeat earlier ones (SURVEY.md §8d "dedup (duplicate files)") -- file i with i % 8 == 5 is file i - 3
    again, byte for byte; file i with i % 8 == 7 is file i - 5 SHIFTED: the same byte stream entered 20 KiB + 16 (i % 64) bytes later,
    so its first chunks are new, the cuts then fall back into step (content-defined chunking) and the rest of its chunks are known
    ones -- blocks whose unique chunks are not one byte range of the data (device block assembly, src/longtail.c:4640-4721)."""
    if kind_of_tree == "files":
        n = max(1, total_bytes // file_bytes)
        sizes = np.full(n, file_bytes, dtype=np.uint64)
    else:
        # north-star tree: sizes log-uniform in [4 KiB, 4 GiB] (capped at a quarter of the tree), fixed seed
        rng = np.random.default_rng(0xA55E7)
        budget, hi, out = total_bytes, min(4 << 30, max(total_bytes // 4, 8192)), []
        while budget > 0:
            sz = max(1, min(int(np.exp(rng.uniform(np.log(4096), np.log(hi)))), budget))
            out.append(sz)
            budget -= sz
        sizes = np.asarray(out, dtype=np.uint64)
        n = len(out)
    names = [f"dir{i % 256:03d}/file{i:06d}.bin" for i in range(n)]
    lens = np.fromiter((len(s) + 1 for s in names), dtype=np.int64, count=n)
    path_offsets = np.zeros(n, np.uint32)
    np.cumsum(lens[:-1], out=path_offsets[1:])
    path_data = ("\0".join(names) + "\0").encode()
    seed_of, shift = np.arange(n, dtype=np.int64), np.zeros(n, np.uint64)
    if dups:
        idx = np.arange(n, dtype=np.int64)
        whole = (idx % 8 == 5) & (idx >= 3)
        moved = (idx % 8 == 7) & (idx >= 5)
        seed_of[whole] = idx[whole] - 3
        seed_of[moved] = idx[moved] - 5
        shift[moved] = (20480 + 16 * (idx[moved] % 64)).astype(np.uint64)
        if kind_of_tree != "files":  # a copy is as long as its original is: the generator's stream has no end, the tree's sizes stay
            pass
    return dict(sizes=sizes, path_data=path_data, path_offsets=path_offsets, perms=np.full(n, 0o644, np.uint16), nfiles=n,
                seed_of=seed_of, shift=shift, dups=bool(dups))


def plain_comm(lib, ctx, rank, world, idfile=None, timeout_s=120):
    """The torch-free handshake (tools/run8.sh, --launch plain): rank 0 makes the communicator id (lthip_comm_unique_id) and publishes
    it by an atomic rename, the others wait for the file, every rank creates its communicator with it (lthip_comm_create returns
    when all ranks are there).  `ctx` None: the shared-memory transport with host pointers (tests without a GPU)."""
    from longtail_amd.lib import Comm

    idfile = idfile or os.environ.get("LTHIP_COMM_ID_FILE", "/tmp/lthip_comm_id")
    if rank == 0:
        with open(idfile + ".tmp", "wb") as f:
            f.write(Comm.unique_id(lib))
        os.replace(idfile + ".tmp", idfile)
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > timeout_s:
            raise SystemExit(f"rank {rank}: no communicator id in {idfile} after {timeout_s} s")
        time.sleep(0.01)
    return Comm(ctx, world, rank, open(idfile, "rb").read(), lib=lib)


def comm_library(lib):
    """(path, rule) of the RCCL the C ABI binds to (comm.hip: $LTHIP_RCCL_PATH, a copy already mapped -- torch's --, the loader's path,
    beside libtorch), or what was tried."""
    d = lib.dll
    d.lthip_comm_library.restype = ctypes.c_char_p
    d.lthip_comm_library.argtypes = [ctypes.POINTER(ctypes.c_char_p)]
    how = ctypes.c_char_p()
    path = d.lthip_comm_library(ctypes.byref(how))
    return (path or b"").decode(), (how.value or b"").decode()


def _report_path(rank):
    key = os.path.basename(os.environ.get("LTHIP_COMM_ID_FILE", "")) or ("port" + os.environ.get("MASTER_PORT", "0"))
    return f"/tmp/lthip_preflight.{key}.rank{rank}.json"


def preflight_report(rank, world, stage, error=None, **extra):
    """One rank's account of the first-contact steps of an N > 1 launch: the stage it reached (or failed in), its device, the transport's
    own error text.  Written to a per-rank file the launcher (or rank 0) merges; a FAILING rank 0 also prints it as the run's one JSON
    line, so that a failed multi-GPU record is a diagnosis and not the tail of a traceback."""
    rep = {"rank": rank, "stage": stage, "ok": error is None, "error": None if error is None else str(error)[:800],
           "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()}
    rep.update(extra)
    try:
        with open(_report_path(rank) + ".tmp", "w") as f:
            json.dump(rep, f)
        os.replace(_report_path(rank) + ".tmp", _report_path(rank))
    except OSError:
        pass
    return rep


def preflight_failure_line(world, mine, wait_s=3.0):
    """The ONE line of a failed first contact: the failing stage, the transport's text, and what every rank that got far enough to
    write its report says about itself (device ids included)."""
    t0 = time.time()
    ranks = {}
    while time.time() - t0 < wait_s and len(ranks) < world:
        for r in range(world):
            if r not in ranks and os.path.exists(_report_path(r)):
                try:
                    ranks[r] = json.load(open(_report_path(r)))
                except (OSError, ValueError):
                    pass
        time.sleep(0.05)
    failed = [v for _, v in sorted(ranks.items()) if not v.get("ok")] or ([mine] if mine else [])
    first = failed[0] if failed else {}
    return {"handshake": False, "n_gpus_asked": world, "failed_stage": first.get("stage"), "error": first.get("error"),
            "failed_ranks": [v["rank"] for v in failed], "ranks": [ranks.get(r, {"rank": r, "stage": "no report (the process did not get that far)"}) for r in range(world)],
            "rccl": first.get("rccl") or (mine or {}).get("rccl")}


def handshake_only(args):
    """--handshake-only: everything a plain launch does BEFORE the first kernel -- the id file, lthip_comm_create, a barrier, the
    reductions bench.py makes (max / sum through an all-gather) and one all-to-all -- and a JSON line from rank 0.  With
    LTHIP_COMM_TRANSPORT=shm it needs no GPU (tests/test_comm_shm.py); on an N-GPU node it is the 10-second check that RCCL sees N
    ranks before a lease is spent on the measurement.  A failure is reported with the STAGE it happened in (device / id hand-over /
    lthip_comm_create = ncclCommInitRank / first all-gather / all-to-all / verdict), the transport's own text and every rank's device:
    one JSON line, exit code 1."""
    import torch

    from longtail_amd.lib import Context, load

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    lib = load()
    shm = os.environ.get("LTHIP_COMM_TRANSPORT") == "shm"
    extra = {"rccl": None if shm else dict(zip(("path", "found_by"), comm_library(lib))), "transport_asked": "host-shm" if shm else "rccl"}
    stage = "device"
    try:
        if os.path.exists(_report_path(rank)):
            os.unlink(_report_path(rank))
        ctx = None
        if torch.cuda.is_available() and lib.device_count() > 0:
            local = int(os.environ.get("LOCAL_RANK", "0"))
            dev = local % torch.cuda.device_count() if shm else local
            if dev >= torch.cuda.device_count():
                raise RuntimeError(f"LOCAL_RANK {local} but this node shows {torch.cuda.device_count()} GPU(s)")
            torch.cuda.set_device(dev)
            ctx = Context(dev)
            props = torch.cuda.get_device_properties(dev)
            extra.update(device_index=dev, device_name=props.name, devices_visible=torch.cuda.device_count(),
                         pci_bus_id=getattr(props, "pci_bus_id", None), visible_devices_env=os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"))
        elif not shm:
            raise RuntimeError("no GPU: the handshake needs RCCL's devices (LTHIP_COMM_TRANSPORT=shm is the stand-in)")
        device = "cuda" if ctx is not None else "cpu"
        if os.environ.get("LTHIP_HANDSHAKE_FAIL_AT") == f"{rank}:device":  # (tests: a rank that dies before it reaches the others)
            raise RuntimeError("injected failure (LTHIP_HANDSHAKE_FAIL_AT)")
        stage = "id hand-over + lthip_comm_create" + ("" if shm else " (ncclCommInitRank)")
        comm = plain_comm(lib, ctx, rank, world, timeout_s=int(os.environ.get("LTHIP_COMM_TIMEOUT_S", "120")))
        info = comm.info()
        stage = "first all-gather"
        vals = torch.tensor([float(rank + 1), 10.0 * (rank + 1)], dtype=torch.float64, device=device)
        g = comm.allgather(vals).view(world, -1)
        comm.sync()
        ok = bool(g[:, 0].max().item() == world and g[:, 1].sum().item() == 10.0 * world * (world + 1) / 2)
        if not ok:
            raise RuntimeError(f"the all-gather delivered {g.tolist()}")
        stage = "all-to-all"
        # all-to-all: rank r sends (r + 1) * (p + 1) copies of the value 1000 * r + p to rank p
        sc = [(rank + 1) * (p + 1) for p in range(world)]
        rc = [(p + 1) * (rank + 1) for p in range(world)]
        send = torch.cat([torch.full((sc[p],), 1000 * rank + p, dtype=torch.int64, device=device) for p in range(world)])
        recv = comm.alltoallv(send, sc, rc)
        comm.sync()
        want = torch.cat([torch.full((rc[p],), 1000 * p + rank, dtype=torch.int64, device=device) for p in range(world)])
        ok = bool(torch.equal(recv, want))
        stage = "verdict all-gather"
        oks = comm.allgather(torch.tensor([1 if ok else 0], dtype=torch.int32, device=device))
        comm.sync()
        all_ok = bool(oks.min().item() == 1)
        if not all_ok:
            raise RuntimeError(f"all-to-all results wrong on ranks {[r for r, v in enumerate(oks.tolist()) if v != 1]}")
        comm.close()
    except BaseException as e:  # (SystemExit of the id hand-over's timeout included)
        mine = preflight_report(rank, world, stage, error=f"{type(e).__name__}: {e}", **extra)
        if rank == 0 and os.environ.get("LTHIP_PREFLIGHT_MERGED_BY_LAUNCHER") != "1":
            print(json.dumps(preflight_failure_line(world, mine)), flush=True)
        else:
            print(json.dumps(mine), file=sys.stderr, flush=True)
        raise SystemExit(1)
    preflight_report(rank, world, "done", **extra)
    if rank == 0:
        print(json.dumps({"handshake": True, "n_gpus": world, "comm": info, "device": device, "rccl": extra["rccl"]}), flush=True)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here.  --launch torch (default) gives them the
    environment torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; collectives = RCCL through
    torch.distributed), --launch plain the one of tools/run8.sh (no torch.distributed; every collective through the C ABI's
    lthip_comm_*).  Rank 0's stdout is this process's; any rank failing stops the others and fails the run -- a line is printed by
    N ranks or not at all."""
    import socket
    import subprocess
    import tempfile

    n = args.gpus
    env = dict(os.environ, WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    idfile = None
    if args.launch == "plain":
        idfile = tempfile.mktemp(prefix="lthip_comm_id.", dir="/tmp")
        env.update(LONGTAIL_LAUNCH="plain", LTHIP_COMM_ID_FILE=idfile)
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n))
    def run_ranks(argv, env, quiet=False):
        procs = []
        for r in range(n):
            procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), *argv], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                          stdout=(subprocess.PIPE if quiet else None) if r == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.DEVNULL if quiet else None, text=True))
        rc, pending = 0, set(range(n))
        try:
            while pending:
                for r in sorted(pending):
                    code = procs[r].poll()
                    if code is None:
                        continue
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        if not quiet:
                            print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                        for q in pending:
                            procs[q].terminate()  # exactly the processes started above
                time.sleep(0.05)
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        return rc, (procs[0].stdout.read() if quiet else None)

    # ---- first contact, by itself: the handshake (id hand-over, lthip_comm_create = ncclCommInitRank, an all-gather, an all-to-all)
    # BEFORE any measurement.  A failure is ONE JSON line naming the stage, the transport's text and every rank's device; rc != 0.
    hs_id = tempfile.mktemp(prefix="lthip_comm_id.", dir="/tmp")
    hs_env = dict(env, LONGTAIL_LAUNCH="plain", LTHIP_COMM_ID_FILE=hs_id, LTHIP_PREFLIGHT_MERGED_BY_LAUNCHER="1")
    # (LONGTAIL_DIST_BACKEND=gloo is the functional stand-in on a box with fewer GPUs: its collectives are not RCCL's, nothing to shake hands with)
    do_handshake = args.handshake_only or args.launch == "plain" or os.environ.get("LONGTAIL_DIST_BACKEND", "nccl") == "nccl"
    try:
        rc, out0 = run_ranks(["--gpus", str(n), "--handshake-only"], hs_env, quiet=True) if do_handshake else (0, "")
        if rc != 0:
            os.environ["LTHIP_COMM_ID_FILE"] = hs_id  # (the key of the ranks' report files)
            print(json.dumps(preflight_failure_line(n, None)), flush=True)
            return rc
        if args.handshake_only:
            sys.stdout.write(out0)
            sys.stdout.flush()
            return 0
    finally:
        for f in [hs_id] + [f"/tmp/lthip_preflight.{os.path.basename(hs_id)}.rank{r}.json" for r in range(n)]:
            if os.path.exists(f):
                os.unlink(f)
    try:
        rc, _ = run_ranks(sys.argv[1:], env)
    finally:
        if idfile and os.path.exists(idfile):
            os.unlink(idfile)
    return rc


class Bench:
    """Device buffers shared by the configurations measured in one process."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        from longtail_amd.lib import Context, load

        self.torch, self.dist, self.args = torch, dist, args
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            # (main() starts the ranks itself when WORLD_SIZE is unset; a line whose n_gpus is not --gpus is never printed)
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        self.lib = load()
        if not torch.cuda.is_available() or self.lib.device_count() == 0:
            raise SystemExit("bench.py needs a GPU and liblongtail_hip.so: there is no CPU fallback")
        # LONGTAIL_DIST_BACKEND=gloo runs the multi-rank flow with all ranks on the GPUs that exist (rank % device_count) and the
        # exchange staged through host memory: a functional check of the N>1 path on a 1-GPU box, not a measurement.  The same for
        # the torch-free launch: LTHIP_COMM_TRANSPORT=shm (the C ABI's shared-memory stand-in for RCCL, comm.hip)
        self.backend = os.environ.get("LONGTAIL_DIST_BACKEND", "nccl")
        self.plain = self.world > 1 and os.environ.get("LONGTAIL_LAUNCH") == "plain"
        stand_in = self.backend != "nccl" or (self.plain and os.environ.get("LTHIP_COMM_TRANSPORT") == "shm")
        if self.world > 1 and not stand_in and torch.cuda.device_count() < self.world:
            raise SystemExit(f"--gpus {self.world} but this node has {torch.cuda.device_count()} GPU(s): one process per GPU "
                             "(functional stand-ins on fewer GPUs: LONGTAIL_DIST_BACKEND=gloo, or LONGTAIL_LAUNCH=plain LTHIP_COMM_TRANSPORT=shm)")
        dev_index = local_rank % torch.cuda.device_count() if stand_in else local_rank
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        # LONGTAIL_LAUNCH=plain (tools/run8.sh, or bench.py's own self-launch with --launch plain): N processes, no torch.distributed
        # at all -- every collective is the C ABI's (lthip_comm_allgather / lthip_comm_alltoallv = RCCL, comm.hip), the communicator
        # id travels through a file
        if self.world > 1 and not self.plain:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # first contact under a launcher (the driver's torch.distributed.run): staged, so that a failure is one JSON line naming
            # the stage (rendezvous + ncclCommInitRank / first all-reduce), the error text and this rank's device -- not a traceback
            import datetime

            stage = f"init_process_group({self.backend})"
            extra = {"device_index": dev_index, "devices_visible": torch.cuda.device_count(), "device_name": torch.cuda.get_device_name(dev_index),
                     "backend": self.backend, "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                     "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
            try:
                if self.backend == "nccl":
                    dist.init_process_group("nccl", device_id=self.dev, timeout=datetime.timedelta(seconds=int(os.environ.get("LTHIP_COMM_TIMEOUT_S", "300"))))
                else:
                    dist.init_process_group(self.backend, timeout=datetime.timedelta(seconds=int(os.environ.get("LTHIP_COMM_TIMEOUT_S", "300"))))
                stage = "first all-reduce"
                t = torch.ones(1, dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
                dist.all_reduce(t)
                if self.backend == "nccl":
                    torch.cuda.synchronize(self.dev)
                if int(t.item()) != self.world:
                    raise RuntimeError(f"the all-reduce of ones over {self.world} ranks gave {t.item()}")
                preflight_report(self.rank, self.world, "done", **extra)
            except BaseException as e:
                mine = preflight_report(self.rank, self.world, stage, error=f"{type(e).__name__}: {e}", **extra)
                if self.rank == 0:
                    print(json.dumps(preflight_failure_line(self.world, mine)), flush=True)
                else:
                    print(json.dumps(mine), file=sys.stderr, flush=True)
                raise SystemExit(1)
        self.ctx = Context(dev_index)
        self.bufs = {}
        # --collective c: the exchange's collectives through the C ABI instead of torch.distributed; the unique id travels through the
        # process group the ranks were launched with
        self.comm = None
        if self.plain:
            self.comm = plain_comm(self.lib, self.ctx, self.rank, self.world)
        elif self.world > 1 and args.collective == "c" and self.backend == "nccl":
            from longtail_amd.lib import Comm

            box = [Comm.unique_id(self.lib) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            self.comm = Comm(self.ctx, self.world, self.rank, box[0])
        self.comm_info = None
        if self.comm is not None:
            self.comm_info = self.comm.info()  # what the TRANSPORT says (ncclCommCount), not what it was asked for
            if self.comm_info["nranks"] != self.world:
                raise SystemExit(f"rank {self.rank}: the communicator has {self.comm_info['nranks']} ranks, expected {self.world}")
        elif self.world > 1:
            self.comm_info = {"nranks": dist.get_world_size(), "rank": dist.get_rank(), "transport": f"torch.distributed/{dist.get_backend()}"}
        if self.comm_info is not None and any(t in str(self.comm_info.get("transport")) for t in ("rccl", "nccl")):
            # which RCCL: the file the C ABI binds (comm.hip: $LTHIP_RCCL_PATH, a copy already mapped -- torch's --, the loader's path)
            path, how = comm_library(self.lib)
            self.comm_info["rccl_library"] = {"path": path, "found_by": how}

    def measure_peak(self, gib=4):
        """What this box's memory system delivers to plain streaming kernels (tools/hbm_peak.py, SURVEY.md §8d "use the measured peak
        as denominator too"): device copy (read + write) and read-only reduction, GB/s, best of 3 on `gib` GiB."""
        torch = self.torch
        n = int(gib) << 30
        a = self.buf("data", n + 256)[:n]
        b = self.buf("arena", n)[:n]

        def best(f):
            f()
            torch.cuda.synchronize(self.dev)
            t = None
            for _ in range(3):
                t0 = time.perf_counter()
                f()
                torch.cuda.synchronize(self.dev)
                dt = time.perf_counter() - t0
                t = dt if t is None or dt < t else t
            return t

        c = best(lambda: b.copy_(a))
        r = best(lambda: a.view(torch.int64).sum())
        self.peak_measured = {"copy_GBps": round(2 * n / c / 1e9, 1), "read_GBps": round(n / r / 1e9, 1),
                              "how": f"torch device copy / int64 sum over {gib} GiB, best of 3 (tools/hbm_peak.py)"}
        return self.peak_measured

    def reduce(self, values, op="sum"):
        """All-reduce of a few host numbers over the ranks (max of the wall time, sums of the result counters)."""
        torch = self.torch
        if self.world == 1:
            return list(values)
        if self.plain:
            t = torch.tensor(list(values), dtype=torch.float64, device=self.dev)
            g = self.comm.allgather(t).view(self.world, -1)
            self.ctx.sync()
            return (g.max(dim=0).values if op == "max" else g.sum(dim=0)).cpu().tolist()
        on = self.dev if self.backend == "nccl" else "cpu"
        t = torch.tensor(list(values), dtype=torch.float64, device=on)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return t.cpu().tolist()

    def buf(self, name, nbytes, dtype=None, pinned=False):
        """A named buffer of at least nbytes, kept across configurations."""
        torch = self.torch
        dtype = dtype or torch.uint8
        n = (nbytes + dtype.itemsize - 1) // dtype.itemsize if hasattr(dtype, "itemsize") else nbytes
        t = self.bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            self.bufs[name] = None
            t = torch.empty(max(n, 16), dtype=dtype, pin_memory=True) if pinned else torch.empty(max(n, 16), dtype=dtype, device=self.dev)
            self.bufs[name] = t
        return t

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            if self.plain:
                self.reduce([0.0])  # an all-gather every rank has to join
            else:
                self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def restore_rates(self, gib=4.0, block_bytes=8 << 20):
        """Restore side: `gib` of the compressible workload as stored blocks of `block_bytes`, compressed on the device,
        then DEcoded on the device -- GB/s of output with every payload resident in HBM and in flight at once, round trip verified."""
        torch, ctx = self.torch, self.ctx
        FILE = 1 << 20
        nfiles = int(gib * (1 << 30)) // FILE
        n = nfiles * FILE
        data = self.buf("data", n + 256)
        ctx.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE), np.full(nfiles, FILE, np.uint64), asset_seeds(1, 0, nfiles), KINDS["mixed"])
        nb = n // block_bytes
        b_off = np.arange(nb, dtype=np.int64) * block_bytes
        b_size = np.full(nb, block_bytes, np.int64)
        out = {"workload": f"{nb} stored blocks of {block_bytes >> 20} MiB of the compressible workload, payloads in HBM", "unit": "GB/s of output",
               "note": "payloads of this library's encoders; stores written by the reference's: profiles/r03d_decode_rate.txt"}
        for name, comp, dec, bound in (("lz4", ctx.lz4_compress_blocks, ctx.lz4_decompress_blocks, b_size + b_size // 255 + 16),
                                       ("zstd", ctx.zstd_compress_blocks, ctx.zstd_decompress_blocks, b_size + (b_size >> 8) + 64)):
            d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
            arena = self.buf("restore_arena", int(bound.sum()) + nb * 64 + 64)
            back = self.buf("restore_back", n + 64)
            sz = comp(data, b_off, b_size, arena, d_offs, bound).cpu().numpy().view(np.uint32).astype(np.int64)
            ctx.sync()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                got = dec(arena, d_offs, sz, back, b_off, b_size)
                ctx.sync()
                t = time.perf_counter() - t0
                best = t if best is None or t < best else best
            ok = bool((got.cpu().numpy().view(np.uint32) == b_size).all()) and bool(torch.equal(back[:n], data[:n]))
            one = None  # latency of ONE block (what a restore that asks block by block sees)
            for _ in range(3):
                t0 = time.perf_counter()
                dec(arena, d_offs[:1], sz[:1], back, b_off[:1], b_size[:1])
                ctx.sync()
                t = time.perf_counter() - t0
                one = t if one is None or t < one else one
            out[name] = {"value": round(n / best / 1e9, 1), "ms": round(best * 1e3, 2), "one_block_ms": round(one * 1e3, 2),
                         "ratio": round(n / float(sz.sum()), 3), "round_trip": ok}
        # ONE stored block written by the REFERENCE's encoders (what a restore that asks block by block sees on an existing store,
        # lib/compressblockstore/longtail_compressblockstore.c:271-338), on the GPU and on one host core with the reference decoder
        try:
            from tests._libs import have_ref, ref

            if have_ref():
                r = ref()
                raw = data[:block_bytes].cpu().numpy()
                back = self.buf("restore_back", n + 64)
                single = {"what": f"one {block_bytes >> 20} MiB block of the compressible workload compressed by the reference encoder: decode latency, best of 5"}
                for name, codec, settings, dec in (("lz4", 0, r.lz4_type, ctx.lz4_decompress_blocks), ("zstd", 1, r.zstd_default, ctx.zstd_decompress_blocks)):
                    comp = r.compress(codec, settings, raw)
                    dev = torch.from_numpy(comp).to(self.dev)
                    cpu_best = gpu_best = None
                    for _ in range(5):
                        t0 = time.perf_counter()
                        err, dec_out = r.decompress(codec, comp, block_bytes)
                        t = time.perf_counter() - t0
                        cpu_best = t if cpu_best is None or t < cpu_best else cpu_best
                    for _ in range(5):
                        t0 = time.perf_counter()
                        got = dec(dev, np.zeros(1, np.int64), np.array([len(comp)], np.int64), back, np.zeros(1, np.int64), np.array([block_bytes], np.int64))
                        ctx.sync()
                        t = time.perf_counter() - t0
                        gpu_best = t if gpu_best is None or t < gpu_best else gpu_best
                    ok1 = err == 0 and int(got.cpu().numpy().view(np.uint32)[0]) == block_bytes and bool((back[:block_bytes].cpu().numpy() == raw).all())
                    single[name] = {"gpu_ms": round(gpu_best * 1e3, 2), "one_host_core_ms": round(cpu_best * 1e3, 2), "ratio": round(block_bytes / len(comp), 3), "round_trip": ok1}
                single["note"] = ("a reference zstd frame is one chain of dependent sequences per 128 KiB block: the GPU's floor for it is a host core's time "
                                  "(DESIGN.md §9 lead 6, INTEGRATION.md: batch the blocks of a restore -- 512 blocks in one call decode at the rates above)")
                out["one_reference_made_block"] = single
        except Exception as e:
            out["one_reference_made_block"] = {"error": repr(e)}
        return out

    def host_fed_rates(self, kinds=("random", "mixed"), slices=2, slice_gib=8.0, rounds=3):
        """secondary.host_fed -- the ingest path fed from HOST memory (what the reference's path starts from: StorageAPI.Read,
        src/longtail.c:1923-1960; block assembly :4640-4721), as the double-buffered loop INTEGRATION.md recommends to an embedder:
        pinned slice k+1 -> HBM on a copy stream | lthip_chunk_hash + lthip_ingest_index / _write / _finish of slice k on the compute
        stream | stored-block images of slice k-1 (lthip_ingest_images) -> pinned host memory on a second copy stream.
        Each direction can be moved by the copy engines (hipMemcpyAsync; the images are packed on the device first) or by the compute
        units (lthip_link_copy in, lthip_gather_ranges straight into pinned host memory out).  Which pair moves both directions at
        once fastest differs from box to box and from process to process (tools/pcie_duplex_probe.py: two hipMemcpyAsync streams
        were served one after the other in one lease -- a slice took the SUM of its two copies -- and side by side in the next; kernel
        copies reached 46 GB/s per direction in the first and 34 in the second), so the loop times the four pairs on its own streams
        first and takes the fastest.  Every slice is a session of its own (its VersionIndex / StoreIndex land in pinned memory as in
        the headline); `slices` distinct slices of the headline tree live in pinned host memory and are streamed `rounds` times.
        PCIe is the ceiling: the value is reported as a fraction of what the link did for the same bytes in the same directions."""
        torch, ctx, args = self.torch, self.ctx, self.args
        from longtail_amd.lib import Context, Ingest, chunker_params

        FILE = 1 << 20
        nfiles = int(slice_gib * (1 << 30)) // FILE
        n = nfiles * FILE
        mn, av, mx = chunker_params(args.target_chunk_size)
        tree = make_tree("files", n, FILE)
        p_off = np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE)
        p_size = np.full(nfiles, FILE, np.uint64)
        limit = args.block_size + args.block_size // 10
        arena_bytes = n + n // 128 + (n // args.block_size + 4) * (16384 + 64) + 2 * (limit + limit // 128 + 16384)
        t0 = time.perf_counter()
        host_in = [self.buf(f"hf_in{i}", n, pinned=True) for i in range(slices)]
        host_out = [self.buf(f"hf_out{i}", arena_bytes, pinned=True) for i in range(2)]
        pin_s = time.perf_counter() - t0
        data = [self.buf("data", n + 256), self.buf("hf_data1", n + 256)]
        arena = [self.buf("arena", arena_bytes), self.buf("hf_arena1", arena_bytes)]
        plan = ctx.make_plan(p_off, p_size, mn, av, mx)
        cap = max(1, plan.capacity)
        out_offs, out_lens, out_hash = self.buf("offs", cap * 8, torch.int64), self.buf("lens", cap * 4, torch.int32), self.buf("hash", cap * 8, torch.int64)
        out_first = self.buf("first", (nfiles + 1) * 4, torch.int32)
        h_first = self.buf("first_host", (nfiles + 1) * 4, torch.int32, pinned=True)
        vi_cap = int(self.lib.dll.lthip_version_index_size(nfiles, cap, cap, len(tree["path_data"]))) + 64
        h_vi, h_si = self.buf("vi", vi_cap, pinned=True), self.buf("si", 16 + 32 * cap + 64, pinned=True)
        job_asset = np.arange(nfiles, dtype=np.uint32)
        h2d, d2h = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        cur = torch.cuda.current_stream(self.dev)
        ctx_in = Context(self.dev.index, stream=h2d.cuda_stream, lib=self.lib)
        ctx_out = Context(self.dev.index, stream=d2h.cuda_stream, lib=self.lib)

        def move_in(mode, dst, src, nbytes):  # on the h2d stream
            if mode == "kernel":
                ctx_in.link_copy(dst, src, nbytes)
            else:
                with torch.cuda.stream(h2d):
                    dst[:nbytes].copy_(src[:nbytes], non_blocking=True)

        def move_out(mode, dst, src, nbytes):  # on the d2h stream
            if mode == "kernel":
                ctx_out.link_copy(dst, src, nbytes)
            else:
                with torch.cuda.stream(d2h):
                    dst[:nbytes].copy_(src[:nbytes], non_blocking=True)

        def link_time(*legs):
            best = None
            for _ in range(3):
                torch.cuda.synchronize(self.dev)
                t0 = time.perf_counter()
                for f in legs:
                    f()
                torch.cuda.synchronize(self.dev)
                t = time.perf_counter() - t0
                best = t if best is None or t < best else best
            return best

        out = {"workload": f"{slices} x {slice_gib:g} GiB slices of the headline tree in pinned host memory, streamed {rounds} x: slice in | chunk+hash+index+write | "
                           "stored-block images out to pinned host memory, double buffered on three streams; a session (VersionIndex + StoreIndex + images) per slice",
               "unit": "GB/s of input", "pinned_alloc_s": round(pin_s, 2)}
        # ---- the link, measured on the loop's own streams: each direction alone, then the four pairs ----
        link = {"bytes_per_direction": n}
        for m in ("engine", "kernel"):
            link[f"in_{m}_GBps"] = round(n / link_time(lambda: move_in(m, data[0], host_in[0], n)) / 1e9, 2)
            link[f"out_{m}_GBps"] = round(n / link_time(lambda: move_out(m, host_out[0], data[1], n)) / 1e9, 2)
        pairs = {}
        for mi in ("engine", "kernel"):
            for mo in ("engine", "kernel"):
                pairs[(mi, mo)] = link_time(lambda: move_in(mi, data[0], host_in[0], n), lambda: move_out(mo, host_out[0], data[1], n))
                link[f"both_in_{mi}_out_{mo}_GBps_per_direction"] = round(n / pairs[(mi, mo)] / 1e9, 2)
        mode_in, mode_out = min(pairs, key=pairs.get)
        link["chosen"] = {"in": mode_in, "out": mode_out}
        out["link"] = link
        out["h2d_GBps"] = max(link["in_engine_GBps"], link["in_kernel_GBps"])
        out["d2h_GBps"] = max(link["out_engine_GBps"], link["out_kernel_GBps"])
        duplex = n / pairs[(mode_in, mode_out)] / 1e9
        packed = [self.buf("hf_packed0", arena_bytes), self.buf("hf_packed1", arena_bytes)] if mode_out == "engine" else None
        for kind in kinds:
            for i in range(slices):
                ctx.synth_fill(data[0], p_off, p_size, asset_seeds(0x10C0FFEE, i * nfiles, nfiles), KINDS[kind])
                ctx.sync()
                host_in[i][:n].copy_(data[0][:n])
            torch.cuda.synchronize(self.dev)
            ing = Ingest(ctx, args.target_chunk_size, args.block_size, args.max_chunks_per_block, "lz4", batch_bytes=n)
            ev_in = [torch.cuda.Event() for _ in range(2)]
            ev_done = [torch.cuda.Event() for _ in range(2)]
            ev_out = [torch.cuda.Event() for _ in range(2)]
            total_slices = slices * rounds
            image_bytes = 0
            res = None
            keep_alive = [None, None]

            def upload(k):
                b = k % 2
                h2d.wait_event(ev_done[b])  # (the session of slice k - 2 has read data[b] to the end)
                move_in(mode_in, data[b], host_in[k % slices], n)
                ev_in[b].record(h2d)

            def one_pass(count):
                nonlocal image_bytes, res
                image_bytes = 0
                upload(0)
                for k in range(count):
                    b = k % 2
                    if k + 1 < count:
                        upload(k + 1)
                    cur.wait_event(ev_in[b])
                    cur.wait_event(ev_out[b])  # (the images of slice k - 2 have left arena[b] / packed[b] for the host)
                    plan.reaim(p_off, p_size)
                    ctx.chunk_hash(plan, data[b], outputs=(out_offs, out_lens, out_hash, out_first), sync=False)
                    ctx._check(self.lib.dll.lthip_copy_d2h(ctx.h, h_first.data_ptr(), out_first.data_ptr(), (nfiles + 1) * 4), "lthip_copy_d2h")
                    ctx.sync()
                    first_host = h_first.numpy()[: nfiles + 1].view(np.uint32)
                    total = int(first_host[nfiles])
                    tr, keep = Ingest.tree(tree["sizes"], tree["path_offsets"], tree["perms"], tree["path_data"], job_asset, first_host.astype(np.uint64), None)
                    ing.index(tr, out_hash, out_lens, total, out_offs, out_first, total, h_vi)
                    ing.write(data[b], arena[b])
                    res = ing.finish(h_si)
                    _, offs, sizes = ing.images()
                    assert len(offs) == res.blocks, "a slice must fit one codec batch"
                    dst = np.zeros(len(offs) + 1, np.int64)
                    np.cumsum((sizes.astype(np.int64) + 7) // 8 * 8, out=dst[1:])
                    tot = int(dst[-1])
                    t_offs, t_sizes = torch.from_numpy(offs.view(np.int64)).to(self.dev), torch.from_numpy(sizes.view(np.int32)).to(self.dev)
                    t_dst = torch.from_numpy(dst[:-1].copy()).to(self.dev)
                    if mode_out == "engine":  # pack on the compute stream, then one copy
                        ctx.gather_ranges(arena[b], t_offs, t_sizes, packed[b], t_dst)
                    ev_done[b].record(cur)
                    d2h.wait_event(ev_done[b])
                    if mode_out == "engine":
                        move_out("engine", host_out[b], packed[b], tot)
                    else:
                        ctx_out.gather_ranges(arena[b], t_offs, t_sizes, host_out[b], t_dst)  # device ranges -> pinned host memory
                    ev_out[b].record(d2h)
                    keep_alive[b] = (t_offs, t_sizes, t_dst)  # (read by the gather on the d2h stream)
                    image_bytes += tot
                torch.cuda.synchronize(self.dev)

            one_pass(2)  # warm-up: allocations of the session, the plan, the streams
            t0 = time.perf_counter()
            one_pass(total_slices)
            dt = time.perf_counter() - t0
            ing.close()
            gbps = total_slices * n / dt / 1e9
            back = image_bytes / (total_slices * n)
            # the link's ceiling for this slice: n bytes in beside back * n bytes out -- the both-directions rate while both run
            # (the shorter transfer's length), the one-direction rate for the rest
            one_way = out["h2d_GBps"]
            t_link = min(back, 1.0) * n / duplex / 1e9 + abs(1.0 - back) * n / one_way / 1e9
            out[kind] = {"value": round(gbps, 2), "ms_per_slice": round(dt / total_slices * 1e3, 2), "ratio": round(res.raw_bytes / max(1, res.compressed_bytes), 3),
                         "image_bytes_per_input_byte": round(back, 4), "frac_of_h2d": round(gbps / one_way, 3),
                         "link_ms_per_slice": round(t_link * 1e3, 2), "frac_of_link": round(t_link / (dt / total_slices), 3)}
        ctx_in.close()
        ctx_out.close()
        plan.close()
        for k in [k for k in self.bufs if k.startswith("hf_")]:
            self.bufs[k] = None
        return out

    # ------------------------------------------------------------------------------------------------------------
    def run(self, cfg, steps, warmup):
        """cfg: dict(tree, kind, codec, gib, file_mib, scaling, partition).  Returns the measurements of this configuration."""
        torch, args, ctx, world, rank = self.torch, self.args, self.ctx, self.world, self.rank
        from longtail_amd.dist import JobPartition, exchange_chunks, sharded_first_seen
        from longtail_amd.lib import Ingest, chunker_params

        mn, av, mx = chunker_params(args.target_chunk_size)
        total_bytes = int(cfg["gib"] * (1 << 30)) * (world if cfg["scaling"] == "weak" else 1)
        file_bytes = int(cfg["file_mib"] * (1 << 20))
        file_bytes -= file_bytes % 16
        tree = make_tree(cfg["tree"], total_bytes, file_bytes, dups=cfg.get("dups", False))
        tree_bytes = int(tree["sizes"].sum())
        part = JobPartition(tree["sizes"], args.target_chunk_size, world, cfg["partition"], self.lib)
        mine = part.jobs_of(rank)
        # ---- the rank's parts, back to back at 16-byte aligned offsets: consecutive parts of one asset stay contiguous ----
        p_size = part.job_size[mine]
        p_off = np.zeros(len(mine), np.uint64)
        if len(mine) > 1:
            np.cumsum(((p_size + np.uint64(15)) // np.uint64(16) * np.uint64(16))[:-1], out=p_off[1:])
        my_bytes = int(p_size.sum())
        arena_in = int(p_off[-1] + p_size[-1]) if len(mine) else 0
        data = self.buf("data", arena_in + 256)
        seeds = asset_seeds(0x10C0FFEE, 0, tree["nfiles"])[tree["seed_of"]]  # (a repeated file has the seed of its original)
        ctx.synth_fill(data, p_off, p_size, seeds[part.job_asset[mine]], KINDS[cfg["kind"]],
                       skips=part.job_offset[mine] + tree["shift"][part.job_asset[mine]])
        ctx.sync()

        # ---- output arrays and arenas (allocated once; nothing is allocated in steady state) ----
        # (the plan's device tables too: every step recomputes the part tables and rewrites them -- lthip_plan_reaim, the job set-up
        # of ChunkAssets stays inside the timed region -- but does not hipMalloc / hipFree them, LTHIP_BENCH_NEW_PLAN=1: a plan per step)
        plan = ctx.make_plan(p_off, p_size, mn, av, mx)
        cap = max(1, plan.capacity)
        new_plan = os.environ.get("LTHIP_BENCH_NEW_PLAN") == "1" or len(mine) == 0
        out_offs = self.buf("offs", cap * 8, torch.int64)
        out_lens = self.buf("lens", cap * 4, torch.int32)
        out_hash = self.buf("hash", cap * 8, torch.int64)
        out_first = self.buf("first", (len(mine) + 1) * 4, torch.int32)
        h_first = self.buf("first_host", (len(mine) + 1) * 4, torch.int32, pinned=True)
        batch_bytes = int(args.batch_gib * (1 << 30))
        limit = args.block_size + args.block_size // 10
        arena_bytes = batch_bytes + batch_bytes // 128 + (batch_bytes // args.block_size + 4) * (16384 + 64) + 2 * (limit + limit // 128 + 16384)
        arena = self.buf("arena", arena_bytes)
        est_chunks = cap * (world if world > 1 else 1)
        vi_cap = int(self.lib.dll.lthip_version_index_size(tree["nfiles"], est_chunks, est_chunks, len(tree["path_data"]))) + 64
        h_vi = self.buf("vi", vi_cap, pinned=True) if rank == 0 else None
        h_si = self.buf("si", 16 + 32 * cap + 64, pinned=True)
        ctype = None
        if cfg["codec"] == "zstd" and args.zstd_settings != 2:
            ctype = 0x7A746430 + args.zstd_settings  # 'ztd3' / 'ztd4' (lib/zstd/longtail_zstd.c:11-28): the max / high parse
        ing = Ingest(ctx, args.target_chunk_size, args.block_size, args.max_chunks_per_block, cfg["codec"], compression_type=ctype,
                     batch_bytes=batch_bytes)
        stats = {}
        from longtail_amd.dist import StepProfile

        xprof = {"host": 0.0, "device": 0.0, "transport": 0.0, "detail": {}, "steps": 0}

        def step():
            prof = StepProfile(ctx) if (args.exchange_profile and world > 1) else None
            t0 = time.perf_counter()
            if new_plan:
                step_plan = ctx.make_plan(p_off, p_size, mn, av, mx)
            else:
                plan.reaim(p_off, p_size)
                step_plan = plan
            # (the part -> first chunk table is what the host builds the tree's job table from: its copy to pinned memory is queued
            # behind the kernels, so that the step's one wait after chunk + hash delivers it together with the chunk count)
            ctx.chunk_hash(step_plan, data, outputs=(out_offs, out_lens, out_hash, out_first), sync=False)
            ctx._check(self.lib.dll.lthip_copy_d2h(ctx.h, h_first.data_ptr(), out_first.data_ptr(), (len(mine) + 1) * 4), "lthip_copy_d2h")
            ctx.sync()
            first_host = h_first.numpy()[: len(mine) + 1].view(np.uint32)
            total = int(first_host[len(mine)]) if len(mine) else 0
            if new_plan:
                step_plan.close()
            t1 = time.perf_counter()
            if world > 1:
                if prof:
                    prof.start()
                counts = out_first[1 : len(mine) + 1] - out_first[: len(mine)]
                ex = exchange_chunks(part, counts, out_hash, out_lens, total, ctx, comm=self.comm, rank=rank, prof=prof)
                all_hash, all_lens, job_first = ex["hashes"], ex["lens"], ex["job_first"].astype(np.uint64)
                my_jobs = mine
                if args.dedup == "sharded":
                    # the first-seen table sharded by hash: this rank inserts its 1/N of the hash space, not every rank's chunks
                    first_all, uniq_all = sharded_first_seen(part, ex, out_hash, total, ctx, comm=self.comm, rank=rank, prof=prof)
                    ing.set_first_seen(first_all, uniq_all)
            else:
                all_hash, all_lens = out_hash, out_lens
                job_first = first_host.astype(np.uint64)
                my_jobs = None
            n_all = int(job_first[-1])
            t2 = time.perf_counter()
            tr, keep = Ingest.tree(tree["sizes"], tree["path_offsets"], tree["perms"], tree["path_data"], part.job_asset, job_first, my_jobs)
            t2b = time.perf_counter()
            if prof:
                prof.mark("host", "index: job / asset tables of the tree (O(jobs))")
            ing.index(tr, all_hash, all_lens, n_all, out_offs, out_first, total, h_vi)
            t3 = time.perf_counter()
            if prof:
                prof.mark("device", "index: lthip_ingest_index (first-seen compaction, ownership, packing of the first batch)")
                for k in ("host", "device", "transport"):
                    xprof[k] += prof.ms[k]
                for k, v in prof.detail.items():
                    xprof["detail"][k] = xprof["detail"].get(k, 0.0) + v
                xprof["steps"] += 1
            if os.environ.get("LTHIP_BENCH_TRACE"):
                print(f"step: chunk_hash {1e3*(t1-t0):.2f} first {1e3*(t2-t1):.2f} tree {1e3*(t2b-t2):.2f} index {1e3*(t3-t2b):.2f} ms", file=sys.stderr)
            if not args.no_compress:
                ing.write(data, arena)
            t3b = time.perf_counter()
            res = ing.finish(h_si)
            t4 = time.perf_counter()
            if os.environ.get("LTHIP_BENCH_TRACE"):
                print(f"step: write (host) {1e3*(t3b-t3):.2f} finish {1e3*(t4-t3b):.2f} ms", file=sys.stderr)
            stats.update(res=res, t=(t1 - t0, t2 - t1, t3 - t2, t4 - t3))

        for _ in range(warmup):
            step()
        ctx.timing(True)
        ctx.timing_reset()
        phase = np.zeros(4)
        xprof.update(host=0.0, device=0.0, transport=0.0, detail={}, steps=0)
        step_ms = []
        self.barrier()
        t_start = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            step()  # (ends with lthip_ingest_finish: the session's full synchronisation)
            step_ms.append((time.perf_counter() - ts) * 1e3)
            phase += stats["t"]
        self.barrier()
        elapsed = time.perf_counter() - t_start
        ktimes = ctx.timing_get()
        ctx.timing(False)
        plan_slices = plan.slices
        res = stats["res"]
        comp, blocks, raw = res.compressed_bytes, res.blocks, res.raw_bytes
        if world > 1:
            elapsed = float(self.reduce([elapsed], "max")[0])
            comp, blocks, raw = (int(round(x)) for x in self.reduce([comp, blocks, raw]))
        sizes = ing.compressed_sizes(res.blocks)
        ing.close()
        plan.close()

        # ---- per-kernel rates: ALGORITHMIC bytes per launch / average launch duration (HIP events on the stream) ----
        # phase 1 = N read per kernel; LZ4 match finder / zstd entropy stage = N read + N_out written, N = the bytes
        # this rank WRITES (unique chunks: what reaches the codec), N_out its payload bytes
        codec_in = int(res.raw_bytes)
        alg_bytes = {"buzhash": my_bytes, "blake3_leaf": my_bytes, "lz4_segments": codec_in + int(res.compressed_bytes),
                     "zstd_encode": codec_in + int(res.compressed_bytes)}
        alg_input_only = {"buzhash": my_bytes, "blake3_leaf": my_bytes, "lz4_segments": codec_in, "zstd_encode": codec_in}
        kern = {}
        for name, (ms, n) in ktimes.items():
            if n:
                per_step = ms / steps
                kern[name] = {"ms_per_step": round(per_step, 3), "launches_per_step": n / steps}
                if name in alg_bytes:
                    kern[name]["GBps"] = round(alg_bytes[name] / (per_step * 1e-3) / 1e9, 1)
        # lthip_chunk_hash runs a large plan as TWO slices on two streams (the candidate scan of the second half beside the leaf hashing
        # of the first): the event times of those kernels overlap each other -- their sum exceeds the phase's wall time -- and say
        # what a kernel took while it SHARED the device, not what it takes.  They are marked; the dominant kernel of the roofline
        # block is the longest one that had the device to itself (K1 / K3 alone: profiles/*_kernel_stats.csv of an ablation-build run
        # with LTHIP_SLICES=1)
        slices = plan_slices
        if slices > 1:
            for k in ("buzhash", "select", "compact", "blake3_leaf", "blake3_parent"):
                if k in kern:
                    kern[k]["overlapped"] = True
                    kern[k].pop("GBps", None)
            kern["chunk_hash_phase"] = {"ms_per_step": round(phase[0] / steps * 1e3, 3), "launches_per_step": 1.0, "slices": slices,
                                        "GBps": round(2 * my_bytes / (phase[0] / steps) / 1e9, 1),
                                        "note": "wall time of lthip_chunk_hash (scan + cut selection + leaf hashing + parents, two slices on two streams); "
                                                "GBps = 2 N (the input read by the scan and by the hashing) / that time"}
        dom = max((k for k in kern if k in alg_bytes and not kern[k].get("overlapped")), key=lambda k: kern[k]["ms_per_step"], default=None)
        if dom is None:
            dom = max((k for k in kern if k in alg_bytes), key=lambda k: kern[k]["ms_per_step"], default=None)
        roofline = None
        if dom:
            launches = kern[dom]["launches_per_step"]
            avg_ms = kern[dom]["ms_per_step"] / launches
            achieved = alg_bytes[dom] / launches / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "algorithmic_bytes_per_launch": int(alg_bytes[dom] / launches), "avg_launch_ms": round(avg_ms, 3),
                        "algorithmic_bytes": "SURVEY.md §8(d): N read per phase-1 kernel; N read + N_out written for the codec kernels",
                        "frac_over_input_only": round(alg_input_only[dom] / launches / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if slices > 1:
                roofline["dominant_kernel_choice"] = (
                    f"the longest kernel that had the device to itself; the candidate scan and the leaf hashing run CONCURRENTLY on {slices} slices of the "
                    f"parts (lthip_chunk_hash), so their event / rocprof durations overlap each other -- buzhash {kern.get('buzhash', {}).get('ms_per_step')} + "
                    f"blake3_leaf {kern.get('blake3_leaf', {}).get('ms_per_step')} ms inside a phase of {kern['chunk_hash_phase']['ms_per_step']} ms; run one "
                    "after the other (ablation build, LTHIP_SLICES=1: profiles/*_one_slice_kernel_stats.csv) they take 24.0 and 25.7 ms per step, both below "
                    "the classification pass")
            pm = getattr(self, "peak_measured", None)
            if pm:
                # the box's own memory system beside the nominal figure: a device copy for kernels that read and write (the codec),
                # a read-only reduction for the ones that only read (scan, hash)
                ref = pm["copy_GBps"] if dom in ("lz4_segments", "zstd_encode") else pm["read_GBps"]
                roofline["peak_measured"] = dict(pm, frac=round(achieved / ref, 4), against="copy" if dom in ("lz4_segments", "zstd_encode") else "read")
            if dom in VALU_INSTR_PER_4KIB:
                roof = VALU_ISSUE_PER_S / VALU_INSTR_PER_4KIB[dom] * 4096 / 1e9
                roofline["valu"] = {"instr_per_4KiB_wave_tile": VALU_INSTR_PER_4KIB[dom], "issue_roof_GBps": round(roof, 1),
                                    "valu_frac": round(achieved / roof, 4),
                                    "note": "integer-issue roof: one VALU instruction per SIMD per 4 cycles at 2.4 GHz, 1024 SIMDs"}
            # HBM traffic from the PMC counters: collected offline (rocprofv3 cannot wrap itself).  Round 3: tools/pmc_exact_traffic.sh
            # -> profiles/*xtraffic*.json, bytes from the L2's request-SIZE counters (128 * RDREQ_128B + 64 * RDREQ_64B + 32 * RDREQ_32B,
            # 64 / 32 per write request): exact for any access pattern, where 2 * FETCH_SIZE is only right for wide coalesced reads.
            # Per input byte, scaled to this run's launch size.
            pipeline_traffic = None
            try:
                tj, tsrc = getattr(self, "traffic", None), "measured in this run (rocprofv3 --kernel-trace --pmc, 8 GiB of the same workload)"
                if tj is None or cfg["kind"] != self.traffic_cfg["kind"] or cfg["codec"] != self.traffic_cfg["codec"] or cfg["tree"] != self.traffic_cfg["tree"]:
                    # the newest committed measurement OF THIS WORKLOAD: ..._xtraffic_8g[_zstd][_mixed].json
                    want = "_xtraffic_8g" + ("_zstd" if cfg["codec"] == "zstd" else "") + ("_mixed" if cfg["kind"] != "random" else "") + ".json"
                    tfiles = sorted(f for f in (ROOT / "profiles").glob("*xtraffic*.json") if f.name.endswith(want) and
                                    ("zstd" in f.name) == (cfg["codec"] == "zstd") and ("mixed" in f.name) == (cfg["kind"] != "random"))
                    tj, tsrc = json.load(open(tfiles[-1])), f"profiles/{tfiles[-1].name} (committed measurement)"
                prefix = {"buzhash": ("k_buzhash",), "blake3_leaf": ("k_blake3_leaves",), "lz4_segments": ("k_lz4_segments<", "k_lz4_lanes2<"),
                          "zstd_encode": ("k_zstd_encode",)}

                def fmt_of(k):  # k_lz4_segments<G, TAB, FMT, MODE, CLS, ...> / k_lz4_lanes2<FMT>: the LZ4 flavours have FMT 0
                    args = k[k.index("<") + 1 : k.rindex(">")].split(",")
                    return int(args[2]) if len(args) >= 3 else int(args[-1])

                want_fmt = 0 if cfg["codec"] == "lz4" else 1

                def per_byte(k):
                    return tj["kernels"][k]["read_per_input_byte"] + tj["kernels"][k]["write_per_input_byte"]

                keys = [k for k in tj["kernels"] if k.startswith(prefix[dom]) and (dom != "lz4_segments" or fmt_of(k) == want_fmt)]
                if dom == "zstd_encode":  # (its match finder runs under the lz4_segments timer; the entropy stage is what dominates)
                    keys = [k for k in tj["kernels"] if k.startswith("k_zstd_encode")]
                if not keys:
                    raise KeyError(dom)
                # the match finder is two kernels (classification pass + lane parser) under one timer: their traffic adds up
                ratio = round(sum(per_byte(k) for k in keys), 4)
                roofline["traffic"] = int(ratio * my_bytes / launches)
                roofline["traffic_source"] = (f"{tsrc}: {ratio} memory-side bytes per input byte (L2 request-size counters, reads + writes), "
                                              f"workload {tj.get('workload', 'the default workload at 8 GiB')}")
                pipeline_traffic = round(sum(per_byte(k) for k in tj["kernels"] if not k.startswith("k_synth")), 4)
            except Exception:
                pass
            # the whole step against the same roof: algorithmic bytes = N read once + N_out written once (SURVEY.md §8d "combined minimal")
            step_s = elapsed / steps
            alg_pipeline = my_bytes + int(res.compressed_bytes)
            roofline["pipeline"] = {"algorithmic_bytes": alg_pipeline, "ms": round(step_s * 1e3, 3),
                                    "achieved": round(alg_pipeline / step_s / 1e9, 1), "frac": round(alg_pipeline / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                    "frac_over_input_only": round(my_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                    "traffic_per_input_byte": pipeline_traffic,
                                    "traffic_over_algorithmic": None if pipeline_traffic is None else round(pipeline_traffic * my_bytes / alg_pipeline, 3),
                                    "note": "all kernels of one step (three serial passes over the input: scan, hash, codec); traffic = sum of the kernels' "
                                            "memory-side bytes from the same PMC file"}
        dup_note = ", a quarter of the files repeated (whole and shifted)" if cfg.get("dups") else ""
        label = {"files": f"{cfg['gib']:g} GiB tree of {cfg['file_mib']:g} MiB {cfg['kind']} files{dup_note}",
                 "mixed-sizes": f"{cfg['gib']:g} GiB tree of {cfg['kind']} files, 4 KiB..4 GiB log-uniform (north-star tree)"}[cfg["tree"]]
        per = "per GPU" if cfg["scaling"] == "weak" else "in total"
        which = ""
        zset = f" 'ztd{args.zstd_settings}'" if cfg["codec"] == "zstd" and args.zstd_settings != 2 else ""
        if cfg["tree"] == "files" and cfg["kind"] == "random" and cfg["codec"] == "lz4" and abs(cfg["file_mib"] - 1.0) < 1e-9:
            which = " (BASELINE.json configs[2])" if world == 1 else " (BASELINE.json configs[3])"
        elif cfg["tree"] == "files" and cfg["codec"] == "zstd" and cfg["file_mib"] >= 16384:
            which = " (BASELINE.json configs[4] shape)"
        exchange_profile = None
        if xprof["steps"]:
            k = xprof["steps"]
            mx = self.reduce([xprof["host"] / k, xprof["device"] / k, xprof["transport"] / k], "max")
            exchange_profile = {"what": "phase_ms.exchange + index of a PROFILED step (the device is waited for at every mark: a breakdown, slower "
                                        "than the plain step), per step; max over ranks, and rank 0's items",
                                "host_ms": round(mx[0], 3), "device_ms": round(mx[1], 3), "transport_ms": round(mx[2], 3),
                                "rank0_detail_ms": {a: round(b / k, 3) for a, b in xprof["detail"].items()}}
        sm = sorted(step_ms)
        return {
            "value": tree_bytes * steps / elapsed / 1e9,
            "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_spread": {"min": round(sm[0], 3), "median": round(sm[len(sm) // 2], 3), "max": round(sm[-1], 3),
                                   "note": "this rank's wall time of each timed step"},
            "exchange_profile": exchange_profile,
            "workload": f"{label} {per}, CreateVersionIndex + CreateMissingContent + WriteContent, chunk+BLAKE3+{cfg['codec'].upper()}{zset}{which}",
            "tree_bytes": tree_bytes, "bytes_this_rank": my_bytes, "files": tree["nfiles"], "jobs": int(part.job_count),
            "jobs_this_rank": int(len(mine)), "min_avg_max": [mn, av, mx],
            "roofline": roofline, "kernels": kern,
            "phase_ms": {"chunk_hash": round(phase[0] / steps * 1e3, 2), "exchange": round(phase[1] / steps * 1e3, 2),
                         "index": round(phase[2] / steps * 1e3, 2), "write_finish": round(phase[3] / steps * 1e3, 2)},
            "result": {"chunks": int(res.chunks_all), "unique_chunks": int(res.unique_all), "blocks": int(blocks),
                       "raw_bytes_written": int(raw), "compressed_bytes": int(comp),
                       "ratio": round(raw / comp, 4) if comp else None, "gathered_blocks_rank0": int(res.gathered_blocks),
                       "gathered_bytes_rank0": int(res.gathered_bytes),
                       "gather_GBps_rank0": (round(res.gathered_bytes / (kern["gather"]["ms_per_step"] * 1e-3) / 1e9, 1)
                                             if res.gathered_bytes and "gather" in kern else None),
                       "version_index_bytes": int(res.version_index_size), "store_index_bytes_rank0": int(res.store_index_size)},
        }


def make_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=64.0, help="GiB of assets per GPU (weak) or in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--partition", choices=["range", "lpt", "mod"], default="range", help="lthip_partition_jobs policy (N > 1)")
    ap.add_argument("--file-mib", type=float, default=1.0)
    ap.add_argument("--tree", choices=["files", "mixed-sizes"], default="files", help="files: equal files (configs[2]); mixed-sizes: 4 KiB..4 GiB log-uniform")
    ap.add_argument("--kind", choices=sorted(KINDS), default="random")
    ap.add_argument("--target-chunk-size", type=int, default=65536)
    ap.add_argument("--block-size", type=int, default=8 << 20)
    ap.add_argument("--max-chunks-per-block", type=int, default=1024)
    ap.add_argument("--batch-gib", "--lz4-batch-gib", dest="batch_gib", type=float, default=8.0)
    ap.add_argument("--codec", choices=["lz4", "zstd"], default="lz4", help="block codec (BASELINE.json configs[4] uses zstd)")
    ap.add_argument("--zstd-settings", type=int, choices=[2, 3, 4], default=2,
                    help="the reference's zstd settings id 'ztd2' (default) / 'ztd3' (max) / 'ztd4' (high): the parse the session runs")
    ap.add_argument("--no-compress", action="store_true", help="diagnostic: skip WriteContent (the line is then not the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dedup", choices=["sharded", "replicated"], default="sharded",
                    help="N > 1: first-seen table sharded by hash (all-to-all, every rank inserts 1/N of the chunks) or rebuilt on every rank")
    ap.add_argument("--collective", choices=["torch", "c"], default="torch",
                    help="N > 1: the exchange's all-gathers by torch.distributed (RCCL backend) or by the C ABI's lthip_comm_allgather")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the newest profiles/*xtraffic*.json instead of a live rocprofv3 --pmc run of this script on 8 GiB")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: print how the tree's (asset, part) jobs fall onto --gpus ranks and what the exchange moves")
    ap.add_argument("--no-secondary", action="store_true", help="skip the compressible / mixed-size-tree measurements")
    ap.add_argument("--dups", action="store_true", help="a quarter of the tree's files repeat earlier ones (whole-file and shifted duplicates)")
    ap.add_argument("--launch", choices=["torch", "plain"], default="torch",
                    help="--gpus N > 1 started without a launcher (WORLD_SIZE unset): how this process starts its N ranks -- 'torch': the "
                         "environment of torch.distributed.run (collectives by torch.distributed = RCCL), 'plain': no torch.distributed, "
                         "every collective through the C ABI (lthip_comm_*), the id through a file (what tools/run8.sh does)")
    ap.add_argument("--handshake-only", action="store_true",
                    help="plain launch only: id file + lthip_comm_create + barrier + reductions + one all-to-all, then a JSON line; no kernels")
    ap.add_argument("--cpu-gib", type=float, default=8.0, help="sample size of the CPU baseline")
    ap.add_argument("--exchange-profile", action="store_true",
                    help="N > 1: split the exchange + index phases of every step into host / device / transport time (waits for the device at "
                         "every mark: a breakdown, not the metric; tools/exchange_cost.py)")
    return ap


def main():
    args = make_parser().parse_args()
    if args.dry_run:
        print(json.dumps(dry_run(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.handshake_only:
            args.launch = "plain"  # (the handshake IS the plain launch's)
        raise SystemExit(self_launch(args))
    if args.handshake_only:
        handshake_only(args)
        return

    # the traffic leg runs FIRST, in a child process under rocprofv3 (a profiler cannot wrap the process it runs in), rank 0 of a
    # single-GPU run only; --no-live-traffic (and the child itself) take the committed measurement instead
    traffic = None
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    plain_default = (args.tree == "files" and args.kind == "random" and args.codec == "lz4" and not args.no_compress and not args.no_secondary
                     and not args.no_cpu_baseline)  # the driver's command; every tool / test / profiler run passes one of these flags
    under_profiler = any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ)
    if world_env == 1 and not args.no_live_traffic and plain_default and not under_profiler:
        traffic = live_traffic(args)
    b = Bench(args)
    b.traffic, b.traffic_cfg = traffic, dict(kind=args.kind, codec=args.codec, tree=args.tree)
    cfg = dict(tree=args.tree, kind=args.kind, codec=args.codec, gib=args.gib, file_mib=args.file_mib, scaling=args.scaling,
               partition=args.partition, dups=args.dups)
    if b.world == 1 and not under_profiler:
        b.measure_peak(4 if args.gib >= 8 else 1)
    main_res = b.run(cfg, args.steps, args.warmup)
    secondary = None
    default_headline = args.tree == "files" and args.kind == "random" and args.codec == "lz4" and not args.no_compress
    # the reference's CPU path beside EVERY number ("same tree on W cores"): one CpuReference for the headline and the
    # secondary workloads, rank 0 of a single-GPU run only
    cr = CpuReference(b, args) if (b.rank == 0 and b.world == 1 and not args.no_cpu_baseline) else None
    cpu_baseline = None
    try:
        if not args.no_secondary and default_headline:
            # "report both" -- the compressible variant (match path + a real ratio) and the north-star tree; and
            # BASELINE.json configs[4]'s shape (4 x 16 GiB PAK-style files, ZStd 'ztd2') on data the codec can compress
            secondary = {}
            legs = (("compressible", dict(kind="mixed"), 4.0, True), ("dedup", dict(kind="mixed", dups=True), 2.0, False),
                    ("north_star_tree", dict(tree="mixed-sizes"), 2.0, False),
                    ("north_star_tree_compressible", dict(tree="mixed-sizes", kind="mixed"), 2.0, False),
                    ("zstd_compressible", dict(kind="mixed", codec="zstd"), 4.0, True),  # (the settings id of --zstd-settings: 'ztd2' by default)
                    ("pak_zstd", dict(kind="mixed", codec="zstd", file_mib=16384.0), 4.0, True))
            for name, over, cpu_gib, with_drop_in in legs:
                if name == "pak_zstd" and args.gib < 16:
                    continue  # (the shape needs at least one whole 16 GiB file)
                c = dict(cfg, **over)
                r = b.run(c, max(1, min(args.steps, 2)), 1)
                secondary[name] = {"value": round(r["value"], 3), "unit": "GB/s", "ms_per_step": round(r["ms_per_step"], 3),
                                   "workload": r["workload"], "ratio": r["result"]["ratio"], "phase_ms": r["phase_ms"],
                                   "dominant_kernel": r["roofline"] and {k: r["roofline"][k] for k in ("kernel", "achieved", "frac")}}
                if over.get("dups"):
                    # first-seen hits, CreateMissingContent with chunks to drop, blocks assembled from non-contiguous unique chunks
                    secondary[name].update({k: r["result"][k] for k in ("chunks", "unique_chunks", "blocks", "raw_bytes_written",
                                                                        "gathered_blocks_rank0", "gathered_bytes_rank0", "gather_GBps_rank0")})
                    secondary[name]["dedup_table"] = args.dedup if b.world > 1 else "single rank"
                if cr is not None:
                    cb = cr.leg(c, min(cpu_gib, args.cpu_gib, args.gib), drop_in=with_drop_in)
                    if "drop_in" in cb:
                        secondary[name]["drop_in"] = cb.pop("drop_in")
                    secondary[name]["reference_ratio"] = cb.get("reference_ratio")
                    secondary[name]["cpu_baseline"] = cb
                    if cb.get("value"):
                        secondary[name]["vs_cpu_baseline"] = round(secondary[name]["value"] / cb["value"], 1)
            if b.world == 1:
                secondary["restore"] = b.restore_rates()
                secondary["host_fed"] = b.host_fed_rates()
        if cr is not None:
            sweep = sorted(({min(32, cr.ncpu), cr.physical, cr.ncpu} | ({max(2, int(cr.quota))} if cr.quota else set())) - {1}) or [1]
            cpu_baseline = cr.leg(cfg, min(args.cpu_gib, args.gib), sweep=sweep, drop_in=True, one_gib=1.0)
            if "error" not in cpu_baseline and cr.have_ref:
                cpu_baseline["configs0_one_256MiB_file"] = cr.configs0(sorted({sweep[0], cr.ncpu}))
            if cpu_baseline.get("drop_in") is not None:
                if secondary is None:
                    secondary = {}
                secondary["drop_in"] = cpu_baseline.pop("drop_in")
    finally:
        if cr is not None:
            cr.close()
    if b.rank == 0:
        line = {
            "metric": "ingest GB/s (chunk+hash+compress)",
            "value": round(main_res["value"], 3),
            "unit": "GB/s",
            "n_gpus": b.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(main_res["ms_per_step"], 3),
            "ms_per_step_spread": main_res["ms_per_step_spread"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8/u32",
            "data": "synthetic",
            "config": {
                "workload": main_res["workload"],
                "target_chunk_size": args.target_chunk_size, "min_avg_max": main_res["min_avg_max"], "block_size": args.block_size,
                "max_chunks_per_block": args.max_chunks_per_block, "tree_bytes": main_res["tree_bytes"], "files": main_res["files"],
                "jobs": main_res["jobs"], "bytes_rank0": main_res["bytes_this_rank"], "jobs_rank0": main_res["jobs_this_rank"],
                "device_block_assembly": main_res["result"]["gathered_blocks_rank0"] > 0,
                "sharding": (f"(asset, part) jobs over {b.world} ranks by lthip_partition_jobs('{args.partition}'), RCCL all-gather of "
                             "per-job chunk counts + hashes + lengths") if b.world > 1 else "single GPU",
                "comm": b.comm_info, "dedup_table": (args.dedup if b.world > 1 else "single rank"),
                "metric_definition": "wall time of CreateVersionIndex + CreateMissingContent + WriteContent equivalents (SURVEY.md §8d), "
                                     "serialized VersionIndex and StoreIndex delivered to host memory, stored-block images to a null sink",
            },
            "roofline": main_res["roofline"],
            "cpu_baseline": cpu_baseline,
            "secondary": secondary,
            "kernels": main_res["kernels"],
            "phase_ms": main_res["phase_ms"],
            "result": main_res["result"],
        }
        if main_res.get("exchange_profile"):
            line["exchange_profile"] = main_res["exchange_profile"]
        print(json.dumps(line))
    if b.world > 1 and not b.plain:
        b.dist.destroy_process_group()


def live_traffic(args, timeout_s=150):
    """Memory-side traffic per kernel, measured NOW: this script again on 8 GiB under `rocprofv3 --kernel-trace --pmc <L2 request-size
    counters>` (two passes: reads, writes; counters and arithmetic of tools/pmc_exact_traffic.sh).  Returns the same structure as
    profiles/*xtraffic*.json or None (no rocprofv3, a failure, a timeout): the caller then falls back to the committed file."""
    import collections
    import csv
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("rocprofv3"):
        return None
    inp = 8 << 30
    base = [sys.executable, str(ROOT / "bench.py"), "--gib", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary",
            "--no-live-traffic", "--kind", args.kind, "--codec", args.codec, "--tree", args.tree, "--file-mib", str(min(args.file_mib, 1024.0)),
            "--target-chunk-size", str(args.target_chunk_size)]
    passes = (["TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum"], ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, ctrs in enumerate(passes):
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "p", "--", *base],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                files = list(Path(d).rglob("p_counter_collection.csv"))
                if r.returncode != 0 or not files:
                    return None
                for row in csv.DictReader(open(files[0])):
                    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    if i == 0 and row["Counter_Name"] == ctrs[0]:
                        launches[k] += 1
    except Exception:
        return None
    out = {"_about": "measured by this run: bench.py --gib 8 --steps 1 under rocprofv3 --kernel-trace --pmc (L2 request-size counters)",
           "input_bytes": inp, "workload": f"8 GiB, kind {args.kind}, codec {args.codec}, tree {args.tree}", "kernels": {}}
    for k, v in agg.items():
        rd = 128 * v["TCC_EA0_RDREQ_128B_sum"] + 64 * v["TCC_EA0_RDREQ_64B_sum"] + 32 * v["TCC_EA0_RDREQ_32B_sum"]
        wr = 64 * v["TCC_EA0_WRREQ_64B_sum"] + 32 * (v["TCC_EA0_WRREQ_sum"] - v["TCC_EA0_WRREQ_64B_sum"])
        if rd + wr >= 0.005 * inp:
            out["kernels"][k] = {"launches": launches[k], "read_per_input_byte": round(rd / inp, 4), "write_per_input_byte": round(wr / inp, 4)}
    return out if out["kernels"] else None


def dry_run(args):
    """What a --gpus N run WOULD do, computed on the host alone (the partitioner is plain C in liblongtail_hip.so and needs no GPU):
    the jobs of the tree exactly as ChunkAssets lists them (src/longtail.c:2399-2457), their assignment, the balance, and the bytes
    the exchange carries -- for BASELINE.json configs[3] (default tree, --scaling strong) and configs[4] (--file-mib 16384)."""
    from longtail_amd.dist import JobPartition

    world = args.gpus
    per_gpu = int(args.gib * (1 << 30))
    total = per_gpu * (world if args.scaling == "weak" else 1)
    file_bytes = int(args.file_mib * (1 << 20))
    tree = make_tree(args.tree, total, file_bytes)
    part = JobPartition(tree["sizes"], args.target_chunk_size, world, args.partition)
    rb = part.rank_bytes.astype(np.int64)
    mean_chunk = 32.1 * 1024 * args.target_chunk_size / 65536  # observed mean on random data
    chunks_rank = rb / mean_chunk
    chunks_all = float(chunks_rank.sum())
    count_stride = int(part.jobs_per_rank.max())
    straddle = int(sum(1 for a in range(len(tree["sizes"])) if len(set(part.job_rank[part.job_asset == a].tolist())) > 1)) if len(tree["sizes"]) <= 4096 else None
    return {
        "dry_run": True, "n_gpus": world, "scaling": args.scaling, "partition": args.partition, "tree_bytes": int(tree["sizes"].sum()),
        "files": int(tree["nfiles"]), "jobs": int(part.job_count), "jobs_per_rank": [int(x) for x in part.jobs_per_rank],
        "bytes_per_rank": [int(x) for x in rb], "imbalance_max_over_mean": round(float(rb.max() / max(rb.mean(), 1)), 4),
        "assets_whose_parts_straddle_ranks": straddle, "rank_major_is_job_order": part.is_rank_major(),
        "expected_chunks_total": int(chunks_all), "expected_chunks_largest_rank": int(chunks_rank.max()),
        "exchange_bytes_received_per_rank": {
            "allgather_job_counts": 4 * count_stride * world,
            "allgather_hashes_and_lengths": int(12 * chunks_rank.max() * world),
            "sharded_first_seen (all-to-all of hash + ordinal, reply, all-gather of the first ordinals)": int(12 * chunks_rank.max() + 4 * chunks_rank.max() + 4 * chunks_all),
        },
        "first_seen_table_inserts_per_rank": {"replicated": int(chunks_all), "sharded": int(chunks_all / world)},
        "note": "expected_* assume the 32.1 KiB mean chunk of random data at target 65536; the all-gathers are padded to the largest rank",
    }


class CpuReference:
    """SURVEY.md §8(d) protocol, for the headline AND for every secondary workload: the reference's own code (oracle/_ref:
    Longtail_CreateVersionIndex + Longtail_CreateMissingContent + Longtail_WriteContent with the reference hpcdc + BLAKE3 + LZ4/ZStd
    plugins and Longtail_CreateBikeshedJobAPI(W, 0)), source tree on tmpfs behind the reference's file storage, null block sink, 3
    repetitions, median -- on a bounded SAMPLE of the same tree (the same generator, seeds and kind; the bytes are synthesized by the
    GPU's k_synth_fill, bit-identical to the oracle's generator, and copied to the host: the sample is not what is timed).  Beside it,
    on the same files: the reference codec's ratio (`reference_ratio`) and the unmodified core with this library's plugin objects
    (`drop_in`).  Without the reference build: the single-thread C restatement (oracle/)."""

    REPS = 3

    def __init__(self, b, args):
        from tests._libs import have_ref, oracle

        self.b, self.args, self.o = b, args, oracle()
        self.ncpu = os.cpu_count() or 1
        self.physical = max(1, self.ncpu // 2)
        self.w = min(32, self.ncpu)
        # what the CONTAINER grants, whatever os.cpu_count() says: the cgroup's CPU bandwidth (cpu.max "quota period": quota / period
        # CPU-seconds per second for all threads of the process together) and the affinity mask
        self.quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                self.quota = round(int(q) / int(per), 2)
        except (OSError, ValueError):
            try:
                q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    self.quota = round(q / per, 2)
            except (OSError, ValueError):
                pass
        try:
            self.affinity = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            self.affinity = self.ncpu
        self.have_ref = have_ref()
        self.storage = "reference in-memory storage"
        self.hip = None
        if self.have_ref:
            from tests._libs import ref

            self.r = ref()
            shm = "/dev/shm"
            try:
                st = os.statvfs(shm)
                if os.access(shm, os.W_OK) and st.f_bavail * st.f_frsize > (24 << 30):
                    self.r.dll.refh_set_tree_dir.argtypes = [ctypes.c_char_p]
                    self.r.dll.refh_set_tree_dir(shm.encode())
                    self.storage = f"reference file storage on tmpfs ({shm})"
            except OSError:
                pass

    def cores_text(self):
        quota = (f"; the container's cgroup grants this process {self.quota:g} CPUs' worth of time (cpu.max) whatever the {self.ncpu} visible hardware "
                 f"threads suggest -- THAT is the core count of this baseline, and why more than a few dozen workers are no faster (by_workers)") if self.quota else ""
        return (f"W = {self.w} bikeshed workers on a host with {self.ncpu} hardware threads ({self.physical} cores){quota}: the ratio to this "
                "figure is 'vs the reference at its best worker count on what the box grants', not 'vs every hardware thread busy'")

    def plugins(self, codec):
        """This library's plugin objects for the drop-in legs (made once, kept until close())."""
        if self.hip is None:
            from longtail_amd.lib import load

            d = load().dll
            self.hip = {"chunker": d.Longtail_CreateHipChunkerAPI(), "hash": d.Longtail_CreateHipBlake3HashAPI(),
                        "lz4": d.Longtail_CreateHipLZ4CompressionAPI(), "zstd": d.Longtail_CreateHipZStdCompressionAPI()}
            if not all(self.hip.values()):
                raise RuntimeError("a HIP plugin constructor returned NULL")
        return self.hip["chunker"], self.hip["hash"], self.hip[codec]

    def close(self):
        if self.hip:
            dispose = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
            for ptr in self.hip.values():
                dispose(ctypes.c_void_p.from_address(ptr).value)(ptr)
            self.hip = None
        if self.b is not None:
            self.b.bufs["cpu_sample"] = None

    def sample_files(self, cfg, sample_bytes):
        """[(path, bytes)] of a tree of `sample_bytes` made by the SAME generator as the measured tree: tree 'files' -> the first files of
        it (a file larger than the sample -- the 16 GiB PAK files -- is represented by its own first `sample_bytes`), 'mixed-sizes' -> the
        log-uniform size generator run to that total, dups -> the same repeat / shift pattern."""
        file_bytes = int(cfg["file_mib"] * (1 << 20))
        file_bytes -= file_bytes % 16
        if cfg["tree"] == "files" and file_bytes > sample_bytes:
            file_bytes = sample_bytes - sample_bytes % 16
        tree = make_tree(cfg["tree"], sample_bytes, file_bytes, dups=cfg.get("dups", False))
        sizes, n = tree["sizes"], tree["nfiles"]
        offs = np.zeros(n, np.uint64)
        if n > 1:
            np.cumsum(((sizes + np.uint64(15)) // np.uint64(16) * np.uint64(16))[:-1], out=offs[1:])
        total = int(offs[-1] + sizes[-1])
        seeds = asset_seeds(0x10C0FFEE, 0, n)[tree["seed_of"]]
        names = [f"dir{i % 256:03d}/file{i:06d}.bin" for i in range(n)]
        kind = KINDS[cfg["kind"]]
        if self.b is not None:
            b = self.b
            dev = b.buf("data", total + 256)
            b.ctx.synth_fill(dev, offs, sizes, seeds, kind, skips=tree["shift"])
            b.ctx.sync()
            host = b.buf("cpu_sample", total, pinned=True)
            host[:total].copy_(dev[:total])
            b.torch.cuda.synchronize(b.dev)
            arr = host.numpy()
            return [(names[i], arr[int(offs[i]) : int(offs[i]) + int(sizes[i])]) for i in range(n)], int(sizes.sum())
        return [(names[i], self.o.synth(int(sizes[i]), int(seeds[i]), kind, int(tree["shift"][i]))) for i in range(n)], int(sizes.sum())

    def _median(self, res, workers, nbytes):
        out = {}
        for wi, w in enumerate(workers):
            tot = res["seconds"][wi].sum(axis=1)
            med = int(np.argsort(tot)[self.REPS // 2])
            sec = res["seconds"][wi][med]
            out[str(w)] = {"GBps": round(nbytes / float(tot[med]) / 1e9, 3), "median_s": round(float(tot[med]), 4),
                           "index_s": round(float(sec[0]), 4), "missing_s": round(float(sec[1]), 4), "write_s": round(float(sec[2]), 4)}
        return out

    def leg(self, cfg, sample_gib, sweep=None, drop_in=True, one_gib=0.0):
        """cpu_baseline (+ reference_ratio, + drop_in) of one workload.  sweep: worker counts (default: W = 32 only)."""
        args = self.args
        if not self.have_ref:
            return self.port_leg(cfg)
        r = self.r
        codec = cfg["codec"]
        tag = r.lz4_type if codec == "lz4" else r.zstd_default
        sample_bytes = int(sample_gib * (1 << 30))
        self._sample_request_bytes = sample_bytes
        files, nbytes = self.sample_files(cfg, sample_bytes)
        sweep = sorted(set(sweep or [self.w]))
        common = (args.target_chunk_size, args.block_size, args.max_chunks_per_block, tag)
        tree = r.tree_create(files, tag)
        try:
            res = r.ingest_sweep_tree(tree, *common, sweep, self.REPS)
            if res["err"]:
                return {"error": f"reference ingest failed: errno {res['err']}"}
            by_w = self._median(res, sweep, nbytes)
            payload = res["stored_bytes"] - 8 * res["blocks"]  # ([raw][compressed] heads each stored block, compressblockstore.c:134-136)
            ref_ratio = round(res["raw_bytes"] / payload, 4) if payload > 0 else None
            best = max(by_w, key=lambda k: by_w[k]["GBps"])
            out = {"value": by_w[best]["GBps"], "unit": "GB/s", "cores": int(best), "kind": "reference",
                   "sample": f"{len(files)} file(s), {nbytes} B = {nbytes / (1 << 30):.2f} GiB of this workload's tree (same generator, seeds, kind '{cfg['kind']}'"
                             f"{', repeats' if cfg.get('dups') else ''}); Longtail_CreateVersionIndex + Longtail_CreateMissingContent + Longtail_WriteContent, "
                             f"reference hpcdc+BLAKE3+{codec.upper()}{' level 3 (ztd2)' if codec == 'zstd' else ''}, {self.storage}, null block sink; "
                             f"median of {self.REPS}; " + self.cores_text(),
                   "by_workers": by_w, "physical_cores": self.physical, "host_cpus": self.ncpu, "cpu_quota_cores": self.quota,
                   "affinity_cpus": self.affinity,
                   "reference_ratio": ref_ratio, "reference_chunks": res["chunks"], "reference_blocks": res["blocks"],
                   "sample_fraction_of_tree": round(nbytes / float(int(cfg["gib"] * (1 << 30))), 4)}
            if drop_in and str(self.w) in by_w:
                try:
                    out["drop_in"] = self.drop_in(tree, files, nbytes, cfg, common, by_w[str(self.w)], ref_ratio)
                except Exception as e:  # (the baseline itself must not fail with it)
                    out["drop_in"] = {"error": repr(e)}
        finally:
            r.tree_destroy(tree)
        if one_gib and 1 not in sweep:
            # the single-thread leg on a smaller sample (0.5 GB/s would take 16 s per repetition on 8 GiB)
            n1 = max(1, min(len(files), int(one_gib * (1 << 30)) // max(1, len(files[0][1]))))
            one = r.ingest_sweep(files[:n1], *common, [1], self.REPS)
            if not one["err"]:
                out["by_workers"]["1"] = dict(self._median(one, [1], sum(len(d) for _, d in files[:n1]))["1"], sample_files=n1)
        return out

    def drop_in(self, tree, files, nbytes, cfg, common, cpu, ref_ratio):
        """secondary.*.drop_in: the SAME sample (generator, seeds, size), storage and worker count with this library's plugin objects in the
        unmodified core -- what a longtail embedder gets by switching constructors and nothing else (host buffers in, host buffers out:
        PCIe, the pull-style per-chunk API and the core's own file reads are all inside) -- all three, and chunker + hash with the CPU
        codec.  Measured in a PROCESS OF ITS OWN (tools/drop_in_child.py) that makes Longtail_Hip_SetBlockingWaits(1) its first call, as
        include/longtail_hip.h tells an embedder to: the wait policy has to be set before a process touches the GPU, and this one long has.
        If the child fails or takes too long, the measurement is made here, with the runtime's default (polling) waits, and says so."""
        w = self.w
        more = [x for x in (2 * w,) if x <= max(self.ncpu, w)]
        if os.environ.get("LTHIP_BENCH_DROPIN_SWEEP", "1") not in ("0", "1"):
            more = [int(x) for x in os.environ["LTHIP_BENCH_DROPIN_SWEEP"].split(",")]
        child, how = None, "a process of its own, Longtail_Hip_SetBlockingWaits(1) first (tools/drop_in_child.py)"
        if os.environ.get("LTHIP_BENCH_DROPIN_INPROCESS") != "1":
            import subprocess

            req = {"cfg": {k: cfg[k] for k in ("tree", "kind", "codec", "file_mib", "gib") if k in cfg} | {"dups": bool(cfg.get("dups"))},
                   "sample_bytes": int(nbytes if cfg["tree"] != "files" else max(nbytes, 1)), "workers": w, "more_workers": more, "reps": self.REPS,
                   "target_chunk_size": self.args.target_chunk_size, "block_size": self.args.block_size, "max_chunks_per_block": self.args.max_chunks_per_block}
            req["sample_bytes"] = self._sample_request_bytes
            try:
                res = subprocess.run([sys.executable, str(ROOT / "tools" / "drop_in_child.py"), json.dumps(req)], capture_output=True, text=True, timeout=240,
                                     env={k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "ROCPROF"))})
                lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
                if res.returncode == 0 and lines:
                    child = json.loads(lines[-1])
                    if "hip" not in child or child.get("nbytes") != nbytes:
                        child = None
            except Exception:
                child = None
        if child is None:
            how = "in this process, the runtime's default (polling) waits"
            child = self._drop_in_here(tree, files, nbytes, cfg, common, more)
            if "error" in child:
                return child
        h = child["hip"][str(w)]
        payload = child["hip_stored_bytes"] - 8 * child["hip_blocks"]
        out = {"what": "the unmodified reference core (oracle/_ref) with Longtail_CreateHipChunkerAPI + HipBlake3HashAPI + Hip"
                       f"{cfg['codec'].upper()}CompressionAPI against its own CPU plugins: same files, same storage, W = {w}, median of {self.REPS}",
               "measured": how, "blocking_waits_rc": child.get("blocking_waits_rc"),
               "workers": w, "sample_files": len(files),
               "upsync_GBps": {"hip_plugins": h["GBps"], "cpu_plugins": cpu["GBps"], "ratio": round(h["GBps"] / cpu["GBps"], 3)},
               "create_version_index_GBps": {"hip_plugins": round(nbytes / h["index_s"] / 1e9, 3), "cpu_plugins": round(nbytes / cpu["index_s"] / 1e9, 3),
                                             "ratio": round(cpu["index_s"] / h["index_s"], 3)},
               "write_content_GBps": {"hip_plugins": round(nbytes / h["write_s"] / 1e9, 3), "cpu_plugins": round(nbytes / cpu["write_s"] / 1e9, 3),
                                      "ratio": round(cpu["write_s"] / h["write_s"], 3)},
               "ratio_hip_codec": round(child["hip_raw_bytes"] / payload, 4) if payload > 0 else None, "ratio_reference_codec": ref_ratio,
               "seconds": {"hip_plugins": h, "cpu_plugins": cpu}}
        if len(child["hip"]) > 1:
            # the HIP plugins' callers WAIT (for the link, for a submission): the same three objects with more workers than the CPU plugins can use
            out["hip_plugins_by_workers"] = {k: {"upsync_GBps": v["GBps"], "index_GBps": round(nbytes / v["index_s"] / 1e9, 3), "write_GBps": round(nbytes / v["write_s"] / 1e9, 3)}
                                             for k, v in child["hip"].items()}
        m = child.get("hip_chunker_hash_cpu_codec")
        if m:  # what INTEGRATION.md recommends where one block per Compress call does not feed a GPU codec: HIP chunker + hash, CPU codec
            out["upsync_GBps"]["hip_chunker_hash_cpu_codec"] = m["GBps"]
            out["upsync_GBps"]["ratio_hip_chunker_hash_cpu_codec"] = round(m["GBps"] / cpu["GBps"], 3)
            out["seconds"]["hip_chunker_hash_cpu_codec"] = m
        return out

    def _drop_in_here(self, tree, files, nbytes, cfg, common, more):
        r, w = self.r, self.w
        chunker, hasher, codec_api = self.plugins(cfg["codec"])
        r.version_index(files[: min(256, len(files))] if len(files) > 1 else [(files[0][0], files[0][1][: 256 << 20])],
                        self.args.target_chunk_size, w, 0, chunker, hasher)  # warm-up: contexts, window pool
        ws = [w] + list(more)
        hip = r.ingest_sweep_tree(tree, *common, ws, self.REPS, chunker, hasher, codec_api)
        if hip["err"]:
            return {"error": f"errno {hip['err']}"}
        out = {"hip": self._median(hip, ws, nbytes), "hip_raw_bytes": hip["raw_bytes"], "hip_stored_bytes": hip["stored_bytes"], "hip_blocks": hip["blocks"],
               "blocking_waits_rc": None}
        mixed = r.ingest_sweep_tree(tree, *common, [w], self.REPS, chunker, hasher, None)
        if not mixed["err"]:
            out["hip_chunker_hash_cpu_codec"] = self._median(mixed, [w], nbytes)[str(w)]
        return out

    def configs0(self, sweep):
        """BASELINE.json configs[0]: one 256 MiB random file (only 4 non-empty jobs)."""
        if not self.have_ref:
            return None
        r, args = self.r, self.args
        files, nbytes = self.sample_files(dict(tree="files", kind="random", codec="lz4", file_mib=256.0, gib=0.25), 268_435_456)
        res = r.ingest_sweep(files, args.target_chunk_size, args.block_size, args.max_chunks_per_block, r.lz4_type, sweep, self.REPS)
        return None if res["err"] else self._median(res, sweep, nbytes)

    def port_leg(self, cfg):
        from tests._libs import IngestResult

        o, args = self.o, self.args
        file_bytes = int(cfg["file_mib"] * (1 << 20))
        file_bytes = min(file_bytes - file_bytes % 16, 1 << 20)
        seeds = asset_seeds(0x10C0FFEE, 0, 64)
        blob = np.concatenate([o.synth(file_bytes, int(seeds[i]), KINDS[cfg["kind"]]) for i in range(64)])
        out = IngestResult()
        err = o.dll.lto_ingest(blob.ctypes.data, len(blob), file_bytes, args.target_chunk_size, args.block_size, 1, out)
        secs = out.seconds_chunk + out.seconds_hash + out.seconds_compress
        return {"value": round(len(blob) / secs / 1e9, 3) if not err else None, "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": f"64 x {file_bytes} B files, single-thread C restatement (oracle/), LZ4", "host_cpus": self.ncpu}


if __name__ == "__main__":
    main()
