"""What the N-rank step adds to a single-GPU step, priced on ONE GPU (no 8-GPU node has been available: SCALE_r01..r04 are skipped).

Three legs, one JSON object (profiles/r05_exchange_cost.json):

  flow8   the real flow with R = 8 processes on the one GPU: `bench.py --gpus 8 --launch plain` over the shared-memory stand-in for
          RCCL (LTHIP_COMM_TRANSPORT=shm), 8 x --gib of the headline tree with --exchange-profile (exchange + index split into
          host / device / transport; the "transport" here is the host-staged stand-in, NOT xGMI).  Eight contexts time-slice one GPU, so device and transport figures of this leg say
          "the flow runs at R = 8 and where its time goes", not what 8 GPUs would do.
  full    ONE rank's exchange + index at the FULL weak-scaling size (8 x 64 GiB: 524 288 jobs, ~17.3 M chunks, 2.16 M of them this
          rank's) with the collectives replaced by local copies of the right sizes (a loopback communicator): the host part and the
          device part of the N = 8 step, each at its real size, nothing extrapolated.  Chunk lists are synthetic (33 chunks per job,
          random digests): the exchange never looks at asset bytes.
  vi      rank 0's serialized VersionIndex of the 8 x 64 GiB tree (lthip_ingest_index + lthip_ingest_finish without a write): what the
          O(N) part of the index costs, and how much of it the other ranks wait for.

Transport at N = 8 is then priced from the bytes (bench.py --dry-run) and xGMI's per-link rate; DESIGN.md §7 holds the arithmetic.

    python tools/exchange_cost.py [--legs flow8,full,vi] [--gib 8] [--out profiles/r05_exchange_cost.json]
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def leg_flow8(gib: float, world: int):
    env = dict(os.environ, PYTHONPATH=str(ROOT), LTHIP_COMM_TRANSPORT="shm")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    base = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--launch", "plain", "--gib", str(gib), "--batch-gib", "2", "--steps", "2",
            "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic"]
    out = {"note": "only the PROFILED flow is run: unprofiled, eight processes contend for one GPU's hardware queues out of step, and every "
                   "host-staged copy of the stand-in transport waits for a time slice behind another process's kernels (round 5's first "
                   "run: 45 s of 'exchange' per step against 0.28 s in step) -- an artefact of one GPU standing in for eight, not of the flow"}
    for name, extra in (("profiled", ["--exchange-profile"]),):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            out[name] = {"error": r.stderr[-1500:]}
            continue
        j = json.loads(lines[-1])
        out[name] = {"n_ranks": j["n_gpus"], "comm": j["config"]["comm"], "tree_bytes": j["config"]["tree_bytes"], "jobs": j["config"]["jobs"],
                     "chunks": j["result"]["chunks"], "unique_chunks": j["result"]["unique_chunks"], "ms_per_step": j["ms_per_step"],
                     "phase_ms": j["phase_ms"], "exchange_profile": j.get("exchange_profile")}
    return out


class Loopback:
    """A communicator of `nranks` whose peers are copies of this rank: every collective returns tensors of the sizes the real one would
    (the all-gather tiles the contribution -- digests get the source rank mixed in so that the tree has nranks x as many distinct
    chunks --, the all-to-all hands back as many elements as the counts say).  Costs a device copy where RCCL would move the bytes."""

    def __init__(self, ctx, nranks):
        self.ctx, self.nranks, self.rank = ctx, nranks, 0
        self.bytes = {"allgather": 0, "alltoallv": 0}

    def sync(self):
        self.ctx.sync()

    def allgather(self, send, recv=None):
        import torch

        out = send.contiguous().repeat(self.nranks)
        if send.dtype == torch.int64 and send.numel() > 4096:
            v = out.view(self.nranks, -1)
            v ^= (torch.arange(self.nranks, device=send.device, dtype=torch.int64) * 0x9E3779B97F4A7C1).view(-1, 1)
        self.bytes["allgather"] += out.numel() * out.element_size()
        return out

    def alltoallv(self, send, send_counts, recv_counts, recv=None):
        n = int(sum(recv_counts))
        reps = -(-n // max(1, send.numel()))
        out = send.repeat(max(1, reps))[:n].contiguous()
        self.bytes["alltoallv"] += n * send.element_size()
        return out


def leg_full(world: int, gib_per_rank: float, steps: int = 5):
    import torch

    import bench as B
    from longtail_amd.dist import JobPartition, StepProfile, exchange_chunks, sharded_first_seen
    from longtail_amd.lib import Context, Ingest, load

    lib = load()
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    file_bytes = 1 << 20
    tree = B.make_tree("files", int(gib_per_rank * (1 << 30)) * world, file_bytes)
    t0 = time.perf_counter()
    part = JobPartition(tree["sizes"], 65536, world, "range", lib)
    t_part = (time.perf_counter() - t0) * 1e3
    mine = part.jobs_of(0)
    per_job = 33
    total = per_job * len(mine)
    g = torch.Generator(device=dev).manual_seed(5)
    out_hash = torch.randint(-(1 << 62), 1 << 62, (total,), dtype=torch.int64, device=dev, generator=g)
    out_lens = torch.randint(8192, 65536, (total,), dtype=torch.int32, device=dev, generator=g)
    out_offs = torch.cumsum(out_lens.to(torch.int64), 0) - out_lens
    out_first = (torch.arange(len(mine) + 1, dtype=torch.int32, device=dev) * per_job).contiguous()
    counts = out_first[1:] - out_first[:-1]
    comm = Loopback(ctx, world)
    est_chunks = total * world
    vi_cap = int(lib.dll.lthip_version_index_size(tree["nfiles"], est_chunks, est_chunks, len(tree["path_data"]))) + 64
    h_vi = torch.empty(vi_cap, dtype=torch.uint8).pin_memory()
    h_si = torch.empty(16 + 32 * total + 64, dtype=torch.uint8).pin_memory()
    ing = Ingest(ctx, 65536, 8 << 20, 1024, "lz4")
    acc = {"host": 0.0, "device": 0.0, "transport(loopback copies)": 0.0}
    detail = {}
    plain_ms, index_ms, finish_ms, tree_ms = [], [], [], []
    res = None
    for it in range(steps + 1):
        # (a) plain: no marks, one wait at the end -- what the step pays
        ctx.sync()
        t0 = time.perf_counter()
        ex = exchange_chunks(part, counts, out_hash, out_lens, total, ctx, comm=comm, rank=0)
        first_all, uniq_all = sharded_first_seen(part, ex, out_hash, total, ctx, comm=comm, rank=0)
        ctx.sync()
        t1 = time.perf_counter()
        # (b) profiled: the same with a wait at every mark
        prof = StepProfile(ctx)
        prof.start()
        ex = exchange_chunks(part, counts, out_hash, out_lens, total, ctx, comm=comm, rank=0, prof=prof)
        sharded_first_seen(part, ex, out_hash, total, ctx, comm=comm, rank=0, prof=prof)
        # (c) the index with a VALID first-seen array of the gathered lists (the loopback's answers are not one), rank 0 = the rank that
        # serializes the VersionIndex
        n_all = int(ex["job_first"][-1])
        valid_first, uniq = ctx.dedup_first_seen(ex["hashes"])
        ctx.sync()
        uniq = int(uniq.item())
        t2 = time.perf_counter()
        tr, keep = Ingest.tree(tree["sizes"], tree["path_offsets"], tree["perms"], tree["path_data"], part.job_asset, ex["job_first"].astype(np.uint64), mine)
        t3 = time.perf_counter()
        ing.set_first_seen(valid_first, uniq)
        ing.index(tr, ex["hashes"], ex["lens"], n_all, out_offs, out_first, total, h_vi)
        ctx.sync()
        t4 = time.perf_counter()
        res = ing.finish(h_si)
        t5 = time.perf_counter()
        if it == 0:
            continue  # warm-up: allocations
        plain_ms.append((t1 - t0) * 1e3)
        tree_ms.append((t3 - t2) * 1e3)
        index_ms.append((t4 - t3) * 1e3)
        finish_ms.append((t5 - t4) * 1e3)
        for k, v in prof.ms.items():
            acc["transport(loopback copies)" if k == "transport" else k] += v
        for k, v in prof.detail.items():
            detail[k] = detail.get(k, 0.0) + v
    # the same index WITHOUT the VersionIndex (what ranks != 0 do)
    other_ms = []
    for it in range(3):
        ing.set_first_seen(valid_first, uniq)
        ctx.sync()
        t0 = time.perf_counter()
        ing.index(tr, ex["hashes"], ex["lens"], n_all, out_offs, out_first, total, None)
        ctx.sync()
        ing.finish(h_si)
        other_ms.append((time.perf_counter() - t0) * 1e3)
    med = lambda a: round(float(np.median(a)), 3)
    ing.close()
    return {
        "tree": f"{world} x {gib_per_rank:g} GiB of 1 MiB files", "jobs": int(part.job_count), "jobs_this_rank": int(len(mine)),
        "chunks_all": int(n_all), "chunks_this_rank": int(total), "unique_all": int(uniq),
        "job_partition_ms_once": round(t_part, 2),
        "exchange_plus_first_seen_ms": {"plain_median": med(plain_ms), "plain_min": round(min(plain_ms), 3),
                                        "profiled_host": round(acc["host"] / steps, 3), "profiled_device": round(acc["device"] / steps, 3),
                                        "profiled_loopback_copies": round(acc["transport(loopback copies)"] / steps, 3),
                                        "detail": {k: round(v / steps, 3) for k, v in detail.items()}},
        "collective_bytes_received_per_step": comm.bytes and {k: int(v / (2 * (steps + 1))) for k, v in comm.bytes.items()},
        "index_ms": {"host_tree_tables": med(tree_ms), "rank0_index_call": med(index_ms), "rank0_finish_joins_version_index": med(finish_ms),
                     "other_ranks_index_plus_finish": med(other_ms), "version_index_bytes": int(res.version_index_size)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default="flow8,full")
    ap.add_argument("--gib", type=float, default=8.0, help="flow8: GiB per rank")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--full-gib", type=float, default=64.0, help="full: GiB per rank of the emulated tree")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    out = {"_about": "tools/exchange_cost.py: the N-rank step's exchange + index priced on one GPU (see the tool's docstring)"}
    legs = args.legs.split(",")
    if "full" in legs:
        out["full"] = leg_full(args.world, args.full_gib)
    if "flow8" in legs:
        out["flow8"] = leg_flow8(args.gib, args.world)
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
