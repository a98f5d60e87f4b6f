#!/usr/bin/env python3
"""bench.py -- ingest throughput (chunk + BLAKE3 + LZ4) of the HIP hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2]): a synthetic tree of 1 MiB files, 64 GiB PER GPU (weak scaling:
rank r owns files [r*F, (r+1)*F) of the N*64 GiB tree), bytes from include/longtail_synth.h, already resident in HBM
when the timed region starts.  One step = the whole hot path over the rank's shard:
  plan -> buzhash candidate scan -> cut selection -> compaction -> BLAKE3 leaves/parents            (phase 1)
  -> [N>1: RCCL all-gather of per-rank chunk hashes] -> first-seen dedup                              (exchange)
  -> greedy block packing of unique chunks (src/longtail.c:6801-6860, 8 MiB x 1.1, <= 1024 chunks)
  -> per-block LZ4 (segments + stitch) into a bounded output arena                                    (phase 2)
value = bytes of all ranks / max-over-ranks wall time of K steps (barrier + synchronize on both sides).
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
MASK64 = (1 << 64) - 1


def synth_mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def asset_seeds(tree_seed: int, first: int, count: int) -> np.ndarray:
    """lt_synth_asset_seed (include/longtail_synth.h) vectorised."""
    with np.errstate(over="ignore"):
        idx = np.arange(first, first + count, dtype=np.uint64) + np.uint64(1)
        return synth_mix(np.uint64(tree_seed) + np.uint64(0x9E3779B97F4A7C15) * idx)


def pack_blocks(lens: np.ndarray, max_block: int, max_chunks: int):
    """Greedy packing of Longtail_CreateStoreIndex (src/longtail.c:6801-6860): returns block start indices (+ end)."""
    n = len(lens)
    cs = np.cumsum(lens, dtype=np.int64)
    limit = max_block + max_block // 10
    starts = []
    i = 0
    while i < n:
        base = int(cs[i - 1]) if i else 0
        j = int(np.searchsorted(cs, base + limit, side="right"))
        j = max(min(j, i + max_chunks), i + 1)
        starts.append(i)
        i = j
    starts.append(n)
    return np.asarray(starts, dtype=np.int64), cs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=64.0, help="GiB of assets per GPU")
    ap.add_argument("--file-mib", type=float, default=1.0)
    ap.add_argument("--tree", choices=["files", "mixed-sizes"], default="files", help="files: equal files (configs[2]); mixed-sizes: 4 KiB..4 GiB log-uniform")
    ap.add_argument("--kind", choices=["random", "mixed", "zero", "records", "tokens", "lines"], default="random")
    ap.add_argument("--target-chunk-size", type=int, default=65536)
    ap.add_argument("--block-size", type=int, default=8 << 20)
    ap.add_argument("--max-chunks-per-block", type=int, default=1024)
    ap.add_argument("--lz4-batch-gib", "--batch-gib", dest="lz4_batch_gib", type=float, default=8.0)
    ap.add_argument("--codec", choices=["lz4", "zstd"], default="lz4", help="block codec of phase 2 (BASELINE.json configs[4] uses zstd)")
    ap.add_argument("--segment-log2", type=int, default=0)
    ap.add_argument("--no-compress", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from longtail_amd.dist import allgather_hashes
    from longtail_amd.lib import Context, chunker_params, load
    from longtail_amd.lib import BatchPacker

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    lib = load()
    if not torch.cuda.is_available() or lib.device_count() == 0:
        raise SystemExit("bench.py needs a GPU and liblongtail_hip.so: there is no CPU fallback")
    # LONGTAIL_DIST_BACKEND=gloo runs the multi-rank flow with all ranks on the GPUs that exist (rank % device_count) and
    # the exchange staged through host memory: a functional check of the N>1 path on a 1-GPU box, not a measurement
    backend = os.environ.get("LONGTAIL_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    ctx = Context(dev_index)
    kind = {"random": 0, "mixed": 1, "zero": 2, "records": 11, "tokens": 12, "lines": 13}[args.kind]
    mn, av, mx = chunker_params(args.target_chunk_size)
    part_bytes = args.target_chunk_size * 1024  # ChunkAssets part size (src/longtail.c:2396)
    if args.tree == "files":
        file_bytes = int(args.file_mib * (1 << 20))
        file_bytes -= file_bytes % 16
        nfiles = max(1, int(args.gib * (1 << 30)) // file_bytes)
        file_sizes = np.full(nfiles, file_bytes, dtype=np.uint64)
    else:
        # north-star tree: sizes log-uniform in [4 KiB, 4 GiB] (capped at a quarter of the shard), fixed seed
        rng = np.random.default_rng(0xA55E7 + rank)
        budget = int(args.gib * (1 << 30))
        hi = min(4 << 30, max(budget // 4, 8192))
        sizes = []
        while budget > 0:
            sz = int(np.exp(rng.uniform(np.log(4096), np.log(hi))))
            sz = max(1, min(sz, budget))
            sizes.append(sz)
            budget -= sz
        file_sizes = np.asarray(sizes, dtype=np.uint64)
        nfiles = len(sizes)
        file_bytes = int(file_sizes.mean())
    file_offsets = np.zeros(nfiles, dtype=np.uint64)
    np.cumsum(((file_sizes + np.uint64(15)) // np.uint64(16) * np.uint64(16))[:-1], out=file_offsets[1:])
    shard_bytes = int(file_sizes.sum())
    arena_bytes = int(file_offsets[-1] + file_sizes[-1])
    # parts: every asset is cut into target*1024-byte segments, each chunked from a fresh state (:2396-2458)
    nparts_per = (file_sizes + np.uint64(part_bytes - 1)) // np.uint64(part_bytes)
    rep = np.repeat(np.arange(nfiles), nparts_per.astype(np.int64))
    within = np.arange(len(rep), dtype=np.uint64) - np.repeat(np.cumsum(nparts_per) - nparts_per, nparts_per.astype(np.int64))
    part_offsets = file_offsets[rep] + within * np.uint64(part_bytes)
    part_sizes = np.minimum(file_sizes[rep] - within * np.uint64(part_bytes), np.uint64(part_bytes))
    nparts = len(part_offsets)

    # ---- inputs resident in HBM (untimed) ----
    data = torch.empty(arena_bytes + 256, dtype=torch.uint8, device=dev)
    seeds = asset_seeds(0x10C0FFEE, rank * 10_000_000, nfiles)
    ctx.synth_fill(data, file_offsets, file_sizes, seeds, kind)
    ctx.sync()

    # ---- output arenas (allocated once; the hot path never allocates in steady state) ----
    probe = ctx.make_plan(part_offsets, part_sizes, mn, av, mx)
    cap = max(1, probe.capacity)
    probe.close()
    out_offs = torch.empty(cap, dtype=torch.int64, device=dev)
    out_lens = torch.empty(cap, dtype=torch.int32, device=dev)
    out_hash = torch.empty(cap, dtype=torch.int64, device=dev)
    out_first = torch.empty(nparts + 1, dtype=torch.int32, device=dev)
    batch_bytes = int(args.lz4_batch_gib * (1 << 30))
    limit = args.block_size + args.block_size // 10
    dst_arena_bytes = batch_bytes + batch_bytes // 255 + (batch_bytes // args.block_size + 2) * 64 + 2 * (limit + limit // 255 + 64)
    dst = torch.empty(dst_arena_bytes, dtype=torch.uint8, device=dev)
    gather_arena = None  # allocated on first use: only trees whose blocks are not contiguous ranges need it
    h_lens = torch.empty(cap, dtype=torch.int32).pin_memory()
    h_offs = torch.empty(cap, dtype=torch.int64).pin_memory()
    stats = {}

    def compress(src, s_offs, s_sizes, dst_t, d_offs, caps):
        if args.codec == "zstd":
            return ctx.zstd_compress_blocks(src, s_offs, s_sizes, dst_t, d_offs, caps)
        return ctx.lz4_compress_blocks(src, s_offs, s_sizes, dst_t, d_offs, caps, args.segment_log2)

    def step():
        t0 = time.perf_counter()
        plan = ctx.make_plan(part_offsets, part_sizes, mn, av, mx)
        total, _, _, _, _ = ctx.chunk_hash(plan, data, outputs=(out_offs, out_lens, out_hash, out_first), sync=True)
        plan.close()
        t1 = time.perf_counter()
        # ---- exchange + dedup (src/longtail.c:2951-2970) ----
        all_hashes, my_base, _counts = allgather_hashes(out_hash, total)
        mine_first, uniq = ctx.dedup_first_seen_range(all_hashes, my_base, total)  # table over all ranks, answers for mine
        unique_mask = mine_first == torch.arange(my_base, my_base + total, dtype=torch.int32, device=dev)
        n_unique_local = int(unique_mask.sum().item())
        t2 = time.perf_counter()
        comp_bytes = 0
        nblocks = 0
        if not args.no_compress:
            # chunk lists to pinned host memory (one async copy each, one sync), then the serial packing in C
            h_lens[:total].copy_(out_lens[:total], non_blocking=True)
            h_offs[:total].copy_(out_offs[:total], non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            lens_u32 = h_lens[:total].numpy().view(np.uint32)
            offs_h = h_offs[:total].numpy().view(np.int64)
            if n_unique_local != total:
                keep = unique_mask.cpu().numpy()
                lens_u32, offs_h = lens_u32[keep], offs_h[keep]
            # greedy packing, one batch at a time: batch k+1 is packed on the host while the device compresses batch k
            lz = args.codec == "lz4"
            packer = BatchPacker(lens_u32, args.block_size, args.max_chunks_per_block, batch_bytes, dst_arena_bytes,
                                 255 if lz else 256, 16 if lz else 64, lib)
            d_offs_u = d_lens_u = None
            size_tensors, b_size_all = [], []
            stats["gather"] = False
            while True:
                nxt = packer.next()
                if nxt is None:
                    break
                starts, b_size = nxt
                b_first, b_last = starts[:-1], starts[1:] - 1
                nblocks += len(b_first)
                bounds = b_size + b_size // 255 + 16 if lz else b_size + (b_size >> 8) + 64
                aligned = (bounds + 63) // 64 * 64
                d_offs = np.concatenate([[0], np.cumsum(aligned)[:-1]])
                is_range = (offs_h[b_last] + lens_u32[b_last]) - offs_h[b_first] == b_size  # block == one contiguous byte range
                if bool(is_range.all()):
                    b_size_all.append(b_size)
                    size_tensors.append(compress(data, offs_h[b_first], b_size, dst, d_offs, bounds))
                else:
                    # block assembly on the device (WriteContentBlockJob, src/longtail.c:4640-4721) for the blocks that
                    # are not one range (they span assets): gather their chunks back to back, then each of them is a
                    # contiguous range of the gather arena; the others are compressed where they lie
                    stats["gather"] = True
                    nonlocal gather_arena
                    if gather_arena is None:
                        gather_arena = torch.empty(batch_bytes + 2 * limit + 256, dtype=torch.uint8, device=dev)
                    if d_offs_u is None:
                        if n_unique_local != total:
                            d_offs_u, d_lens_u = out_offs[:total][unique_mask], out_lens[:total][unique_mask]
                        else:
                            d_offs_u, d_lens_u = out_offs[:total], out_lens[:total]
                    r, g = np.flatnonzero(is_range), np.flatnonzero(~is_range)
                    cnt = (starts[g + 1] - starts[g]).astype(np.int64)
                    first_of = np.repeat(starts[g] - (np.cumsum(cnt) - cnt), cnt)
                    chunk_idx = torch.from_numpy(first_of + np.arange(int(cnt.sum()), dtype=np.int64)).to(dev)
                    lens_d = d_lens_u[chunk_idx]
                    dst_off = torch.cumsum(lens_d.to(torch.int64), 0) - lens_d.to(torch.int64)
                    ctx.gather_ranges(data, d_offs_u[chunk_idx].contiguous(), lens_d.contiguous(), gather_arena, dst_off)
                    if len(r):
                        b_size_all.append(b_size[r])
                        size_tensors.append(compress(data, offs_h[b_first[r]], b_size[r], dst, d_offs[r], bounds[r]))
                    b_size_all.append(b_size[g])
                    size_tensors.append(compress(gather_arena, np.cumsum(b_size[g]) - b_size[g], b_size[g], dst, d_offs[g], bounds[g]))
            b_size = np.concatenate(b_size_all) if b_size_all else np.zeros(0, np.int64)
            sizes = (torch.cat(size_tensors).cpu().numpy().view(np.uint32).astype(np.int64) if size_tensors else np.zeros(0, np.int64))  # one D2H, waits for the codec
            comp_bytes = int(sizes.sum())
            # blocks without any match are laid out by the match finder itself (it writes their literals): count them
            stats["placed_by_matcher"] = int(b_size[sizes >= b_size].sum()) if args.codec == "lz4" else 0
            if int((sizes == 0).sum()) != 0:
                raise SystemExit("a block did not fit its bound: encoder bug")
        ctx.sync()
        t3 = time.perf_counter()
        stats.update(chunks=total, unique_local=n_unique_local, unique_global=int(uniq.item()), blocks=nblocks,
                     compressed_bytes=comp_bytes, t_phase1=t1 - t0, t_exchange=t2 - t1, t_phase2=t3 - t2)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    ctx.timing(True)
    ctx.timing_reset()
    phase = np.zeros(3)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        phase += [stats["t_phase1"], stats["t_exchange"], stats["t_phase2"]]
    barrier()
    elapsed = time.perf_counter() - t_start
    ktimes = ctx.timing_get()
    ctx.timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_bytes = shard_bytes * world
    value = total_bytes * args.steps / elapsed / 1e9

    # ---- roofline of the dominant kernel (algorithmic bytes / launch; DESIGN.md "Measurement") ----
    comp = stats["compressed_bytes"]
    alg_bytes = {
        "buzhash": shard_bytes,                 # N read once
        "blake3_leaf": shard_bytes,             # N read once
        "lz4_segments": shard_bytes + stats.get("placed_by_matcher", 0),  # N read + the blocks it lays out itself (all-literal)
        "zstd_encode": shard_bytes + comp,      # literals + sequences read, pieces written
    }
    # lz4_stitch has no fixed algorithmic figure any more: for blocks without matches it only writes a header
    kern = {}
    for name, (ms, n) in ktimes.items():
        if n:
            per_step_ms = ms / args.steps
            kern[name] = {"ms_per_step": round(per_step_ms, 3), "launches_per_step": n / args.steps}
            if name in alg_bytes:
                kern[name]["GBps"] = round(alg_bytes[name] / (per_step_ms * 1e-3) / 1e9, 1)
    dom = max((k for k in kern if k in alg_bytes), key=lambda k: kern[k]["ms_per_step"], default=None)
    roofline = None
    if dom:
        launches = kern[dom]["launches_per_step"]
        avg_ms = kern[dom]["ms_per_step"] / launches
        achieved = alg_bytes[dom] / launches / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": int(alg_bytes[dom] / launches), "avg_launch_ms": round(avg_ms, 3)}

    # HBM traffic from the PMC counters: collected offline (rocprofv3 cannot wrap itself), tools/pmc_traffic.sh ->
    # profiles/*pmc_traffic*.json; scaled by input bytes to this run's launch size
    if roofline:
        try:
            tfiles = sorted((ROOT / "profiles").glob("*pmc_traffic*.json"))
            tj = json.load(open(tfiles[-1]))
            names = {"buzhash": "k_buzhash_candidates<0>", "blake3_leaf": "k_blake3_leaves", "lz4_segments": "k_lz4_segments<",
                     "lz4_stitch": "k_lz4_stitch_copy"}
            if dom == "lz4_segments" and args.codec != "lz4":
                raise KeyError("no PMC pass for the sequence-output variant of the match finder")
            key = next(k for k in tj["kernels"] if k.startswith(names[dom]) and (dom != "lz4_segments" or k.endswith(", 0>")))
            ratio = tj["kernels"][key]["corrected_per_input_byte"]
            roofline["traffic"] = int(ratio * shard_bytes / kern[dom]["launches_per_step"])
            roofline["traffic_source"] = f"profiles/{tfiles[-1].name}: {ratio} HBM bytes per input byte (2*FETCH_SIZE+WRITE_SIZE)"
        except Exception:
            pass

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(args, kind, file_bytes)

    if rank == 0:
        line = {
            "metric": "ingest GB/s (chunk+hash+compress)",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/u32",
            "data": "synthetic",
            "config": {
                "workload": (f"{args.gib:g} GiB tree of {args.file_mib:g} MiB {args.kind} files per GPU, chunk+BLAKE3+{args.codec.upper()} "
                             f"(BASELINE.json configs[2]{'/[3]' if world > 1 else ''})") if args.tree == "files" else
                            (f"{args.gib:g} GiB tree of {nfiles} {args.kind} files, 4 KiB..4 GiB log-uniform, per GPU, chunk+BLAKE3+{args.codec.upper()} "
                             f"(north-star tree)"),
                "target_chunk_size": args.target_chunk_size, "min_avg_max": [mn, av, mx], "block_size": args.block_size,
                "max_chunks_per_block": args.max_chunks_per_block, "bytes_per_gpu": shard_bytes, "files_per_gpu": nfiles, "parts_per_gpu": nparts, "device_block_assembly": bool(stats.get("gather")),
                "sharding": "by file, RCCL all-gather of chunk hashes for dedup" if world > 1 else "single GPU",
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "kernels": kern,
            "phase_ms": {"chunk_hash": round(phase[0] / args.steps * 1e3, 2), "exchange_dedup": round(phase[1] / args.steps * 1e3, 2),
                         "pack_compress": round(phase[2] / args.steps * 1e3, 2)},
            "result": {"chunks": stats["chunks"], "unique_chunks_global": stats["unique_global"], "blocks": stats["blocks"],
                       "compressed_bytes": stats["compressed_bytes"],
                       "ratio": round(shard_bytes / stats["compressed_bytes"], 4) if stats["compressed_bytes"] else None},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def run_cpu_baseline(args, kind, file_bytes):
    """The reference's bikeshed-threaded CPU path (oracle/_ref) -- or the single-thread port (oracle/) when the
    reference build is not present -- on a bounded sample of the SAME workload, timed on this host's cores."""
    from tests._libs import have_ref, oracle, ref

    o = oracle()
    ncores = os.cpu_count() or 1
    target = args.cpu_seconds

    def make_files(n):
        seeds = asset_seeds(0x10C0FFEE, 0, n)
        return [(f"dir{i % 256:03d}/file{i:05d}.bin", o.synth(file_bytes, int(seeds[i]), kind)) for i in range(n)]

    if have_ref():
        r = ref()
        workers = int(r.dll.refh_cpu_count())
        # source tree on tmpfs through the reference's file storage ("page-cache-warm, from tmpfs"); its
        # in-memory storage serialises reads behind one lock and is only the fallback
        storage = "in-memory storage"
        shm = "/dev/shm"
        try:
            st = os.statvfs(shm)
            if os.access(shm, os.W_OK) and st.f_bavail * st.f_frsize > (20 << 30):
                r.dll.refh_set_tree_dir.argtypes = [ctypes.c_char_p]
                r.dll.refh_set_tree_dir(shm.encode())
                storage = f"file storage on tmpfs ({shm})"
        except OSError:
            pass
        n = 256
        best = None
        spent = 0.0
        while True:
            files = make_files(n)
            t0 = time.perf_counter()
            res = r.ingest_time(files, args.target_chunk_size, args.block_size, args.max_chunks_per_block, r.lz4_type, workers)
            wall = time.perf_counter() - t0
            if res["err"]:
                return {"error": res["err"]}
            secs = res["seconds_index"] + res["seconds_write"]
            best = (n, secs, res)
            spent += wall
            if secs * 4 > target / 2 or n * file_bytes >= (8 << 30) or spent > target:
                break
            n *= 4
        n, secs, res = best
        return {"value": round(n * file_bytes / secs / 1e9, 3), "unit": "GB/s", "cores": workers, "kind": "reference",
                "sample": f"{n} x {file_bytes} B files of the same tree; Longtail_CreateVersionIndex {res['seconds_index']:.3f} s + "
                          f"Longtail_WriteContent {res['seconds_write']:.3f} s (reference hpcdc+BLAKE3+LZ4, bikeshed {workers} workers, "
                          f"{storage}, null block sink)",
                "host_cpus": ncores}
    from tests._libs import IngestResult

    n = 64
    files = make_files(n)
    blob = np.concatenate([d for _, d in files])
    out = IngestResult()
    err = o.dll.lto_ingest(blob.ctypes.data, len(blob), file_bytes, args.target_chunk_size, args.block_size, 1, out)
    secs = out.seconds_chunk + out.seconds_hash + out.seconds_compress
    return {"value": round(len(blob) / secs / 1e9, 3) if not err else None, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{n} x {file_bytes} B files, single-thread C restatement (oracle/)", "host_cpus": ncores}


if __name__ == "__main__":
    main()
