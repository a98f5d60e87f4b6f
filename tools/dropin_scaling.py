#!/usr/bin/env python
"""How the drop-in path scales with the job system's worker count, and what the workers do while they wait: the unmodified reference
core with the HIP chunker + hash (+ codec) on one tmpfs tree, W swept; per W the three phases' rates, the process's CPU seconds per wall
second (threads that spin show up here, threads that sleep do not) and the batchers' windows / blocks per submission.
usage: tools/dropin_scaling.py [kind] [codec] [gib] [W,W,...]"""
import ctypes as C
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "random"
codec = sys.argv[2] if len(sys.argv) > 2 else "lz4"
gib = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
ws = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "8,32,64,128").split(",")]
args = bench.make_parser().parse_args(["--gib", str(max(gib, 1.0)), "--kind", kind, "--codec", codec, "--no-secondary", "--no-live-traffic"])
b = bench.Bench(args)
if os.environ.get("SCHED_FLAGS"):  # experiment: hipSetDeviceFlags(1 spin / 2 yield / 4 blocking sync) on the runtime torch has loaded
    path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
    rc = C.CDLL(path).hipSetDeviceFlags(int(os.environ["SCHED_FLAGS"]))
    print("hipSetDeviceFlags", os.environ["SCHED_FLAGS"], "->", rc, path)
cr = bench.CpuReference(b, args)
cfg = dict(tree="files", kind=kind, codec=codec, gib=gib, file_mib=1.0)
files, nbytes = cr.sample_files(cfg, int(gib * (1 << 30)))
r = cr.r
tag = r.lz4_type if codec == "lz4" else r.zstd_default
tree = r.tree_create(files, tag)
chunker, hasher, codec_api = cr.plugins(codec)
d = b.lib.dll
common = (args.target_chunk_size, args.block_size, args.max_chunks_per_block, tag)


def stats():
    a, c, e, f = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    d.Longtail_Hip_BatchStats(C.byref(a), C.byref(c))
    d.Longtail_Hip_CodecBatchStats.argtypes = [C.c_void_p, C.c_void_p]
    d.Longtail_Hip_CodecBatchStats(C.byref(e), C.byref(f))
    return a.value, c.value, e.value, f.value


print(f"{len(files)} files, {nbytes / (1 << 30):.1f} GiB, kind {kind}, codec {codec}; host threads {os.cpu_count()}")
for who, apis in (("hip plugins", (chunker, hasher, codec_api)), ("cpu plugins", (None, None, None))):
    for w in ws:
        s0, t0, c0 = stats(), time.perf_counter(), os.times()
        res = r.ingest_sweep_tree(tree, *common, [w], 3, *apis)
        s1, t1, c1 = stats(), time.perf_counter(), os.times()
        m = cr._median(res, [w], nbytes)[str(w)]
        cpu = (c1.user - c0.user) + (c1.system - c0.system)
        sub_w, win, sub_c, blk = (s1[i] - s0[i] for i in range(4))
        print(f"{who} W={w:3d}: upsync {m['GBps']:6.2f} GB/s  index {nbytes / m['index_s'] / 1e9:6.2f}  write {nbytes / m['write_s'] / 1e9:6.2f} | cpu-seconds per wall-second {cpu / (t1 - t0):6.1f}"
              f" (user {c1.user - c0.user:.1f} sys {c1.system - c0.system:.1f}) | windows per submission {win / max(1, sub_w):5.1f}  blocks per codec submission {blk / max(1, sub_c):5.1f}", flush=True)
r.tree_destroy(tree)
cr.close()
