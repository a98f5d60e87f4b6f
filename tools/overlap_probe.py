#!/usr/bin/env python
"""Would chunk+hash (VALU bound) and the LZ4 match finder (latency / HBM bound) overlap if they ran on two streams?
Two contexts with private streams, one host thread each: chunk_hash over half A of a tree, lz4 over half B; wall time of
both together against each alone.  usage: tools/overlap_probe.py [gib_per_half] [kind]"""
import sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import asset_seeds
from longtail_amd.lib import Context, chunker_params

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = {"random": 0, "mixed": 1}[sys.argv[2] if len(sys.argv) > 2 else "random"]
FILE, BLOCK = 1 << 20, 8 << 20
nfiles = int(gib * (1 << 30)) // FILE
a, b = Context(0, stream=None), Context(0, stream=None)
halves = []
for h in range(2):
    d = torch.empty(nfiles * FILE + 256, dtype=torch.uint8, device="cuda")
    a.synth_fill(d, np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE), np.full(nfiles, FILE, np.uint64), asset_seeds(7 + h, 0, nfiles), kind)
    halves.append(d)
a.sync()
mn, av, mx = chunker_params(65536)
offs = np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE)
sizes = np.full(nfiles, FILE, np.uint64)
n = nfiles * FILE
nb = n // BLOCK
b_off = np.arange(nb, dtype=np.int64) * BLOCK
b_size = np.full(nb, BLOCK, np.int64)
bound = b_size + b_size // 255 + 16
d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
arena = torch.empty(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")

def hash_half():
    plan = a.make_plan(offs, sizes, mn, av, mx)
    a.chunk_hash(plan, halves[0])
    plan.close()
    a.sync()

def lz4_half():
    b.lz4_compress_blocks(halves[1], b_off, b_size, arena, d_offs, bound)
    b.sync()

def timed(fns):
    ts = [threading.Thread(target=f) for f in fns]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return (time.perf_counter() - t0) * 1e3

for rep in range(3):
    th, tl, tb = timed([hash_half]), timed([lz4_half]), timed([hash_half, lz4_half])
    print(f"{gib:g} GiB halves, kind {kind}: chunk+hash {th:.2f} ms, lz4 {tl:.2f} ms, both concurrently {tb:.2f} ms (sum {th + tl:.2f}, max {max(th, tl):.2f})")
