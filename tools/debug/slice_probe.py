#!/usr/bin/env python
"""What would a slice-wise pipeline buy?  Hash (BLAKE3 ranges) then LZ4-compress the SAME slice while it may still sit in the
256 MiB memory-side cache, slice after slice, against one hash pass + one codec pass over everything.  Kernel times from the
library's own HIP-event timers, so the extra launches of the slice loop do not count.  usage: tools/debug/slice_probe.py [gib]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from bench import asset_seeds
from longtail_amd.lib import Context

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Context(0)
FILE, BLOCK, RANGE = 1 << 20, 8 << 20, 32 << 10
nfiles = int(gib * (1 << 30)) // FILE
n = nfiles * FILE
data = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
ctx.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE), np.full(nfiles, FILE, np.uint64), asset_seeds(3, 0, nfiles), 0)
r_off = torch.arange(0, n, RANGE, dtype=torch.int64, device="cuda")
r_len = torch.full((len(r_off),), RANGE, dtype=torch.int32, device="cuda")
nb = n // BLOCK
b_off = np.arange(nb, dtype=np.int64) * BLOCK
b_size = np.full(nb, BLOCK, np.int64)
bound = b_size + b_size // 255 + 16
d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
arena = torch.empty(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")

def run(slice_mib):
    ctx.timing(True); ctx.timing_reset()
    if slice_mib == 0:
        ctx.hash_ranges(data, r_off, r_len, RANGE)
        ctx.lz4_compress_blocks(data, b_off, b_size, arena, d_offs, bound)
    else:
        bs = max(1, (slice_mib << 20) // BLOCK)            # blocks per slice
        rs = bs * BLOCK // RANGE                            # ranges per slice
        for i in range(0, nb, bs):
            j = min(nb, i + bs)
            ctx.hash_ranges(data, r_off[i * BLOCK // RANGE : i * BLOCK // RANGE + (j - i) * BLOCK // RANGE], r_len[: (j - i) * BLOCK // RANGE], RANGE)
            ctx.lz4_compress_blocks(data, b_off[i:j], b_size[i:j], arena, d_offs[i:j], bound[i:j])
    ctx.sync()
    t = ctx.timing_get()
    return {k: round(v[0], 2) for k, v in t.items() if v[0] > 0.005}

run(0)
for s in (0, 8, 16, 32, 64, 128, 256, 0):
    print(f"slice {s:4d} MiB" if s else "one pass each ", run(s))
