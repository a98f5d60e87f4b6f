#!/usr/bin/env python3
"""Is the LZ4 / zstd payload of a block a function of the block alone?  Compresses the same blocks several times (alone, in another
order of the call's block list, with tickets and with the fixed stride) and compares the payload bytes.
python tools/k5_determinism.py [gib] [codec] [ragged|-] [zstd quality 0..2]"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import KINDS, asset_seeds  # noqa: E402
from longtail_amd.lib import Context  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
codec = sys.argv[2] if len(sys.argv) > 2 else "lz4"
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"  # block sizes that are no multiples of the 64 KiB groups (what stores hold)
ctx = Context(0)
BLOCK = 8 << 20
n = int(gib * (1 << 30)) // BLOCK * BLOCK
nb = n // BLOCK
data = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
nfiles = n // (1 << 20)
ctx.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(1 << 20), np.full(nfiles, 1 << 20, np.uint64), asset_seeds(0xBEEF, 0, nfiles), KINDS["mixed"])
ctx.sync()
b_size = np.full(nb, BLOCK, np.int64)
if ragged:
    b_size = BLOCK - np.random.default_rng(5).integers(1, 700000, nb).astype(np.int64)
full = np.full(nb, BLOCK, np.int64)
bound = full + full // 255 + 16 if codec == "lz4" else full + (full >> 8) + 64  # (of a whole block: the lists are permuted)
d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
quality = int(sys.argv[4]) if len(sys.argv) > 4 else 0
fn = ctx.lz4_compress_blocks if codec == "lz4" else (lambda *a: ctx.zstd_compress_blocks(*a, quality=quality))


def run(order, dbg):
    os.environ["LTHIP_LZ4_DBG"] = str(dbg)
    ctx.lib.dll.lthip_debug_reload_env()
    arena = torch.zeros(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    b_off = np.asarray(order, np.int64) * BLOCK
    sizes = fn(data, b_off, b_size[np.asarray(order)], arena, d_offs[: len(order)], bound[: len(order)])
    ctx.sync()
    sz = sizes.cpu().numpy().view(np.uint32).astype(np.int64)
    host = arena.cpu().numpy()
    return {int(b): bytes(host[int(d_offs[i]) : int(d_offs[i]) + int(sz[i])]) for i, b in enumerate(order)}


ident = list(range(nb))
base = run(ident, 0)
for name, order, dbg in [("again", ident, 0), ("reversed block list", ident[::-1], 0), ("fixed stride", ident, 1 << 27),
                         ("fixed stride, reversed", ident[::-1], 1 << 27), ("first half only", ident[: nb // 2], 0)]:
    other = run(order, dbg)
    diff = [b for b in other if other[b] != base[b]]
    print(f"{codec}{' q' + str(quality) if quality else ''}{' ragged' if ragged else ''} {name}: {len(diff)} of {len(other)} payloads differ" + (f" (first: block {diff[0]}, {len(base[diff[0]])} vs {len(other[diff[0]])} bytes)" if diff else ""))
