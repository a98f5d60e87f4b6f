#!/usr/bin/env python
"""Decode throughput of the HIP LZ4 / ZStd block decoders on device-resident payloads (8 MiB blocks).
usage: tools/decode_rate.py [gib] [kind]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import asset_seeds
from longtail_amd.lib import Context

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
kind = {"random": 0, "mixed": 1, "records": 11, "tokens": 12, "lines": 13}[sys.argv[2] if len(sys.argv) > 2 else "mixed"]
ctx = Context(0)
FILE, BLOCK = 1 << 20, 8 << 20
nfiles = int(gib * (1 << 30)) // FILE
data = torch.empty(nfiles * FILE + 256, dtype=torch.uint8, device="cuda")
ctx.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE), np.full(nfiles, FILE, np.uint64), asset_seeds(1, 0, nfiles), kind)
n = nfiles * FILE
nb = n // BLOCK
b_off = np.arange(nb, dtype=np.int64) * BLOCK
b_size = np.full(nb, BLOCK, np.int64)
for name, comp, dec, bound in (("lz4", ctx.lz4_compress_blocks, ctx.lz4_decompress_blocks, b_size + b_size // 255 + 16),
                               ("zstd", ctx.zstd_compress_blocks, ctx.zstd_decompress_blocks, b_size + (b_size >> 8) + 64)):
    d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
    arena = torch.empty(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    sz = comp(data, b_off, b_size, arena, d_offs, bound).cpu().numpy().view(np.uint32).astype(np.int64)
    ctx.sync()
    for rep in range(2):
        t0 = time.perf_counter()
        out = dec(arena, d_offs, sz, back, b_off, b_size)
        ctx.sync()
        t = time.perf_counter() - t0
    ok = bool((out.cpu().numpy().view(np.uint32) == b_size).all()) and torch.equal(back[:n], data[:n])
    print(f"{name}: {nb} blocks of 8 MiB, ratio {n / sz.sum():.3f}, decode {t * 1e3:.1f} ms = {n / t / 1e9:.1f} GB/s of output, round trip {'ok' if ok else 'MISMATCH'}")
