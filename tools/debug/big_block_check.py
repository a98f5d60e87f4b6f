#!/usr/bin/env python
"""One block above 1 GiB through the LZ4 codec (the 64-bit index flavour of the decoders): round trip on the device."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from bench import asset_seeds
from longtail_amd.lib import Context
ctx = Context(0)
n = (1 << 30) + (37 << 20) + 12345
FILE = 1 << 20
nf = (n + FILE - 1) // FILE
data = torch.empty(nf * FILE + 256, dtype=torch.uint8, device="cuda")
ctx.synth_fill(data, np.arange(nf, dtype=np.uint64) * np.uint64(FILE), np.full(nf, FILE, np.uint64), asset_seeds(1, 0, nf), 1)
bound = n + n // 255 + 16
arena = torch.empty(bound + 64, dtype=torch.uint8, device="cuda")
sz = int(ctx.lz4_compress_blocks(data, [0], [n], arena, [0], [bound]).cpu().numpy().view(np.uint32)[0])
back = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
t0 = time.perf_counter()
out = int(ctx.lz4_decompress_blocks(arena, [0], [sz], back, [0], [n]).cpu().numpy().view(np.uint32)[0])
ctx.sync()
print(f"block {n} bytes -> {sz}; decoded {out} in {(time.perf_counter() - t0) * 1e3:.1f} ms;", "ok" if out == n and torch.equal(back[:n], data[:n]) else "MISMATCH")
# the same block through the zstd codec (8 488 pieces in one frame: more than one round of the piece decoders)
zb = n + (n >> 8) + 64
arena = None
arena = torch.empty(zb + 64, dtype=torch.uint8, device="cuda")
sz = int(ctx.zstd_compress_blocks(data, [0], [n], arena, [0], [zb]).cpu().numpy().view(np.uint32)[0])
back.zero_()
t0 = time.perf_counter()
out = int(ctx.zstd_decompress_blocks(arena, [0], [sz], back, [0], [n]).cpu().numpy().view(np.uint32)[0])
ctx.sync()
print(f"zstd: block {n} bytes -> {sz}; decoded {out} in {(time.perf_counter() - t0) * 1e3:.1f} ms;", "ok" if out == n and torch.equal(back[:n], data[:n]) else "MISMATCH", ctx.zstd_last_decode_stats())
