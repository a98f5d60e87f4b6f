#!/usr/bin/env python
"""What the three zstd parses cost on the RESTORE side: ratio and decode rate of this library's own frames at the default / high / max
setting, 8 MiB blocks of one kind, 512 / 64 / 1 blocks per call (frames of the "max" setting carry history across their 128 KiB pieces:
trailer version 4, the pieces are one chain for the decoder).  usage: tools/zstd_quality_rates.py [kinds, comma separated]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import asset_seeds
from longtail_amd.lib import Context

kinds = (sys.argv[1] if len(sys.argv) > 1 else "mixed,tokens,records").split(",")
KIND = {"random": 0, "mixed": 1, "records": 11, "tokens": 12, "lines": 13}
ctx = Context(0)
FILE, BLOCK = 1 << 20, 8 << 20
for kind in kinds:
    nfiles = 4096
    data = torch.empty(nfiles * FILE + 256, dtype=torch.uint8, device="cuda")
    ctx.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE), np.full(nfiles, FILE, np.uint64), asset_seeds(1, 0, nfiles), KIND[kind])
    n = nfiles * FILE
    nb = n // BLOCK
    b_off = np.arange(nb, dtype=np.int64) * BLOCK
    b_size = np.full(nb, BLOCK, np.int64)
    bound = b_size + (b_size >> 8) + 64
    d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
    arena = torch.empty(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    for q, qname in ((0, "default"), (1, "high"), (2, "max")):
        sz = ctx.zstd_compress_blocks(data, b_off, b_size, arena, d_offs, bound, quality=q).cpu().numpy().view(np.uint32).astype(np.int64)
        ctx.sync()
        line = f"{kind:8s} {qname:8s} ratio {n / sz.sum():6.3f}  decode:"
        for count in (512, 64, 1):
            best = None
            for rep in range(3):
                back.zero_()
                ctx.sync()
                t0 = time.perf_counter()
                out = ctx.zstd_decompress_blocks(arena, d_offs[:count], sz[:count], back, b_off[:count], b_size[:count])
                ctx.sync()
                t = time.perf_counter() - t0
                best = t if best is None or t < best else best
            ok = bool((out.cpu().numpy().view(np.uint32) == BLOCK).all()) and torch.equal(back[: count * BLOCK], data[: count * BLOCK])
            stats = ctx.zstd_last_decode_stats()
            line += f"  {count:3d} blocks {best * 1e3:7.2f} ms = {count * BLOCK / best / 1e9:6.1f} GB/s{'' if ok else ' MISMATCH'} (serial payloads {stats[2]})"
        print(line, flush=True)
