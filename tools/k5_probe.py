#!/usr/bin/env python3
"""K5 experiment loop: LZ4-compress N GiB of each synthetic kind as 8 MiB blocks, print ratio and the match finder's time for a
list of LTHIP_LZ4_DBG settings (read per call).  python tools/k5_probe.py [gib] [dbg,dbg,...] [kinds]"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import KINDS, asset_seeds  # noqa: E402
from longtail_amd.lib import Context  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
dbgs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
kinds = (sys.argv[3] if len(sys.argv) > 3 else "mixed,records,tokens,lines,random").split(",")
codec = sys.argv[4] if len(sys.argv) > 4 else "lz4"
shift = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # blocks start `shift` bytes into the data (phase experiments)
ctx = Context(0)
BLOCK = 8 << 20
n = int(gib * (1 << 30)) // BLOCK * BLOCK
data = torch.empty(n + BLOCK + 256, dtype=torch.uint8, device="cuda")
nb = n // BLOCK
b_off = np.arange(nb, dtype=np.int64) * BLOCK + shift
b_size = np.full(nb, BLOCK, np.int64)
bound = b_size + b_size // 255 + 16 if codec == "lz4" else b_size + (b_size >> 8) + 64
d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
arena = torch.empty(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
print(f"parser={os.environ.get('LTHIP_LZ4_PARSER', 'lanes')} {gib} GiB, {codec}")
for kind in kinds:
    nfiles = (n + BLOCK) // (1 << 20)
    ctx.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(1 << 20), np.full(nfiles, 1 << 20, np.uint64), asset_seeds(0xBEEF, 0, nfiles), KINDS[kind])
    ctx.sync()
    for dbg in dbgs:
        os.environ["LTHIP_LZ4_DBG"] = str(dbg)
        fn = ctx.lz4_compress_blocks if codec == "lz4" else ctx.zstd_compress_blocks
        fn(data, b_off, b_size, arena, d_offs, bound)
        ctx.sync()
        ctx.timing(True)
        ctx.timing_reset()
        sizes = fn(data, b_off, b_size, arena, d_offs, bound)
        sz = sizes.cpu().numpy().view(np.uint32).astype(np.int64)
        t = ctx.timing_get()
        ctx.timing(False)
        dec = ctx.lz4_decompress_blocks if codec == "lz4" else ctx.zstd_decompress_blocks
        out = dec(arena, d_offs, sz, back, b_off - shift, b_size)
        good = bool((out.cpu().numpy().view(np.uint32) == b_size).all()) and all(torch.equal(back[int(o) - shift:int(o) - shift + BLOCK], data[int(o):int(o) + BLOCK]) for o in b_off[:4])
        k5 = t["lz4_segments"][0]
        rest = {k: round(v[0], 2) for k, v in t.items() if v[1] and k != "lz4_segments"}
        print(f"{kind:8s} dbg={dbg:<5d} ratio {n / sz.sum():7.4f}  K5 {k5:8.2f} ms = {n / k5 / 1e6:7.1f} GB/s  roundtrip={'ok' if good else 'FAIL'}  {rest}")
