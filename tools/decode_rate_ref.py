#!/usr/bin/env python
"""LZ4 decode rate on payloads with a SLIDING window (the reference's parse, made by the oracle's bit-exact restatement of LZ4_compress_fast):
the units of the block-parallel decoder wait for each other.  usage: tools/decode_rate_ref.py [blocks] [kind]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle
from tests.gpu_util import to_device, u32
from longtail_amd.lib import Context
o, ctx = get_oracle(), Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kind = {"mixed": 1, "records": 11, "tokens": 12, "lines": 13}[sys.argv[2] if len(sys.argv) > 2 else "mixed"]
BLOCK = 8 << 20
raws = [np.concatenate([o.synth(1 << 20, 1000 * b + f, kind) for f in range(8)]) for b in range(nb)]
comps = [o.lz4_compress(r) for r in raws]
dev, offs = to_device(comps)
back = torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device="cuda")
b_off = [i * BLOCK for i in range(nb)]
for rep in range(2):
    t0 = time.perf_counter()
    out = ctx.lz4_decompress_blocks(dev, offs, [len(c) for c in comps], back, b_off, [BLOCK] * nb)
    ctx.sync()
    t = time.perf_counter() - t0
ok = (u32(out) == BLOCK).all() and all((back[i * BLOCK:(i + 1) * BLOCK].cpu().numpy() == raws[i]).all() for i in range(0, nb, max(1, nb // 4)))
print(f"lz4 (sliding-window payloads): {nb} blocks of 8 MiB, ratio {nb * BLOCK / sum(len(c) for c in comps):.3f}, decode {t * 1e3:.1f} ms = {nb * BLOCK / t / 1e9:.2f} GB/s, {'ok' if ok else 'MISMATCH'}")

# the same for zstd frames written by the REFERENCE encoder (longtail's default setting): blocks that depend on each other (window =
# the frame, repeat offsets, repeated tables) -- one wave per payload, the serial core
try:
    from tests._libs import ref as get_ref
    r = get_ref()
    zc = [r.compress(1, r.zstd_default, x) for x in raws]
    zdev, zoffs = to_device(zc)
    for rep in range(2):
        t0 = time.perf_counter()
        out = ctx.zstd_decompress_blocks(zdev, zoffs, [len(c) for c in zc], back, b_off, [BLOCK] * nb)
        ctx.sync()
        t = time.perf_counter() - t0
    ok = (u32(out) == BLOCK).all() and all((back[i * BLOCK:(i + 1) * BLOCK].cpu().numpy() == raws[i]).all() for i in range(0, nb, max(1, nb // 4)))
    print(f"zstd (reference encoder's frames): {nb} blocks of 8 MiB, ratio {nb * BLOCK / sum(len(c) for c in zc):.3f}, decode {t * 1e3:.1f} ms = {nb * BLOCK / t / 1e9:.2f} GB/s, {'ok' if ok else 'MISMATCH'}")
except Exception as e:  # oracle/_ref not built
    print("zstd reference frames: skipped:", e)
