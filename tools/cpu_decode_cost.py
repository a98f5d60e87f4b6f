#!/usr/bin/env python
"""What the sub-block frame layout costs a CPU-ONLY consumer: the REFERENCE decoder (oracle/_ref, ZSTD_decompressDCtx behind longtail's
CompressionAPI, longtail_zstd.c:144-177) timed on one host core over frames of this library's encoder in both layouts (one zstd block per
4 KiB unit sharing the piece's tables = the default; one block per 128 KiB piece = LTHIP_ZSTD_SUB=0) and over the reference encoder's own
frames of the same data; LZ4 payloads of both encoders beside it.  usage: tools/cpu_decode_cost.py [blocks] [kind]"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle, ref as get_ref
from longtail_amd.lib import Context

o, r, ctx = get_oracle(), get_ref(), Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = {"mixed": 1, "records": 11, "tokens": 12, "lines": 13}[sys.argv[2] if len(sys.argv) > 2 else "mixed"]
BLOCK = 8 << 20
raws = [np.concatenate([o.synth(1 << 20, 1000 * b + f, kind) for f in range(8)]) for b in range(nb)]
data = torch.from_numpy(np.concatenate(raws)).cuda()
b_off = np.arange(nb, dtype=np.int64) * BLOCK
b_size = np.full(nb, BLOCK, np.int64)


def gpu_frames(fn, bound):
    d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
    arena = torch.zeros(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    sz = fn(data, b_off, b_size, arena, d_offs, bound).cpu().numpy().view(np.uint32).astype(np.int64)
    ctx.sync()
    host = arena.cpu().numpy()
    return [host[int(o_) : int(o_) + int(s)].copy() for o_, s in zip(d_offs, sz)]


def cpu_time(codec, frames):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for f, raw in zip(frames, raws):
            err, out = r.decompress(codec, f, BLOCK)
            assert err == 0 and len(out) == BLOCK
        t = time.perf_counter() - t0
        best = t if best is None or t < best else best
    assert (out == raws[-1]).all()
    return best


rows = []
for sub in ("1", "0"):
    os.environ["LTHIP_ZSTD_SUB"] = sub
    fr = gpu_frames(ctx.zstd_compress_blocks, b_size + (b_size >> 8) + 64)
    rows.append((f"zstd, this library, {'one block per 4 KiB unit (default)' if sub == '1' else 'one block per 128 KiB piece (LTHIP_ZSTD_SUB=0)'}", 1, fr))
os.environ.pop("LTHIP_ZSTD_SUB")
rows.append(("zstd, reference encoder (longtail default setting)", 1, [r.compress(1, r.zstd_default, x) for x in raws]))
rows.append(("lz4, this library", 0, gpu_frames(ctx.lz4_compress_blocks, b_size + b_size // 255 + 16)))
rows.append(("lz4, reference encoder (LZ4_compress_fast restated by the oracle)", 0, [o.lz4_compress(x) for x in raws]))
print(f"{nb} blocks of 8 MiB, kind {sys.argv[2] if len(sys.argv) > 2 else 'mixed'}: the REFERENCE decoder on ONE host core")
for name, codec, fr in rows:
    t = cpu_time(codec, fr)
    print(f"  {name:75s} ratio {nb * BLOCK / sum(len(f) for f in fr):6.3f}  {nb * BLOCK / t / 1e9:6.2f} GB/s")
