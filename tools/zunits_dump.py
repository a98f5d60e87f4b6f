#!/usr/bin/env python
"""Dumps the zstd match finder's units (meta / literals / records, ZbInput of zstd_block_core.h) of a few blocks per data kind, so that
changes to the ENTROPY stage can be tried on the CPU with the host model (oracle/zstd_model.c compiles the same zstd_block_core.h):
tools/zunits_eval.py reads what this writes.  usage (GPU box): tools/zunits_dump.py [MiB per kind] [quality] -> gpurun_out/zunits_q<quality>.npz
The blocks start at an odd offset inside the data, like bench.py's blocks (chunk boundaries), not in phase with the data's structures."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle
from longtail_amd.lib import Context

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2
quality = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o, ctx = get_oracle(), Context(0)
BLOCK = mib << 20
SHIFT = 12345


def text_block(seed, n):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, int(rng.integers(3, 12))).astype(np.uint8)) for _ in range(4096)]
    idx = np.minimum((rng.pareto(1.1, 2_000_000)).astype(np.int64), 4095)
    out = b" ".join(words[i] for i in idx[: n // 4])
    return np.frombuffer(out[:n].ljust(n, b" "), np.uint8).copy()


out = {}
for name, kind in (("mixed", 1), ("records", 11), ("tokens", 12), ("lines", 13), ("text", -1)):
    n = BLOCK + (1 << 20)
    raw = np.concatenate([o.synth(1 << 20, 77 + f, kind) for f in range(n >> 20)]) if kind >= 0 else text_block(7, n)
    data = torch.from_numpy(raw).cuda()
    bound = BLOCK + (BLOCK >> 8) + 64
    arena = torch.zeros(bound + 128, dtype=torch.uint8, device="cuda")
    sz = ctx.zstd_compress_blocks(data, [SHIFT], [BLOCK], arena, [0], [bound], quality=quality).cpu().numpy().view(np.uint32)
    ctx.sync()
    nunits = (BLOCK + 4095) // 4096
    meta, lits, recs = ctx.zstd_debug_units(0, nunits)
    # records beyond a unit's count are stale: zero them (the file compresses better)
    for u in range(nunits):
        recs[u, int(meta[u, 0]):] = 0
        lits[u, int(meta[u, 1]):] = 0
    out[name + "_raw"] = raw[SHIFT : SHIFT + BLOCK]
    out[name + "_meta"] = meta
    out[name + "_lits"] = lits
    out[name + "_recs"] = recs
    out[name + "_frame"] = np.array([int(sz[0])])
    print(name, "frame", int(sz[0]), "ratio", BLOCK / int(sz[0]), "sequences", int(meta[:, 0].sum()))
Path("gpurun_out").mkdir(exist_ok=True)
np.savez_compressed(f"gpurun_out/zunits_q{quality}.npz", **out)
