#!/usr/bin/env python3
"""LZ4 ratio of the lane parser against the number of one-byte steps a lane takes after a hit or at its start before it only probes
address-aligned positions (LTHIP_LZ4_DBG bits 29-30: 4, 2, 1, 0), on data whose structure is NOT aligned to the addresses: word soup
(text), and the synthetic kinds copied to an address that is 1 mod 4.  python tools/ablations/dense_probe.py"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from bench import KINDS, asset_seeds  # noqa: E402
from longtail_amd.lib import Context  # noqa: E402
from tests.gpu_util import u32  # noqa: E402

ctx = Context(0)
BLOCK = 8 << 20
nb = 16
n = nb * BLOCK
rng = np.random.default_rng(5)
voc = [bytes(rng.integers(97, 123, int(rng.integers(3, 12)), dtype=np.uint8)) for _ in range(4000)]


def text(nbytes):
    z = rng.zipf(1.3, nbytes // 4) % 4000
    out = b" ".join(voc[int(w)] for w in z)
    return np.frombuffer(out[:nbytes].ljust(nbytes, b" "), np.uint8).copy()


def run(data, label, shift):
    b_off = np.arange(nb, dtype=np.int64) * BLOCK + shift
    b_size = np.full(nb, BLOCK, np.int64)
    bound = b_size + b_size // 255 + 16
    d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
    arena = torch.empty(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    row = []
    for x in (0, 1, 2, 3):
        os.environ["LTHIP_LZ4_DBG"] = str(x << 29)
        ctx.lz4_compress_blocks(data, b_off, b_size, arena, d_offs, bound)
        ctx.sync()
        ctx.timing(True)
        ctx.timing_reset()
        sz = u32(ctx.lz4_compress_blocks(data, b_off, b_size, arena, d_offs, bound)).astype(np.int64)
        t = ctx.timing_get()
        ctx.timing(False)
        row.append(f"{n / sz.sum():7.4f} ({t['lz4_segments'][0]:5.2f} ms)")
    print(f"{label:28s} " + "  ".join(row))


print(f"{nb} blocks of 8 MiB; one-byte steps before the aligned ones:       4                 2                 1                 0")
t = torch.from_numpy(np.concatenate([text(BLOCK) for _ in range(nb)])).cuda()
pad = torch.zeros(n + 256, dtype=torch.uint8, device="cuda")
pad[:n] = t
run(pad, "text", 0)
for kind in ("mixed", "records", "tokens", "lines"):
    data = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
    nfiles = n // (1 << 20)
    ctx.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(1 << 20), np.full(nfiles, 1 << 20, np.uint64), asset_seeds(0xBEEF, 0, nfiles), KINDS[kind])
    ctx.sync()
    run(data, kind + " (aligned)", 0)
    moved = torch.zeros(n + 512, dtype=torch.uint8, device="cuda")
    moved[17 : 17 + n] = data[:n]
    run(moved, kind + " (address 1 mod 4)", 17)

# the reference's LZ4_compress_default on the same text, for scale (oracle restatement, one 8 MiB block)
from tests._libs import oracle as get_oracle  # noqa: E402

o = get_oracle()
tb = t[:BLOCK].cpu().numpy()
print(f"reference LZ4_compress_default on the first text block: {BLOCK / len(o.lz4_compress(tb)):.4f}")
