#!/usr/bin/env python
"""Debugging aid for the two-stage zstd piece decoder: own frames of several kinds / sizes, first mismatch per block."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle
from tests.gpu_util import layout, to_device, u32
from longtail_amd.lib import Context

o = get_oracle()
gpu = Context(0)
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [131073, 400000, 1 << 20]
for kind in (1, 2, 11, 12, 13, 0):
    for n in sizes:
        raw = o.synth(n, 60 + n, kind)
        dev, offs = to_device([raw])
        bound = n + (n >> 8) + 64
        dst = torch.zeros(bound + 64, dtype=torch.uint8, device="cuda")
        sz = int(u32(gpu.zstd_compress_blocks(dev, offs, [n], dst, [0], [bound]))[0])
        back = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        out = int(u32(gpu.zstd_decompress_blocks(dst, [0], [sz], back, [0], [n]))[0])
        gpu.sync()
        b = back.cpu().numpy()[:n]
        bad = np.nonzero(b != raw)[0]
        print(f"kind {kind} n {n}: frame {sz}, decoded {out if out != 0xFFFFFFFF else 'ERROR'}, "
              + ("ok" if out == n and len(bad) == 0 else f"MISMATCH first {bad[0] if len(bad) else '-'} count {len(bad)} (piece {bad[0] // 131072 if len(bad) else '-'}, in-piece {bad[0] % 131072 if len(bad) else '-'})"), flush=True)
