#!/usr/bin/env python
"""Debugging aid for the block-parallel LZ4 decoder: one block per call, kind by kind (run with LTHIP_LZ4_PD_TRACE=1 LTHIP_LZ4_PD_STATS=1)."""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle
from tests.gpu_util import layout, to_device, u32
from longtail_amd.lib import Context

o = get_oracle()
gpu = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
for kind in (0, 1, 2, 11):
    raw = o.synth(n, 900 + n, kind)
    comp = o.lz4_compress(raw)
    dev, offs = to_device([comp])
    dst = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    print("kind", kind, "payload", len(comp), flush=True)
    t0 = time.perf_counter()
    sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(comp)], dst, [0], [n]))
    gpu.sync()
    ok = int(sizes[0]) == n and (dst.cpu().numpy()[:n] == raw).all()
    print("   size", int(sizes[0]), "ok" if ok else "MISMATCH", f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
