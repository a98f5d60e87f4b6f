#!/usr/bin/env python
"""debug: sub-block frames of the HIP zstd encoder through the reference decoder"""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle, ref as get_ref
from longtail_amd.lib import Context
import tests.test_gpu_codecs as T
o = get_oracle(); r = get_ref(); gpu = Context(0)
blocks = [o.synth(n, 40 + n, k) for k in (0, 1, 2, 11, 12, 13) for n in (0, 1, 100, 5000, 131071, 131072, 131073, 400000)]
blocks.append(o.synth((8 << 20) + 12345, 7, 1))
rng = np.random.default_rng(3)
blocks.append((np.abs(rng.normal(128, 20, 700000)).astype(np.int64) % 256).astype(np.uint8))
blocks.append(np.frombuffer(b"the quick brown fox jumps over the lazy dog. " * 9000, np.uint8).copy())
frames = T.gpu_zstd(gpu, blocks)
for i, (b, f) in enumerate(zip(blocks, frames)):
    err, out = r.decompress(1, f, len(b))
    ok = err == 0 and len(out) == len(b) and (out == b).all()
    if not ok:
        print(i, len(b), len(f), f"FAIL err={err}", flush=True)
        alone = T.gpu_zstd(gpu, [b])[0]
        e2, o2 = r.decompress(1, alone, len(b))
        print("  alone:", len(alone), e2, "same bytes" if len(alone) == len(f) and (alone == f).all() else "differs")
        if len(alone) == len(f):
            d = np.flatnonzero(alone != f)
            print("  first diffs at", d[:10], "of", len(d))
        try:
            ps = T.zstd_pieces(f)
            print("  pieces", [(t, len(p)) for t, p in ps][:6])
        except AssertionError as e:
            print("  pieces parse failed")
print("done")
