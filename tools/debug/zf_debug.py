#!/usr/bin/env python
"""debug: which reference-made zstd frames go back to the serial decoder"""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle, ref as get_ref
from longtail_amd.lib import Context
import tests.test_gpu_codecs as T
o = get_oracle(); r = get_ref(); gpu = Context(0)
rng = np.random.default_rng(33)
datas = [o.synth(n, 290 + n, k) for k, n in ((2, 400000), (1, 600000), (12, 300000))]
datas.append(np.concatenate([rng.integers(0, 256, 200000, dtype=np.uint8), o.synth(300000, 4, 1), np.zeros(150000, np.uint8)]))
for i, d in enumerate(datas):
    for w in range(5):
        if len(d) > (1 << 20) and w not in (0, 3):
            continue
        f = r.compress(1, r.dll.refh_zstd_type(w), d)
        out = T.gpu_zstd_decode(gpu, [f], [len(d)])
        st = gpu.zstd_last_decode_stats()
        # block structure
        fhd = int(f[4]); pos = 5 + (0 if fhd & 0x20 else 1) + (0, 1, 2, 4)[fhd & 3] + ((1, 2, 4, 8)[fhd >> 6] if (fhd >> 6) or (fhd & 0x20) else 0)
        nb = 0; types = []
        while True:
            h = int(f[pos]) | int(f[pos + 1]) << 8 | int(f[pos + 2]) << 16
            n = 1 if (h >> 1) & 3 == 1 else h >> 3
            types.append((h >> 1) & 3); pos += 3 + n; nb += 1
            if h & 1: break
        print(i, w, len(d), len(f), "fhd %02x" % fhd, "blocks", nb, "types", sorted(set(types)), "stats", st, "ok" if (out[0] is not None and (out[0] == d).all()) else "WRONG")
