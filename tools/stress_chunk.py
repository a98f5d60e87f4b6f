"""Randomised GPU stress (not part of the pytest suite): usage  python tools/<this>.py [seed]"""
import sys, numpy as np, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from tests._libs import oracle
from tests.gpu_util import gpu_chunk_hash, check_part
from longtail_amd.lib import Context
o, ctx = oracle(), Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = 0
for it in range(60):
    mn = int(rng.choice([48, 64, 100, 1000, 4096, 8192, 65536, 200000]))
    av = int(mn * rng.choice([1, 1.5, 2, 4, 8]))
    mx = int(av * rng.choice([1, 1.3, 2, 4, 16]))
    parts = []
    for _ in range(int(rng.integers(1, 12))):
        size = int(rng.choice([rng.integers(0, 300), rng.integers(0, 70000), rng.integers(0, 3 << 20), rng.integers(0, 9 << 20)]))
        parts.append(o.synth(size, int(rng.integers(1, 1 << 30)), int(rng.choice([0, 1, 2, 11, 12, 13]))))
    got = gpu_chunk_hash(ctx, parts, mn, av, mx)
    for i, (p, g) in enumerate(zip(parts, got)):
        check_part(o, p, g, mn, av, mx, f"it {it} cfg {(mn, av, mx)} part {i} size {len(p)}")
        n += len(g[1])
print("ok", n, "chunks bit-exact")
