"""Where does a K1 flavour disagree with itself / with the first run?  usage: LTHIP_K1=... python tools/ablations/k1_diff.py [gib] [part_mib] [runs]"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import sys, numpy as np, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent.parent))
from longtail_amd.lib import Context
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
part = int(float(sys.argv[2]) * (1 << 20)) if len(sys.argv) > 2 else (1 << 20)
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = Context(0)
n = int(gib * (1 << 30)) // part
sizes = np.full(n, part, np.uint64)
offs = (np.arange(n, dtype=np.uint64) * np.uint64((part + 15) // 16 * 16))
data = torch.empty(int(offs[-1]) + part + 64, dtype=torch.uint8, device="cuda")
ctx.synth_fill(data, offs, sizes, np.arange(1, n + 1, dtype=np.uint64) * 7919, 0)
plan = ctx.make_plan(offs, sizes, 8192, 32768, 131072)
ref = None
for r in range(runs):
    total, o, l, h, f = ctx.chunk_hash(plan, data, want_hashes=False)
    o = o[:total].cpu().numpy().view(np.uint64); f = f.cpu().numpy().view(np.uint32).astype(np.int64)
    if ref is None:
        ref = (total, o, f); print("run 0:", total, "chunks"); continue
    if total == ref[0] and (o == ref[1]).all():
        print("run", r, "same"); continue
    # first differing chunk per part
    bad = 0
    for p in range(n):
        a, b = ref[1][ref[2][p]:ref[2][p + 1]], o[f[p]:f[p + 1]]
        if len(a) != len(b) or (a != b).any():
            k = next(i for i in range(min(len(a), len(b))) if a[i] != b[i]) if len(a) and len(b) and (a[:min(len(a), len(b))] != b[:min(len(a), len(b))]).any() else min(len(a), len(b))
            ra = int(a[k] - offs[p]) if k < len(a) else -1
            rb = int(b[k] - offs[p]) if k < len(b) else -1
            sa, sb = set(a.tolist()), set(b.tolist())
            extra = sorted(int(x - offs[p]) for x in sb - sa)[:3]; missing = sorted(int(x - offs[p]) for x in sa - sb)[:3]
            print(f"   spurious cuts at {extra} (mod 4096: {[e % 4096 for e in extra]}), missing cuts at {missing} (mod 4096: {[e % 4096 for e in missing]})")
            print(f"run {r} part {p}: chunk {k} starts at {ra} (ref) vs {rb}; rel pos mod 4096 = {ra % 4096 if ra>=0 else -1} / {rb % 4096 if rb>=0 else -1}; tile {ra//4096} / {rb//4096}; part size {part}")
            bad += 1
            if bad > 12: break
    print("run", r, "differs in", bad, "parts (listed up to 12)")
