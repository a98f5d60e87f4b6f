"""K1 differential over >= 1 TiB of scanned bytes (once per round; the log goes to profiles/): the shipped candidate scan -- prefix-XOR
formulation fed by LDS-DMA with hand-written exec-masked LDS reads -- against the INDEPENDENT rolling-window kernel of rounds 1-2
(LTHIP_K1=roll, another formulation of hpcdcchunker.c:266-306) on fresh data every pass, cut for cut, plus the shipped kernel against
itself three times per pass (a stale-register hand-off showed as one wrong cut in ~10^6 chunks, and not in every run).  The contract is
bit-exact (lib/hpcdcchunker/longtail_hpcdcchunker.c:266-306); the rolling kernel is pinned against the reference by the -m gpu tests.
usage: python tools/ablations/k1_stress_tib.py [tib] [gib_per_pass]"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from bench import asset_seeds  # noqa: E402
from longtail_amd.lib import Context  # noqa: E402

tib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 32.0
ctx = Context(0)
PART = 1 << 20
n = int(gib * (1 << 30)) // PART
sizes = np.full(n, PART, np.uint64)
offs = np.arange(n, dtype=np.uint64) * np.uint64(PART)
data = torch.empty(n * PART + 256, dtype=torch.uint8, device="cuda")
params = [(8192, 32768, 131072), (4096, 16384, 65536), (16384, 65536, 262144), (48, 64, 128)]
passes = int(np.ceil(tib * 1024 / gib))
scanned = chunks = 0
t0 = time.time()


def cuts(flavour, mn, av, mx):
    if flavour:
        os.environ["LTHIP_K1"] = flavour
    else:
        os.environ.pop("LTHIP_K1", None)
    ctx.lib.dll.lthip_debug_reload_env()
    plan = ctx.make_plan(offs[:m], sizes[:m], mn, av, mx)
    total, o, l, h, f = ctx.chunk_hash(plan, data, want_hashes=False)
    plan.close()
    return total, o[:total].clone(), f.clone()


for it in range(passes):
    kind = [0, 0, 0, 1, 2, 12][it % 6]  # random mostly; mixed, all-zero (no candidate at all), tokens
    mn, av, mx = params[it % len(params)] if it % 5 else params[0]
    m = n if mn >= 4096 else n // 16  # (48-byte chunks: 16x the chunk tables)
    ctx.synth_fill(data, offs[:m], sizes[:m], asset_seeds(0x51DE + it, 0, m), kind)
    ref_total, ref_o, ref_f = cuts("roll", mn, av, mx)
    for rep in range(3):
        total, o, f = cuts(None, mn, av, mx)
        if total != ref_total or not torch.equal(o, ref_o) or not torch.equal(f, ref_f):
            bad = int((o[: min(total, ref_total)] != ref_o[: min(total, ref_total)]).nonzero()[0]) if min(total, ref_total) else 0
            print(f"MISMATCH pass {it} rep {rep} kind {kind} params {(mn, av, mx)}: {total} vs {ref_total} chunks, first differing chunk {bad}")
            sys.exit(1)
        scanned += m * PART
    chunks += ref_total
    if it % 8 == 7 or it + 1 == passes:
        print(f"pass {it + 1}/{passes}: {scanned / (1 << 40):.3f} TiB scanned by the shipped K1 (+ {(it + 1) * gib / 1024:.3f} TiB by the rolling kernel), "
              f"{chunks} chunks per flavour, all cuts equal, {time.time() - t0:.0f} s", flush=True)
print(f"ok: {scanned / (1 << 40):.3f} TiB, {3 * chunks} chunk boundaries of the shipped kernel equal to the rolling-window kernel's")
