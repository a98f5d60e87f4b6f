#!/usr/bin/env python
"""Would the candidate scan (K1: 4 waves per SIMD, LDS-heavy, VALU 72 % busy) of one SLICE of the tree and the leaf hashing (K3: VALU
bound, no LDS) of the slice before it overlap?  chunk_hash over a tree of 1 MiB files as one call, and as S slices on S contexts with
private streams (launched back to back from one thread: every slice's kernels are in order on its own stream, the slices' streams run
concurrently).  Optional K3 occupancy limit through the ablation build's LTHIP_B3_PAD_LDS (unused LDS per workgroup).
usage: tools/k1k3_overlap_probe.py [gib] [slices,slices,...]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import asset_seeds
from longtail_amd.lib import Context, chunker_params, load, load_ablations

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
slices_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
lib = load_ablations() if os.environ.get("LTHIP_B3_PAD_LDS") else load()
FILE = 1 << 20
nfiles = int(gib * (1 << 30)) // FILE
main = Context(0, stream=None, lib=lib)
data = torch.empty(nfiles * FILE + 256, dtype=torch.uint8, device="cuda")
main.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE), np.full(nfiles, FILE, np.uint64), asset_seeds(7, 0, nfiles), 0)
main.sync()
mn, av, mx = chunker_params(65536)
ctxs = [Context(0, stream=None, lib=lib) for _ in range(max(slices_list))]
ref_hash = None
for S in slices_list:
    per = nfiles // S
    plans, outs = [], []
    for s in range(S):
        f0, f1 = s * per, (nfiles if s == S - 1 else (s + 1) * per)
        offs = np.arange(f0, f1, dtype=np.uint64) * np.uint64(FILE)
        sizes = np.full(f1 - f0, FILE, np.uint64)
        plan = ctxs[s].make_plan(offs, sizes, mn, av, mx)
        cap = max(1, plan.capacity)
        outs.append((torch.empty(cap, dtype=torch.int64, device="cuda"), torch.empty(cap, dtype=torch.int32, device="cuda"),
                     torch.empty(cap, dtype=torch.int64, device="cuda"), torch.empty(f1 - f0 + 1, dtype=torch.int32, device="cuda")))
        plans.append(plan)
    best = None
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(S):
            ctxs[s].chunk_hash(plans[s], data, outputs=outs[s], sync=False)
        for s in range(S):
            ctxs[s].sync()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    # the hashes of all slices, in order, must be the one-call result
    hs = []
    for s in range(S):
        total = int(outs[s][3][-1].item())
        hs.append(outs[s][2][:total].cpu())
    h = torch.cat(hs)
    if ref_hash is None:
        ref_hash = h
    same = bool(h.numel() == ref_hash.numel() and torch.equal(h, ref_hash))
    print(f"{gib:g} GiB, {S} slice(s): chunk + hash {best:.2f} ms = {gib * 1.073741824 / best * 1e3:.0f} GB/s  hashes {'identical' if same else 'DIFFER'}", flush=True)
    for p in plans:
        p.close()
