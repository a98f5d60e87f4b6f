#!/usr/bin/env python
"""zstd ratio of this library's encoder (ONE parse whatever the setting) next to the reference encoder at longtail's settings
ztd1..ztd4 (levels 3 / 3 / 22 / 8, lib/zstd/longtail_zstd.c:11-28) on the synthetic kinds, 8 MiB blocks.  The reference frames
are checked through the HIP decoder, ours through the reference decoder.  usage: tools/zstd_ratio_table.py [blocks per kind]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle, ref as get_ref
from tests.gpu_util import to_device, u32
from longtail_amd.lib import Context

o, r, ctx = get_oracle(), get_ref(), Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BLOCK = 8 << 20
print(f"{nb} blocks of 8 MiB per kind; ratio = input / frames (the reference's settings: ztd1 = level 3, ztd2 = 3 (default), ztd3 = 22, ztd4 = 8)")
print(f"{'kind':8s} {'this library':>13s} {'ztd1':>8s} {'ztd2':>8s} {'ztd3':>8s} {'ztd4':>8s}   reference seconds per block at ztd2 / ztd3 / ztd4 (one host core)")
for name, kind in (("mixed", 1), ("records", 11), ("tokens", 12), ("lines", 13)):
    raws = [np.concatenate([o.synth(1 << 20, 1000 * b + f, kind) for f in range(8)]) for b in range(nb)]
    data = torch.from_numpy(np.concatenate(raws)).cuda()
    b_off = np.arange(nb, dtype=np.int64) * BLOCK
    b_size = np.full(nb, BLOCK, np.int64)
    bound = b_size + (b_size >> 8) + 64
    d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
    arena = torch.zeros(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    sz = ctx.zstd_compress_blocks(data, b_off, b_size, arena, d_offs, bound).cpu().numpy().view(np.uint32).astype(np.int64)
    ctx.sync()
    host = arena.cpu().numpy()
    for i in range(nb):
        err, out = r.decompress(1, host[int(d_offs[i]) : int(d_offs[i]) + int(sz[i])].copy(), BLOCK)
        assert err == 0 and (out == raws[i]).all()
    ours = nb * BLOCK / float(sz.sum())
    ratios, secs = [], []
    for w in range(4):
        t0 = time.perf_counter()
        frames = [r.compress(1, r.dll.refh_zstd_type(w), x) for x in raws]
        secs.append((time.perf_counter() - t0) / nb)
        ratios.append(nb * BLOCK / sum(len(f) for f in frames))
        if w in (1, 2):
            dev, offs = to_device(frames)
            back = torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device="cuda")
            got = ctx.zstd_decompress_blocks(dev, offs, [len(f) for f in frames], back, list(b_off), [BLOCK] * nb)
            ctx.sync()
            assert (u32(got) == BLOCK).all() and (back[: nb * BLOCK].cpu().numpy() == np.concatenate(raws)).all()
    print(f"{name:8s} {ours:13.3f} {ratios[0]:8.3f} {ratios[1]:8.3f} {ratios[2]:8.3f} {ratios[3]:8.3f}   {secs[1]:.2f} / {secs[2]:.2f} / {secs[3]:.2f}")
