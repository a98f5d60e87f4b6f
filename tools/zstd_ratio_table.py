#!/usr/bin/env python
"""zstd ratio of this library's encoder at its three parses (LTHIP_ZSTD_Q_DEFAULT = 'ztd1'/'ztd2', _HIGH = 'ztd4', _MAX = 'ztd3'/'ztd5';
LTHIP_ZSTD_REP=0 in the environment: without repeat-offset codes) next to the reference encoder at longtail's settings
ztd1..ztd4 (levels 3 / 3 / 22 / 8, lib/zstd/longtail_zstd.c:11-28) on the synthetic kinds and on text, 8 MiB blocks.  The reference frames
are checked through the HIP decoder, ours through the reference decoder.  usage: tools/zstd_ratio_table.py [blocks per kind]"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle, ref as get_ref
from tests.gpu_util import to_device, u32
from longtail_amd.lib import Context

o, r, ctx = get_oracle(), get_ref(), Context(0)
ours_only = "--ours-only" in sys.argv  # skip the reference legs (parameter sweeps)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
nb = int(args[0]) if args else 2
BLOCK = 8 << 20
print(f"{nb} blocks of 8 MiB per kind; ratio = input / frames (the reference's settings: ztd1 = level 3, ztd2 = 3 (default), ztd3 = 22, ztd4 = 8)")
print(f"{'kind':8s} {'ours default':>13s} {'ours high':>10s} {'ours max':>9s} {'ztd1':>8s} {'ztd2':>8s} {'ztd3':>8s} {'ztd4':>8s}   reference seconds per block at ztd2 / ztd3 / ztd4 (one host core)")


def text_block(seed):
    """word soup: 4096 words of 3..11 letters drawn with a Zipf-like law, separated by spaces (tools/text_ratio_probe.py's data)"""
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, int(rng.integers(3, 12))).astype(np.uint8)) for _ in range(4096)]
    idx = np.minimum((rng.pareto(1.1, 2_000_000)).astype(np.int64), 4095)
    out = b" ".join(words[i] for i in idx[: 1_400_000])
    return np.frombuffer(out[:BLOCK].ljust(BLOCK, b" "), np.uint8).copy()


for name, kind in (("mixed", 1), ("records", 11), ("tokens", 12), ("lines", 13), ("text", -1)):
    raws = [np.concatenate([o.synth(1 << 20, 1000 * b + f, kind) for f in range(8)]) if kind >= 0 else text_block(b) for b in range(nb)]
    data = torch.from_numpy(np.concatenate(raws)).cuda()
    b_off = np.arange(nb, dtype=np.int64) * BLOCK
    b_size = np.full(nb, BLOCK, np.int64)
    bound = b_size + (b_size >> 8) + 64
    d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
    arena = torch.zeros(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    ours_q = []
    for q in (0, 1, 2):
        sz = ctx.zstd_compress_blocks(data, b_off, b_size, arena, d_offs, bound, quality=q).cpu().numpy().view(np.uint32).astype(np.int64)
        ctx.sync()
        host = arena.cpu().numpy()
        for i in range(nb):
            err, out = r.decompress(1, host[int(d_offs[i]) : int(d_offs[i]) + int(sz[i])].copy(), BLOCK)
            assert err == 0 and (out == raws[i]).all()
        back = torch.zeros(nb * BLOCK + 64, dtype=torch.uint8, device="cuda")
        got = ctx.zstd_decompress_blocks(arena, d_offs, sz, back, list(b_off), [BLOCK] * nb)  # ... and through this library's decoder
        ctx.sync()
        assert (u32(got) == BLOCK).all() and torch.equal(back[: nb * BLOCK], data[: nb * BLOCK])
        ours_q.append(nb * BLOCK / float(sz.sum()))
    ratios, secs = [], []
    if ours_only:
        print(f"{name:8s} {ours_q[0]:13.3f} {ours_q[1]:10.3f} {ours_q[2]:9.3f}")
        continue
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
    print(f"{name:8s} {ours_q[0]:13.3f} {ours_q[1]:10.3f} {ours_q[2]:9.3f} {ratios[0]:8.3f} {ratios[1]:8.3f} {ratios[2]:8.3f} {ratios[3]:8.3f}   {secs[1]:.2f} / {secs[2]:.2f} / {secs[3]:.2f}")
