#!/usr/bin/env python
"""What the zstd front end (the lane parser, FMT 1) finds on one 8 MiB block, per setting: sequences and literals per 4 KiB unit, match
lengths, offsets, and an order-0 price of what it found (literals by their entropy; a sequence by the entropies of its three codes plus
its extra bits) next to the frame's real size.  usage: tools/zstd_parse_stats.py [text|tokens|records|mixed ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from tests._libs import oracle as get_oracle
from tests.gpu_util import to_device, layout, u32
from longtail_amd.lib import Context

o, ctx = get_oracle(), Context(0)
BLOCK = 8 << 20
KIND = {"mixed": 1, "records": 11, "tokens": 12, "lines": 13}


def text_block(seed):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, int(rng.integers(3, 12))).astype(np.uint8)) for _ in range(4096)]
    idx = np.minimum((rng.pareto(1.1, 2_000_000)).astype(np.int64), 4095)
    out = b" ".join(words[i] for i in idx[: 1_400_000])
    return np.frombuffer(out[:BLOCK].ljust(BLOCK, b" "), np.uint8).copy()


def H(counts):
    c = counts[counts > 0].astype(np.float64)
    return float((c * np.log2(c.sum() / c)).sum())


def code_of(v, base_codes):  # zstd's length codes: small values map to themselves, large ones to log2 buckets (approximation: highbit + const)
    return np.where(v < base_codes, v, base_codes + np.floor(np.log2(np.maximum(v, 1))).astype(np.int64))


for name in sys.argv[1:] or ["text", "tokens"]:
    raw = text_block(0) if name == "text" else np.concatenate([o.synth(1 << 20, f, KIND[name]) for f in range(8)])
    dev, offs = to_device([raw])
    for q, qn in ((0, "default"), (1, "high"), (2, "max")):
        cap = BLOCK + (BLOCK >> 8) + 64
        dst = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
        cs = int(u32(ctx.zstd_compress_blocks(dev, offs, [BLOCK], dst, [0], [cap], quality=q))[0])
        nun = BLOCK // 4096
        meta, lits, recs = ctx.zstd_debug_units(0, nun)
        nseq, nlit = meta[:, 0].astype(np.int64), meta[:, 1].astype(np.int64)
        rr = np.concatenate([recs[u, : nseq[u]] for u in range(nun)])
        ll, ml, off = (rr & 0xFFFF).astype(np.int64), ((rr >> 16) & 0xFFFF).astype(np.int64), (rr >> 32).astype(np.int64)
        litbytes = np.concatenate([lits[u, : nlit[u]] if nseq[u] else raw[u * 4096 : (u + 1) * 4096] for u in range(nun)])
        lit_bits = H(np.bincount(litbytes, minlength=256))
        oc = np.floor(np.log2(off + 3)).astype(np.int64)
        llc, mlc = code_of(ll, 16), code_of(ml - 3, 32)
        seq_bits = H(np.bincount(oc)) + oc.sum() + H(np.bincount(llc)) + H(np.bincount(mlc)) + np.maximum(llc - 16, 0).sum() + np.maximum(mlc - 32, 0).sum()
        print(f"{name:8s} {qn:8s} frame {cs:8d} (ratio {BLOCK / cs:6.3f})  seqs/unit {nseq.mean():6.1f}  lits/unit {nlit.mean():7.1f}  mean ml {ml.mean():6.2f}  "
              f"median off {int(np.median(off)):6d}  off<64 {100.0 * (off < 64).mean():4.1f}%  price: literals {lit_bits / 8:9.0f} B  sequences {seq_bits / 8:9.0f} B "
              f"({seq_bits / max(len(rr), 1):5.2f} bits each, offset part {(H(np.bincount(oc)) + oc.sum()) / max(len(rr), 1):5.2f})")
