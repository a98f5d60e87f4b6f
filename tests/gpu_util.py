"""Helpers shared by the -m gpu parity tests: lay numpy parts out in one device buffer and call the C ABI."""
from __future__ import annotations

import numpy as np
import torch


def layout(parts, align=16):
    offs, pos = [], 0
    for p in parts:
        offs.append(pos)
        pos += (len(p) + align - 1) // align * align
    return offs, max(pos, 16)


def to_device(parts, pad=64):
    offs, total = layout(parts)
    host = np.zeros(total + pad, np.uint8)
    for o, p in zip(offs, parts):
        host[o : o + len(p)] = p
    return torch.from_numpy(host).cuda(), offs


def u64(t):
    return t.cpu().numpy().view(np.uint64)


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def gpu_chunk_hash(ctx, parts, mn, av, mx, hashes=True):
    """-> list per part of (offsets rel. to the part, lens, hashes)"""
    dev, offs = to_device(parts)
    plan = ctx.make_plan(offs, [len(p) for p in parts], mn, av, mx)
    total, d_off, d_len, d_hash, d_first = ctx.chunk_hash(plan, dev, want_hashes=hashes)
    first = u32(d_first)
    assert int(first[-1]) == total
    assert total <= plan.capacity
    o, l = u64(d_off)[:total], u32(d_len)[:total]
    h = u64(d_hash)[:total] if hashes else None
    out = []
    for i, p in enumerate(parts):
        a, b = int(first[i]), int(first[i + 1])
        out.append((o[a:b] - np.uint64(offs[i]), l[a:b], None if h is None else h[a:b]))
    plan.close()
    return out


def check_part(oracle, data, got, mn, av, mx, what=""):
    offs, lens, hashes = got
    e_off, e_len, e_hash = oracle.chunk_and_hash(data, mn, av, mx)
    assert len(lens) == len(e_len), f"{what}: chunk count {len(lens)} != {len(e_len)}"
    assert (lens == e_len).all(), f"{what}: chunk lengths differ at {np.nonzero(lens != e_len)[0][:5]}"
    assert (offs == e_off).all(), f"{what}: chunk offsets differ"
    if hashes is not None:
        bad = np.nonzero(hashes != e_hash)[0]
        assert len(bad) == 0, f"{what}: {len(bad)} chunk hashes differ, first at chunk {bad[:5]} lens {e_len[bad[:5]]}"
