"""Randomised GPU stress (not part of the pytest suite): usage  python tools/<this>.py [seed]"""
import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import sys, numpy as np, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from tests._libs import oracle, ref
from tests.gpu_util import layout, to_device, u32
from longtail_amd.lib import Context
o, r, ctx = oracle(), ref(), Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
kinds = [0, 1, 2, 11, 12, 13]
tot = 0
for rnd in range(12):
    blocks = []
    for _ in range(150):
        k = int(rng.choice(kinds)); n = int(rng.choice([rng.integers(0, 200), rng.integers(0, 9000), rng.integers(0, 300000), rng.integers(0, 1200000)]))
        b = o.synth(n, int(rng.integers(1, 1 << 30)), k)
        if n > 16 and rng.integers(0, 4) == 0:  # splice: repeat an earlier part of the block
            a = int(rng.integers(0, n // 2)); l = int(rng.integers(1, n - a)); p = int(rng.integers(0, n - l + 1)); b = b.copy(); b[p:p + l] = b[a:a + l][: len(b[p:p + l])]
        blocks.append(b)
    dev, offs = to_device(blocks)
    sizes = [len(b) for b in blocks]
    for codec in ("lz4", "zstd"):
        caps = [s + s // 255 + 16 if codec == "lz4" else s + (s >> 8) + 64 for s in sizes]
        d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
        dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        zq = rnd % 3  # the zstd setting's parse: default / high (history halves) / max, round by round
        fn = ctx.lz4_compress_blocks if codec == "lz4" else (lambda *a: ctx.zstd_compress_blocks(*a, quality=zq))
        cs = u32(fn(dev, offs, sizes, dst, d_offs, caps)).astype(np.int64)
        assert (cs > 0).all(), codec
        b_offs, btot = layout([np.zeros(s, np.uint8) for s in sizes])
        back = torch.zeros(btot + 64, dtype=torch.uint8, device="cuda")
        dfn = ctx.lz4_decompress_blocks if codec == "lz4" else ctx.zstd_decompress_blocks
        ds = u32(dfn(dst, d_offs, cs, back, b_offs, sizes))
        host, bh = dst.cpu().numpy(), back.cpu().numpy()
        for i, b in enumerate(blocks):
            assert int(ds[i]) == len(b), (codec, i, int(ds[i]), len(b))
            assert (bh[b_offs[i]: b_offs[i] + len(b)] == b).all(), (codec, i)
            if i % 5 == 0:
                err, out = r.decompress(0 if codec == "lz4" else 1, host[d_offs[i]: d_offs[i] + int(cs[i])].copy(), len(b))
                assert err == 0 and len(out) == len(b) and (out == b).all(), (codec, i, "reference decoder")
        tot += len(blocks)
    # the LZ4 decoder on payloads with a sliding window (the reference's parse: units of the block-parallel path wait for each other)
    # and on damaged ones: size, bytes and verdict are the oracle decoder's
    big = [b for b in blocks if len(b) >= 100000][:40]
    comps, caps = [], []
    for b in big:
        c = o.lz4_compress(b)
        kind = int(rng.integers(0, 5))
        if kind == 1:
            c = c[: int(rng.integers(1, len(c)))]
        elif kind == 2:
            c = c.copy(); c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 3:
            c = c.copy(); c[int(rng.integers(0, len(c))):] = 0
        comps.append(c); caps.append(len(b) if kind != 4 else max(1, len(b) + int(rng.integers(-70000, 70000))))
    if comps:
        cdev, coffs = to_device(comps)
        b_offs, btot = layout([np.zeros(c, np.uint8) for c in caps])
        back = torch.zeros(btot + 64, dtype=torch.uint8, device="cuda")
        ds = u32(ctx.lz4_decompress_blocks(cdev, coffs, [len(c) for c in comps], back, b_offs, caps))
        bh = back.cpu().numpy()
        for i, (c, cap) in enumerate(zip(comps, caps)):
            n, out = o.lz4_decompress(c, cap)
            if n < 0:
                assert int(ds[i]) == 0xFFFFFFFF, ("lz4 verdict", i, int(ds[i]))
            else:
                assert int(ds[i]) == n and (bh[b_offs[i]: b_offs[i] + n] == out[:n]).all(), ("lz4 sliding window", i, int(ds[i]), n)
        tot += len(comps)
    # the zstd decoders on damaged frames of this encoder (sub-block layout with its directory, and one block per piece): the
    # lane-parallel decoders must give the serial decoder's verdict and bytes (LTHIP_ZSTD_DBG=1: one wave per payload)
    import os
    for sub in ("1", "0", "ref"):
        if sub != "ref":
            os.environ["LTHIP_ZSTD_SUB"] = sub
        src = [b for b in blocks if len(b) >= 3000][:60]
        if not src:
            continue
        if sub == "ref":
            # frames of the REFERENCE encoder (any of longtail's five settings): decoded block-parallel (k_zstd_blk_entropy / _sequences /
            # k_zstd_execute_payload), anything it declines by the serial decoder
            src = src[:24]
            ssz = [len(b) for b in src]
            made = [r.compress(1, r.dll.refh_zstd_type(int(rng.integers(0, 5))), b) for b in src]
            cs = np.array([len(m) for m in made], np.int64)
            d_offs, total = layout(made)
            host = np.zeros(total + 64, np.uint8)
            for o_, m in zip(d_offs, made):
                host[o_: o_ + len(m)] = m
        else:
            sdev, soffs = to_device(src)
            ssz = [len(b) for b in src]
            caps = [n + (n >> 8) + 64 for n in ssz]
            d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
            dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
            cs = u32(ctx.zstd_compress_blocks(sdev, soffs, ssz, dst, d_offs, caps, quality=(rnd + 1) % 3)).astype(np.int64)
            host = dst.cpu().numpy()
        frames, fcaps = [], []
        for i, b in enumerate(src):
            f = host[d_offs[i]: d_offs[i] + int(cs[i])].copy()
            nu = (len(b) + 4095) // 4096
            tl = 12 + 2 * nu if sub == "1" else 12 if sub == "0" else 0
            frames.append(f)  # (the undamaged frame too)
            fcaps.append(len(b))
            for _ in range(6):
                x = f.copy()
                kind = int(rng.integers(0, 6))
                if kind == 0 and len(x) > tl + 20:
                    x = np.concatenate([x[: int(rng.integers(13, len(x) - tl))], x[-tl:]])
                elif kind == 1:
                    x = x[: int(rng.integers(1, len(x)))]
                elif kind == 2 and sub == "1":
                    x[len(x) - 2 * nu + int(rng.integers(0, 2 * nu))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                elif kind == 3:
                    a = int(rng.integers(0, len(x))); x[a: a + int(rng.integers(1, 64))] = int(rng.integers(0, 256))
                else:
                    for _ in range(int(rng.integers(1, 5))):
                        x[int(rng.integers(0, len(x)))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                frames.append(x)
                fcaps.append(len(b) if rng.integers(0, 8) else max(1, len(b) + int(rng.integers(-5000, 5000))))
        fdev, foffs = to_device(frames)
        b_offs, btot = layout([np.zeros(c, np.uint8) for c in fcaps])
        outs = []
        for dbg in (None, "1"):
            if dbg:
                os.environ["LTHIP_ZSTD_DBG"] = dbg
            ctx.lib.dll.lthip_debug_reload_env()  # (the library caches its switches)
            back = torch.zeros(btot + 64, dtype=torch.uint8, device="cuda")
            ds = u32(ctx.zstd_decompress_blocks(fdev, foffs, [len(f) for f in frames], back, b_offs, fcaps))
            outs.append((ds.copy(), back.cpu().numpy()))
            os.environ.pop("LTHIP_ZSTD_DBG", None)
        ctx.lib.dll.lthip_debug_reload_env()
        (fs, fb), (ss, sb) = outs
        for i in range(len(frames)):
            # sub-block frames the lane-parallel decoder declines go to the serial one: same verdict.  One-block pieces are decoded
            # strictly (a damaged offset that reaches into the piece before is an error there): it may reject more, never accept more
            if sub != "0" or int(fs[i]) != 0xFFFFFFFF:
                assert int(fs[i]) == int(ss[i]), ("zstd verdict", sub, i, int(fs[i]), int(ss[i]))
            if int(fs[i]) != 0xFFFFFFFF:
                n = int(fs[i])
                assert (fb[b_offs[i]: b_offs[i] + n] == sb[b_offs[i]: b_offs[i] + n]).all(), ("zstd bytes", sub, i)
        tot += len(frames)
    os.environ.pop("LTHIP_ZSTD_SUB", None)
print("ok", tot, "payloads")
