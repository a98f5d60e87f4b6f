import _ablations  # noqa: F401  (first: the LTHIP_* switches used here exist in the ablation build only)
import sys
sys.path.insert(0,'/root/repo')
import os, numpy as np, torch
from tests.gpu_util import to_device, layout, u32
from longtail_amd.lib import Context
ctx = Context(0)
rng = np.random.default_rng(5)
# "text": words of 3..11 bytes from a vocabulary of 4000, separated by spaces: repeats at every distance
voc = [bytes(rng.integers(97, 123, int(rng.integers(3, 12)), dtype=np.uint8)) for _ in range(4000)]
def text(n):
    out = bytearray()
    z = rng.zipf(1.3, n // 4) % 4000
    for w in z:
        out += voc[int(w)] + b" "
        if len(out) >= n: break
    return np.frombuffer(bytes(out[:n]), np.uint8).copy()
blocks = [text(4 << 20) for _ in range(8)]
dev, offs = to_device(blocks)
sizes = [len(b) for b in blocks]
for codec in ("lz4", "zstd"):
    for dbg in (1 << 16, 0, 3 << 17, 2 << 17, 1 << 17, 3 << 29):  # fixed four one-byte steps; adaptive with 6 (default), 3, 2, 1 quiet rounds; no one-byte steps
        os.environ["LTHIP_LZ4_DBG"] = str(dbg)
        ctx.lib.dll.lthip_debug_reload_env()
        caps = [s + s // 255 + 16 if codec == "lz4" else s + (s >> 8) + 64 for s in sizes]
        d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
        dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        fn = ctx.lz4_compress_blocks if codec == "lz4" else ctx.zstd_compress_blocks
        cs = u32(fn(dev, offs, sizes, dst, d_offs, caps)).astype(np.int64)
        print(codec, "dbg", hex(dbg), "ratio %.4f" % (sum(sizes) / cs.sum()))
