import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from bench import KINDS, asset_seeds
from longtail_amd.lib import Context
ctx = Context(0)
BLOCK = 8 << 20
n = 64 << 20
data = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
for file_mib in (1, 64):
    nf = n // (file_mib << 20)
    ctx.synth_fill(data, np.arange(nf, dtype=np.uint64) * np.uint64(file_mib << 20), np.full(nf, file_mib << 20, np.uint64), asset_seeds(0xBEEF, 0, nf), KINDS["mixed"])
    ctx.sync()
    nb = n // BLOCK
    b_off = np.arange(nb, dtype=np.int64) * BLOCK
    b_size = np.full(nb, BLOCK, np.int64)
    bound = b_size + b_size // 255 + 16
    d_offs = np.concatenate([[0], np.cumsum((bound + 63) // 64 * 64)[:-1]])
    arena = torch.empty(int(bound.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
    batch = ctx.lz4_compress_blocks(data, b_off, b_size, arena, d_offs, bound).cpu().numpy().view(np.uint32).astype(np.int64)
    single = np.array([int(ctx.lz4_compress_blocks(data, b_off[i:i+1], b_size[i:i+1], arena, d_offs[i:i+1], bound[i:i+1]).cpu().numpy().view(np.uint32)[0]) for i in range(nb)])
    print(file_mib, "MiB files: batch ratio", n / batch.sum(), "single-block ratio", n / single.sum(), (batch == single).all())
