import sys
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests._libs import oracle as get_oracle, ref as get_ref
from tests.gpu_util import to_device, u32
from longtail_amd.lib import Context
o, ctx, r = get_oracle(), Context(0), get_ref()
n = int(sys.argv[1]) if len(sys.argv)>1 else 1<<20
kind = int(sys.argv[2]) if len(sys.argv)>2 else 1
raw = np.concatenate([o.synth(min(n,1<<20), 7+f, kind) for f in range(max(1,n>>20))])[:n]
zc = [r.compress(1, r.zstd_default, raw)]
dev, offs = to_device(zc)
back = torch.zeros(len(raw)+64, dtype=torch.uint8, device="cuda")
out = ctx.zstd_decompress_blocks(dev, offs, [len(zc[0])], back, [0], [len(raw)])
ctx.sync()
print("stats", ctx.zstd_last_decode_stats(), "size", u32(out), "equal", bool((back[:len(raw)].cpu().numpy()==raw).all()))
