#!/usr/bin/env python
"""Does a slice-sized working set stay in the 256 MiB memory-side cache between passes?  Sum (read-only) and copy of
buffers of 32 MiB .. 8 GiB, repeated back to back: bandwidth per size.  If small sizes run well above the 8 GiB rate, a
slice-by-slice pipeline (chunk+hash then compress the same slice) would read the slice from cache in its second pass."""
import time, torch
def t(f, reps):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
big = torch.empty(8 << 30, dtype=torch.uint8, device="cuda"); big.random_(0, 255)
dst = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
for mib in (32, 64, 128, 192, 256, 512, 2048, 8192):
    n = mib << 20
    a = big[:n]; ai = a.view(torch.int64); b = dst[:n]
    reps = max(3, min(200, (16 << 30) // n))
    s = t(lambda: ai.sum(), reps)
    c = t(lambda: b.copy_(a), reps)
    # read slice (hot) -> write to a fresh region each time (streaming output like the codec's)
    k = [0]
    def stream():
        o = (k[0] * n) % ((8 << 30) - n + 1); k[0] += 1
        dst[o:o + n].copy_(a)
    w = t(stream, reps)
    print(f"{mib:5d} MiB: read-only {n/s/1e12:5.2f} TB/s | copy in place {2*n/c/1e12:5.2f} TB/s (r+w) | hot source -> streaming destination {2*n/w/1e12:5.2f} TB/s (r+w)")
