#!/usr/bin/env python3
"""Does the box move bytes over PCIe in both directions at once, and by which engine?

secondary.host_fed (bench.py) streams slices host -> device and stored-block images device -> host.  The measured
slice time was the SUM of the two copies, not their maximum, so this probe times, on 4 GiB of pinned host memory:
  h2d / d2h alone           hipMemcpyAsync through torch (the SDMA engines)
  h2d + d2h together        two streams
  kernel d2h alone          lthip_gather_ranges with a pinned HOST destination (the CUs store over PCIe)
  kernel h2d alone          lthip_gather_ranges with a pinned HOST source
  SDMA h2d + kernel d2h     two streams
  kernel h2d + SDMA d2h     two streams
  kernel h2d + kernel d2h   two streams
Prints one JSON object (GB/s per direction and the aggregate)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from longtail_amd.lib import Context  # noqa: E402


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    n = int(gib * (1 << 30))
    dev = torch.device("cuda:0")
    h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_in.fill_(7)
    d_a = torch.empty(n, dtype=torch.uint8, device=dev)
    d_b = torch.full((n,), 3, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ctx1 = Context(0, stream=s1.cuda_stream)
    ctx2 = Context(0, stream=s2.cuda_stream)
    # ranges of 8 MiB (what a stored block is)
    R = 8 << 20
    offs = torch.arange(0, n, R, dtype=torch.int64, device=dev)
    lens = torch.full((offs.numel(),), R, dtype=torch.int32, device=dev)

    def sdma_h2d(stream):
        with torch.cuda.stream(stream):
            d_a.copy_(h_in, non_blocking=True)

    def sdma_d2h(stream):
        with torch.cuda.stream(stream):
            h_out.copy_(d_b, non_blocking=True)

    def kern_d2h(ctx):
        ctx.gather_ranges(d_b, offs, lens, h_out, offs)

    def kern_h2d(ctx):
        ctx.gather_ranges(h_in, offs, lens, d_a, offs)

    def timed(*legs):
        best = None
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for f in legs:
                f()
            torch.cuda.synchronize(dev)
            t = time.perf_counter() - t0
            best = t if best is None or t < best else best
        return best

    out = {"GiB": gib}

    def leg(name, *fs):
        t = timed(*fs)
        out[name] = {"ms": round(t * 1e3, 1), "GBps_per_direction": round(n / t / 1e9, 1), "GBps_total": round(len(fs) * n / t / 1e9, 1)}

    leg("sdma_h2d", lambda: sdma_h2d(s1))
    leg("sdma_d2h", lambda: sdma_d2h(s2))
    leg("sdma_h2d+sdma_d2h", lambda: sdma_h2d(s1), lambda: sdma_d2h(s2))
    leg("kernel_d2h", lambda: kern_d2h(ctx2))
    assert bool((h_out[:: 1 << 20] == 3).all()), "kernel d2h did not land"
    leg("kernel_h2d", lambda: kern_h2d(ctx1))
    torch.cuda.synchronize(dev)
    assert bool((d_a[:: 1 << 20] == 7).all().item()), "kernel h2d did not land"
    leg("sdma_h2d+kernel_d2h", lambda: sdma_h2d(s1), lambda: kern_d2h(ctx2))
    leg("kernel_h2d+sdma_d2h", lambda: kern_h2d(ctx1), lambda: sdma_d2h(s2))
    leg("kernel_h2d+kernel_d2h", lambda: kern_h2d(ctx1), lambda: kern_d2h(ctx2))
    # a compute kernel beside the kernel copy: does the copy starve it?  (an int64 sum over 4 GiB on the default stream)
    x = d_b.view(torch.int64)
    t_alone = timed(lambda: x.sum())
    t_with = timed(lambda: kern_d2h(ctx2), lambda: x.sum())
    out["sum_4GiB_alone_ms"] = round(t_alone * 1e3, 2)
    out["kernel_d2h_with_sum_ms"] = round(t_with * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
