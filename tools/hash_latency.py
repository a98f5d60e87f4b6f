#!/usr/bin/env python
"""Latency of ONE Longtail_HashAPI::HashBuffer call of the HIP plugin on a host buffer that no chunker handed out (what
Longtail_CreateMissingContent does per block with the block's chunk hashes, src/longtail.c:6801-6860, and CreateVersionIndex per path):
the small-input path of plugin_hash.c -- copy into the thread's pinned block, one launch that reads it over the link and writes the
digest back, one wait.  usage: tools/hash_latency.py"""
import ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch  # noqa: F401  (first: the library binds to torch's HIP runtime)
from longtail_amd.lib import load
from tests._libs import oracle

d = load().dll
d.Longtail_CreateHipBlake3HashAPI.restype = C.c_void_p
api = d.Longtail_CreateHipBlake3HashAPI()
fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64))(C.cast(api + 8 * 5, C.POINTER(C.c_void_p))[0])
o = oracle()
rng = np.random.default_rng(1)
for n in (16, 256, 1024, 8192, 65536):
    bufs = [rng.integers(0, 256, n, dtype=np.uint8) for _ in range(64)]
    out = C.c_uint64(0)
    for b in bufs[:8]:
        assert fn(api, n, b.ctypes.data, C.byref(out)) == 0
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 0.5:
        for b in bufs:
            fn(api, n, b.ctypes.data, C.byref(out))
        reps += len(bufs)
    dt = (time.perf_counter() - t0) / reps
    fn(api, n, bufs[3].ctypes.data, C.byref(out))
    ok = out.value == int(o.blake3(bufs[3]))
    print(f"HashBuffer {n:6d} bytes: {dt * 1e6:7.1f} us per call {'ok' if ok else 'MISMATCH'}")
