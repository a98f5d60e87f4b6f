#!/usr/bin/env python
"""Latency of ONE Longtail_CompressionAPI::Compress / Decompress call of the HIP plugins on host buffers (what WriteContentBlockJob and
the compress block store do per block, src/longtail.c:4559-4758, compressblockstore.c:271-338), one caller, 8 MiB blocks.
usage: tools/codec_call_latency.py"""
import ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch  # noqa: F401
from longtail_amd.lib import load
from tests._libs import oracle

d = load().dll
o = oracle()
FN_C = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t))
FN_D = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t))
FN_B = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_uint32, C.c_size_t)
for name, ctor, tag in (("lz4", d.Longtail_CreateHipLZ4CompressionAPI, 0x6C7A3432), ("zstd", d.Longtail_CreateHipZStdCompressionAPI, 0x7A746432)):
    ctor.restype = C.c_void_p
    api = ctor()
    vt = C.cast(api, C.POINTER(C.c_void_p))
    bound, comp, dec = FN_B(vt[1]), FN_C(vt[2]), FN_D(vt[3])
    for kind, kname in ((0, "random"), (1, "mixed")):
        n = 8 << 20
        src = o.synth(n, 7, kind)
        cap = bound(api, tag, n)
        out = np.zeros(cap + 8, np.uint8)
        back = np.zeros(n + 8, np.uint8)
        got, m = C.c_size_t(0), C.c_size_t(0)
        for _ in range(3):
            assert comp(api, tag, src.ctypes.data, out.ctypes.data, n, cap, C.byref(got)) == 0
        t0 = time.perf_counter()
        for _ in range(10):
            comp(api, tag, src.ctypes.data, out.ctypes.data, n, cap, C.byref(got))
        tc = (time.perf_counter() - t0) / 10
        for _ in range(3):
            assert dec(api, out.ctypes.data, back.ctypes.data, got.value, n, C.byref(m)) == 0
        t0 = time.perf_counter()
        for _ in range(10):
            dec(api, out.ctypes.data, back.ctypes.data, got.value, n, C.byref(m))
        td = (time.perf_counter() - t0) / 10
        assert m.value == n and (back[:n] == src).all()
        print(f"{name:4s} {kname:6s} 8 MiB: Compress {tc * 1e3:6.2f} ms ({n / tc / 1e9:5.2f} GB/s), Decompress {td * 1e3:6.2f} ms ({n / td / 1e9:5.2f} GB/s), ratio {n / got.value:.3f}")
