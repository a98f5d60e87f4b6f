#!/usr/bin/env python
"""Leak check of the HIP plugins with the reference's memtracer (run as its own process: the allocator must be installed
before the first plugin allocation).  Prints `outstanding <n>` after all API objects are disposed."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch  # noqa: F401

from longtail_amd.lib import load
from tests._libs import oracle, ref

o, r, lib = oracle(), ref(), load()
d, rd = lib.dll, r.dll
for f in ("refh_alloc_ptr", "refh_free_ptr"):
    getattr(rd, f).restype = C.c_void_p
rd.refh_memtrace_outstanding.restype = C.c_uint64
rd.refh_memtrace_begin()
d.Longtail_Hip_SetAllocator.argtypes = [C.c_void_p, C.c_void_p]
d.Longtail_Hip_SetAllocator(rd.refh_alloc_ptr(), rd.refh_free_ptr())
for f in ("Longtail_CreateHipChunkerAPI", "Longtail_CreateHipBlake3HashAPI", "Longtail_CreateHipLZ4CompressionAPI",
          "Longtail_CreateHipZStdCompressionAPI"):
    getattr(d, f).restype = C.c_void_p
base = rd.refh_memtrace_outstanding()
objs = [d.Longtail_CreateHipChunkerAPI(), d.Longtail_CreateHipBlake3HashAPI(), d.Longtail_CreateHipLZ4CompressionAPI(),
        d.Longtail_CreateHipZStdCompressionAPI()]
assert all(objs)
files = [(f"d{i % 3}/f{i:02d}.bin", o.synth(int(n), 40 + i, i % 3)) for i, n in enumerate([0, 100, 70000, 1 << 20, 3 << 20, 5, 2 << 20])]
for workers in (0, 4):
    vi_hip, _ = r.version_index(files, 65536, workers, r.lz4_type, objs[0], objs[1])
    vi_cpu, _ = r.version_index(files, 65536, workers, r.lz4_type)
    assert vi_hip == vi_cpu
    for codec, tag in ((objs[2], r.lz4_type), (objs[3], r.zstd_default)):
        res = r.ingest_roundtrip(files, 65536, 1 << 20, 64, tag, workers, objs[0], objs[1], codec)
        assert res["err"] == 0, res
during = rd.refh_memtrace_outstanding()
Dispose = C.CFUNCTYPE(None, C.c_void_p)
for p in objs:
    Dispose(C.c_void_p.from_address(p).value)(p)
print("outstanding", rd.refh_memtrace_outstanding() - base, "held while alive", during - base)
