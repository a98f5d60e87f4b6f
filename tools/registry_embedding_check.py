#!/usr/bin/env python
"""The embedding exactly as INTEGRATION.md prints it, executed (own process: the allocator goes in before the first plugin
allocation).  For LZ4- and zstd-tagged trees at W = 0 and 4: Longtail_CreateDefaultCompressionRegistry over the EXPORTED
Longtail_CompressionRegistry_CreateForHipLZ4 / ...CreateForHipZstd (+ the reference's LZ4 factory) and a Longtail_CreateDefaultHashRegistry
entry for Longtail_GetBlake3HashType() holding Longtail_CreateHipBlake3HashAPI() (oracle/ref_harness.c refh_ingest_registry_embedding;
lib/compressionregistry/longtail_compression_registry.c:50-146, lib/hashregistry/longtail_hash_registry.c:41-67), UpSync, restore
through a reference-only registry, byte compare; the registries dispose the objects.  Afterwards nothing is pinned and the reference's
memtracer has nothing outstanding.  Prints `embedding ok ...`."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401

from longtail_amd.lib import load
from tests._libs import oracle, ref

o, r, lib = oracle(), ref(), load()
d, rd = lib.dll, r.dll
for f in ("refh_alloc_ptr", "refh_free_ptr"):
    getattr(rd, f).restype = C.c_void_p
rd.refh_memtrace_outstanding.restype = C.c_uint64
d.Longtail_Hip_PinnedBytes.restype = C.c_uint64
rd.refh_memtrace_begin()
d.Longtail_Hip_SetAllocator.argtypes = [C.c_void_p, C.c_void_p]
d.Longtail_Hip_SetAllocator(rd.refh_alloc_ptr(), rd.refh_free_ptr())
base = rd.refh_memtrace_outstanding()
files = [(f"d{i % 3}/f{i:02d}.bin", o.synth(int(n), 90 + i, (1, 0, 2, 12)[i % 4])) for i, n in enumerate([0, 100, 70000, 1 << 20, 3 << 20, 5, 2 << 20, 9 << 20])]
runs = 0
for workers in (0, 4):
    for tag, name in ((r.lz4_type, "lz42"), (r.zstd_default, "ztd2"), (int(rd.refh_zstd_type(3)), "ztd4")):
        want = r.ingest_roundtrip(files, 65536, 1 << 20, 64, tag, workers)
        got = r.ingest_registry_embedding(d, files, 65536, 1 << 20, 64, tag, workers)
        assert want["err"] == 0 and got["err"] == 0, (name, workers, want, got)
        assert got["chunks"] == want["chunks"] and got["blocks"] == want["blocks"], (name, got, want)
        assert got["apis_created"] == 1, got  # ONE CompressionAPI per type used, created lazily by the registry (compression_registry.c:69-97)
        # disposed with the registry: the codec dispatcher, the per-thread contexts of the bikeshed workers that have exited, their pins
        assert d.Longtail_Hip_PinnedBytes() == 0, (name, workers, d.Longtail_Hip_PinnedBytes())
        runs += 1
out = rd.refh_memtrace_outstanding() - base
print("embedding ok", runs, "outstanding", out, "pinned", d.Longtail_Hip_PinnedBytes())
sys.exit(0 if out == 0 else 1)
