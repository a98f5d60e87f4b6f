#!/usr/bin/env python
"""PCIe-inclusive rate of the DROP-IN path: the unmodified reference core (oracle/_ref) driving the HIP
ChunkerAPI + HashAPI (+ LZ4 / ZStd CompressionAPI) through host buffers, next to the reference's own CPU plugins, same tree,
same worker count.  usage: tools/plugin_rate.py [files] [file_mib] [workers] [kind] [codec] [tmpfs|mem] [index]
(`index`: CreateVersionIndex only, one repetition after a warm-up on the first 1024 files -- the 64 GiB headline tree)"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401  (loads the HIP runtime first)

from bench import KINDS
from longtail_amd.lib import load
from tests._libs import oracle, ref

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 32
kind = sys.argv[4] if len(sys.argv) > 4 else "random"
codec = sys.argv[5] if len(sys.argv) > 5 else "lz4"
o, r, lib = oracle(), ref(), load()
d = lib.dll
storage = "reference in-memory storage (one lock around every read)"
if len(sys.argv) > 6 and sys.argv[6] == "tmpfs":
    r.dll.refh_set_tree_dir.argtypes = [C.c_char_p]
    r.dll.refh_set_tree_dir(b"/dev/shm")
    storage = "reference file storage on tmpfs"
for f in ("Longtail_CreateHipChunkerAPI", "Longtail_CreateHipBlake3HashAPI", "Longtail_CreateHipLZ4CompressionAPI", "Longtail_CreateHipZStdCompressionAPI"):
    getattr(d, f).restype = C.c_void_p
chunker, hasher = d.Longtail_CreateHipChunkerAPI(), d.Longtail_CreateHipBlake3HashAPI()
codec_api = d.Longtail_CreateHipLZ4CompressionAPI() if codec == "lz4" else d.Longtail_CreateHipZStdCompressionAPI()
tag = r.lz4_type if codec == "lz4" else r.zstd_default
assert chunker and hasher and codec_api
files = [(f"d{i % 8}/f{i:04d}.bin", o.synth(mib << 20, 500 + i, KINDS[kind])) for i in range(nfiles)]
total = sum(len(b) for _, b in files)
print(f"{nfiles} x {mib} MiB {kind} files, {workers} bikeshed workers, {codec}, source tree in {storage}")
index_only = len(sys.argv) > 7 and sys.argv[7] == "index"
if index_only:
    r.version_index(files[:1024], 65536, workers, 0, chunker, hasher)  # warm-up: contexts, window pool, dispatcher
    vi_hip, t_hip = r.version_index(files, 65536, workers, 0, chunker, hasher)
    vi_cpu, t_cpu = r.version_index(files, 65536, workers, 0)
else:
    for rep in range(2):
        vi_hip, t_hip = r.version_index(files, 65536, workers, 0, chunker, hasher)
        vi_cpu, t_cpu = r.version_index(files, 65536, workers, 0)
assert vi_hip == vi_cpu
print(f"CreateVersionIndex {total / 2**30:.1f} GiB: HIP plugins {total / t_hip / 1e9:.2f} GB/s, reference CPU plugins {total / t_cpu / 1e9:.2f} GB/s "
      f"(VersionIndex identical); pinned window memory held by the HIP chunkers: {d.Longtail_Hip_PinnedBytes() / 2**20:.0f} MiB")
bs = (C.c_uint64 * 4)()
d.Longtail_Hip_BatchStats(C.byref(bs, 0), C.byref(bs, 8))
d.Longtail_Hip_MemoStats(C.byref(bs, 16), C.byref(bs, 24))
if bs[0]:
    print(f"small-window batcher: {bs[1]} windows in {bs[0]} submissions ({bs[1] / bs[0]:.1f} per submission); content-hash memo: {bs[2]} digest arrays kept, {bs[3]} HashBuffer calls answered")
if index_only:
    sys.exit(0)
for rep in range(2):
    res_h = r.ingest_roundtrip(files, 65536, 8 << 20, 1024, tag, workers, chunker, hasher, codec_api)
    res_c = r.ingest_roundtrip(files, 65536, 8 << 20, 1024, tag, workers)
cb = (C.c_uint64 * 2)()
d.Longtail_Hip_CodecBatchStats(C.byref(cb, 0), C.byref(cb, 8))
if cb[0]:
    print(f"codec dispatcher: {cb[1]} blocks in {cb[0]} submissions ({cb[1] / cb[0]:.1f} per submission)")
for name, res in (("HIP plugins", res_h), ("reference CPU plugins", res_c)):
    s = res["seconds_index"] + res["seconds_write"]
    print(f"UpSync (index + WriteContent, restore verified) {name}: err {res['err']}, index {res['seconds_index']:.2f} s, write {res['seconds_write']:.2f} s "
          f"({total / res['seconds_write'] / 1e9:.2f} GB/s), stored {res['stored_bytes'] / 2**20:.0f} MiB -> {total / s / 1e9:.2f} GB/s")
