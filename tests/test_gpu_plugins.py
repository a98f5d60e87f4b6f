"""-m gpu: the drop-in boundary.  The HIP Longtail_ChunkerAPI / HashAPI / CompressionAPI objects are plugged into the
UNMODIFIED reference core (oracle/_ref/liblongtail_ref.so, built from /root/reference by oracle/Makefile) exactly
where the CPU plugins go; results must be byte-identical (VersionIndex) / round-trip through the reference decoder."""
import ctypes as C
import errno

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def params(target):
    return max(48, target // 8), max(48, target // 2), max(48, target * 2)


class HashAPIStruct(C.Structure):
    _fields_ = [
        ("Dispose", C.CFUNCTYPE(None, C.c_void_p)),
        ("GetIdentifier", C.CFUNCTYPE(C.c_uint32, C.c_void_p)),
        ("BeginContext", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p))),
        ("Hash", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p)),
        ("EndContext", C.CFUNCTYPE(C.c_uint64, C.c_void_p, C.c_void_p)),
        ("HashBuffer", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64))),
    ]


class CompressionAPIStruct(C.Structure):
    _fields_ = [
        ("Dispose", C.CFUNCTYPE(None, C.c_void_p)),
        ("GetMaxCompressedSize", C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_uint32, C.c_size_t)),
        ("Compress", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t))),
        ("Decompress", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t))),
    ]


class ChunkerAPIStruct(C.Structure):
    _fields_ = [
        ("Dispose", C.CFUNCTYPE(None, C.c_void_p)),
        ("GetMinChunkSize", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32))),
    ]


@pytest.fixture(scope="module")
def plugins(gpu, hiplib):
    d = hiplib.dll
    chunker = d.Longtail_CreateHipChunkerAPI()
    hasher = d.Longtail_CreateHipBlake3HashAPI()
    lz4 = d.Longtail_CreateHipLZ4CompressionAPI()
    zstd = d.Longtail_CreateHipZStdCompressionAPI()
    assert chunker and hasher and lz4 and zstd, "plugin constructors returned NULL on a GPU box"
    yield dict(chunker=chunker, hash=hasher, lz4=lz4, zstd=zstd)
    for k, st in (("chunker", ChunkerAPIStruct), ("hash", HashAPIStruct), ("lz4", CompressionAPIStruct), ("zstd", CompressionAPIStruct)):
        ptr = dict(chunker=chunker, hash=hasher, lz4=lz4, zstd=zstd)[k]
        st.from_address(ptr).Dispose(ptr)


def test_identifiers(plugins, golden, hiplib):
    h = HashAPIStruct.from_address(plugins["hash"])
    assert h.GetIdentifier(plugins["hash"]) == golden["tests"]["blake3_id"] == 0x626C6B33  # longtail_blake3.c:6
    assert hiplib.dll.Longtail_GetHipLZ4DefaultQuality() == golden["tests"]["lz4_type"]      # longtail_lz4.c:10
    mn = C.c_uint32(0)
    c = ChunkerAPIStruct.from_address(plugins["chunker"])
    assert c.GetMinChunkSize(plugins["chunker"], C.byref(mn)) == 0 and mn.value == 48       # hpcdcchunker.c:343


def test_hash_api_entry_points(plugins, oracle, golden):
    ptr = plugins["hash"]
    h = HashAPIStruct.from_address(ptr)
    kat = golden["tests"]["blake3_kat"]
    s = kat["string_plus_nul"].encode() + b"\0"
    out = C.c_uint64(0)
    assert h.HashBuffer(ptr, len(s), s, C.byref(out)) == 0
    assert "%016x" % out.value == kat["hash_hex"]
    empty = b"\0"
    assert h.HashBuffer(ptr, 0, empty, C.byref(out)) == 0  # empty asset content hash (src/longtail.c:2521-2522)
    assert out.value == oracle.blake3(np.zeros(0, np.uint8))
    data = oracle.synth(300000, 8, 0)
    for n in (1, 64, 1024, 1025, 70000, 300000):
        assert h.HashBuffer(ptr, n, data.ctypes.data, C.byref(out)) == 0
        assert out.value == oracle.blake3(data[:n]), n
    # streaming trio (longtail_blake3.c:24-79)
    ctx = C.c_void_p()
    assert h.BeginContext(ptr, C.byref(ctx)) == 0
    for a, b in ((0, 1000), (1000, 1001), (1001, 150000), (150000, 300000)):
        h.Hash(ptr, ctx, b - a, data[a:b].ctypes.data)
    assert h.EndContext(ptr, ctx) == oracle.blake3(data)


@pytest.mark.parametrize("which", ["hip+hip", "hip+cpu", "cpu+hip"])
def test_chunker_hash_pairings_through_reference_driver(plugins, ref, oracle, which, golden):
    """NextChunk/HashBuffer exactly as DynamicChunking drives them (src/longtail.c:2231-2296), all plugin pairings."""
    chunker = plugins["chunker"] if which.startswith("hip") else None
    hasher = plugins["hash"] if which.endswith("hip") else None
    ci = golden["tests"]["chunker_input"]
    offs, lens, hashes = ref.chunk_and_hash(golden["chunker_input"], ci["min"], ci["avg"], ci["max"], chunker, hasher)
    assert [(int(a), int(b)) for a, b in zip(offs, lens)] == [tuple(r) for r in ci["ranges"]]
    for kind, size, target in [(0, 3 << 20, 65536), (1, (2 << 20) + 99, 32768), (2, 1 << 20, 65536), (0, 1000, 65536), (0, 0, 65536),
                               (1, 100000, 16)]:
        data = oracle.synth(size, 321 + size, kind)
        mn, av, mx = params(target)
        e = ref.chunk_and_hash(data, mn, av, mx)
        g = ref.chunk_and_hash(data, mn, av, mx, chunker, hasher)
        assert all((a == b).all() for a, b in zip(e, g)) and len(e[1]) == len(g[1]), (which, kind, size, target)


def test_chunker_streams_longer_than_one_window(plugins, ref, oracle):
    """> 64 MiB through ONE chunker: windows must reproduce the reference's stream semantics."""
    data = oracle.synth((64 << 20) * 2 + 12345, 99, 1)
    mn, av, mx = params(65536)
    e = ref.chunk_and_hash(data, mn, av, mx)
    g = ref.chunk_and_hash(data, mn, av, mx, plugins["chunker"], plugins["hash"])
    assert len(e[1]) == len(g[1]) and all((a == b).all() for a, b in zip(e, g))


def test_next_chunk_from_buffer(plugins, ref, oracle):
    for kind, size, target in [(0, 2 << 20, 65536), (1, 1 << 20, 32768), (0, 3000, 16)]:
        data = oracle.synth(size, 55 + size, kind)
        mn, av, mx = params(target)
        e = ref.chunk_from_buffer(data, mn, av, mx)
        g = ref.chunk_from_buffer(data, mn, av, mx, plugins["chunker"])
        assert len(e) == len(g) and (e == g).all()


def make_tree(oracle, spec, seed=1):
    files = []
    for i, (size, kind) in enumerate(spec):
        files.append((f"dir{i % 3}/sub{i % 2}/file{i:04d}.bin", oracle.synth(size, oracle.asset_seed(seed, i), kind)))
    return files


@pytest.mark.parametrize("workers", [0, 4])
def test_version_index_is_byte_identical(plugins, ref, oracle, workers):
    """Longtail_CreateVersionIndex with HIP plugins == with CPU plugins, serialized byte for byte."""
    spec = [(0, 0), (1, 0), (47, 1), (48, 0), (49, 1), (1000, 0), (65536, 1), (200000, 0), (1 << 20, 1), (3 << 20, 0), (1 << 20, 2),
            (1 << 20, 1)]
    files = make_tree(oracle, spec)
    files.append(("dup/copy.bin", files[8][1].copy()))  # duplicate content -> dedup path (src/longtail.c:2951-2970)
    for target in (65536, 4096):  # 4096 -> 4 MiB parts: the 3 MiB+ assets stay single-part, use 1024 for multi-part
        cpu, _ = ref.version_index(files, target, workers=workers, tag=ref.lz4_type)
        hip, _ = ref.version_index(files, target, workers=workers, tag=ref.lz4_type, chunker_api=plugins["chunker"],
                                   hash_api=plugins["hash"])
        assert cpu == hip, f"VersionIndex differs (target {target}, workers {workers})"
    cpu, _ = ref.version_index(files, 1024, workers=workers)  # 1 MiB parts -> multi-part assets, empty trailing parts
    hip, _ = ref.version_index(files, 1024, workers=workers, chunker_api=plugins["chunker"], hash_api=plugins["hash"])
    assert cpu == hip


def test_write_content_roundtrips_through_reference(plugins, ref, oracle):
    """UpSync sequence with HIP chunker+hash+LZ4, then restore with a REFERENCE-ONLY registry and compare files."""
    spec = [(0, 0), (100, 1), (70000, 1), (1 << 20, 1), (2 << 20, 0), (1 << 20, 2), (300000, 1), (5, 0)]
    files = make_tree(oracle, spec, seed=7)
    for block_size, per_block in ((8 << 20, 1024), (262144, 64)):
        base = ref.ingest_roundtrip(files, 65536, block_size, per_block, ref.lz4_type)
        assert base["err"] == 0
        got = ref.ingest_roundtrip(files, 65536, block_size, per_block, ref.lz4_type, workers=2, chunker_api=plugins["chunker"],
                                   hash_api=plugins["hash"], codec_api=plugins["lz4"])
        assert got["err"] == 0, got
        assert got["chunks"] == base["chunks"] and got["blocks"] == base["blocks"]
        assert got["stored_bytes"] <= base["stored_bytes"] * 1.25 + 4096


def test_zstd_plugin_compress_decodes_with_reference(plugins, ref, oracle):
    api = CompressionAPIStruct.from_address(plugins["zstd"])
    tag = ref.zstd_default
    for n, kind in ((0, 0), (1000, 1), (400000, 1), (400000, 2)):
        d = oracle.synth(n, 12 + n, kind)
        cap = api.GetMaxCompressedSize(plugins["zstd"], tag, n)
        assert cap == ref.dll.refh_codec_bound(1, tag, n)  # ZSTD_COMPRESSBOUND
        out = np.zeros(cap + 8, np.uint8)
        got = C.c_size_t(0)
        assert api.Compress(plugins["zstd"], tag, d.ctypes.data if n else out.ctypes.data, out.ctypes.data, n, cap, C.byref(got)) == 0
        err, back = ref.decompress(1, out[: got.value].copy(), n)
        assert err == 0 and len(back) == n and (back == d).all()


def test_zstd_plugin_settings_select_the_parse(plugins, ref, oracle):
    """'ztd4' and 'ztd3' through Longtail_CompressionAPI::Compress: smaller than 'ztd2' on data with redundancy beyond a half's own
    32 KiB (a vocabulary of tokens), decoded by the reference; an unknown 'ztd?' id compresses like the default (longtail_zstd.c:43-60:
    `default: return 0`)."""
    api = CompressionAPIStruct.from_address(plugins["zstd"])
    d = oracle.synth(3 << 20, 21, 12)
    sizes = {}
    for w, name in ((1, "ztd2"), (3, "ztd4"), (2, "ztd3")):
        tag = int(ref.dll.refh_zstd_type(w))
        cap = api.GetMaxCompressedSize(plugins["zstd"], tag, len(d))
        out = np.zeros(cap + 8, np.uint8)
        got = C.c_size_t(0)
        assert api.Compress(plugins["zstd"], tag, d.ctypes.data, out.ctypes.data, len(d), cap, C.byref(got)) == 0
        err, back = ref.decompress(1, out[: got.value].copy(), len(d))
        assert err == 0 and (back == d).all()
        # ... and by this plugin's own Decompress, one block per call ('ztd3' frames: chains of pieces, k_zstd.hip)
        mine = np.zeros(len(d) + 8, np.uint8)
        n_back = C.c_size_t(0)
        assert api.Decompress(plugins["zstd"], out.ctypes.data, mine.ctypes.data, got.value, len(d), C.byref(n_back)) == 0
        assert n_back.value == len(d) and (mine[: len(d)] == d).all()
        sizes[name] = got.value
    assert sizes["ztd4"] < 0.95 * sizes["ztd2"] and sizes["ztd3"] < 0.95 * sizes["ztd4"], sizes  # (history inside the piece; across the pieces)
    tag = 0x7A746439  # 'ztd9'
    out = np.zeros(len(d) + (len(d) >> 8) + 72, np.uint8)
    got = C.c_size_t(0)
    assert api.Compress(plugins["zstd"], tag, d.ctypes.data, out.ctypes.data, len(d), len(out) - 8, C.byref(got)) == 0
    assert got.value == sizes["ztd2"]


def test_zstd_plugin_decompress_reads_reference_frames(plugins, ref, oracle):
    """ZStdCompressionAPI_Decompress on the GPU: frames written by the REFERENCE encoder at every longtail setting
    ('ztd1'..'ztd5', lib/zstd/longtail_zstd.c:12-22) and frames written by the HIP encoder itself; malformed -> EINVAL."""
    ptr = plugins["zstd"]
    api = CompressionAPIStruct.from_address(ptr)
    for n, kind in ((0, 0), (1000, 1), (400000, 1), (400000, 2), (700000, 12), ((1 << 20) + 3, 11)):
        d = oracle.synth(n, 12 + n, kind)
        frames = [ref.compress(1, ref.dll.refh_zstd_type(w), d) for w in range(5)]
        cap = api.GetMaxCompressedSize(ptr, ref.zstd_default, n)
        own = np.zeros(cap + 8, np.uint8)
        got = C.c_size_t(0)
        assert api.Compress(ptr, ref.zstd_default, d.ctypes.data if n else own.ctypes.data, own.ctypes.data, n, cap, C.byref(got)) == 0
        frames.append(own[: got.value].copy())
        for f in frames:
            back = np.zeros(n + 8, np.uint8)
            m = C.c_size_t(0)
            assert api.Decompress(ptr, f.ctypes.data, back.ctypes.data, len(f), n, C.byref(m)) == 0
            assert m.value == n and (back[:n] == d).all()
        if n > 1000:
            bad = frames[0].copy()
            bad[len(bad) // 2 :] = 0
            m = C.c_size_t(0)
            assert api.Decompress(ptr, bad.ctypes.data, np.zeros(n + 8, np.uint8).ctypes.data, len(bad), n, C.byref(m)) == errno.EINVAL


def test_lz4_plugin_entry_points(plugins, ref, oracle):
    ptr = plugins["lz4"]
    api = CompressionAPIStruct.from_address(ptr)
    tag = ref.lz4_type
    for n, kind in ((1, 0), (5858, 2), (100000, 1), (1 << 20, 0), ((8 << 20) + 800000, 1)):
        d = oracle.synth(n, 3 + n, kind)
        cap = api.GetMaxCompressedSize(ptr, tag, n)
        assert cap == n + n // 255 + 16  # LZ4_COMPRESSBOUND
        out = np.zeros(cap + 12, np.uint8)
        got = C.c_size_t(0)
        # destination only 4-byte aligned like &header[2] (compressblockstore.c:117-125)
        assert api.Compress(ptr, tag, d.ctypes.data, out.ctypes.data + 4, n, cap, C.byref(got)) == 0
        payload = out[4 : 4 + got.value].copy()
        err, back = ref.decompress(0, payload, n)
        assert err == 0 and len(back) == n and (back == d).all()
        # HIP Decompress on a REFERENCE payload (tags are persisted: both directions must interoperate)
        refp = ref.compress(0, tag, d)
        back2 = np.zeros(n + 8, np.uint8)
        got2 = C.c_size_t(0)
        assert api.Decompress(ptr, refp.ctypes.data, back2.ctypes.data, len(refp), n, C.byref(got2)) == 0
        assert got2.value == n and (back2[:n] == d).all()
    # errors: too small a destination -> ENOMEM (longtail_lz4.c:70-74); garbage -> EBADF (:95-99)
    d = oracle.synth(50000, 1, 0)
    out = np.zeros(60000, np.uint8)
    got = C.c_size_t(0)
    assert api.Compress(ptr, tag, d.ctypes.data, out.ctypes.data, len(d), 100, C.byref(got)) == errno.ENOMEM
    junk = np.full(100, 0xFF, np.uint8)
    assert api.Decompress(ptr, junk.ctypes.data, out.ctypes.data, len(junk), 1000, C.byref(got)) == errno.EBADF


def test_plugins_do_not_leak_under_the_reference_memtracer():
    """Own process: Longtail_Hip_SetAllocator(Longtail_Alloc, Longtail_Free) + lib/memtracer around VersionIndex and UpSync
    runs with 0 and 4 bikeshed workers; after disposing the four HIP API objects nothing they allocated is outstanding."""
    import subprocess
    import sys

    from tests._libs import ROOT

    out = subprocess.run([sys.executable, str(ROOT / "tools" / "memtrace_check.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("outstanding")][-1].split()
    assert int(line[1]) == 0, line
    assert int(line[-1]) >= 4  # the objects were really counted while alive


def test_registry_embedding_as_integration_md_prints_it():
    """INTEGRATION.md's embedding, executed: the reference's Longtail_CreateDefaultCompressionRegistry builds its CompressionAPI objects
    through the EXPORTED Longtail_CompressionRegistry_CreateForHipLZ4 / ...HipZstd factories and owns them, the hash API comes out of a
    Longtail_CreateDefaultHashRegistry entry for Longtail_GetBlake3HashType(); UpSync + reference-only restore on LZ4- and zstd-tagged
    trees at W = 0 and 4; after the registries are disposed nothing is pinned and the memtracer has nothing outstanding
    (lib/compressionregistry/longtail_compression_registry.c:50-146, lib/hashregistry/longtail_hash_registry.c:41-67)."""
    import subprocess
    import sys

    from tests._libs import ROOT

    out = subprocess.run([sys.executable, str(ROOT / "tools" / "registry_embedding_check.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("embedding ok")][-1].split()
    assert int(line[2]) == 6 and int(line[4]) == 0 and int(line[6]) == 0, line


def test_blocking_waits_first_call_then_the_drop_in_sequence():
    """Longtail_Hip_SetBlockingWaits(1) as an embedder's FIRST call (include/longtail_hip.h: the device's wait policy, set before the
    process touches the GPU), then the three plugin objects in the unmodified core -- in a process of its own, which is how bench.py
    measures its drop_in legs (tools/drop_in_child.py): the call succeeds, UpSync + the CPU-codec pairing run, the HIP codec's stored
    bytes are a real compression of the compressible sample."""
    import json
    import subprocess
    import sys

    from tests._libs import ROOT

    req = {"cfg": {"tree": "files", "kind": "mixed", "codec": "lz4", "file_mib": 1.0, "gib": 0.125, "dups": False}, "sample_bytes": 128 << 20,
           "workers": 8, "more_workers": [16], "reps": 1, "target_chunk_size": 65536, "block_size": 8 << 20, "max_chunks_per_block": 1024}
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "drop_in_child.py"), json.dumps(req)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["blocking_waits_rc"] == 0 and "error" not in j, j
    assert set(j["hip"]) == {"8", "16"} and j["hip"]["8"]["GBps"] > 0 and j["hip_chunker_hash_cpu_codec"]["GBps"] > 0
    assert j["nbytes"] == 128 << 20 and 0 < j["hip_stored_bytes"] < 0.7 * j["hip_raw_bytes"]


def _stream(api_ptr, pieces):
    """BeginContext / Hash... / EndContext over an iterable of numpy byte arrays, as longtail_blake3.c:24-79 is driven."""
    h = HashAPIStruct.from_address(api_ptr)
    ctx = C.c_void_p()
    assert h.BeginContext(api_ptr, C.byref(ctx)) == 0
    for p in pieces:
        if len(p):
            h.Hash(api_ptr, ctx, len(p), p.ctypes.data)
    return h.EndContext(api_ptr, ctx)


def _cut(data, step):
    return [data[o : o + step] for o in range(0, len(data), step)] if len(data) else []


def test_streaming_context_matches_the_reference_hasher(plugins, ref, oracle, hiplib):
    """The streaming trio with O(1) state (round 3: batches of 1 MiB reduced on the device, the subtree stack in device memory) against
    the reference's own Blake3Hash_BeginContext/_Hash/_EndContext: every size around the leaf, the one-launch limit and the batch."""
    ref.dll.Longtail_CreateBlake3HashAPI.restype = C.c_void_p
    cpu = ref.dll.Longtail_CreateBlake3HashAPI()
    assert cpu
    try:
        data = oracle.synth((17 << 20) + 1, 4242, 1)
        mib = 1 << 20
        for n in (0, 1, 63, 64, 1023, 1024, 1025, 65535, 65536, 65537, mib - 1, mib, mib + 1, 2 * mib - 1, 2 * mib, 2 * mib + 1, 3 * mib + 5,
                  4 * mib, 8 * mib + 1023, 16 * mib, 17 * mib + 1):
            for step in (n or 1, 700001, 4096 + 17):
                if step > 100000 or n <= 3 * mib + 5:
                    got = _stream(plugins["hash"], _cut(data[:n], step))
                    want = _stream(cpu, _cut(data[:n], step))
                    assert got == want == oracle.blake3(data[:n]), (n, step)
        assert hiplib.dll.Longtail_Hip_GetLastError() == 0
    finally:
        HashAPIStruct.from_address(cpu).Dispose(cpu)


def test_streaming_context_5_gib(plugins, ref, oracle, hiplib):
    """More than 4 GiB through one context (round 2 stopped there with EFBIG and buffered the whole stream on the host): 5 GiB + 3 bytes
    fed in 64 MiB pieces, against the reference's hasher fed the same pieces."""
    ref.dll.Longtail_CreateBlake3HashAPI.restype = C.c_void_p
    cpu = ref.dll.Longtail_CreateBlake3HashAPI()
    assert cpu
    try:
        base = oracle.synth(64 << 20, 99, 0)

        def pieces():
            for i in range(80):  # 80 x 64 MiB = 5 GiB, every piece different (rotated)
                yield np.roll(base, i * 4099)
            yield base[:3]

        got = _stream(plugins["hash"], pieces())
        want = _stream(cpu, pieces())
        assert got == want and got != 0
        assert hiplib.dll.Longtail_Hip_GetLastError() == 0
    finally:
        HashAPIStruct.from_address(cpu).Dispose(cpu)


def test_codec_plugins_many_threads_at_once(plugins, ref, oracle, hiplib):
    """Compress / Decompress of both HIP codec objects from sixteen threads at once: the calls meet in the codec dispatcher
    (plugin_codec_batch.c: one bulk submission for whatever is queued, offsets relative to the lowest queued address).  Every payload
    decodes with the reference decoder and through the plugin itself; reference-made payloads (matches across the pieces: the origin
    passes) come back too."""
    import threading

    jobs = []
    for i in range(16):
        codec = "zstd" if i % 2 else "lz4"
        n = 200000 + 300007 * i
        jobs.append((codec, oracle.synth(n, 900 + i, (1, 11, 12)[i % 3])))
    errors = []

    def work(codec, d):
        try:
            ptr = plugins[codec]
            api = CompressionAPIStruct.from_address(ptr)
            tag = ref.zstd_default if codec == "zstd" else ref.lz4_type
            n = len(d)
            cap = api.GetMaxCompressedSize(ptr, tag, n)
            for rnd in range(3):
                out = np.zeros(cap + 8, np.uint8)
                got = C.c_size_t(0)
                assert api.Compress(ptr, tag, d.ctypes.data, out.ctypes.data + 4, n, cap, C.byref(got)) == 0
                own = out[4 : 4 + got.value].copy()
                err, back = ref.decompress(1 if codec == "zstd" else 0, own, n)
                assert err == 0 and len(back) == n and (back == d).all()
                theirs = ref.compress(1, tag, d) if codec == "zstd" else oracle.lz4_compress(d)
                for f in (own, theirs):
                    res = np.zeros(n + 8, np.uint8)
                    m = C.c_size_t(0)
                    assert api.Decompress(ptr, f.ctypes.data, res.ctypes.data + 4, len(f), n, C.byref(m)) == 0
                    assert m.value == n and (res[4 : 4 + n] == d).all()
        except BaseException as e:  # noqa: BLE001 (collected for the main thread)
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    subs, blocks = C.c_uint64(0), C.c_uint64(0)
    hiplib.dll.Longtail_Hip_CodecBatchStats(C.byref(subs), C.byref(blocks))
    assert blocks.value >= 16 * 3 * 3 and 1 <= subs.value <= blocks.value
