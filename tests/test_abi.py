"""CPU-only checks of the drop-in boundary: the library loads, exports exactly what include/longtail_hip.h declares,
refuses to work without a GPU (no CPU fallback), and the host arithmetic the kernels rely on is exact."""
import ctypes as C
import errno
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "longtail_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#define LTHIP_EXPORT.*", "", text)
    names = re.findall(r"LTHIP_EXPORT[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)
    return sorted(set(names))


def test_header_symbols_are_exported(hiplib):
    names = declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(hiplib.dll, n)]
    assert not missing, missing


def test_product_reads_only_its_documented_switches(hiplib):
    """README.md "Environment": the product library reads eleven variables.  Every other LTHIP_* switch of rounds 1-4 (kernel flavours,
    debug masks, experiments) is compiled by the ablation build only (make ablations, csrc/ablations/*.inc) -- neither their names
    nor the kernels they select are in the shipped binary; the ablation build has both and the same C interface."""
    import subprocess

    def names(path):
        out = subprocess.run(["strings", "-a", str(path)], capture_output=True, text=True, check=True).stdout
        return sorted(set(re.findall(r"^(?:LTHIP|LONGTAIL_HIP)_[A-Z0-9_]+$", out, flags=re.M)))

    documented = sorted(re.findall(r"`((?:LTHIP|LONGTAIL_HIP)_[A-Z0-9_]+)`", (ROOT / "README.md").read_text().split("## Environment")[1].split("Everything else")[0]))
    documented += ["LONGTAIL_HIP_LARGE_WINDOWS", "LTHIP_COMM_SHM_SLOT"] if "LONGTAIL_HIP_LARGE_WINDOWS" not in documented else []
    got = names(hiplib.path)
    assert got == sorted(set(documented)), (got, documented)
    assert len(got) == 11
    symbols = subprocess.run(["strings", "-a", str(hiplib.path)], capture_output=True, text=True, check=True).stdout
    for gone in ("k_buzhash_candidates", "k_lz4_segments_modes", "k_lz4_pd_units", "k_zstd_prepare", "k_zstd_execute_payload", "k_blake3_parents_small"):
        assert gone not in symbols, f"{gone} is an ablation-only kernel"
    from longtail_amd.lib import ABLATIONS_LIB_PATH

    if ABLATIONS_LIB_PATH.exists():  # (built by make all / __graft_entry__.build())
        abl = C.CDLL(str(ABLATIONS_LIB_PATH))
        missing = [n for n in declared_symbols() if not hasattr(abl, n)]
        assert not missing, missing
        abl_names = names(ABLATIONS_LIB_PATH)
        assert set(got) <= set(abl_names) and len(abl_names) >= 35, abl_names
        abl_syms = subprocess.run(["strings", "-a", str(ABLATIONS_LIB_PATH)], capture_output=True, text=True, check=True).stdout
        assert "k_lz4_segments_modes" in abl_syms and "k_buzhash_candidates" in abl_syms


def test_library_is_built_from_this_tree(hiplib):
    """lthip_build_id() is the hash of csrc/ + include/ the Makefile baked in; a stale .so (round 1: k_zstd.o older than
    zstd_decode_core.h) must fail here and in tests/test_gpu_build_id.py on the GPU box."""
    import sys

    sys.path.insert(0, str(ROOT / "tools"))
    from build_id import build_id

    assert hiplib.build_id() == build_id(ROOT), "liblongtail_hip.so was not built from the sources beside it: run make"


def test_no_gpu_means_loud_failure(hiplib):
    """In the build container there is no GPU: every constructor must refuse instead of falling back to the CPU."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("this check is for the GPU-less container")
    d = hiplib.dll
    assert d.lthip_device_count() == 0
    h = C.c_void_p()
    assert d.lthip_ctx_create(0, C.c_void_p(-1), C.byref(h)) == errno.ENODEV and not h.value
    assert not d.Longtail_CreateHipChunkerAPI()
    assert not d.Longtail_CreateHipBlake3HashAPI()
    assert not d.Longtail_CreateHipLZ4CompressionAPI()
    assert not d.Longtail_CreateHipZStdCompressionAPI()
    st = C.c_uint32(0)
    assert not d.Longtail_CompressionRegistry_CreateForHipLZ4(0x6C7A3432, C.byref(st))


def test_product_does_not_link_or_reference_the_oracle(hiplib):
    import subprocess

    out = subprocess.run(["ldd", str(hiplib.path)], capture_output=True, text=True).stdout
    assert "oracle" not in out and "longtail_ref" not in out
    for src in (ROOT / "longtail_amd").rglob("*"):
        if src.suffix in (".hip", ".c", ".h", ".py") and src.is_file():
            t = src.read_text()
            assert "oracle/" not in t.replace("oracle/Makefile", "").replace("oracle/_ref", "").replace("oracle/*.c", "") or src.name in ("__init__.py",), src


def test_bounds(hiplib, golden):
    d = hiplib.dll
    for n in (0, 1, 254, 255, 256, 65536, (8 << 20) + 800000, 0x7E000000):
        assert d.lthip_lz4_bound(n) == n + n // 255 + 16      # lz4.h:215
    assert d.lthip_lz4_bound(0x7E000001) == 0
    for n in (0, 1, 1000, 131071, 131072, 131073, 9 << 20):
        exp = n + (n >> 8) + (((128 << 10) - n) >> 11 if n < (128 << 10) else 0)
        assert d.lthip_zstd_bound(n) == exp                   # zstd.h:232
    assert d.Longtail_GetHipLZ4DefaultQuality() == golden["tests"]["lz4_type"]


def test_division_free_cut_test_is_exact(hiplib, oracle):
    """`hash % d == d-1` (hpcdcchunker.c:298) as multiply-add + rotate + compare, for every discriminator shape."""
    f = hiplib.dll.lthip_divtest_eval
    rng = np.random.default_rng(0)
    ds = [oracle.dll.lto_hpcdc_discriminator(a) for a in (48, 64, 100, 2048, 16384, 32768, 65536, 131072, 1 << 20)]
    ds += [1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 96, 1024, 65536, 12345, 99991, 2**31, 2**31 + 1, 0xFFFFFFFF, 0xFFFFFFFE]
    ds += [int(x) for x in rng.integers(1, 2**32, size=40)]
    assert 24680 in ds and 49535 in ds  # target 65536 / 131072
    for d in ds:
        hs = [0, 1, d - 1, d, max(d - 2, 0), (2 * d - 1) & 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFE, 0x7FFFFFFF, 0x80000000]
        hs += [int(x) for x in rng.integers(0, 2**32, size=200)]
        hs += [int(k) * d + d - 1 for k in rng.integers(0, max(1, 2**32 // d), size=200) if int(k) * d + d - 1 < 2**32]
        for h in hs:
            assert f(d, h & 0xFFFFFFFF) == (1 if (h & 0xFFFFFFFF) % d == d - 1 else 0), (d, h)


def test_pack_blocks_matches_reference_store_index(hiplib, oracle, ref):
    """lthip_pack_blocks (host logic of the bulk path) against Longtail_CreateMissingContent -> Longtail_CreateStoreIndex
    (src/longtail.c:6745-6880) run by the reference itself on the same chunk list: same number of blocks."""
    from longtail_amd.lib import pack_blocks

    rng = np.random.default_rng(21)
    for target, max_block, max_chunks in [(65536, 8 << 20, 1024), (32768, 1 << 20, 16), (16384, 262144, 1024), (65536, 300000, 3)]:
        files = [(f"d/f{i:03d}.bin", oracle.synth(int(rng.integers(1, 3 << 20)), 900 + i, 0)) for i in range(12)]
        res = ref.ingest_time(files, target, max_block, max_chunks, ref.lz4_type, 2)
        assert res["err"] == 0
        mn, av, mx = max(48, target // 8), max(48, target // 2), max(48, target * 2)
        lens = []
        for name, data in sorted(files, key=lambda f: f[0]):  # version index order = sorted paths; random data: no duplicates
            part = target * 1024
            for s in range(0, len(data), part):
                lens.append(oracle.chunk_and_hash(data[s : s + part], mn, av, mx)[1])
        lens = np.concatenate(lens).astype(np.uint32)
        assert len(lens) == res["chunks"]
        starts = pack_blocks(lens, max_block, max_chunks, hiplib)
        assert len(starts) - 1 == res["blocks"], (target, max_block, max_chunks)


def test_pack_blocks_batch_resumes_to_the_same_blocks(hiplib):
    """lthip_pack_blocks_batch: the batches concatenated are exactly lthip_pack_blocks' blocks, every batch respects the byte
    and arena limits (or holds a single block), sizes are the sums of the chunk lengths."""
    from longtail_amd.lib import BatchPacker, pack_blocks

    rng = np.random.default_rng(5)
    for n, max_block, max_chunks, batch, arena in [(20000, 1 << 20, 1024, 16 << 20, 17 << 20), (5000, 300000, 3, 1 << 20, 1 << 30),
                                                   (3000, 65536, 1024, 1, 1), (1, 8 << 20, 1024, 8 << 30, 9 << 30), (0, 8 << 20, 1024, 1, 1),
                                                   (40000, 8 << 20, 4, 4 << 20, 1 << 20)]:
        lens = rng.integers(1, 200000, n).astype(np.uint32)
        want = pack_blocks(lens, max_block, max_chunks, hiplib)
        packer = BatchPacker(lens, max_block, max_chunks, batch, arena, 255, 16, hiplib)
        got, sizes, nbatches = [], [], 0
        while (nxt := packer.next()) is not None:
            starts, bs = nxt
            assert len(starts) == len(bs) + 1 and len(bs) >= 1
            if got:
                assert starts[0] == got[-1][-1]
            bounds = (bs + bs // 255 + 16 + 63) // 64 * 64
            assert len(bs) == 1 or (int(bs.sum()) <= batch and int(bounds.sum()) <= arena)
            got.append(starts)
            sizes.append(bs)
            nbatches += 1
        flat = np.concatenate([g[:-1] for g in got] + [[n]]) if got else np.array([0])
        assert (flat == want).all(), (n, max_block, max_chunks)
        if n:
            cs = np.concatenate([[0], np.cumsum(lens, dtype=np.int64)])
            assert (np.concatenate(sizes) == cs[want[1:]] - cs[want[:-1]]).all()
