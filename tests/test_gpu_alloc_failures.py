"""-m gpu: device-side failure injection -- the counterpart of the reference's FailableStorageAPI tests (test/test.cpp:5677-5752,
SURVEY.md §5 "HIP errors -> EIO / ENOMEM, never abort").

The ABLATION build of the library (build/ablations/liblongtail_hip.so, the same sources with -DLTHIP_ABLATIONS) routes every device /
pinned allocation through a counter; lthip_debug_fail_alloc(after, count) makes allocations after+1 .. after+count fail with
out-of-memory.  Checked, with the allocation that fails swept over a cold run's allocations:

  * Longtail_CreateVersionIndex with the HIP chunker + hash in the UNMODIFIED reference core returns ENOMEM (not EIO, no hang, no
    abort), produces no index; the next call on the SAME plugin objects succeeds with the byte-identical VersionIndex;
  * Longtail_WriteContent with the HIP codec objects: the same, LZ4 and zstd; the restore through the reference decoders is intact;
  * the bulk session (lthip_chunk_hash + lthip_ingest_*): ENOMEM out of the failing call, a fresh session on the same context works;
  * the chunker pool's pinned memory returns to zero once the objects are disposed (Longtail_Hip_PinnedBytes).
"""
import ctypes as C
import errno

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DISPOSE = C.CFUNCTYPE(None, C.c_void_p)
FOREVER = 1 << 60


def dispose(ptr):
    DISPOSE(C.c_void_p.from_address(ptr).value)(ptr)


@pytest.fixture()
def abl(gpu_abl):
    d = gpu_abl.lib.dll
    assert d.lthip_debug_fail_alloc(-1, 0) == 0, "the ablation build must have the injection switch"
    yield d
    d.lthip_debug_fail_alloc(-1, 0)


def calls(d):
    failed = C.c_int64(0)
    n = d.lthip_debug_alloc_calls(C.byref(failed))
    assert n >= 0
    return int(n), int(failed.value)


def test_product_library_has_no_injection(hiplib):
    assert hiplib.dll.lthip_debug_fail_alloc(0, 1) == errno.ENOTSUP
    assert hiplib.dll.lthip_debug_alloc_calls(None) == -1


def tree(oracle):
    sizes = [0, 100, 70000, 1 << 20, 3 << 20, 5, 2 << 20, 5 << 20]
    return [(f"d{i % 3}/f{i:02d}.bin", oracle.synth(int(n), 60 + i, (1, 0, 2)[i % 3])) for i, n in enumerate(sizes)]


def version_index_err(ref, files, workers, chunker, hasher):
    try:
        blob, _ = ref.version_index(files, 65536, workers, ref.lz4_type, chunker, hasher)
        return 0, blob
    except RuntimeError as e:
        return int(str(e).rsplit(" ", 1)[1]), None


def sweep_points(n):
    return sorted({k for k in (*range(min(n, 10)), n // 3, n // 2, (2 * n) // 3, n - 2, n - 1) if 0 <= k < n})


@pytest.mark.parametrize("workers", [0, 4])
def test_create_version_index_enomem_then_recovers(abl, ref, oracle, workers):
    files = tree(oracle)
    cpu, _ = ref.version_index(files, 65536, workers, ref.lz4_type)

    def objects():
        c, h = abl.Longtail_CreateHipChunkerAPI(), abl.Longtail_CreateHipBlake3HashAPI()
        assert c and h
        return c, h

    c, h = objects()
    n0, _ = calls(abl)
    err, blob = version_index_err(ref, files, workers, c, h)
    assert err == 0 and blob == cpu
    n_cold = calls(abl)[0] - n0
    dispose(c), dispose(h)
    assert abl.Longtail_Hip_PinnedBytes() == 0  # the last ChunkerAPI took the window pool with it
    assert n_cold > 0, "a cold run allocates its windows"
    enomem = 0
    for k in sweep_points(n_cold):
        c, h = objects()
        abl.lthip_debug_fail_alloc(k, FOREVER)
        try:
            err, blob = version_index_err(ref, files, workers, c, h)
        finally:
            abl.lthip_debug_fail_alloc(-1, 0)
        # a run that never reached its (k+1)-th allocation (W > 0: the order is the scheduler's) may succeed -- then it must be right
        assert err in (0, errno.ENOMEM), f"allocation {k + 1} of {n_cold} failing gave errno {err} (want ENOMEM)"
        assert (blob == cpu) if err == 0 else (blob is None)
        enomem += err == errno.ENOMEM
        err, blob = version_index_err(ref, files, workers, c, h)  # the SAME objects, right away
        assert err == 0 and blob == cpu, f"after the failure at allocation {k + 1}: errno {err}"
        dispose(c), dispose(h)
        assert abl.Longtail_Hip_PinnedBytes() == 0
    # (the batch dispatcher allocates on its own thread, concurrently with the caller's: which allocation is "number k + 1" differs
    # slightly from run to run even at W = 0, and a point near the end may fall behind the run's last allocation)
    assert enomem >= (len(sweep_points(n_cold)) - 4 if workers == 0 else 1), (enomem, n_cold)


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_write_content_enomem_then_recovers(abl, ref, oracle, codec):
    """Reference chunker + hash (CPU), HIP codec object: every allocation of the library in this run is the codec plugin's, inside
    Longtail_WriteContent (lib/compressblockstore/longtail_compressblockstore.c:67-141 -> Compress)."""
    files = tree(oracle)
    tag = ref.lz4_type if codec == "lz4" else ref.zstd_default
    make = abl.Longtail_CreateHipLZ4CompressionAPI if codec == "lz4" else abl.Longtail_CreateHipZStdCompressionAPI
    base = ref.ingest_roundtrip(files, 65536, 1 << 20, 64, tag, 2)
    assert base["err"] == 0
    for workers in (0, 2):
        api = make()
        n0, _ = calls(abl)
        got = ref.ingest_roundtrip(files, 65536, 1 << 20, 64, tag, workers, None, None, api)
        assert got["err"] == 0 and got["blocks"] == base["blocks"]
        n_cold = calls(abl)[0] - n0
        dispose(api)
        assert n_cold > 0
        enomem = 0
        for k in sweep_points(n_cold):
            api = make()  # (the last codec object took the dispatcher thread and its context with it: cold again)
            abl.lthip_debug_fail_alloc(k, FOREVER)
            try:
                got = ref.ingest_roundtrip(files, 65536, 1 << 20, 64, tag, workers, None, None, api)
            finally:
                abl.lthip_debug_fail_alloc(-1, 0)
            assert got["err"] in (0, errno.ENOMEM), f"{codec}: allocation {k + 1} of {n_cold} failing gave errno {got['err']}"
            enomem += got["err"] == errno.ENOMEM
            again = ref.ingest_roundtrip(files, 65536, 1 << 20, 64, tag, workers, None, None, api)  # incl. the reference-only restore
            assert again["err"] == 0 and again["blocks"] == base["blocks"], again
            dispose(api)
        assert enomem >= 1, (codec, workers, n_cold)


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_bulk_session_enomem_then_recovers(abl, gpu_abl, ref, oracle, codec):
    from longtail_amd.lib import Context, LongtailHipError
    from tests.test_gpu_ingest import make_files, rank_session

    files = make_files(oracle, 32768)
    tag = ref.lz4_type if codec == "lz4" else ref.zstd_default

    def session(ctx):
        local = rank_session(ctx, ref, files, 32768, 1, 0, "range", codec, 1 << 20, 64, tag)
        lists = {j: (local["d_hash"][local["first"][i] : local["first"][i + 1]], local["d_len"][local["first"][i] : local["first"][i + 1]])
                 for i, j in enumerate(local["mine"])}
        full = rank_session(ctx, ref, files, 32768, 1, 0, "range", codec, 1 << 20, 64, tag, all_lists=lists)
        out = (full["vi"], full["si"], full["comp"].copy())
        full["ing"].close()
        return out

    ctx = Context(0, lib=gpu_abl.lib)
    n0, _ = calls(abl)
    want = session(ctx)
    n_cold = calls(abl)[0] - n0
    ctx.close()
    assert n_cold > 0
    enomem = 0
    for k in sweep_points(n_cold):
        ctx = Context(0, lib=gpu_abl.lib)  # cold scratch pools
        abl.lthip_debug_fail_alloc(k, FOREVER)
        try:
            session(ctx)
            code = 0
        except LongtailHipError as e:
            code = e.code
        finally:
            abl.lthip_debug_fail_alloc(-1, 0)
        assert code in (0, errno.ENOMEM), f"{codec}: allocation {k + 1} of {n_cold} failing gave errno {code}"
        enomem += code == errno.ENOMEM
        got = session(ctx)  # the same context: whatever the failed call left behind must not be in the way
        assert got[0] == want[0] and got[1] == want[1] and (got[2] == want[2]).all()
        ctx.close()
    assert enomem == len(sweep_points(n_cold)), (enomem, n_cold)
