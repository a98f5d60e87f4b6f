"""-m gpu: boundary semantics through the HIP plugins inside the unmodified reference core (oracle/_ref), and BASELINE.json
configs[1] (the 256 MiB file through HIP ChunkerAPI + HashAPI, bit-exact against the CPU plugins).

  * failing feeder -> empty range + ESPIPE, handle still disposable (hpcdcchunker.c:244-248, 420-423; src/longtail.c:2296)
  * cancel: Longtail_CreateVersionIndex with a cancelled token -> ECANCELED, *out == 0 (test/test.cpp:4733-4837), also when the
    token is cancelled while chunking jobs are in flight; the plugins stay usable afterwards
  * configs[1]: 268 435 456 bytes, target 65536 -> 5 parts (4 x 64 MiB + one empty), 8 166 chunks on SURVEY.md §8d's xorshift stream"""
import errno

import numpy as np
import pytest

from tests.test_gpu_plugins import plugins  # noqa: F401  (module-scoped fixture)

pytestmark = pytest.mark.gpu


def parse_version_index_header(blob: bytes):
    v = np.frombuffer(blob[:24], np.uint32)
    return dict(version=int(v[0]), hash_id=int(v[1]), target=int(v[2]), assets=int(v[3]), chunks=int(v[4]), chunk_indexes=int(v[5]))


@pytest.mark.parametrize("fail_at", [0, 1, 100000, (1 << 20) + 17, (3 << 20) - 1])
def test_failing_feeder_is_an_empty_range_and_espipe(plugins, ref, oracle, fail_at):
    mn, av, mx = 8192, 32768, 131072
    data = oracle.synth(3 << 20, 4242, 0)
    full = oracle.chunk(data, mn, av, mx)
    lens, fail = ref.chunk_failing_feeder(data, mn, av, mx, fail_at, errno.EIO, plugins["chunker"])
    assert fail == dict(err=errno.ESPIPE, len=0, offset=0, has_buf=False)
    assert len(lens) <= len(full) and (lens == full[: len(lens)]).all() and int(lens.sum()) <= fail_at
    # the API object works on: same chunker pool, next stream is chunked correctly
    offs, lens2, hashes = ref.chunk_and_hash(data, mn, av, mx, plugins["chunker"], plugins["hash"])
    assert (lens2 == full).all()


def test_failing_feeder_in_a_later_window(plugins, ref, oracle):
    """The failure hits while refilling the SECOND window of a long stream: the chunks of the first window were handed out,
    the failing call still mirrors the reference."""
    mn, av, mx = 8192, 32768, 131072
    data = oracle.synth((64 << 20) + (3 << 20), 77, 1)
    full = oracle.chunk(data, mn, av, mx)
    lens, fail = ref.chunk_failing_feeder(data, mn, av, mx, (64 << 20) + 4096, errno.EACCES, plugins["chunker"])
    assert fail == dict(err=errno.ESPIPE, len=0, offset=0, has_buf=False)
    assert 0 < len(lens) < len(full) and (lens == full[: len(lens)]).all()


@pytest.mark.parametrize("workers,after", [(0, 0), (2, 0), (4, 0), (2, 3), (4, 5)])
def test_cancel_gives_ecanceled_and_no_index(plugins, ref, oracle, workers, after):
    files = [(f"d{i % 2}/f{i:02d}.bin", oracle.synth(300000 + 7 * i, 900 + i, i % 3)) for i in range(12)]
    err, is_null, calls = ref.version_index_cancel(files, 16384, workers, after, plugins["chunker"], plugins["hash"])
    if after and err == 0:
        # The token is cancelled from the progress callback once it has been called `after` times.  Since round 3 the twelve jobs
        # of this tree can all be finished before that (the small windows of all workers go to the GPU in one submission): the
        # reference then returns the index like it would with its own plugins on a fast machine.  Anything but a clean success or
        # a clean cancellation is still a failure.
        assert not is_null
    else:
        assert err == errno.ECANCELED and is_null
    # nothing is left in a bad state: the same plugin objects index the same tree exactly like the CPU plugins
    cpu, _ = ref.version_index(files, 16384, workers=workers)
    hip, _ = ref.version_index(files, 16384, workers=workers, chunker_api=plugins["chunker"], hash_api=plugins["hash"])
    assert cpu == hip


@pytest.mark.parametrize("workers,after", [(2, 3), (4, 5), (4, 40)])
def test_cancel_in_flight_is_always_ecanceled_on_a_large_tree(plugins, ref, oracle, workers, after):
    """The cancellation semantics of the BATCHED path, deterministically: 600 jobs on 2-4 workers -- a worker blocks in its submission, so
    at most `workers` jobs can be in flight when the progress callback cancels after `after` completions, and hundreds are left: the
    call must return ECANCELED and no index (test/test.cpp:4733-4837), whatever the batcher had queued."""
    files = [(f"d{i % 7}/f{i:03d}.bin", oracle.synth(120000 + 11 * i, 5000 + i, i % 3)) for i in range(600)]
    err, is_null, calls = ref.version_index_cancel(files, 16384, workers, after, plugins["chunker"], plugins["hash"])
    assert err == errno.ECANCELED and is_null and calls >= after
    sample = files[:40]
    cpu, _ = ref.version_index(sample, 16384, workers=workers)
    hip, _ = ref.version_index(sample, 16384, workers=workers, chunker_api=plugins["chunker"], hash_api=plugins["hash"])
    assert cpu == hip


def test_cancel_semantics_without_the_batcher():
    """The same two cancel tests with LONGTAIL_HIP_BATCH=0 (every window on its thread's own stream; the switch is read once per process)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    if os.environ.get("LONGTAIL_HIP_BATCH") == "0":
        pytest.skip("already the unbatched run")
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, "-m", "pytest", str(Path(__file__).resolve()), "-q", "-x", "-m", "gpu", "-k",
                          "test_cancel_in_flight or test_cancel_gives"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, LONGTAIL_HIP_BATCH="0", PYTHONPATH=str(root)), cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


@pytest.mark.parametrize("workers", [0, 4])
def test_configs1_256mib_file_is_bit_exact(plugins, ref, oracle, workers):
    """BASELINE.json configs[1]."""
    size = 268_435_456
    data = oracle.synth(size, oracle.asset_seed(0x10C0FFEE, 1), 0)
    files = [("big/file00000.bin", data)]
    cpu, _ = ref.version_index(files, 65536, workers=workers, tag=ref.lz4_type)
    hip, secs = ref.version_index(files, 65536, workers=workers, tag=ref.lz4_type, chunker_api=plugins["chunker"], hash_api=plugins["hash"])
    assert cpu == hip, "VersionIndex of the 256 MiB file differs between HIP and CPU plugins"
    hdr = parse_version_index_header(hip)
    assert hdr["hash_id"] == 0x626C6B33 and hdr["target"] == 65536
    # 1 + size / (target * 1024) = 5 jobs, the fifth one empty (src/longtail.c:2402, 2432-2437); chunk count from the oracle
    part = 65536 * 1024
    assert 1 + size // part == 5
    per_part = [oracle.chunk(data[k * part : (k + 1) * part], 8192, 32768, 131072) for k in range(4)]
    assert hdr["chunk_indexes"] == sum(len(p) for p in per_part)
    assert hdr["chunks"] == hdr["chunk_indexes"]  # random data: no duplicate chunks
    assert 7900 <= hdr["chunks"] <= 8400          # 8 166 for SURVEY.md §8d's xorshift stream (next test); same statistics here


def test_configs1_survey_xorshift_stream(plugins, ref, oracle):
    """The stream SURVEY.md §8(c,d) names: xorshift64 bytes, 256 MiB, target 65536.  The survey quotes 8 170 chunks; the
    reference compiled here (oracle/_ref, Longtail_CreateVersionIndex with its own plugins) gives 8 166 for this stream (8 165
    if it were chunked as one stream instead of 4 parts) -- the reference run is the authority, the HIP plugins must equal it."""
    from tests.survey_vectors import xorshift_stream

    data = oracle.xorshift(268_435_456)
    assert (data[: 1 << 16] == xorshift_stream(1 << 16)).all()  # the C generator is the survey's stream
    files = [("file.bin", data)]
    cpu, _ = ref.version_index(files, 65536, workers=4)
    hip, _ = ref.version_index(files, 65536, workers=4, chunker_api=plugins["chunker"], hash_api=plugins["hash"])
    assert cpu == hip
    assert parse_version_index_header(hip)["chunks"] == parse_version_index_header(cpu)["chunks"] == 8166
