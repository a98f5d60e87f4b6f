"""Boundary semantics of the plugin structs as the REFERENCE implements them, pinned on the CPU (reference plugins inside the
reference core, oracle/_ref): what tests/test_gpu_boundary.py then demands from the HIP plugins.
  * a failing feeder -> empty range {0,0,0} + ESPIPE (hpcdcchunker.c:244-248, 420-423)
  * Longtail_CreateVersionIndex with a cancelled token -> ECANCELED and no index (test/test.cpp:4733-4837)"""
import errno

import numpy as np
import pytest


def boundary_cases(oracle):
    mn, av, mx = 8192, 32768, 131072
    data = oracle.synth(3 << 20, 4242, 0)
    return data, (mn, av, mx)


@pytest.mark.parametrize("fail_at", [0, 1, 100000, (1 << 20) + 17, (3 << 20) - 1])
def test_reference_failing_feeder_is_an_empty_range_and_espipe(ref, oracle, fail_at):
    data, (mn, av, mx) = boundary_cases(oracle)
    full = oracle.chunk(data, mn, av, mx)
    lens, fail = ref.chunk_failing_feeder(data, mn, av, mx, fail_at, errno.EIO)
    assert fail == dict(err=errno.ESPIPE, len=0, offset=0, has_buf=False)
    # whatever was handed out before the failure lies inside the bytes served.  (It need not be a prefix of the true chunk
    # list: the reference chunks a short read as if the stream ended there -- fail_at = 1 yields a 1-byte chunk.)
    assert len(lens) <= len(full) and int(lens.sum()) <= fail_at


@pytest.mark.parametrize("workers,after", [(0, 0), (2, 0), (4, 0), (2, 3)])
def test_reference_cancel_gives_ecanceled_and_no_index(ref, oracle, workers, after):
    files = [(f"d{i % 2}/f{i:02d}.bin", oracle.synth(300000 + 7 * i, 900 + i, i % 3)) for i in range(12)]
    err, is_null, calls = ref.version_index_cancel(files, 16384, workers, after)
    assert err == errno.ECANCELED and is_null
    if after:
        assert calls >= after
