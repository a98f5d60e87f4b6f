"""The oracle (oracle/*.c) against the golden vectors the reference's own tests hold and against vectors produced by
the reference library (tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest


def params(target):
    return max(48, target // 8), max(48, target // 2), max(48, target * 2)


def test_blake3_kat(oracle, golden):
    kat = golden["tests"]["blake3_kat"]
    s = np.frombuffer(kat["string_plus_nul"].encode() + b"\0", dtype=np.uint8)
    assert "%016x" % oracle.blake3(s) == kat["hash_hex"]  # test/test.cpp:465-474


def test_chunker_input_ranges(oracle, golden):
    ci = golden["tests"]["chunker_input"]
    data = golden["chunker_input"]
    assert len(data) == 1 << 20
    for pure in (False, True):
        lens = oracle.chunk(data, ci["min"], ci["avg"], ci["max"], pure=pure)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        assert [(int(a), int(b)) for a, b in zip(offs, lens)] == [tuple(r) for r in ci["ranges"]]  # test.cpp:3423-3445


def test_chunker_input_from_buffer(oracle, golden):
    # test/test.cpp:3467-3539: the mmap-style entry point yields the same 20 ranges on this file
    ci = golden["tests"]["chunker_input"]
    data = golden["chunker_input"]
    pos, got = 0, []
    while pos < len(data):
        n = oracle.dll.lto_hpcdc_next_from_buffer(data[pos:].ctypes.data, len(data) - pos, ci["min"], ci["avg"], ci["max"])
        got.append((pos, int(n)))
        pos += int(n)
    assert got == [tuple(r) for r in ci["ranges"]]


def test_lz4_golden_payload(oracle, golden):
    lb = golden["tests"]["lz4_block"]
    blk = np.concatenate([np.full(n, v, np.uint8) for n, v in lb["runs"]])
    comp = oracle.lz4_compress(blk)
    assert len(comp) == 38  # pinned by test/test.cpp:2092-2192
    assert comp.tobytes().hex() == lb["payload_hex"]
    n, out = oracle.lz4_decompress(comp, len(blk))
    assert n == len(blk) and (out == blk).all()


def test_chunk_and_hash_vectors(oracle, golden):
    v = golden["vec"]
    for name, kind, size, target, seed in golden["cases"]:
        data = oracle.synth(size, seed, kind)
        mn, av, mx = params(target)
        offs, lens, hashes = oracle.chunk_and_hash(data, mn, av, mx)
        assert (lens == v[name + "_lens"]).all(), name
        assert (hashes == v[name + "_hashes"]).all(), name
        assert (oracle.chunk(data, mn, av, mx, pure=True) == lens).all(), name
        # NextChunkFromBuffer restatement (quirk included)
        pos, fb = 0, []
        while pos < size:
            n = int(oracle.dll.lto_hpcdc_next_from_buffer(data[pos:].ctypes.data, size - pos, mn, av, mx))
            fb.append(n)
            pos += n
        assert fb == list(v[name + "_frombuf"]), name


def test_blake3_length_vectors(oracle, golden):
    v = golden["vec"]
    data = oracle.synth(300000, 99, 0)
    for n, h in zip(v["blake3_lengths"], v["blake3_hashes"]):
        assert oracle.blake3(data[: int(n)]) == int(h), int(n)
    for s, h in zip(range(1, 9), v["blake3_unaligned"]):
        assert oracle.blake3(data[s : s + 70000]) == int(h)


def test_synth_is_position_pure(oracle):
    a = oracle.synth(10000, 5, 1)
    b = oracle.synth(3000, 5, 1, offset=1237)
    assert (a[1237:4237] == b).all()
    assert not (oracle.synth(4096, 6, 0) == a[:4096]).all()
    assert (oracle.synth(4096, 6, 2) == 0).all()


def test_lz4_reference_sizes(oracle, golden):
    for kind, size, seed, ref_size in golden["vec"]["lz4_ref_sizes"]:
        d = oracle.synth(int(size), int(seed), int(kind))
        c = oracle.lz4_compress(d)
        assert len(c) == int(ref_size)
        n, out = oracle.lz4_decompress(c, len(d))
        assert n == len(d) and (out == d).all()


@pytest.mark.parametrize("nbytes,target", [(1000, 65536), (1 << 20, 32768), (64 << 20, 65536)])
def test_survey_probed_reference_values(oracle, nbytes, target):
    """chunk counts, first chunks and hashes the survey measured on the reference for xorshift64 streams."""
    from longtail_amd.lib import chunker_params
    from tests.survey_vectors import EXPECTED, xorshift_stream

    count, first = EXPECTED[(nbytes, target)]
    off, lens, hashes = oracle.chunk_and_hash(xorshift_stream(nbytes), *chunker_params(target))
    assert len(lens) == count
    got = [(int(o), int(l), int(h)) for o, l, h in zip(off[: len(first)], lens[: len(first)], hashes[: len(first)])]
    assert got == first
