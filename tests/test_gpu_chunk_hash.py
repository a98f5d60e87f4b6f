"""-m gpu: the HIP chunk+hash path (through the C ABI) against the oracle and the reference's golden vectors:
chunk boundaries and 64-bit BLAKE3 chunk hashes must be bit-exact."""
import numpy as np
import pytest
import torch

from tests.gpu_util import check_part, gpu_chunk_hash, to_device, u32, u64

pytestmark = pytest.mark.gpu


def params(target):
    return max(48, target // 8), max(48, target // 2), max(48, target * 2)


def test_golden_chunker_input(gpu, oracle, golden):
    ci = golden["tests"]["chunker_input"]
    data = golden["chunker_input"]
    (offs, lens, hashes), = gpu_chunk_hash(gpu, [data], ci["min"], ci["avg"], ci["max"])
    assert [(int(a), int(b)) for a, b in zip(offs, lens)] == [tuple(r) for r in ci["ranges"]]  # test/test.cpp:3423-3445
    check_part(oracle, data, (offs, lens, hashes), ci["min"], ci["avg"], ci["max"], "chunker.input")


def test_golden_blake3_kat(gpu, golden):
    kat = golden["tests"]["blake3_kat"]
    s = np.frombuffer(kat["string_plus_nul"].encode() + b"\0", dtype=np.uint8).copy()
    dev, offs = to_device([s])
    o = torch.tensor([0], dtype=torch.int64, device="cuda")
    l = torch.tensor([len(s)], dtype=torch.int32, device="cuda")
    h = gpu.hash_ranges(dev, o, l, max_len=len(s))
    assert "%016x" % int(u64(h)[0]) == kat["hash_hex"]  # test/test.cpp:465-474


def test_reference_vectors(gpu, oracle, golden):
    v = golden["vec"]
    for name, kind, size, target, seed in golden["cases"]:
        data = oracle.synth(size, seed, kind)
        mn, av, mx = params(target)
        (offs, lens, hashes), = gpu_chunk_hash(gpu, [data], mn, av, mx)
        assert (lens == v[name + "_lens"]).all(), name
        assert (hashes == v[name + "_hashes"]).all(), name


def test_blake3_lengths(gpu, oracle, golden):
    v = golden["vec"]
    data = oracle.synth(300000, 99, 0)
    dev, _ = to_device([data])
    lengths = [int(x) for x in v["blake3_lengths"]]
    o = torch.zeros(len(lengths), dtype=torch.int64, device="cuda")
    l = torch.tensor(lengths, dtype=torch.int32, device="cuda")
    h = u64(gpu.hash_ranges(dev, o, l, max_len=0))  # unknown bound -> level-synchronous parent kernels
    assert (h == v["blake3_hashes"]).all(), np.nonzero(h != v["blake3_hashes"])
    # small-tree path on lengths <= 256 KiB
    sel = [i for i, n in enumerate(lengths) if n <= 262144]
    h = u64(gpu.hash_ranges(dev, o[sel], l[sel], max_len=262144))
    assert (h == v["blake3_hashes"][sel]).all()
    # unaligned starts
    o2 = torch.arange(1, 9, dtype=torch.int64, device="cuda")
    l2 = torch.full((8,), 70000, dtype=torch.int32, device="cuda")
    assert (u64(gpu.hash_ranges(dev, o2, l2, max_len=70000)) == v["blake3_unaligned"]).all()


def test_hash_ranges_fuzz(gpu, oracle):
    rng = np.random.default_rng(5)
    data = oracle.synth(1 << 20, 123, 0)
    dev, _ = to_device([data])
    n = 600
    lens = np.where(rng.random(n) < 0.5, rng.integers(0, 3000, n), rng.integers(0, 200000, n)).astype(np.uint32)
    offs = np.array([rng.integers(0, len(data) - int(l) + 1) for l in lens], np.uint64)
    exp = oracle.blake3_many(data, offs, lens)
    got = u64(gpu.hash_ranges(dev, torch.from_numpy(offs.view(np.int64)).cuda(), torch.from_numpy(lens.view(np.int32)).cuda(),
                              max_len=200000))
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, (bad[:5], lens[bad[:5]], offs[bad[:5]])


@pytest.mark.parametrize("cfg", [(8192, 32768, 131072), (48, 48, 48), (4096, 16384, 65536), (48, 100, 300), (16384, 65536, 262144),
                                 (48, 64, 64), (64, 128, 4096), (2048, 2048, 8192)])
def test_params_and_kinds(gpu, oracle, cfg):
    rng = np.random.default_rng(cfg[0] + cfg[2])
    parts = []
    for kind in (0, 1, 2):
        size = int(rng.integers(1, 3 << 20))
        parts.append(oracle.synth(size, 10 * cfg[0] + kind, kind))
    for (got, data) in zip(gpu_chunk_hash(gpu, parts, *cfg), parts):
        check_part(oracle, data, got, *cfg, what=f"cfg={cfg} size={len(data)}")


def test_edge_sizes(gpu, oracle):
    cfg = (8192, 32768, 131072)
    base = oracle.synth(600000, 77, 0)
    sizes = [0, 1, 2, 15, 16, 17, 47, 48, 49, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385, 131071, 131072,
             131073, 262144, 262145, 500001]
    parts = [base[:s].copy() for s in sizes]
    res = gpu_chunk_hash(gpu, parts, *cfg)
    for s, got, data in zip(sizes, res, parts):
        check_part(oracle, data, got, *cfg, what=f"size={s}")
    assert len(res[0][1]) == 0  # empty part -> no chunk (src/longtail.c:2015-2019)


def test_no_candidates_means_max_chunks(gpu, oracle):
    cfg = (8192, 32768, 131072)
    for fill in (0, 255):
        data = np.full((1 << 20) + 5, fill, np.uint8)
        (got,) = gpu_chunk_hash(gpu, [data], *cfg)
        check_part(oracle, data, got, *cfg, what=f"fill={fill}")


def test_many_small_parts(gpu, oracle):
    cfg = (8192, 32768, 131072)
    rng = np.random.default_rng(9)
    sizes = [int(x) for x in rng.integers(0, 70000, 300)] + [1 << 20] * 3
    parts = [oracle.synth(s, 3000 + i, i % 3) for i, s in enumerate(sizes)]
    for i, (got, data) in enumerate(zip(gpu_chunk_hash(gpu, parts, *cfg), parts)):
        check_part(oracle, data, got, *cfg, what=f"part {i} size {len(data)}")


def test_part_boundaries_are_independent(gpu, oracle):
    """A 64 MiB+ asset is cut into target*1024-byte parts, each chunked from a fresh state (src/longtail.c:2396-2458)."""
    cfg = (1024, 4096, 16384)  # target 8192 -> parts of 8 MiB
    data = oracle.synth((20 << 20) + 777, 4242, 1)
    part = 8192 * 1024
    parts = [data[i : i + part] for i in range(0, len(data), part)]
    res = gpu_chunk_hash(gpu, parts, *cfg)
    for got, d in zip(res, parts):
        check_part(oracle, np.ascontiguousarray(d), got, *cfg)


def test_chunk_from_buffer_quirk(gpu, oracle, golden):
    v = golden["vec"]
    for name, kind, size, target, seed in golden["cases"]:
        if size < 100:
            continue
        data = oracle.synth(size, seed, kind)
        dev, _ = to_device([data])
        mn, av, mx = params(target)
        pos, got = 0, []
        limit = 40  # a few calls per case: each is a synchronous round trip
        while pos < size and len(got) < limit:
            n = gpu.chunk_from_buffer(dev[pos:], size - pos, mn, av, mx)
            got.append(n)
            pos += n
        assert got == [int(x) for x in v[name + "_frombuf"][: len(got)]], name


def test_dedup_first_seen(gpu):
    rng = np.random.default_rng(1)
    base = rng.integers(0, 2**63, 5000, dtype=np.int64)
    h = base[rng.integers(0, len(base), 40000)]
    h[7] = -1  # the table's empty-key sentinel must still work as a value
    h[900] = -1
    first, uniq = gpu.dedup_first_seen(torch.from_numpy(h).cuda())
    exp, seen = np.zeros(len(h), np.uint32), {}
    for i, x in enumerate(h.tolist()):
        exp[i] = seen.setdefault(x, i)
    assert (u32(first) == exp).all()
    assert int(uniq.item()) == len(seen)
    # multi-GPU form: same table, answers only for one rank's range, distinct count from the insertions
    for lo, cnt in ((0, 0), (0, 100), (12345, 20000), (39999, 1), (0, len(h))):
        part, uniq2 = gpu.dedup_first_seen_range(torch.from_numpy(h).cuda(), lo, cnt)
        assert (u32(part) == exp[lo : lo + cnt]).all() and int(uniq2.item()) == len(seen)


@pytest.mark.parametrize("nbytes,target", [(1000, 65536), (1 << 20, 32768), (64 << 20, 65536)])
def test_survey_probed_reference_values_on_gpu(gpu, oracle, nbytes, target):
    """SURVEY.md §8(c)'s reference-measured chunk counts / first chunks / hashes for xorshift64 streams, through the C ABI; the
    whole list equals the oracle's."""
    from longtail_amd.lib import chunker_params
    from tests.survey_vectors import EXPECTED, xorshift_stream

    count, first = EXPECTED[(nbytes, target)]
    data = xorshift_stream(nbytes)
    mn, av, mx = chunker_params(target)
    ((off, lens, hashes),) = gpu_chunk_hash(gpu, [data], mn, av, mx)
    assert len(lens) == count
    assert [(int(o), int(l), int(h)) for o, l, h in zip(off[: len(first)], lens[: len(first)], hashes[: len(first)])] == first
    e_off, e_len, e_hash = oracle.chunk_and_hash(data, mn, av, mx)
    assert (lens == e_len).all() and (hashes == e_hash).all()
