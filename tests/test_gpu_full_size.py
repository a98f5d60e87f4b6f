"""-m gpu: the hot path at BASELINE.json's FULL size (configs[2]: 64 GiB of 1 MiB files on one MI355X), checked through
properties that do not need a CPU to redo 64 GiB:

  * every part's chunks tile it exactly (sum of lengths, contiguous offsets), lengths obey min/max (hpcdcchunker.c:257-309);
  * a random sample of files is re-chunked and re-hashed by the oracle: boundaries and BLAKE3 digests bit-exact;
  * checksum of checksums: XOR and wrapping SUM of all 2.1 M chunk hashes are identical when the SAME bytes are processed
    as one plan or as two half plans (parts are independent, src/longtail.c:2429-2457);
  * a tree whose second half repeats the first dedups to exactly half (first-seen order, src/longtail.c:2951-2970);
  * every LZ4 payload of the full tree decodes, with the HIP decoder, to the bytes it came from (encode -> decode round
    trip on the device), a sample also with the oracle's strict LZ4_decompress_safe restatement; compressible data
    (8 GiB "mixed") the same;
  * a sample of ZStd frames of full-size blocks decodes with the reference decoder.

Set LONGTAIL_FULL_GIB to shrink it (default 64)."""
import os

import numpy as np
import pytest
import torch

from bench import asset_seeds
from longtail_amd.lib import chunker_params, pack_blocks
from tests._libs import have_ref, ref as get_ref

pytestmark = pytest.mark.gpu

GIB = float(os.environ.get("LONGTAIL_FULL_GIB", "64"))
FILE = 1 << 20
TARGET = 65536
BLOCK = 8 << 20


@pytest.fixture(scope="module")
def tree(gpu):
    free, _ = torch.cuda.mem_get_info()
    gib = min(GIB, max(1.0, (free / (1 << 30) - 40) / 1.6))  # data + codec arenas and scratch must fit
    nfiles = int(gib * (1 << 30)) // FILE
    sizes = np.full(nfiles, FILE, np.uint64)
    offs = np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE)
    data = torch.empty(nfiles * FILE + 256, dtype=torch.uint8, device="cuda")
    seeds = asset_seeds(0x10C0FFEE, 0, nfiles)
    gpu.synth_fill(data, offs, sizes, seeds, 0)
    gpu.sync()
    return dict(data=data, offs=offs, sizes=sizes, seeds=seeds, nfiles=nfiles)


def run_plan(gpu, tree, lo, hi):
    mn, av, mx = chunker_params(TARGET)
    plan = gpu.make_plan(tree["offs"][lo:hi], tree["sizes"][lo:hi], mn, av, mx)
    total, d_off, d_len, d_hash, d_first = gpu.chunk_hash(plan, tree["data"])
    plan.close()
    return total, d_off[:total], d_len[:total], d_hash[:total], d_first


def test_full_tree_chunks_tile_parts_and_sample_matches_oracle(gpu, oracle, tree):
    mn, av, mx = chunker_params(TARGET)
    n = tree["nfiles"]
    total, d_off, d_len, d_hash, d_first = run_plan(gpu, tree, 0, n)
    first = d_first.cpu().numpy().view(np.uint32).astype(np.int64)
    lens = d_len.to(torch.int64)
    assert int(first[-1]) == total and int(lens.sum().item()) == n * FILE
    # per part: lengths sum to the part, offsets are the running sum from the part's start
    csum = torch.cumsum(lens, 0)
    ends = csum[torch.from_numpy(first[1:] - 1).cuda()]
    assert torch.equal(ends, torch.arange(1, n + 1, dtype=torch.int64, device="cuda") * FILE)
    assert torch.equal(d_off, csum - lens)  # parts are laid out back to back here, so offsets are global running sums
    last = torch.zeros(total, dtype=torch.bool, device="cuda")
    last[torch.from_numpy(first[1:] - 1).cuda()] = True
    assert int((lens > mx).sum().item()) == 0
    assert int(((lens <= mn) & ~last).sum().item()) == 0  # only a part's last chunk may be <= min (hpcdcchunker.c:257-264)
    # sample: the oracle redoes 48 files
    rng = np.random.default_rng(7)
    off_h, len_h, hash_h = (t.cpu().numpy() for t in (d_off, d_len, d_hash))
    for f in rng.choice(n, 48, replace=False):
        host = tree["data"][int(f) * FILE : (int(f) + 1) * FILE].cpu().numpy()
        assert (host == oracle.synth(FILE, int(tree["seeds"][f]), 0)).all()  # the generator kernel == the C generator
        e_off, e_len, e_hash = oracle.chunk_and_hash(host, mn, av, mx)
        a, b = first[f], first[f + 1]
        assert (len_h[a:b].view(np.uint32) == e_len).all() and (off_h[a:b] - f * FILE == e_off.astype(np.int64)).all()
        assert (hash_h[a:b].view(np.uint64) == e_hash).all()


def test_checksum_of_checksums_is_independent_of_batching(gpu, tree):
    n = tree["nfiles"]
    total, _, _, h_all, _ = run_plan(gpu, tree, 0, n)
    t1, _, _, h1, _ = run_plan(gpu, tree, 0, n // 3)
    h1 = h1.clone()
    t2, _, _, h2, _ = run_plan(gpu, tree, n // 3, n)
    assert t1 + t2 == total
    both = torch.cat([h1, h2])
    assert torch.equal(both, h_all)  # stronger than the checksums, but they are what a user would compare:
    xor = lambda t: int(np.bitwise_xor.reduce(t.cpu().numpy().view(np.uint64)))
    assert xor(both) == xor(h_all) and int(both.sum().item()) == int(h_all.sum().item())


def test_duplicated_half_dedups_to_half(gpu, tree):
    n = tree["nfiles"] // 2 * 2
    total, _, _, h, first = run_plan(gpu, tree, 0, n // 2)
    twice = torch.cat([h, h])
    first_idx, uniq = gpu.dedup_first_seen(twice)
    assert int(uniq.item()) == int(torch.unique(h).numel())
    fi = first_idx.to(torch.int64)
    assert torch.equal(fi[total:], fi[:total])  # the copy points at the original
    assert int((fi[:total] > torch.arange(total, device="cuda")).sum().item()) == 0


def _compress_decode_check(gpu, oracle, data, nbytes, kind_name, sample=6):
    """LZ4-compress [0, nbytes) of `data` as 8 MiB blocks in batches, decode every payload on the GPU, compare."""
    nblocks = (nbytes + BLOCK - 1) // BLOCK
    b_off = np.arange(nblocks, dtype=np.int64) * BLOCK
    b_size = np.minimum(BLOCK, nbytes - b_off).astype(np.int64)
    bound = b_size + b_size // 255 + 16
    per = 512  # blocks per batch (4 GiB)
    arena = torch.empty(int(bound[:per].sum()) + per * 64 + 64, dtype=torch.uint8, device="cuda")
    back = torch.empty(per * BLOCK + 64, dtype=torch.uint8, device="cuda")
    rng = np.random.default_rng(3)
    comp_total = 0
    for i in range(0, nblocks, per):
        j = min(nblocks, i + per)
        d_offs = np.concatenate([[0], np.cumsum((bound[i:j] + 63) // 64 * 64)[:-1]])
        sizes = gpu.lz4_compress_blocks(data, b_off[i:j], b_size[i:j], arena, d_offs, bound[i:j])
        sz = sizes.cpu().numpy().view(np.uint32).astype(np.int64)
        assert (sz > 0).all()
        comp_total += int(sz.sum())
        out = gpu.lz4_decompress_blocks(arena, d_offs, sz, back, b_off[i:j] - b_off[i], b_size[i:j])
        assert (out.cpu().numpy().view(np.uint32) == b_size[i:j]).all()
        assert torch.equal(back[: int(b_size[i:j].sum())], data[int(b_off[i]) : int(b_off[i]) + int(b_size[i:j].sum())])
        for k in rng.choice(j - i, min(sample, j - i), replace=False) if i == 0 else []:
            payload = arena[int(d_offs[k]) : int(d_offs[k]) + int(sz[k])].cpu().numpy()
            raw = data[int(b_off[i + k]) : int(b_off[i + k]) + int(b_size[i + k])].cpu().numpy()
            nn, dec = oracle.lz4_decompress(payload, len(raw))
            assert nn == len(raw) and (dec == raw).all(), f"{kind_name}: oracle decoder disagrees on block {i + k}"
    return comp_total


def test_lz4_full_tree_roundtrip_on_device(gpu, oracle, tree):
    nbytes = tree["nfiles"] * FILE
    comp = _compress_decode_check(gpu, oracle, tree["data"], nbytes, "random")
    assert nbytes <= comp <= nbytes + nbytes // 255 + 16 * (nbytes // BLOCK + 1)  # incompressible: literal runs only


def test_lz4_and_zstd_compressible_roundtrip(gpu, oracle):
    n = min(8 << 30, int(GIB * (1 << 30)))
    nfiles = n // FILE
    data = torch.empty(nfiles * FILE + 256, dtype=torch.uint8, device="cuda")
    gpu.synth_fill(data, np.arange(nfiles, dtype=np.uint64) * np.uint64(FILE), np.full(nfiles, FILE, np.uint64),
                   asset_seeds(0xBEEF, 0, nfiles), 1)
    gpu.sync()
    comp = _compress_decode_check(gpu, oracle, data, nfiles * FILE, "mixed")
    assert comp < 0.7 * nfiles * FILE
    if have_ref():
        r = get_ref()
        nb = 24
        b_off = np.arange(nb, dtype=np.int64) * BLOCK
        b_size = np.full(nb, BLOCK, np.int64)
        caps = b_size + (b_size >> 8) + 64
        d_offs = np.concatenate([[0], np.cumsum((caps + 63) // 64 * 64)[:-1]])
        arena = torch.empty(int(caps.sum()) + nb * 64 + 64, dtype=torch.uint8, device="cuda")
        sz = gpu.zstd_compress_blocks(data, b_off, b_size, arena, d_offs, caps).cpu().numpy().view(np.uint32)
        assert (sz > 0).all() and int(sz.astype(np.int64).sum()) < 0.65 * nb * BLOCK
        for k in range(nb):
            frame = arena[int(d_offs[k]) : int(d_offs[k]) + int(sz[k])].cpu().numpy()
            err, out = r.decompress(1, frame, BLOCK)
            assert err == 0 and len(out) == BLOCK
            assert (out == data[int(b_off[k]) : int(b_off[k]) + BLOCK].cpu().numpy()).all()


@pytest.mark.parametrize("target", [65536, 4 << 20])
def test_one_gigabyte_part_through_one_chunker(gpu, oracle, target):
    """Maximum sizes: a single part of 1 GiB + 12345 bytes (what one reference chunker would see for a huge target), with
    the default chunk sizes (one wave walks ~27 000 cuts) and with 8 MiB maximum chunks (level-synchronous BLAKE3 trees)."""
    n = (1 << 30) + 12345
    data = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
    gpu.synth_fill(data, np.array([0], np.uint64), np.array([n], np.uint64), asset_seeds(5, 0, 1), 1)
    mn, av, mx = chunker_params(target)
    plan = gpu.make_plan([0], [n], mn, av, mx)
    total, d_off, d_len, d_hash, _ = gpu.chunk_hash(plan, data)
    plan.close()
    e_off, e_len, e_hash = oracle.chunk_and_hash(data[:n].cpu().numpy(), mn, av, mx)
    assert total == len(e_len)
    assert (d_len[:total].cpu().numpy().view(np.uint32) == e_len).all()
    assert (d_hash[:total].cpu().numpy().view(np.uint64) == e_hash).all()


def vi_asset_chunks(blob):
    """Per-asset chunk lists of a serialized VersionIndex (layout src/longtail.c:2551-2584): (asset sizes, chunk counts per asset,
    chunk hashes in (asset, chunk) order, chunk sizes in that order)."""
    h = np.frombuffer(blob[:24], np.uint32)
    na, nu, ni = int(h[3]), int(h[4]), int(h[5])
    o = 24 + na * 16
    sizes = np.frombuffer(blob[o : o + na * 8], np.uint64); o += na * 8
    counts = np.frombuffer(blob[o : o + na * 4], np.uint32); o += na * 4
    starts = np.frombuffer(blob[o : o + na * 4], np.uint32); o += na * 4
    idx = np.frombuffer(blob[o : o + ni * 4], np.uint32); o += ni * 4
    hashes = np.frombuffer(blob[o : o + nu * 8], np.uint64); o += nu * 8
    csz = np.frombuffer(blob[o : o + nu * 4], np.uint32)
    assert (starts.astype(np.int64) == np.concatenate([[0], np.cumsum(counts.astype(np.int64))[:-1]])).all()
    return sizes, counts, hashes[idx], csz[idx]


@pytest.mark.skipif(not have_ref(), reason="needs the reference build (oracle/_ref)")
def test_every_chunk_of_the_full_tree_against_the_reference_itself(gpu, tree):
    """ALL chunks of BASELINE.json configs[2], not a sample: the tree leaves the device in slices of 8 GiB, the reference's own
    Longtail_CreateVersionIndex (reference hpcdc chunker + BLAKE3, bikeshed W = 32; lib/hpcdcchunker/longtail_hpcdcchunker.c:266-306,
    src/longtail.c:2343-2550) runs on every slice, and each file's (length, hash) list must equal the device's, pair for pair."""
    n = tree["nfiles"]
    total, d_off, d_len, d_hash, d_first = run_plan(gpu, tree, 0, n)
    first = d_first.cpu().numpy().view(np.uint32).astype(np.int64)
    len_h, hash_h = d_len.cpu().numpy().view(np.uint32), d_hash.cpu().numpy().view(np.uint64)
    r = get_ref()
    per = (8 << 30) // FILE
    workers = min(32, os.cpu_count() or 1)
    checked = 0
    secs = 0.0
    for lo in range(0, n, per):
        hi = min(n, lo + per)
        host = tree["data"][lo * FILE : hi * FILE].cpu().numpy()
        files = [(f"f{lo + i:06d}.bin", host[i * FILE : (i + 1) * FILE]) for i in range(hi - lo)]
        blob, s = r.version_index(files, TARGET, workers=workers)
        secs += s
        sizes, counts, hashes, csz = vi_asset_chunks(blob)
        assert len(sizes) == hi - lo and (sizes == FILE).all()
        a, b = int(first[lo]), int(first[hi])
        assert (counts.astype(np.int64) == np.diff(first[lo : hi + 1])).all(), "chunk counts per file differ from the reference"
        assert (csz == len_h[a:b]).all(), "chunk lengths differ from the reference"
        assert (hashes == hash_h[a:b]).all(), "chunk hashes differ from the reference"
        checked += b - a
    assert checked == total
    print(f"full-size parity: {checked} (length, hash) pairs of {n} files equal to Longtail_CreateVersionIndex's ({secs:.1f} s of reference time at W={workers})")


@pytest.mark.skipif(not have_ref(), reason="needs the reference build (oracle/_ref)")
def test_large_files_against_the_reference_itself(gpu):
    """The configs[4] shape (few huge assets, many parts each) at 15.5 GiB: 8 files of 31 x 64 MiB (32 jobs each, the last one empty:
    an exact multiple of the part size, src/longtail.c:2402), every chunk against the reference.  (Files stay below 2 GiB: the
    reference's in-memory storage, which the harness feeds it from, does not take larger ones.)"""
    part = TARGET * 1024
    nf = 8 if GIB >= 16 else 2
    fsz = 31 * part
    # parts back to back: consecutive parts of one file are contiguous, files at 16-byte aligned offsets
    p_off, p_size, seeds, skips = [], [], [], []
    s4 = asset_seeds(0xC0FFEE4, 0, nf)
    for f in range(nf):
        for k in range(1 + fsz // part):
            p_off.append(f * fsz + k * part)
            p_size.append(min(part, fsz - k * part))
            seeds.append(int(s4[f]))
            skips.append(k * part)
    data = torch.empty(nf * fsz + 256, dtype=torch.uint8, device="cuda")
    gpu.synth_fill(data, np.array(p_off, np.uint64), np.array(p_size, np.uint64), np.array(seeds, np.uint64), 1, skips=np.array(skips, np.uint64))
    mn, av, mx = chunker_params(TARGET)
    plan = gpu.make_plan(p_off, p_size, mn, av, mx)
    total, d_off, d_len, d_hash, d_first = gpu.chunk_hash(plan, data)
    plan.close()
    first = d_first.cpu().numpy().view(np.uint32).astype(np.int64)
    len_h, hash_h = d_len[:total].cpu().numpy().view(np.uint32), d_hash[:total].cpu().numpy().view(np.uint64)
    r = get_ref()
    jobs_per_file = 1 + fsz // part
    for f in range(nf):
        host = data[f * fsz : (f + 1) * fsz].cpu().numpy()
        blob, _ = r.version_index([(f"big{f}.pak", host)], TARGET, workers=min(32, os.cpu_count() or 1))
        sizes, counts, hashes, csz = vi_asset_chunks(blob)
        a, b = int(first[f * jobs_per_file]), int(first[(f + 1) * jobs_per_file])
        assert int(sizes[0]) == fsz and int(counts[0]) == b - a
        assert (csz == len_h[a:b]).all() and (hashes == hash_h[a:b]).all()
