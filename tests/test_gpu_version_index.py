"""-m gpu: SURVEY.md §8 f1 -- the bulk path (lthip_chunk_hash + lthip_build_version_index) must emit the SAME serialized
VersionIndex as the reference's Longtail_CreateVersionIndex with its own CPU plugins (oracle/_ref), byte for byte:
unique chunk list in first-seen order, asset chunk indexes, content hashes, path hashes, names, permissions, tags."""
import numpy as np
import pytest
import torch

from longtail_amd.lib import chunker_params
from tests.gpu_util import to_device

pytestmark = pytest.mark.gpu


def bulk_version_index(gpu, ref, files, target, tag=0):
    by_name = {n: d for n, d in files}
    paths, sizes, offs, perms, path_data = ref.tree_file_infos(files)
    mn, av, mx = chunker_params(target)
    part = target * 1024
    # parts in the reference's job order: assets in FileInfos order, 1 + size/part jobs each (src/longtail.c:2402, 2432-2437)
    blobs, counts_parts = [], []
    for p, sz in zip(paths, sizes):
        data = by_name.get(p, np.zeros(0, np.uint8))
        assert len(data) == int(sz)
        nparts = 1 + len(data) // part
        counts_parts.append(nparts)
        for k in range(nparts):
            blobs.append(data[k * part : (k + 1) * part])
    dev, part_offs = to_device(blobs)
    plan = gpu.make_plan(part_offs, [len(b) for b in blobs], mn, av, mx)
    total, d_off, d_len, d_hash, d_first = gpu.chunk_hash(plan, dev)
    plan.close()
    first = d_first.cpu().numpy().view(np.uint32).astype(np.int64)
    bounds = np.concatenate([[0], np.cumsum(counts_parts)])
    asset_chunk_counts = (first[bounds[1:]] - first[bounds[:-1]]).astype(np.uint32)
    tags = np.full(len(paths), tag, np.uint32) if tag else None
    return gpu.build_version_index(sizes, offs, perms, path_data, asset_chunk_counts, d_hash, d_len, total, target, tags)


@pytest.mark.parametrize("target,tag", [(65536, 0), (4096, 0x6C7A3432), (32768, 0)])
def test_bulk_version_index_is_byte_identical(gpu, oracle, ref, target, tag):
    rng = np.random.default_rng(target)
    files = []
    for i in range(14):
        kind = i % 3
        files.append((f"dir{i % 3}/sub{i % 2}/file{i:02d}.bin", oracle.synth(int(rng.integers(1, 5 << 20)), 70 + i, kind)))
    files.append(("dir0/copy_of_03.bin", files[3][1].copy()))       # duplicate content: shared chunks
    files.append(("empty.bin", np.zeros(0, np.uint8)))
    files.append(("zeros/all_zero.bin", np.zeros(3 << 20, np.uint8)))   # every chunk identical
    files.append(("exact_part.bin", oracle.synth(target * 1024 if target * 1024 <= (8 << 20) else 1 << 20, 5, 0)))  # empty trailing part
    expect, _ = ref.version_index(files, target, 0, tag)
    got = bulk_version_index(gpu, ref, files, target, tag)
    assert len(got) == len(expect)
    assert got == expect


def test_bulk_version_index_empty_tree_of_directories(gpu, ref):
    files = [("only/empty.bin", np.zeros(0, np.uint8))]
    expect, _ = ref.version_index(files, 65536, 0, 0)
    assert bulk_version_index(gpu, ref, files, 65536) == expect


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_bulk_stored_blocks_open_with_the_reference(gpu, oracle, ref, codec):
    """chunk -> dedup -> pack -> device block assembly -> compress -> stored-block images, all on the device.
    Every image must be a stored block the REFERENCE reads (Longtail_ReadStoredBlockFromBuffer): BlockIndex equal to
    Longtail_CreateBlockIndex over the same chunks, payload decoded by the reference codec == the chunks' bytes."""
    import ctypes as C

    from longtail_amd.lib import pack_blocks

    target, max_block, max_chunks = 32768, 1 << 20, 64
    mn, av, mx = chunker_params(target)
    files = [oracle.synth(int(n), 300 + i, i % 3) for i, n in enumerate([3 << 20, 70000, 2 << 20, 1 << 20, 5, 2 << 20])]
    files.append(files[2].copy())  # duplicate content: its chunks must not be stored twice
    dev, offs = to_device(files)
    plan = gpu.make_plan(offs, [len(f) for f in files], mn, av, mx)
    total, d_off, d_len, d_hash, _ = gpu.chunk_hash(plan, dev)
    plan.close()
    first_idx, uniq = gpu.dedup_first_seen(d_hash[:total])
    keep = first_idx.to(torch.int64) == torch.arange(total, device="cuda")
    u_off, u_len, u_hash = d_off[:total][keep].contiguous(), d_len[:total][keep].contiguous(), d_hash[:total][keep].contiguous()
    nu = int(u_len.numel())
    assert nu == int(uniq.item()) < total
    lens_h = u_len.cpu().numpy().view(np.uint32)
    starts = pack_blocks(lens_h, max_block, max_chunks)
    nb = len(starts) - 1
    cs = np.concatenate([[0], np.cumsum(lens_h.astype(np.int64))])
    raw_sizes = (cs[starts[1:]] - cs[starts[:-1]]).astype(np.int64)
    # device block assembly: the unique chunks back to back == the blocks back to back
    gathered = torch.empty(int(cs[-1]) + 64, dtype=torch.uint8, device="cuda")
    gpu.gather_ranges(dev, u_off, u_len, gathered, torch.from_numpy(cs[:-1].copy()).cuda())
    tag = ref.lz4_type if codec == "lz4" else ref.zstd_default
    bound = raw_sizes + raw_sizes // 255 + 16 if codec == "lz4" else raw_sizes + (raw_sizes >> 8) + 64
    hdr = np.array([gpu.lib.dll.lthip_stored_block_header_size(int(starts[b + 1] - starts[b])) for b in range(nb)], np.int64)
    img = np.concatenate([[0], np.cumsum((hdr + bound + 7) // 8 * 8)])
    arena = torch.zeros(int(img[-1]) + 64, dtype=torch.uint8, device="cuda")
    fn = gpu.lz4_compress_blocks if codec == "lz4" else gpu.zstd_compress_blocks
    comp = fn(gathered, cs[starts[:-1]], raw_sizes, arena, img[:-1] + hdr, bound)
    gpu.write_stored_block_headers(starts, u_hash, u_len, tag, raw_sizes, comp, arena, img[:-1])
    gpu.sync()
    host = arena.cpu().numpy()
    comp_h = comp.cpu().numpy().view(np.uint32)
    hashes_h, raw_all = u_hash.cpu().numpy().view(np.uint64), gathered.cpu().numpy()
    for b in range(nb):
        image = host[int(img[b]) : int(img[b]) + int(hdr[b]) + int(comp_h[b])].copy()
        c0, c1 = int(starts[b]), int(starts[b + 1])
        h, s = np.ascontiguousarray(hashes_h[c0:c1]), np.ascontiguousarray(lens_h[c0:c1])
        out = np.zeros(int(raw_sizes[b]) + 8, np.uint8)
        n = C.c_uint64(0)
        err = ref.dll.refh_open_stored_block(image.ctypes.data, len(image), c1 - c0, h.ctypes.data, s.ctypes.data, tag, out.ctypes.data,
                                             int(raw_sizes[b]), C.byref(n))
        assert err == 0, (b, err)
        assert n.value == raw_sizes[b] and (out[: n.value] == raw_all[int(cs[c0]) : int(cs[c1])]).all()


def test_bulk_missing_content_store_index_is_byte_identical(gpu, ref):
    """which chunks does the store lack and how are they blocked -- the serialized StoreIndex must equal
    Longtail_CreateMissingContent + Longtail_WriteStoreIndexToBuffer on the same arrays (first-occurrence order, tag breaks,
    block hashes)."""
    import ctypes as C

    rng = np.random.default_rng(17)
    for n, ne, max_block, max_chunks, ntags in ((5000, 1500, 1 << 20, 64, 1), (20000, 0, 8 << 20, 1024, 3), (300, 300, 65536, 4, 2),
                                                (4000, 9000, 262144, 1024, 1)):
        hashes = rng.integers(1, 2**63, n, dtype=np.int64).astype(np.uint64)
        sizes = rng.integers(1, 131072, n, dtype=np.int64).astype(np.uint32)
        tags = (np.arange(n) * ntags // n).astype(np.uint32) * 7 + 1  # runs of equal tags
        # the store holds a random subset of the version's chunks plus unrelated ones
        existing = np.concatenate([rng.choice(hashes, min(ne, n) // 2, replace=False),
                                   rng.integers(1, 2**63, ne - min(ne, n) // 2, dtype=np.int64).astype(np.uint64)]) if ne else np.zeros(0, np.uint64)
        rng.shuffle(existing)
        buf, size = C.c_void_p(), C.c_uint64(0)
        err = ref.dll.refh_missing_content(existing.ctypes.data if ne else None, ne, hashes.ctypes.data, sizes.ctypes.data, tags.ctypes.data,
                                           n, max_block, max_chunks, C.byref(buf), C.byref(size))
        assert err == 0
        expect = bytes((C.c_ubyte * size.value).from_address(buf.value))
        ref.dll.refh_free(buf)
        got = gpu.create_missing_content(torch.from_numpy(existing.view(np.int64)).cuda() if ne else None,
                                         torch.from_numpy(hashes.view(np.int64)).cuda(), torch.from_numpy(sizes.view(np.int32)).cuda(),
                                         tags, max_block, max_chunks)
        assert got == expect, (n, ne, len(got), len(expect))


def make_store_index(rng, nblocks, max_chunks, pool, ntags=3):
    """A serialized StoreIndex (layout src/longtail.c:8913-8931) whose blocks draw their chunks from `pool` WITH repetition across
    blocks (a chunk can live in several blocks: what GetExistingStoreIndex's most-used-first walk is about)."""
    counts = rng.integers(1, max_chunks + 1, nblocks).astype(np.uint32)
    m = int(counts.sum())
    chunk_hashes = rng.choice(pool, m).astype(np.uint64)
    chunk_sizes = rng.integers(1, 100000, m).astype(np.uint32)
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
    block_hashes = rng.integers(1, 2**63, nblocks, dtype=np.int64).astype(np.uint64)
    tags = rng.integers(1, ntags + 1, nblocks).astype(np.uint32)
    head = np.array([1 << 24, 0x626C6B33, nblocks, m], np.uint32)
    return b"".join(a.tobytes() for a in (head, block_hashes, chunk_hashes, offsets, counts, tags, chunk_sizes))


def test_bulk_existing_store_index_is_byte_identical(gpu, ref):
    """lthip_get_existing_store_index == Longtail_GetExistingStoreIndex + Longtail_WriteStoreIndexToBuffer."""
    rng = np.random.default_rng(23)
    for nblocks, max_chunks, npool, nwant in ((40, 12, 200, 120), (300, 64, 5000, 3000), (8, 4, 10, 10), (100, 1024, 40000, 100),
                                              (50, 20, 300, 0)):
        pool = rng.integers(1, 2**63, npool, dtype=np.int64).astype(np.uint64)
        si = make_store_index(rng, nblocks, max_chunks, pool)
        wanted = np.concatenate([rng.choice(pool, nwant // 2 * 2 // 2), rng.integers(1, 2**63, nwant - nwant // 2, dtype=np.int64).astype(np.uint64)]) \
            if nwant else np.zeros(0, np.uint64)
        wanted = np.concatenate([wanted, wanted[: len(wanted) // 5]])  # duplicates in the request
        rng.shuffle(wanted)
        d_wanted = torch.from_numpy(wanted.view(np.int64).copy()).cuda() if len(wanted) else None
        for pct in (0, 1, 30, 75, 100, 101):
            expect = ref.get_existing_store_index(si, wanted, pct)
            got = gpu.get_existing_store_index(si, d_wanted, pct)
            assert got == expect, (nblocks, max_chunks, npool, nwant, pct, len(got), len(expect))
    # the whole store is wanted: every block that brings something new is taken
    pool = rng.integers(1, 2**63, 1000, dtype=np.int64).astype(np.uint64)
    si = make_store_index(rng, 60, 50, pool)
    everything = torch.from_numpy(pool.view(np.int64).copy()).cuda()
    assert gpu.get_existing_store_index(si, everything, 0) == ref.get_existing_store_index(si, pool, 0)
