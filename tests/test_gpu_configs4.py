"""-m gpu: BASELINE.json configs[4] at its shape on one MI355X -- four 16 GiB PAK-style assets (257 jobs each: 256 x 64 MiB parts
+ the empty trailing job, src/longtail.c:2402), ZStd -- through properties that do not need a CPU to redo 64 GiB:

  * every part's chunks tile it, lengths obey min / max (hpcdcchunker.c:257-309); a sample of parts is re-chunked and re-hashed
    by the oracle: boundaries and BLAKE3 digests bit-exact;
  * intra-file segment sharding: the jobs assigned to each of 8 ranks (lthip_partition_jobs 'range': every asset straddles two
    ranks) processed rank by rank and put back in job order give exactly the single-pass chunk lists;
  * the ingest session's serialized VersionIndex: 4 assets, content hash of each = BLAKE3 of its chunk-hash array (oracle),
    chunk count / sizes consistent; the StoreIndex packs all unique chunks;
  * every ZStd frame of the tree's stored blocks decodes on the device to the bytes it came from; a sample of frames decodes
    with the REFERENCE's ZSTD_decompressDCtx.

Set LONGTAIL_FULL_GIB to shrink it (default 64: 4 x 16 GiB)."""
import os

import numpy as np
import pytest
import torch

from bench import asset_seeds
from longtail_amd.dist import JobPartition
from longtail_amd.lib import Ingest, chunker_params
from tests._libs import have_ref, ref as get_ref

pytestmark = pytest.mark.gpu

GIB = float(os.environ.get("LONGTAIL_FULL_GIB", "64"))
TARGET = 65536
PART = TARGET * 1024
BLOCK = 8 << 20


@pytest.fixture(scope="module")
def pak(gpu):
    free, _ = torch.cuda.mem_get_info()
    gib = min(GIB, max(1.0, (free / (1 << 30) - 60) / 1.2))
    asset = int(gib * (1 << 30)) // 4 // PART * PART  # whole parts: the asset ends with an empty job
    sizes = np.full(4, asset, np.uint64)
    part = JobPartition(sizes, TARGET, 1)
    assert part.job_count == 4 * (asset // PART + 1)
    p_size = part.job_size
    p_off = np.zeros(part.job_count, np.uint64)
    np.cumsum(p_size[:-1], out=p_off[1:])
    data = torch.empty(int(sizes.sum()) + 256, dtype=torch.uint8, device="cuda")
    seeds = asset_seeds(0xFAC, 0, 4)
    gpu.synth_fill(data, p_off, p_size, seeds[part.job_asset], 1, skips=part.job_offset)
    gpu.sync()
    mn, av, mx = chunker_params(TARGET)
    plan = gpu.make_plan(p_off, p_size, mn, av, mx)
    total, d_off, d_len, d_hash, d_first = gpu.chunk_hash(plan, data)
    plan.close()
    return dict(data=data, sizes=sizes, part=part, p_off=p_off, p_size=p_size, seeds=seeds, total=total, d_off=d_off[:total],
                d_len=d_len[:total], d_hash=d_hash[:total], d_first=d_first, first=d_first.cpu().numpy().view(np.uint32).astype(np.int64))


def test_parts_tile_and_sample_matches_oracle(gpu, oracle, pak):
    mn, av, mx = chunker_params(TARGET)
    first, part = pak["first"], pak["part"]
    lens = pak["d_len"].to(torch.int64)
    assert int(first[-1]) == pak["total"] and int(lens.sum().item()) == int(pak["sizes"].sum())
    csum = torch.cumsum(lens, 0)
    assert torch.equal(pak["d_off"], csum - lens)  # parts lie back to back: offsets are the running sum
    nonempty = np.flatnonzero(pak["p_size"] > 0)
    ends = csum[torch.from_numpy(first[nonempty + 1] - 1).cuda()]
    assert torch.equal(ends, torch.from_numpy((pak["p_off"][nonempty] + pak["p_size"][nonempty]).astype(np.int64)).cuda())
    assert (first[1:][pak["p_size"] == 0] == first[:-1][pak["p_size"] == 0]).all()  # the empty trailing jobs have no chunk
    last = torch.zeros(pak["total"], dtype=torch.bool, device="cuda")
    last[torch.from_numpy(first[nonempty + 1] - 1).cuda()] = True
    assert int((lens > mx).sum().item()) == 0 and int(((lens <= mn) & ~last).sum().item()) == 0
    rng = np.random.default_rng(4)
    len_h, hash_h = pak["d_len"].cpu().numpy().view(np.uint32), pak["d_hash"].cpu().numpy().view(np.uint64)
    for j in rng.choice(nonempty, 5, replace=False):
        o, s = int(pak["p_off"][j]), int(pak["p_size"][j])
        host = pak["data"][o : o + s].cpu().numpy()
        # the generator kernel produced the asset's bytes at the part's offset inside the asset
        assert (host[: 1 << 20] == oracle.synth(1 << 20, int(pak["seeds"][part.job_asset[j]]), 1, int(part.job_offset[j]))).all()
        _, e_len, e_hash = oracle.chunk_and_hash(host, mn, av, mx)
        a, b = first[j], first[j + 1]
        assert (len_h[a:b] == e_len).all() and (hash_h[a:b] == e_hash).all(), f"job {j}"


def test_intra_file_segment_sharding_reproduces_the_single_pass(gpu, pak):
    mn, av, mx = chunker_params(TARGET)
    p8 = JobPartition(pak["sizes"], TARGET, 8, "range")
    assert [len(set(p8.job_rank[p8.job_asset == a].tolist())) for a in range(4)] == [2, 2, 2, 2]  # every asset straddles two ranks
    hashes, lens = [None] * p8.job_count, [None] * p8.job_count
    for r in range(8):
        mine = p8.jobs_of(r)
        plan = gpu.make_plan(pak["p_off"][mine], pak["p_size"][mine], mn, av, mx)
        total, _, d_len, d_hash, d_first = gpu.chunk_hash(plan, pak["data"])
        plan.close()
        f = d_first.cpu().numpy().view(np.uint32).astype(np.int64)
        for m, j in enumerate(mine):
            hashes[j], lens[j] = d_hash[f[m] : f[m + 1]].clone(), d_len[f[m] : f[m + 1]].clone()
    assert torch.equal(torch.cat(hashes), pak["d_hash"]) and torch.equal(torch.cat(lens), pak["d_len"])


def test_ingest_session_indexes_and_zstd_frames(gpu, oracle, pak):
    part, total = pak["part"], pak["total"]
    names = [f"paks/pak{i}.pak" for i in range(4)]
    path_data = ("\0".join(names) + "\0").encode()
    path_offs = np.cumsum([0] + [len(n) + 1 for n in names[:-1]]).astype(np.uint32)
    ing = Ingest(gpu, TARGET, BLOCK, 1024, "zstd", batch_bytes=8 << 30)
    tree, keep = Ingest.tree(pak["sizes"], path_offs, np.full(4, 0o644, np.uint16), path_data, part.job_asset, pak["first"].astype(np.uint64))
    vi = torch.zeros(int(gpu.lib.dll.lthip_version_index_size(4, total, total, len(path_data))) + 64, dtype=torch.uint8).pin_memory()
    si = torch.zeros(16 + 32 * total + 64, dtype=torch.uint8).pin_memory()
    ing.index(tree, pak["d_hash"], pak["d_len"], total, pak["d_off"], pak["d_first"], total, vi)
    arena = torch.empty((8 << 30) + (8 << 30) // 64 + (64 << 20), dtype=torch.uint8, device="cuda")
    ing.write(pak["data"], arena)
    res = ing.finish(si)
    blob = bytes(vi.numpy()[: res.version_index_size])
    head = np.frombuffer(blob[:24], np.uint32)
    assert (int(head[1]), int(head[2]), int(head[3]), int(head[5])) == (0x626C6B33, TARGET, 4, total)
    nu = int(head[4])
    assert nu == res.unique_all == res.unique_local <= total
    o = 24
    path_hashes = np.frombuffer(blob[o : o + 32], np.uint64); o += 32
    content = np.frombuffer(blob[o : o + 32], np.uint64); o += 32
    assert (np.frombuffer(blob[o : o + 32], np.uint64) == pak["sizes"]).all(); o += 32
    counts = np.frombuffer(blob[o : o + 16], np.uint32); o += 16
    hash_h = pak["d_hash"].cpu().numpy().view(np.uint64)
    bounds = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    per_asset = [int(pak["first"][np.flatnonzero(part.job_asset == a)[-1] + 1] - pak["first"][np.flatnonzero(part.job_asset == a)[0]]) for a in range(4)]
    assert counts.tolist() == per_asset
    for a in range(4):
        assert int(content[a]) == oracle.blake3(hash_h[bounds[a] : bounds[a + 1]].view(np.uint8).copy())  # src/longtail.c:2518-2537
        assert int(path_hashes[a]) == oracle.blake3(np.frombuffer(names[a].encode(), np.uint8).copy())    # :1269-1300
    o += 16 + total * 4
    uh = np.frombuffer(blob[o : o + nu * 8], np.uint64); o += nu * 8
    us = np.frombuffer(blob[o : o + nu * 4], np.uint32)
    assert len(set(uh.tolist())) == nu and res.raw_bytes == int(us.astype(np.int64).sum())
    assert res.blocks >= res.raw_bytes // (BLOCK + BLOCK // 10) and 0 < res.compressed_bytes < 0.7 * res.raw_bytes
    sizes = ing.compressed_sizes(res.blocks)
    assert int(sizes.astype(np.int64).sum()) == res.compressed_bytes and (sizes > 0).all()
    ing.close()


def test_zstd_frames_of_the_whole_tree_decode(gpu, oracle, pak):
    """8 MiB stored blocks over the whole tree, batch by batch: compress, decode every frame on the device, compare; a sample
    through the reference decoder."""
    nbytes = int(pak["sizes"].sum())
    nblocks = (nbytes + BLOCK - 1) // BLOCK
    b_off = np.arange(nblocks, dtype=np.int64) * BLOCK
    b_size = np.minimum(BLOCK, nbytes - b_off).astype(np.int64)
    caps = b_size + (b_size >> 8) + 64
    per = 512
    arena = torch.empty(int(caps[:per].sum()) + per * 64 + 64, dtype=torch.uint8, device="cuda")
    back = torch.empty(per * BLOCK + 64, dtype=torch.uint8, device="cuda")
    r = get_ref() if have_ref() else None
    comp_total = 0
    for i in range(0, nblocks, per):
        j = min(nblocks, i + per)
        d_offs = np.concatenate([[0], np.cumsum((caps[i:j] + 63) // 64 * 64)[:-1]])
        sz = gpu.zstd_compress_blocks(pak["data"], b_off[i:j], b_size[i:j], arena, d_offs, caps[i:j]).cpu().numpy().view(np.uint32).astype(np.int64)
        assert (sz > 0).all()
        comp_total += int(sz.sum())
        out = gpu.zstd_decompress_blocks(arena, d_offs, sz, back, b_off[i:j] - b_off[i], b_size[i:j])
        assert (out.cpu().numpy().view(np.uint32) == b_size[i:j]).all()
        n = int(b_size[i:j].sum())
        assert torch.equal(back[:n], pak["data"][int(b_off[i]) : int(b_off[i]) + n])
        if r is not None and i == 0:
            for k in (0, 1, (j - i) // 2, j - i - 1):
                frame = arena[int(d_offs[k]) : int(d_offs[k]) + int(sz[k])].cpu().numpy()
                err, dec = r.decompress(1, frame, int(b_size[i + k]))
                assert err == 0 and (dec == pak["data"][int(b_off[i + k]) : int(b_off[i + k]) + int(b_size[i + k])].cpu().numpy()).all()
    assert comp_total < 0.7 * nbytes
