"""-m gpu: the native ingest session (lthip_ingest_*, SURVEY.md §8d's metric = CreateVersionIndex + CreateMissingContent +
WriteContent) against the reference run on the same tree (oracle/_ref):

  * serialized VersionIndex          == Longtail_CreateVersionIndex + Longtail_WriteVersionIndexToBuffer, byte for byte
  * serialized StoreIndex            == Longtail_CreateMissingContent + Longtail_WriteStoreIndexToBuffer on the version's chunks
  * every stored-block image         opens with Longtail_ReadStoredBlockFromBuffer, BlockIndex == Longtail_CreateBlockIndex,
                                       payload decoded by the REFERENCE codec == the block's chunk bytes
  * R ranks (1 process, R sessions fed the job-ordered arrays the exchange would deliver, LPT / range / mod assignments, an
    asset whose parts straddle ranks): rank 0's VersionIndex is the single-rank one; every rank's StoreIndex equals
    Longtail_CreateMissingContent against a store holding the other ranks' chunks; the ranks' chunk sets partition the
    version's unique chunks."""
import ctypes as C

import numpy as np
import pytest
import torch

from longtail_amd.dist import JobPartition
from longtail_amd.lib import Ingest, chunker_params
from tests.gpu_util import to_device

pytestmark = pytest.mark.gpu


def make_files(oracle, target):
    part = target * 1024
    rng = np.random.default_rng(target)
    files = []
    for i in range(12):
        files.append((f"dir{i % 3}/sub{i % 2}/file{i:02d}.bin", oracle.synth(int(rng.integers(1, 3 << 20)), 70 + i, i % 3)))
    files.append(("dir0/copy_of_03.bin", files[3][1].copy()))                # duplicate content
    files.append(("empty.bin", np.zeros(0, np.uint8)))
    files.append(("zeros/all_zero.bin", np.zeros(2 << 20, np.uint8)))         # every chunk identical
    files.append(("big/multi_part.bin", oracle.synth(min(part * 3 + 4321, 9 << 20), 5, 1)))  # several jobs
    if part <= (4 << 20):
        files.append(("big/exact_parts.bin", oracle.synth(part * 2, 6, 0)))  # ends with an empty job
    return files


def rank_session(gpu, ref, files, target, world, rank, policy, codec, max_block, max_chunks, tag, all_lists=None, arena_bytes=None,
                 batch_bytes=0):
    """Chunk + hash the rank's jobs on the GPU; with all_lists (job -> (hashes, lens) of every job) run the session."""
    by_name = {n: d for n, d in files}
    paths, sizes, offs, perms, path_data = ref.tree_file_infos(files)
    part = JobPartition(sizes, target, world, policy)
    mine = part.jobs_of(rank)
    blobs = []
    for j in mine:
        data = by_name.get(paths[int(part.job_asset[j])], np.zeros(0, np.uint8))
        o, s = int(part.job_offset[j]), int(part.job_size[j])
        blobs.append(data[o : o + s])
    dev, part_offs = to_device(blobs) if blobs else (torch.zeros(64, dtype=torch.uint8, device="cuda"), [])
    mn, av, mx = chunker_params(target)
    plan = gpu.make_plan(part_offs, [len(b) for b in blobs], mn, av, mx)
    total, d_off, d_len, d_hash, d_first = gpu.chunk_hash(plan, dev)
    plan.close()
    first = d_first.cpu().numpy().view(np.uint32).astype(np.int64)
    local = dict(part=part, mine=mine, dev=dev, total=total, d_off=d_off, d_len=d_len, d_hash=d_hash, d_first=d_first, first=first,
                 infos=(paths, sizes, offs, perms, path_data))
    if all_lists is None:
        return local
    # the job-ordered arrays of all ranks (what dist.exchange_chunks delivers)
    all_hash = torch.cat([all_lists[j][0] for j in range(part.job_count)]) if part.job_count else torch.zeros(0, dtype=torch.int64, device="cuda")
    all_lens = torch.cat([all_lists[j][1] for j in range(part.job_count)]) if part.job_count else torch.zeros(0, dtype=torch.int32, device="cuda")
    job_first = np.concatenate([[0], np.cumsum([int(all_lists[j][0].numel()) for j in range(part.job_count)])]).astype(np.uint64)
    n_all = int(job_first[-1])
    ing = Ingest(gpu, target, max_block, max_chunks, codec, compression_type=tag, batch_bytes=batch_bytes)
    tree, keep = Ingest.tree(sizes.copy(), offs.copy(), perms.copy(), path_data, part.job_asset.copy(), job_first.copy(),
                             my_jobs=None if world == 1 else mine.copy())
    vi = torch.zeros(gpu.lib.dll.lthip_version_index_size(len(sizes), n_all, n_all, len(path_data)) + 64, dtype=torch.uint8).pin_memory()
    ing.index(tree, all_hash, all_lens, n_all, d_off, d_first, total, vi if rank == 0 else None)
    # the host arrays of the tree belong to the caller again once index() has returned (the helper thread that serializes the
    # VersionIndex reads the session's own copy): scribble over them
    for a in keep:
        if isinstance(a, np.ndarray):
            a.fill(0xEE)
    arena = torch.zeros(arena_bytes or (64 << 20), dtype=torch.uint8, device="cuda")
    ing.write(dev, arena)
    si = torch.zeros(16 + 32 * max(total, 1) + 64, dtype=torch.uint8).pin_memory()
    res = ing.finish(si)
    local.update(ing=ing, res=res, vi=bytes(vi.numpy()[: res.version_index_size]) if rank == 0 else None,
                 si=bytes(si.numpy()[: res.store_index_size]), arena=arena, comp=ing.compressed_sizes(res.blocks),
                 all_hash=all_hash, all_lens=all_lens, job_first=job_first)
    return local


def parse_store_index(blob):
    head = np.frombuffer(blob[:16], np.uint32)
    nb, m = int(head[2]), int(head[3])
    o = 16
    bh = np.frombuffer(blob[o : o + nb * 8], np.uint64); o += nb * 8
    ch = np.frombuffer(blob[o : o + m * 8], np.uint64); o += m * 8
    bo = np.frombuffer(blob[o : o + nb * 4], np.uint32); o += nb * 4
    bc = np.frombuffer(blob[o : o + nb * 4], np.uint32); o += nb * 4
    bt = np.frombuffer(blob[o : o + nb * 4], np.uint32); o += nb * 4
    cs = np.frombuffer(blob[o : o + m * 4], np.uint32); o += m * 4
    assert o == len(blob)
    return dict(block_hashes=bh, chunk_hashes=ch, block_offsets=bo, block_counts=bc, block_tags=bt, chunk_sizes=cs)


def ref_missing_content(ref, existing, hashes, sizes, tags, max_block, max_chunks):
    buf, size = C.c_void_p(), C.c_uint64(0)
    existing = np.ascontiguousarray(existing, dtype=np.uint64)
    hashes, sizes, tags = (np.ascontiguousarray(a) for a in (hashes, sizes, tags))
    err = ref.dll.refh_missing_content(existing.ctypes.data if len(existing) else None, len(existing), hashes.ctypes.data, sizes.ctypes.data,
                                       tags.ctypes.data, len(hashes), max_block, max_chunks, C.byref(buf), C.byref(size))
    assert err == 0
    out = bytes((C.c_ubyte * size.value).from_address(buf.value))
    ref.dll.refh_free(buf)
    return out


def version_unique_lists(vi_blob):
    """(chunk hashes, sizes, tags) of a serialized VersionIndex (layout src/longtail.c:2551-2584)."""
    h = np.frombuffer(vi_blob[:24], np.uint32)
    na, nu, ni = int(h[3]), int(h[4]), int(h[5])
    o = 24 + na * (8 + 8 + 8 + 4 + 4) + ni * 4
    hashes = np.frombuffer(vi_blob[o : o + nu * 8], np.uint64); o += nu * 8
    sizes = np.frombuffer(vi_blob[o : o + nu * 4], np.uint32); o += nu * 4
    tags = np.frombuffer(vi_blob[o : o + nu * 4], np.uint32)
    return hashes, sizes, tags


def check_images(gpu, ref, sess, codec_id, tag, max_block):
    """Every stored-block image of the (single-batch) arena through the reference's reader and codec."""
    si = parse_store_index(sess["si"])
    res, comp = sess["res"], sess["comp"]
    host = sess["arena"].cpu().numpy()
    # owned chunks' bytes in version order: offsets from the session's inputs
    first_idx, _ = gpu.dedup_first_seen(sess["all_hash"])
    pos = 0
    bound = (lambda n: n + n // 255 + 16) if codec_id == 0 else (lambda n: int(gpu.lib.dll.lthip_zstd_bound(n)))
    data_host = sess["dev"].cpu().numpy()
    # owned chunk byte ranges: recompute from local chunk lists + ownership (store index lists them in order)
    l_off = sess["d_off"].cpu().numpy().view(np.uint64)[: sess["total"]]
    l_len = sess["d_len"].cpu().numpy().view(np.uint32)[: sess["total"]]
    l_hash = sess["d_hash"].cpu().numpy().view(np.uint64)[: sess["total"]]
    owned_pos = {}
    for k in range(sess["total"]):
        owned_pos.setdefault(int(l_hash[k]), (int(l_off[k]), int(l_len[k])))
    for b in range(res.blocks):
        c0, n = int(si["block_offsets"][b]), int(si["block_counts"][b])
        raw = int(si["chunk_sizes"][c0 : c0 + n].astype(np.int64).sum())
        hdr = int(gpu.lib.dll.lthip_stored_block_header_size(n))
        image = host[pos : pos + hdr + int(comp[b])].copy()
        pos += (hdr + bound(raw) + 63) // 64 * 64
        h = np.ascontiguousarray(si["chunk_hashes"][c0 : c0 + n])
        s = np.ascontiguousarray(si["chunk_sizes"][c0 : c0 + n])
        out = np.zeros(raw + 8, np.uint8)
        got = C.c_uint64(0)
        err = ref.dll.refh_open_stored_block(image.ctypes.data, len(image), n, h.ctypes.data, s.ctypes.data, int(si["block_tags"][b]),
                                             out.ctypes.data, raw, C.byref(got))
        assert err == 0, (b, err)
        expect = np.concatenate([data_host[owned_pos[int(x)][0] : owned_pos[int(x)][0] + owned_pos[int(x)][1]] for x in h])
        assert got.value == raw and (out[:raw] == expect).all(), b


@pytest.mark.parametrize("target,codec,max_block,max_chunks", [(65536, "lz4", 8 << 20, 1024), (4096, "zstd", 1 << 20, 64), (1024, "lz4", 262144, 16),
                                                               (4096, "zstd:ztd4", 1 << 20, 64)])
def test_ingest_session_single_rank_matches_reference(gpu, oracle, ref, target, codec, max_block, max_chunks):
    files = make_files(oracle, target)
    tag = ref.lz4_type if codec == "lz4" else ref.zstd_default
    if codec == "zstd:ztd4":  # the settings id is the block tag AND selects the parse (lthip_zstd_quality_of_settings): "high"
        codec, tag = "zstd", int(ref.dll.refh_zstd_type(3))
        assert tag == 0x7A746434 and gpu.lib.dll.lthip_zstd_quality_of_settings(tag) == 1
        assert [gpu.lib.dll.lthip_zstd_quality_of_settings(0x7A746430 + k) for k in range(7)] == [0, 0, 0, 2, 1, 2, 0]
        assert gpu.lib.dll.lthip_zstd_quality_of_settings(ref.lz4_type) == 0
    probe = rank_session(gpu, ref, files, target, 1, 0, "range", codec, max_block, max_chunks, tag)
    lists = {int(j): (probe["d_hash"][int(probe["first"][m]) : int(probe["first"][m + 1])], probe["d_len"][int(probe["first"][m]) : int(probe["first"][m + 1])])
             for m, j in enumerate(probe["mine"])}
    sess = rank_session(gpu, ref, files, target, 1, 0, "range", codec, max_block, max_chunks, tag, lists, arena_bytes=96 << 20)
    expect_vi, _ = ref.version_index(files, target, 0, tag)
    assert sess["vi"] == expect_vi, "VersionIndex differs from Longtail_CreateVersionIndex"
    uh, us, ut = version_unique_lists(expect_vi)
    expect_si = ref_missing_content(ref, np.zeros(0, np.uint64), uh, us, ut, max_block, max_chunks)
    assert sess["si"] == expect_si, "StoreIndex differs from Longtail_CreateMissingContent"
    res = sess["res"]
    assert res.chunks_all == res.chunks_local and res.unique_all == res.unique_local == len(uh) < res.chunks_all
    assert res.raw_bytes == int(us.astype(np.int64).sum()) and 0 < res.compressed_bytes == int(sess["comp"].astype(np.int64).sum())
    assert res.gathered_blocks > 0  # the duplicate / zero files leave holes: some blocks are not one byte range
    check_images(gpu, ref, sess, 0 if codec == "lz4" else 1, tag, max_block)


@pytest.mark.parametrize("target,codec,max_block,max_chunks,batch", [(1024, "lz4", 262144, 16, 1 << 20), (4096, "zstd", 1 << 20, 64, 3 << 20),
                                                                       (1024, "lz4", 65536, 1024, 1 << 18)])
def test_ingest_session_packing_in_slices_is_the_packing_at_once(gpu, oracle, ref, target, codec, max_block, max_chunks, batch):
    """A tree of more than two codec batches: lthip_ingest_index packs the blocks of the first batch from the head of the owned chunks'
    lists (the rest of the lists arrives by the side stream), lthip_ingest_write the rest behind its first batch; the VersionIndex is
    serialized by the helper thread meanwhile.  Same VersionIndex, StoreIndex and compressed sizes as the one-batch session."""
    files = make_files(oracle, target)
    tag = ref.lz4_type if codec == "lz4" else ref.zstd_default
    probe = rank_session(gpu, ref, files, target, 1, 0, "range", codec, max_block, max_chunks, tag)
    lists = {int(j): (probe["d_hash"][int(probe["first"][m]) : int(probe["first"][m + 1])], probe["d_len"][int(probe["first"][m]) : int(probe["first"][m + 1])])
             for m, j in enumerate(probe["mine"])}
    once = rank_session(gpu, ref, files, target, 1, 0, "range", codec, max_block, max_chunks, tag, lists, arena_bytes=96 << 20)
    for arena in (96 << 20, 3 * max_block):  # batches bounded by the codec's batch size / by a small arena
        sliced = rank_session(gpu, ref, files, target, 1, 0, "range", codec, max_block, max_chunks, tag, lists, arena_bytes=arena,
                              batch_bytes=batch)
        assert sliced["vi"] == once["vi"] and sliced["si"] == once["si"]
        assert sliced["res"].blocks == once["res"].blocks and sliced["res"].raw_bytes == once["res"].raw_bytes
        assert sliced["res"].compressed_bytes == once["res"].compressed_bytes and (sliced["comp"] == once["comp"]).all()
        assert sliced["res"].gathered_blocks == once["res"].gathered_blocks


@pytest.mark.parametrize("world,policy", [(2, "range"), (3, "lpt"), (4, "range"), (4, "mod")])
def test_ingest_sessions_of_r_ranks_partition_the_reference_result(gpu, oracle, ref, world, policy):
    target, codec, max_block, max_chunks = 1024, "lz4", 262144, 64
    tag = ref.lz4_type
    files = make_files(oracle, target)
    probes = [rank_session(gpu, ref, files, target, world, r, policy, codec, max_block, max_chunks, tag) for r in range(world)]
    lists = {}
    for p in probes:
        for m, j in enumerate(p["mine"]):
            a, b = int(p["first"][m]), int(p["first"][m + 1])
            lists[int(j)] = (p["d_hash"][a:b].clone(), p["d_len"][a:b].clone())
    part = probes[0]["part"]
    assert len(lists) == part.job_count
    multi = np.flatnonzero(np.bincount(part.job_asset) > 2)
    straddling = [a for a in multi if len(set(part.job_rank[part.job_asset == a].tolist())) > 1]
    if policy == "mod":
        assert straddling  # consecutive jobs alternate ranks: the multi-part assets' parts straddle ranks
    sessions = [rank_session(gpu, ref, files, target, world, r, policy, codec, max_block, max_chunks, tag, lists, arena_bytes=64 << 20)
                for r in range(world)]
    expect_vi, _ = ref.version_index(files, target, 0, tag)
    assert sessions[0]["vi"] == expect_vi, "rank 0's VersionIndex differs from the single-process reference"
    uh, us, ut = version_unique_lists(expect_vi)
    owned = [parse_store_index(s["si"])["chunk_hashes"] for s in sessions]
    allc = np.concatenate(owned)
    assert len(allc) == len(uh) and set(allc.tolist()) == set(uh.tolist())  # a partition of the version's unique chunks
    for r, s in enumerate(sessions):
        others = np.concatenate([owned[q] for q in range(world) if q != r]) if world > 1 else np.zeros(0, np.uint64)
        assert s["si"] == ref_missing_content(ref, others, uh, us, ut, max_block, max_chunks), f"rank {r} StoreIndex"
        assert s["res"].unique_all == len(uh) and s["res"].unique_local == len(owned[r])
        check_images(gpu, ref, s, 0, tag, max_block)


def test_sharded_first_seen_table_feeds_the_ingest_session(gpu, oracle):
    """lthip_dedup_min_ordinal (the owner's side of the hash-range-sharded first-seen table) against the serial pass, and a session that
    is GIVEN the first-seen index (lthip_ingest_set_first_seen) against one that builds its own table: same serialized VersionIndex."""
    rng = np.random.default_rng(7)
    n = 200_000
    pool = rng.integers(1, 2**63 - 1, size=60_000, dtype=np.int64)
    h = pool[rng.integers(0, len(pool), size=n)]
    h[1234] = -1  # the table's empty-key value is a legal hash
    h[99_999] = -1
    order = rng.permutation(n).astype(np.int32)  # an owner sees its items in any order, each with its global position
    first, uniq = gpu.dedup_min_ordinal(torch.from_numpy(h[order]).cuda(), torch.from_numpy(order).cuda())
    seen, exp = {}, np.zeros(n, np.int64)
    for i, x in enumerate(h.tolist()):
        exp[i] = seen.setdefault(x, i)
    assert uniq == len(seen)
    assert (first.cpu().numpy().astype(np.int64) == exp[order]).all()
    # two subsets (two "owners") partition the hashes: minima per owner == global minima, distinct counts add up
    own = (h >> 40) & 1
    tot = 0
    for o in (0, 1):
        idx = np.flatnonzero(own == o).astype(np.int32)
        f, u = gpu.dedup_min_ordinal(torch.from_numpy(h[idx]).cuda(), torch.from_numpy(idx).cuda())
        assert (f.cpu().numpy().astype(np.int64) == exp[idx]).all()
        tot += u
    assert tot == len(seen)


def test_duplicate_bearing_tree_at_8_gib_matches_reference(gpu, ref):
    """bench.py's `secondary.dedup` workload (compressible 1 MiB files, a quarter of them repeated: whole-file copies and copies entered
    20 KiB later, whose chunking falls back into step) at 8 GiB: first-seen hits (src/longtail.c:2951-2970), CreateMissingContent with
    chunks to drop (:6801-6860), blocks assembled from non-contiguous unique chunks (:4640-4721).  Serialized VersionIndex and StoreIndex
    byte-identical with the reference's."""
    import os

    from bench import KINDS, asset_seeds, make_tree

    gib = float(os.environ.get("LONGTAIL_DEDUP_GIB", "8"))
    FILE = 1 << 20
    tree = make_tree("files", int(gib * (1 << 30)), FILE, dups=True)
    n = tree["nfiles"]
    data = torch.empty(n * FILE + 256, dtype=torch.uint8, device="cuda")
    seeds = asset_seeds(0x10C0FFEE, 0, n)[tree["seed_of"]]
    gpu.synth_fill(data, np.arange(n, dtype=np.uint64) * np.uint64(FILE), np.full(n, FILE, np.uint64), seeds, KINDS["mixed"], skips=tree["shift"])
    gpu.sync()
    host = data[: n * FILE].cpu().numpy()
    del data
    names = tree["path_data"].decode().split("\0")[:n]
    files = [(names[i], host[i * FILE : (i + 1) * FILE]) for i in range(n)]
    assert (files[5][1] == files[2][1]).all() and not (files[7][1] == files[2][1]).all()  # whole copy / shifted copy
    target, max_block, max_chunks, tag = 65536, 8 << 20, 1024, ref.lz4_type
    probe = rank_session(gpu, ref, files, target, 1, 0, "range", "lz4", max_block, max_chunks, tag)
    lists = {int(j): (probe["d_hash"][int(probe["first"][m]) : int(probe["first"][m + 1])], probe["d_len"][int(probe["first"][m]) : int(probe["first"][m + 1])])
             for m, j in enumerate(probe["mine"])}
    del probe
    batch = 2 << 30
    limit = max_block + max_block // 10
    arena_bytes = batch + batch // 128 + (batch // max_block + 4) * (16384 + 64) + 2 * (limit + limit // 128 + 16384)
    sess = rank_session(gpu, ref, files, target, 1, 0, "range", "lz4", max_block, max_chunks, tag, lists, arena_bytes=arena_bytes, batch_bytes=batch)
    expect_vi, _ = ref.version_index(files, target, min(32, os.cpu_count() or 1), tag)
    assert sess["vi"] == expect_vi, "VersionIndex differs from Longtail_CreateVersionIndex"
    uh, us, ut = version_unique_lists(expect_vi)
    expect_si = ref_missing_content(ref, np.zeros(0, np.uint64), uh, us, ut, max_block, max_chunks)
    assert sess["si"] == expect_si, "StoreIndex differs from Longtail_CreateMissingContent"
    res = sess["res"]
    assert res.unique_all == len(uh) and 0.70 * res.chunks_all < res.unique_all < 0.80 * res.chunks_all  # a quarter of the files repeat
    assert res.gathered_blocks > 0 and res.gathered_bytes > 0 and res.raw_bytes == int(us.astype(np.int64).sum())


@pytest.mark.gpu
def test_link_copy_moves_pinned_host_memory_both_ways(gpu):
    """lthip_link_copy (the host-fed loop's transfer, include/longtail_hip.h): pinned host -> device and device -> pinned host by the
    compute units, sizes with a tail below 16 bytes, on two contexts at once; a misaligned pointer is refused.  lthip_gather_ranges
    writes block images straight into pinned host memory."""
    from longtail_amd.lib import Context, LongtailHipError

    rng = np.random.default_rng(5)
    s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
    c_in, c_out = Context(0, stream=s_in.cuda_stream, lib=gpu.lib), Context(0, stream=s_out.cuda_stream, lib=gpu.lib)
    for n in (1, 15, 16, 17, 4096 + 7, (64 << 20) + 13):
        src = torch.from_numpy(rng.integers(0, 256, n + 32, dtype=np.uint8)).pin_memory()
        back = torch.zeros(n + 32, dtype=torch.uint8).pin_memory()
        dev_a = torch.zeros(n + 32, dtype=torch.uint8, device="cuda")
        dev_b = torch.from_numpy(rng.integers(0, 256, n + 32, dtype=np.uint8)).cuda()
        torch.cuda.synchronize()
        c_in.link_copy(dev_a, src, n)
        c_out.link_copy(back, dev_b, n)
        c_in.sync()
        c_out.sync()
        assert (dev_a[:n].cpu() == src[:n]).all() and int(dev_a[n:].sum()) == 0, n
        assert (back[:n] == dev_b[:n].cpu()).all() and int(back[n:].sum()) == 0, n
    with pytest.raises(LongtailHipError):
        c_in.link_copy(dev_a[1:], src, 64)
    # ranges of a device arena -> pinned host memory
    offs = np.array([16, 100000, 7 << 20], np.int64)
    lens = np.array([4093, 65536, 1 << 20], np.int32)
    dst = np.array([0, 4096, 4096 + 65536], np.int64)
    host = torch.zeros(int(dst[-1] + lens[-1]), dtype=torch.uint8).pin_memory()
    c_out.gather_ranges(dev_b, torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), host, torch.from_numpy(dst).cuda())
    c_out.sync()
    ref_b = dev_b.cpu()
    for o, l, d in zip(offs, lens, dst):
        assert (host[d : d + l] == ref_b[o : o + l]).all()
    c_in.close()
    c_out.close()


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_images_as_a_host_fed_embedder_downloads_them(gpu, oracle, ref, codec):
    """lthip_ingest_images (include/longtail_hip.h: "exactly what Longtail_WriteStoredBlockToBuffer produces"): every image, cut out of
    the arena by the offsets and sizes the call returns, opens with the reference's Longtail_ReadStoredBlockFromBuffer and decodes with
    the reference codec to the block's chunk bytes.  lthip_ingest_finish may be called again (a retry with a larger StoreIndex buffer,
    as res.store_index_size invites): the image sizes stay header + payload, not header + 2 * payload.  The result struct is written
    up to the size the CALLER states and never past it; an unset struct_size is EINVAL before any work."""
    from longtail_amd.lib import IngestResult, LongtailHipError

    target, max_block, max_chunks = 4096, 1 << 20, 64
    files = make_files(oracle, target)
    tag = ref.lz4_type if codec == "lz4" else ref.zstd_default
    probe = rank_session(gpu, ref, files, target, 1, 0, "range", codec, max_block, max_chunks, tag)
    lists = {int(j): (probe["d_hash"][int(probe["first"][m]) : int(probe["first"][m + 1])], probe["d_len"][int(probe["first"][m]) : int(probe["first"][m + 1])])
             for m, j in enumerate(probe["mine"])}
    sess = rank_session(gpu, ref, files, target, 1, 0, "range", codec, max_block, max_chunks, tag, lists, arena_bytes=96 << 20)
    ing, res = sess["ing"], sess["res"]
    first, offs, sizes = ing.images()
    assert first == 0 and len(offs) == res.blocks == len(sizes)
    dll = gpu.lib.dll
    # ---- a second finish: too small a buffer (ENOMEM + the size wanted), then the retry; sizes unchanged ----
    small = torch.zeros(16, dtype=torch.uint8).pin_memory()
    with pytest.raises(LongtailHipError) as e:
        ing.finish(small)
    assert e.value.code == 12
    si2 = torch.zeros(res.store_index_size, dtype=torch.uint8).pin_memory()
    res2 = ing.finish(si2)
    assert bytes(si2.numpy()[: res2.store_index_size]) == sess["si"]
    f2, offs2, sizes2 = ing.images()
    assert f2 == first and (offs2 == offs).all() and (sizes2 == sizes).all(), "a second finish changed the image sizes"
    # ---- struct_size: 16 -> two fields written, the canary behind them untouched; 0 -> EINVAL ----
    buf = (C.c_uint64 * 12)(*([0xA5A5A5A5A5A5A5A5] * 12))
    buf[0] = 16
    assert dll.lthip_ingest_finish(ing.h, None, 0, C.byref(buf)) == 0
    assert buf[0] == 16 and buf[1] == res.chunks_all and all(buf[i] == 0xA5A5A5A5A5A5A5A5 for i in range(2, 12))
    unset = IngestResult()
    assert dll.lthip_ingest_finish(ing.h, None, 0, C.byref(unset)) == 22  # EINVAL
    assert (ing.images()[2] == sizes).all()
    # ---- the images through the reference ----
    si = parse_store_index(sess["si"])
    host = sess["arena"].cpu().numpy()
    data_host = sess["dev"].cpu().numpy()
    l_off = sess["d_off"].cpu().numpy().view(np.uint64)[: sess["total"]]
    l_len = sess["d_len"].cpu().numpy().view(np.uint32)[: sess["total"]]
    l_hash = sess["d_hash"].cpu().numpy().view(np.uint64)[: sess["total"]]
    where = {}
    for k in range(sess["total"]):
        where.setdefault(int(l_hash[k]), (int(l_off[k]), int(l_len[k])))
    for b in range(res.blocks):
        c0, n = int(si["block_offsets"][b]), int(si["block_counts"][b])
        raw = int(si["chunk_sizes"][c0 : c0 + n].astype(np.int64).sum())
        image = host[int(offs[b]) : int(offs[b]) + int(sizes[b])].copy()
        assert int(sizes[b]) == int(dll.lthip_stored_block_header_size(n)) + int(sess["comp"][b])
        h = np.ascontiguousarray(si["chunk_hashes"][c0 : c0 + n])
        s = np.ascontiguousarray(si["chunk_sizes"][c0 : c0 + n])
        out = np.zeros(raw + 8, np.uint8)
        got = C.c_uint64(0)
        err = ref.dll.refh_open_stored_block(image.ctypes.data, len(image), n, h.ctypes.data, s.ctypes.data, int(si["block_tags"][b]),
                                             out.ctypes.data, raw, C.byref(got))
        assert err == 0, (b, err)
        expect = np.concatenate([data_host[where[int(x)][0] : where[int(x)][0] + where[int(x)][1]] for x in h])
        assert got.value == raw and (out[:raw] == expect).all(), b
    ing.close()
