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
