"""Test infrastructure: ctypes loaders for the oracle (own C restatement) and, when present, the reference itself
(oracle/_ref/liblongtail_ref.so).  Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_SO = ROOT / "oracle" / "liblongtail_oracle.so"
REF_SO = ROOT / "oracle" / "_ref" / "liblongtail_ref.so"
GOLDEN = ROOT / "tests" / "golden"

vp, u64, u32, i32, i64, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int64, C.c_size_t


class IngestResult(C.Structure):
    _fields_ = [
        ("chunk_count", u64),
        ("hash_xor", u64),
        ("hash_sum", u64),
        ("compressed_bytes", u64),
        ("seconds_chunk", C.c_double),
        ("seconds_hash", C.c_double),
        ("seconds_compress", C.c_double),
    ]


class Oracle:
    def __init__(self):
        if not ORACLE_SO.exists():
            subprocess.run(["make", "-C", str(ROOT / "oracle"), "oracle"], check=True, capture_output=True)
        d = self.dll = C.CDLL(str(ORACLE_SO))

        def sig(name, res, args):
            f = getattr(d, name)
            f.restype, f.argtypes = res, args

        sig("lto_hpcdc_discriminator", u32, [u32])
        sig("lto_hpcdc_chunk_stream", u64, [vp, u64, u32, u32, u32, vp, u64])
        sig("lto_hpcdc_chunk_pure", u64, [vp, u64, u32, u32, u32, vp, u64])
        sig("lto_hpcdc_next_from_buffer", u64, [vp, u64, u32, u32, u32])
        sig("lto_buzhash_at", u32, [vp, u64])
        sig("lto_blake3_u64", u64, [vp, sz])
        sig("lto_blake3_u64_many", None, [vp, vp, vp, u64, vp])
        sig("lto_lz4_bound", sz, [sz])
        sig("lto_lz4_compress", i32, [vp, i32, vp, i32])
        sig("lto_lz4_decompress", i32, [vp, i32, vp, i32])
        sig("lto_synth_fill", None, [vp, u64, u64, u64, i32])
        sig("lto_synth_asset_seed", u64, [u64, u64])
        sig("lto_xorshift_fill", None, [vp, u64, u64])
        sig("lto_ingest", i32, [vp, u64, u64, u32, u32, i32, C.POINTER(IngestResult)])

    # -- conveniences on numpy uint8 arrays --
    def chunk(self, data: np.ndarray, mn: int, av: int, mx: int, pure: bool = False) -> np.ndarray:
        cap = len(data) // mn + 8
        lens = np.zeros(cap, np.uint32)
        fn = self.dll.lto_hpcdc_chunk_pure if pure else self.dll.lto_hpcdc_chunk_stream
        n = fn(data.ctypes.data, len(data), mn, av, mx, lens.ctypes.data, cap)
        assert n <= cap
        return lens[:n].copy()

    def blake3(self, data: np.ndarray) -> int:
        return int(self.dll.lto_blake3_u64(data.ctypes.data if len(data) else None, len(data)))

    def blake3_many(self, data: np.ndarray, offsets, lens) -> np.ndarray:
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint32)
        out = np.zeros(len(o), np.uint64)
        self.dll.lto_blake3_u64_many(data.ctypes.data, o.ctypes.data, l.ctypes.data, len(o), out.ctypes.data)
        return out

    def chunk_and_hash(self, data: np.ndarray, mn: int, av: int, mx: int):
        lens = self.chunk(data, mn, av, mx)
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64) if len(lens) else np.zeros(0, np.uint64)
        return offs, lens, self.blake3_many(data, offs, lens)

    def lz4_compress(self, data: np.ndarray) -> np.ndarray:
        cap = int(self.dll.lto_lz4_bound(len(data)))
        out = np.zeros(cap + 8, np.uint8)
        n = self.dll.lto_lz4_compress(data.ctypes.data, len(data), out.ctypes.data, cap)
        assert n > 0
        return out[:n].copy()

    def lz4_decompress(self, comp: np.ndarray, cap: int):
        out = np.zeros(cap + 8, np.uint8)
        n = self.dll.lto_lz4_decompress(comp.ctypes.data, len(comp), out.ctypes.data, cap)
        return n, out[: max(n, 0)]

    def synth(self, nbytes: int, seed: int, kind: int, offset: int = 0) -> np.ndarray:
        out = np.zeros(nbytes, np.uint8)
        if nbytes:
            self.dll.lto_synth_fill(out.ctypes.data, nbytes, seed, offset, kind)
        return out

    def xorshift(self, nbytes: int, seed: int = 0x9E3779B97F4A7C15) -> np.ndarray:
        """SURVEY.md §8(c)'s stream (tests/survey_vectors.py holds the slow Python statement of it)."""
        out = np.zeros(nbytes, np.uint8)
        if nbytes:
            self.dll.lto_xorshift_fill(out.ctypes.data, nbytes, seed)
        return out

    def asset_seed(self, tree_seed: int, index: int) -> int:
        return int(self.dll.lto_synth_asset_seed(tree_seed, index))


class Ref:
    """The reference library + oracle/ref_harness.c.  Only exists where oracle/_ref was built."""

    def __init__(self):
        d = self.dll = C.CDLL(str(REF_SO))

        def sig(name, res, args):
            f = getattr(d, name)
            f.restype, f.argtypes = res, args

        sig("refh_chunk_stream", i64, [vp, vp, vp, u64, u32, u32, u32, vp, vp, vp, u64])
        sig("refh_chunk_from_buffer", i64, [vp, vp, u64, u32, u32, u32, vp, u64])
        sig("refh_blake3", u64, [vp, u32])
        sig("refh_blake3_id", u32, [])
        sig("refh_lz4_type", u32, [])
        sig("refh_zstd_type", u32, [i32])
        sig("refh_codec_bound", sz, [i32, u32, sz])
        sig("refh_codec_compress", i32, [i32, u32, vp, sz, vp, sz, C.POINTER(sz)])
        sig("refh_codec_decompress", i32, [i32, vp, sz, vp, sz, C.POINTER(sz)])
        sig("refh_free", None, [vp])
        sig("refh_version_index", i32, [vp, vp, u32, vp, vp, vp, u32, i32, u32, C.POINTER(vp), C.POINTER(u64),
                                       C.POINTER(C.c_double)])
        sig("refh_ingest_roundtrip", i32, [vp, vp, vp, u32, u32, vp, vp, vp, u32, u32, u32, i32, C.POINTER(u64),
                                          C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double)])
        sig("refh_ingest_time", i32, [u32, u32, vp, vp, vp, u32, u32, u32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)])
        sig("refh_cpu_count", i32, [])
        sig("refh_get_existing_store_index", i32, [vp, u64, vp, u32, u32, C.POINTER(vp), C.POINTER(u64)])
        sig("refh_ingest_sweep", i32, [u32, u32, vp, vp, vp, u32, u32, u32, u32, vp, u32, vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)])
        sig("refh_ingest_sweep_apis", i32, [vp, vp, vp, u32, u32, vp, vp, vp, u32, u32, u32, u32, vp, u32, vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)])
        sig("refh_chunk_stream_failing_feeder", i64, [vp, vp, u64, u32, u32, u32, u64, i32, vp, u64, vp])
        sig("refh_version_index_cancel", i32, [vp, vp, u32, vp, vp, vp, u32, i32, u32, C.POINTER(i32), C.POINTER(u32)])
        sig("refh_tree_file_infos", i32, [u32, vp, vp, vp, C.POINTER(vp), C.POINTER(u64)])
        sig("refh_missing_content", i32, [vp, u32, vp, vp, vp, u32, u32, u32, C.POINTER(vp), C.POINTER(u64)])
        sig("refh_open_stored_block", i32, [vp, u64, u32, vp, vp, u32, vp, u64, C.POINTER(u64)])
        sig("refh_ingest_registry_embedding", i32, [vp, vp, vp, vp, u32, u32, vp, vp, vp, u32, u32, u32, i32, C.POINTER(u64), C.POINTER(u64),
                                                   C.POINTER(u64), C.POINTER(i32)])
        sig("refh_last_raw_bytes", u64, [])
        sig("refh_tree_create", vp, [u32, u32, vp, vp, vp])
        sig("refh_tree_destroy", None, [vp])
        sig("refh_ingest_sweep_tree", i32, [vp, vp, vp, vp, u32, u32, u32, u32, u32, vp, u32, vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)])
        self.lz4_type = int(d.refh_lz4_type())
        self.zstd_default = int(d.refh_zstd_type(1))

    def chunk_and_hash(self, data: np.ndarray, mn, av, mx, chunker_api=None, hash_api=None):
        cap = len(data) // mn + 8
        offs, lens, hashes = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint64)
        n = self.dll.refh_chunk_stream(chunker_api, hash_api, data.ctypes.data if len(data) else None, len(data), mn, av, mx,
                                       offs.ctypes.data, lens.ctypes.data, hashes.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(f"refh_chunk_stream failed: {n}")
        return offs[:n].copy(), lens[:n].copy(), hashes[:n].copy()

    def chunk_from_buffer(self, data: np.ndarray, mn, av, mx, chunker_api=None):
        cap = len(data) // mn + 8
        lens = np.zeros(cap, np.uint32)
        n = self.dll.refh_chunk_from_buffer(chunker_api, data.ctypes.data, len(data), mn, av, mx, lens.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(f"refh_chunk_from_buffer failed: {n}")
        return lens[:n].copy()

    def chunk_failing_feeder(self, data: np.ndarray, mn, av, mx, fail_at: int, fail_errno: int, chunker_api=None):
        """NextChunk with a feeder that fails after fail_at bytes -> (lens handed out before, dict of the failing call)."""
        cap = len(data) // mn + 8
        lens = np.zeros(cap, np.uint32)
        fail = np.zeros(4, np.uint64)
        n = self.dll.refh_chunk_stream_failing_feeder(chunker_api, data.ctypes.data, len(data), mn, av, mx, fail_at, fail_errno,
                                                      lens.ctypes.data, cap, fail.ctypes.data)
        if n < 0:
            raise RuntimeError(f"refh_chunk_stream_failing_feeder failed: {n}")
        return lens[:n].copy(), dict(err=int(fail[0]), len=int(fail[1]), offset=int(fail[2]), has_buf=bool(fail[3]))

    def version_index_cancel(self, files, target_chunk_size: int, workers: int, cancel_after_progress: int, chunker_api=None,
                             hash_api=None):
        """Longtail_CreateVersionIndex with a cancel token (test.cpp:4733-4837) -> (errno, index pointer stayed NULL, progress calls)."""
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        is_null, calls = i32(0), u32(0)
        err = self.dll.refh_version_index_cancel(chunker_api, hash_api, n, c_names, c_datas, c_sizes, target_chunk_size, workers,
                                                 cancel_after_progress, C.byref(is_null), C.byref(calls))
        return err, bool(is_null.value), calls.value

    def blake3(self, data: np.ndarray) -> int:
        return int(self.dll.refh_blake3(data.ctypes.data if len(data) else None, len(data)))

    def compress(self, codec: int, settings: int, data: np.ndarray) -> np.ndarray:
        cap = int(self.dll.refh_codec_bound(codec, settings, len(data)))
        out = np.zeros(cap + 8, np.uint8)
        n = sz(0)
        err = self.dll.refh_codec_compress(codec, settings, data.ctypes.data, len(data), out.ctypes.data, cap, C.byref(n))
        if err:
            raise RuntimeError(f"reference compress failed: {err}")
        return out[: n.value].copy()

    def decompress(self, codec: int, comp: np.ndarray, cap: int):
        out = np.zeros(cap + 8, np.uint8)
        n = sz(0)
        err = self.dll.refh_codec_decompress(codec, comp.ctypes.data, len(comp), out.ctypes.data, cap, C.byref(n))
        return err, out[: n.value]

    @staticmethod
    def _tree_args(files):
        names = [n.encode() for n, _ in files]
        datas = [np.ascontiguousarray(d, dtype=np.uint8) for _, d in files]
        n = len(files)
        c_names = (C.c_char_p * n)(*names)
        c_datas = (vp * n)(*[d.ctypes.data if len(d) else None for d in datas])
        c_sizes = (u64 * n)(*[len(d) for d in datas])
        return n, c_names, c_datas, c_sizes, datas

    def version_index(self, files, target_chunk_size: int, workers: int = 0, tag: int = 0, chunker_api=None, hash_api=None):
        """files: [(relative path, uint8 array)] -> (serialized VersionIndex bytes, seconds)."""
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        buf, size, secs = vp(), u64(0), C.c_double(0)
        err = self.dll.refh_version_index(chunker_api, hash_api, n, c_names, c_datas, c_sizes, target_chunk_size, workers,
                                          tag, C.byref(buf), C.byref(size), C.byref(secs))
        if err:
            raise RuntimeError(f"refh_version_index failed: {err}")
        out = bytes((C.c_ubyte * size.value).from_address(buf.value))
        self.dll.refh_free(buf)
        return out, secs.value

    def tree_file_infos(self, files):
        """The reference's Longtail_FileInfos for the tree: (paths list, sizes u64, path_start_offsets u32, permissions u16,
        path_data bytes), in the reference's asset order (directories included, with a trailing '/')."""
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        buf, size = vp(), u64(0)
        err = self.dll.refh_tree_file_infos(n, c_names, c_datas, c_sizes, C.byref(buf), C.byref(size))
        if err:
            raise RuntimeError(f"refh_tree_file_infos failed: {err}")
        raw = bytes((C.c_ubyte * size.value).from_address(buf.value))
        self.dll.refh_free(buf)
        cnt, pd = np.frombuffer(raw[:8], np.uint32)
        cnt, pd = int(cnt), int(pd)
        o = 8
        sizes = np.frombuffer(raw[o : o + cnt * 8], np.uint64).copy(); o += cnt * 8
        offs = np.frombuffer(raw[o : o + cnt * 4], np.uint32).copy(); o += cnt * 4
        perms = np.frombuffer(raw[o : o + cnt * 2], np.uint16).copy(); o += cnt * 2
        path_data = raw[o : o + pd]
        paths = [path_data[int(a) : path_data.index(b"\0", int(a))].decode() for a in offs]
        return paths, sizes, offs, perms, path_data

    def ingest_roundtrip(self, files, target_chunk_size, max_block_size, max_chunks_per_block, tag, workers=0,
                         chunker_api=None, hash_api=None, codec_api=None):
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        nchunks, nblocks, stored = u64(0), u64(0), u64(0)
        t_index, t_write = C.c_double(0), C.c_double(0)
        err = self.dll.refh_ingest_roundtrip(chunker_api, hash_api, codec_api, tag, n, c_names, c_datas, c_sizes,
                                             target_chunk_size, max_block_size, max_chunks_per_block, workers,
                                             C.byref(nchunks), C.byref(nblocks), C.byref(stored), C.byref(t_index),
                                             C.byref(t_write))
        return dict(err=err, chunks=nchunks.value, blocks=nblocks.value, stored_bytes=stored.value,
                    seconds_index=t_index.value, seconds_write=t_write.value)


    def ingest_registry_embedding(self, hip_dll, files, target_chunk_size, max_block_size, max_chunks_per_block, tag, workers=0):
        """UpSync + reference-only restore with the registries INTEGRATION.md prints: Longtail_CreateDefaultCompressionRegistry over the
        exported Longtail_CompressionRegistry_CreateForHipLZ4 / ...HipZstd factories (+ the reference's LZ4 one) and a
        Longtail_CreateDefaultHashRegistry entry holding Longtail_CreateHipBlake3HashAPI(); the registries own and dispose the objects."""
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        fn = lambda name: C.cast(getattr(hip_dll, name), vp)
        nchunks, nblocks, stored, made = u64(0), u64(0), u64(0), i32(0)
        err = self.dll.refh_ingest_registry_embedding(fn("Longtail_CompressionRegistry_CreateForHipLZ4"), fn("Longtail_CompressionRegistry_CreateForHipZstd"),
                                                      fn("Longtail_CreateHipBlake3HashAPI"), fn("Longtail_CreateHipChunkerAPI"), tag, n, c_names,
                                                      c_datas, c_sizes, target_chunk_size, max_block_size, max_chunks_per_block, workers,
                                                      C.byref(nchunks), C.byref(nblocks), C.byref(stored), C.byref(made))
        return dict(err=err, chunks=nchunks.value, blocks=nblocks.value, stored_bytes=stored.value, apis_created=made.value)

    def ingest_time(self, files, target_chunk_size, max_block_size, max_chunks_per_block, tag, workers):
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        nchunks, nblocks, stored = u64(0), u64(0), u64(0)
        t_index, t_write = C.c_double(0), C.c_double(0)
        err = self.dll.refh_ingest_time(tag, n, c_names, c_datas, c_sizes, target_chunk_size, max_block_size,
                                        max_chunks_per_block, workers, C.byref(nchunks), C.byref(nblocks), C.byref(stored),
                                        C.byref(t_index), C.byref(t_write))
        return dict(err=err, chunks=nchunks.value, blocks=nblocks.value, stored_bytes=stored.value,
                    seconds_index=t_index.value, seconds_write=t_write.value)


    def get_existing_store_index(self, store_index: bytes, chunks: np.ndarray, min_block_usage_percent: int) -> bytes:
        chunks = np.ascontiguousarray(chunks, dtype=np.uint64)
        buf, size = vp(), u64(0)
        raw = np.frombuffer(store_index, np.uint8)
        err = self.dll.refh_get_existing_store_index(raw.ctypes.data, len(raw), chunks.ctypes.data if len(chunks) else None, len(chunks),
                                                     min_block_usage_percent, C.byref(buf), C.byref(size))
        if err:
            raise RuntimeError(f"refh_get_existing_store_index failed: {err}")
        out = bytes((C.c_ubyte * size.value).from_address(buf.value))
        self.dll.refh_free(buf)
        return out

    def ingest_sweep(self, files, target_chunk_size, max_block_size, max_chunks_per_block, tag, workers, reps,
                     chunker_api=None, hash_api=None, codec_api=None):
        """One tree, then CreateVersionIndex + CreateMissingContent + WriteContent timed for every W of `workers`, `reps` times.
        -> dict(err, chunks, blocks, stored_bytes, seconds[w][r] = (index, missing, write)).  chunker_api / hash_api / codec_api: the
        embedder's plugin objects in the unmodified core instead of the reference's (the drop-in measurement)."""
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        w = np.ascontiguousarray(workers, dtype=np.int32)
        secs = np.zeros((len(w), reps, 3), np.float64)
        nchunks, nblocks, stored = u64(0), u64(0), u64(0)
        err = self.dll.refh_ingest_sweep_apis(chunker_api, hash_api, codec_api, tag, n, c_names, c_datas, c_sizes, target_chunk_size,
                                              max_block_size, max_chunks_per_block, len(w), w.ctypes.data, reps, secs.ctypes.data,
                                              C.byref(nchunks), C.byref(nblocks), C.byref(stored))
        return dict(err=err, chunks=nchunks.value, blocks=nblocks.value, stored_bytes=stored.value, seconds=secs)


    def tree_create(self, files, tag):
        """The source tree of several sweeps (the reference's file storage on tmpfs when refh_set_tree_dir was called): a handle for
        ingest_sweep_tree / tree_destroy."""
        n, c_names, c_datas, c_sizes, keep = self._tree_args(files)
        h = self.dll.refh_tree_create(tag, n, c_names, c_datas, c_sizes)
        if not h:
            raise RuntimeError("refh_tree_create failed")
        return h

    def tree_destroy(self, handle):
        self.dll.refh_tree_destroy(handle)

    def ingest_sweep_tree(self, handle, target_chunk_size, max_block_size, max_chunks_per_block, tag, workers, reps,
                          chunker_api=None, hash_api=None, codec_api=None):
        """ingest_sweep on a tree made by tree_create; additionally raw_bytes = the bytes of the chunks written (what reached the codec)."""
        w = np.ascontiguousarray(workers, dtype=np.int32)
        secs = np.zeros((len(w), reps, 3), np.float64)
        nchunks, nblocks, stored = u64(0), u64(0), u64(0)
        err = self.dll.refh_ingest_sweep_tree(handle, chunker_api, hash_api, codec_api, tag, target_chunk_size, max_block_size,
                                              max_chunks_per_block, len(w), w.ctypes.data, reps, secs.ctypes.data,
                                              C.byref(nchunks), C.byref(nblocks), C.byref(stored))
        return dict(err=err, chunks=nchunks.value, blocks=nblocks.value, stored_bytes=stored.value, seconds=secs,
                    raw_bytes=int(self.dll.refh_last_raw_bytes()))


_oracle = None
_ref = None


def oracle() -> Oracle:
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def have_ref() -> bool:
    return REF_SO.exists()


def ref() -> Ref:
    global _ref
    if _ref is None:
        _ref = Ref()
    return _ref
