"""ctypes binding of liblongtail_hip.so (include/longtail_hip.h) + thin torch-tensor conveniences.

Nothing here computes anything on the CPU: every call goes through the C ABI into the HIP kernels, and loading
fails loudly when the library (or a GPU, for the compute entry points) is missing.
"""
from __future__ import annotations

import ctypes as C
import errno
import os
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

# (LTHIP_LIB_PATH: another build of the same sources next to the in-tree one, for same-box A/B runs of compile-time constants)
_LIB_PATH = Path(os.environ["LTHIP_LIB_PATH"]) if os.environ.get("LTHIP_LIB_PATH") else Path(__file__).resolve().parent / "liblongtail_hip.so"

KERNEL_IDS = {
    "buzhash": 0,
    "select": 1,
    "compact": 2,
    "blake3_leaf": 3,
    "blake3_parent": 4,
    "lz4_segments": 5,
    "lz4_stitch": 6,
    "other": 7,
    "zstd_encode": 8,
    "gather": 9,
}


class LongtailHipError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str = ""):
        self.code = code
        super().__init__(f"{what}: errno {code} ({errno.errorcode.get(code, '?')}) {detail}")


def _u64arr(a: Sequence[int]) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _u32arr(a: Sequence[int]) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


ABI_VERSION = 3  # include/longtail_hip.h LTHIP_ABI_VERSION


class HipLib:
    """The loaded shared library with argtypes/restypes set."""

    def __init__(self, path: Optional[os.PathLike] = None):
        p = Path(path) if path else _LIB_PATH
        if not p.exists():
            raise FileNotFoundError(
                f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make` first "
                "(there is no CPU fallback)"
            )
        self.path = p
        # torch bundles its own HIP/HSA runtime; when it is going to be used in this process it must be loaded
        # first so that the library binds to the SAME runtime (two runtimes in one process -> no devices).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.dll = C.CDLL(str(p))
        d = self.dll
        vp, u64, u32, i32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_size_t
        P = C.POINTER

        def sig(name, res, args):
            f = getattr(d, name)
            f.restype = res
            f.argtypes = args
            return f

        # --- plugin constructors ---
        sig("Longtail_CreateHipChunkerAPI", vp, [])
        sig("Longtail_CreateHipBlake3HashAPI", vp, [])
        sig("Longtail_CreateHipLZ4CompressionAPI", vp, [])
        sig("Longtail_CompressionRegistry_CreateForHipLZ4", vp, [u32, P(u32)])
        sig("Longtail_GetHipLZ4DefaultQuality", u32, [])
        sig("Longtail_CreateHipZStdCompressionAPI", vp, [])
        sig("Longtail_CompressionRegistry_CreateForHipZstd", vp, [u32, P(u32)])
        sig("Longtail_Hip_SetAllocator", None, [vp, vp])
        sig("Longtail_Hip_SetDevice", i32, [i32])
        sig("Longtail_Hip_GetLastError", i32, [])
        sig("Longtail_Hip_PinnedBytes", u64, [])
        sig("Longtail_Hip_BatchStats", None, [vp, vp])
        sig("Longtail_Hip_MemoStats", None, [vp, vp])
        # --- bulk API ---
        sig("lthip_ctx_create", i32, [i32, vp, P(vp)])
        sig("lthip_ctx_destroy", None, [vp])
        sig("lthip_ctx_sync", i32, [vp])
        sig("lthip_ctx_error", C.c_char_p, [vp])
        sig("lthip_device_count", i32, [])
        sig("lthip_build_id", C.c_char_p, [])
        sig("lthip_abi_version", i32, [])
        if d.lthip_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{p}: binary interface version {d.lthip_abi_version()}, this binding was written for {ABI_VERSION} "
                               "(include/longtail_hip.h LTHIP_ABI_VERSION): rebuild the library")
        sig("lthip_malloc_device", i32, [vp, sz, P(vp)])
        sig("lthip_free_device", None, [vp, vp])
        sig("lthip_malloc_pinned", i32, [vp, sz, P(vp)])
        sig("lthip_free_pinned", None, [vp, vp])
        sig("lthip_copy_h2d", i32, [vp, vp, vp, sz])
        sig("lthip_copy_d2h", i32, [vp, vp, vp, sz])
        sig("lthip_link_copy", i32, [vp, vp, vp, sz])
        sig("lthip_timing_enable", i32, [vp, i32])
        sig("lthip_timing_reset", i32, [vp])
        sig("lthip_timing_get", i32, [vp, i32, P(C.c_double), P(u64)])
        sig("lthip_plan_create", i32, [vp, u32, vp, vp, u32, u32, u32, P(vp)])
        sig("lthip_plan_destroy", None, [vp, vp])
        sig("lthip_plan_resize_single", i32, [vp, vp, u64])
        sig("lthip_plan_chunk_capacity", u64, [vp])
        sig("lthip_plan_slices", u32, [vp])
        sig("lthip_chunk_hash", i32, [vp, vp, vp, vp, vp, vp, vp, P(u64)])
        sig("lthip_chunk_from_buffer", i32, [vp, vp, u64, u32, u32, u32, P(u64)])
        sig("lthip_hash_ranges", i32, [vp, vp, u64, vp, vp, u32, vp])
        sig("lthip_lz4_bound", sz, [sz])
        sig("lthip_lz4_compress_blocks", i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp, i32])
        sig("lthip_lz4_decompress_blocks", i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp])
        sig("lthip_zstd_bound", sz, [sz])
        sig("lthip_zstd_compress_blocks", i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp])
        sig("lthip_zstd_compress_blocks_q", i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp, i32])
        sig("lthip_zstd_quality_of_settings", i32, [u32])
        sig("lthip_zstd_decompress_blocks", i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp])
        sig("lthip_zstd_debug_units", i32, [vp, u64, u64, vp, vp, vp])
        sig("lthip_zstd_last_decode_stats", i32, [vp, vp])
        sig("lthip_debug_reload_env", None, [])
        sig("lthip_debug_fail_alloc", i32, [C.c_int64, C.c_int64])
        sig("lthip_debug_alloc_calls", C.c_int64, [P(C.c_int64)])
        sig("lthip_stored_block_header_size", sz, [u32])
        sig("lthip_write_stored_block_headers", i32, [vp, u32, vp, vp, vp, u32, u32, vp, vp, vp, vp])
        sig("lthip_create_missing_content", i32, [vp, u64, vp, u64, vp, vp, vp, u32, u32, u32, vp, sz, vp])
        sig("lthip_get_existing_store_index", i32, [vp, vp, sz, u64, vp, u32, vp, sz, vp])
        sig("lthip_version_index_size", sz, [u32, u64, u64, u32])
        sig("lthip_build_version_index", i32, [vp, u32, vp, vp, vp, vp, u32, vp, u64, vp, vp, vp, u32, u32, vp, sz, vp])
        sig("lthip_dedup_first_seen", i32, [vp, u64, vp, vp, vp])
        sig("lthip_dedup_min_ordinal", i32, [vp, u64, vp, vp, vp, vp])
        sig("lthip_ingest_set_first_seen", i32, [vp, vp, u64])
        sig("lthip_plan_reaim", i32, [vp, vp, u32, vp, vp])
        sig("lthip_hash_one", i32, [vp, vp, u32, vp])
        sig("lthip_hash_runs_u64", i32, [vp, vp, vp, u32, vp])
        sig("lthip_hash_runs_u64_bounded", i32, [vp, vp, vp, u32, u64, u64, vp])
        sig("lthip_b3_stream_batch", i32, [vp, vp, u64, vp])
        sig("lthip_b3_stream_final", i32, [vp, vp, u32, u64, vp, vp])
        sig("lthip_dedup_first_seen_range", i32, [vp, u64, vp, u64, u64, vp, vp])
        sig("lthip_gather_ranges", i32, [vp, vp, u64, vp, vp, vp, vp])
        sig("lthip_pack_blocks", i32, [u64, vp, u32, u32, vp, u64, P(u64)])
        sig("lthip_pack_blocks_batch", i32, [u64, vp, u64, u32, u32, u64, u64, u32, u32, vp, vp, u64, P(u64), P(u64)])
        sig("lthip_synth_fill", i32, [vp, vp, u32, vp, vp, vp, i32])
        sig("lthip_synth_fill_ranges", i32, [vp, vp, u32, vp, vp, vp, vp, i32])
        sig("lthip_ingest_create", i32, [vp, vp, P(vp)])
        sig("lthip_ingest_destroy", None, [vp])
        sig("lthip_ingest_index", i32, [vp, vp, vp, vp, u64, vp, vp, u64, vp, sz])
        sig("lthip_ingest_write", i32, [vp, vp, vp, u64])
        sig("lthip_ingest_finish", i32, [vp, vp, sz, vp])
        sig("lthip_ingest_compressed_sizes", vp, [vp])
        sig("lthip_ingest_images", i32, [vp, P(u64), P(u64), P(vp), P(vp)])
        sig("lthip_divtest_eval", i32, [u32, u32])
        sig("lthip_job_count", u64, [u32, vp, u32])
        sig("lthip_make_jobs", i32, [u32, vp, u32, u64, vp, vp, vp])
        sig("lthip_partition_jobs", i32, [u64, vp, u32, i32, vp, vp])
        sig("lthip_exchange_layout", i32, [u64, vp, u32, vp, u64, u64, vp, vp, vp])
        sig("lthip_exchange_ranges", u64, [u64, vp, vp, vp, u64, u64, vp, vp, vp])
        sig("lthip_exchange_reorder", i32, [vp, vp, vp, u32, u64, vp, vp, vp])
        sig("lthip_job_ordinals", i32, [vp, u64, vp, vp, u64, vp])
        sig("lthip_comm_unique_id", i32, [vp])
        sig("lthip_comm_create", i32, [vp, i32, i32, vp, P(vp)])
        sig("lthip_comm_destroy", i32, [vp])
        sig("lthip_comm_allgather", i32, [vp, vp, vp, vp, u64, u32])
        sig("lthip_comm_alltoallv", i32, [vp, vp, vp, vp, vp, vp, vp, vp, u32])
        sig("lthip_comm_info", i32, [vp, P(i32), P(i32), P(i32)])

    def device_count(self) -> int:
        return int(self.dll.lthip_device_count())

    def build_id(self) -> str:
        return self.dll.lthip_build_id().decode()


_lib: Optional[HipLib] = None


ABLATIONS_LIB_PATH = Path(__file__).resolve().parent.parent / "build" / "ablations" / "liblongtail_hip.so"
_abl = None


def load_ablations() -> HipLib:
    """The ABLATION build of the same sources (`make ablations`: -DLTHIP_ABLATIONS): earlier formulations of the kernels and debug
    paths behind LTHIP_* switches, kept as second implementations for the differential tests and the A/B tools.  The product
    library has none of them.  A separate handle: load() keeps returning the product library."""
    global _abl
    if _abl is None:
        if os.environ.get("LTHIP_LIB_PATH"):
            _abl = load()  # (the whole process runs on the library named there)
        else:
            if not ABLATIONS_LIB_PATH.exists():
                raise FileNotFoundError(f"{ABLATIONS_LIB_PATH} is missing: run `make ablations` (or __graft_entry__.build())")
            _abl = HipLib(ABLATIONS_LIB_PATH)
    return _abl


def load(path: Optional[os.PathLike] = None) -> HipLib:
    global _lib
    if _lib is None or path is not None:
        _lib = HipLib(path)
    return _lib


def _ptr(t) -> int:
    """device/host pointer of a torch tensor or numpy array (0 for None)."""
    if t is None:
        return 0
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return int(t.data_ptr())


class Context:
    """One lthip_ctx bound to a torch device and (by default) torch's current stream on it."""

    def __init__(self, device: int = 0, stream: "object | None" = "torch", lib: Optional[HipLib] = None):
        import torch

        self.lib = lib or load()
        self.torch = torch
        self.device = device
        if stream == "torch":
            raw_stream = int(torch.cuda.current_stream(device).cuda_stream)  # 0 = the null stream, also valid
        elif stream is None:
            raw_stream = -1  # LTHIP_STREAM_PRIVATE
        else:
            raw_stream = int(stream)
        h = C.c_void_p()
        err = self.lib.dll.lthip_ctx_create(device, C.c_void_p(raw_stream), C.byref(h))
        if err:
            raise LongtailHipError(err, "lthip_ctx_create", "(no GPU?)" if err == errno.ENODEV else "")
        self.h = h

    # -- plumbing --
    def close(self):
        if getattr(self, "h", None):
            self.lib.dll.lthip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, err: int, what: str):
        if err:
            msg = self.lib.dll.lthip_ctx_error(self.h)
            raise LongtailHipError(err, what, msg.decode() if msg else "")

    def sync(self):
        self._check(self.lib.dll.lthip_ctx_sync(self.h), "lthip_ctx_sync")

    def timing(self, on: bool):
        self._check(self.lib.dll.lthip_timing_enable(self.h, 1 if on else 0), "lthip_timing_enable")

    def timing_reset(self):
        self._check(self.lib.dll.lthip_timing_reset(self.h), "lthip_timing_reset")

    def timing_get(self) -> dict:
        out = {}
        for name, kid in KERNEL_IDS.items():
            ms, n = C.c_double(0), C.c_uint64(0)
            self._check(self.lib.dll.lthip_timing_get(self.h, kid, C.byref(ms), C.byref(n)), "lthip_timing_get")
            out[name] = (ms.value, n.value)
        return out

    def _dev(self):
        return self.torch.device("cuda", self.device)

    # -- synthetic data --
    def synth_fill(self, dst, offsets, sizes, seeds, kind: int, skips=None):
        """Ranges [skips[i], skips[i] + sizes[i]) of the assets with the given seeds -> dst + offsets[i] (include/longtail_synth.h)."""
        o, s, sd = _u64arr(offsets), _u64arr(sizes), _u64arr(seeds)
        sk = _u64arr(skips) if skips is not None else None
        self._check(
            self.lib.dll.lthip_synth_fill_ranges(self.h, _ptr(dst), len(o), o.ctypes.data, s.ctypes.data, sd.ctypes.data,
                                                 sk.ctypes.data if sk is not None else None, kind),
            "lthip_synth_fill_ranges",
        )

    # -- phase 1 --
    def make_plan(self, part_offsets, part_sizes, min_chunk: int, avg_chunk: int, max_chunk: int) -> "Plan":
        return Plan(self, part_offsets, part_sizes, min_chunk, avg_chunk, max_chunk)

    def chunk_hash(self, plan: "Plan", data, want_hashes: bool = True, outputs=None, sync: bool = True):
        """-> (total, offsets u64[cap], lens u32[cap], hashes u64[cap] | None, part_first u32[nparts+1]) torch tensors."""
        torch = self.torch
        cap = max(1, plan.capacity)
        if outputs is None:
            dev = self._dev()
            offs = torch.empty(cap, dtype=torch.int64, device=dev)
            lens = torch.empty(cap, dtype=torch.int32, device=dev)
            hashes = torch.empty(cap, dtype=torch.int64, device=dev) if want_hashes else None
            first = torch.empty(plan.nparts + 1, dtype=torch.int32, device=dev)
        else:
            offs, lens, hashes, first = outputs
        total = C.c_uint64(0)
        err = self.lib.dll.lthip_chunk_hash(
            self.h, plan.h, _ptr(data), _ptr(offs), _ptr(lens), _ptr(hashes), _ptr(first), C.byref(total) if sync else None
        )
        self._check(err, "lthip_chunk_hash")
        return (int(total.value) if sync else None), offs, lens, hashes, first

    def hash_ranges(self, data, offsets, lens, max_len: int = 0, out=None):
        torch = self.torch
        n = int(offsets.numel())
        if out is None:
            out = torch.empty(max(1, n), dtype=torch.int64, device=self._dev())
        self._check(
            self.lib.dll.lthip_hash_ranges(self.h, _ptr(data), n, _ptr(offsets), _ptr(lens), max_len, _ptr(out)),
            "lthip_hash_ranges",
        )
        return out[:n]

    def chunk_from_buffer(self, data, size: int, min_chunk: int, avg_chunk: int, max_chunk: int) -> int:
        out = C.c_uint64(0)
        self._check(
            self.lib.dll.lthip_chunk_from_buffer(self.h, _ptr(data), size, min_chunk, avg_chunk, max_chunk, C.byref(out)),
            "lthip_chunk_from_buffer",
        )
        return int(out.value)

    # -- phase 2 --
    def _codec(self, fn, src, src_offsets, src_sizes, dst, dst_offsets, dst_caps, extra=()):
        torch = self.torch
        so, ss = _u64arr(src_offsets), _u32arr(src_sizes)
        do, dc = _u64arr(dst_offsets), _u32arr(dst_caps)
        n = len(so)
        out_sizes = torch.empty(max(1, n), dtype=torch.int32, device=self._dev())
        err = fn(self.h, _ptr(src), n, so.ctypes.data, ss.ctypes.data, _ptr(dst), do.ctypes.data, dc.ctypes.data,
                 _ptr(out_sizes), *extra)
        self._check(err, fn.__name__)
        return out_sizes[:n]

    def lz4_compress_blocks(self, src, src_offsets, src_sizes, dst, dst_offsets, dst_caps, segment_log2: int = 0):
        return self._codec(self.lib.dll.lthip_lz4_compress_blocks, src, src_offsets, src_sizes, dst, dst_offsets,
                           dst_caps, (segment_log2,))

    def lz4_decompress_blocks(self, src, src_offsets, src_sizes, dst, dst_offsets, dst_caps):
        return self._codec(self.lib.dll.lthip_lz4_decompress_blocks, src, src_offsets, src_sizes, dst, dst_offsets,
                           dst_caps)

    def zstd_compress_blocks(self, src, src_offsets, src_sizes, dst, dst_offsets, dst_caps, quality: int = 0):
        """quality: 0 default ('ztd1', 'ztd2'), 1 high ('ztd4'), 2 max ('ztd3', 'ztd5') -- LTHIP_ZSTD_Q_*"""
        return self._codec(self.lib.dll.lthip_zstd_compress_blocks_q, src, src_offsets, src_sizes, dst, dst_offsets,
                           dst_caps, (int(quality),))

    def zstd_decompress_blocks(self, src, src_offsets, src_sizes, dst, dst_offsets, dst_caps):
        return self._codec(self.lib.dll.lthip_zstd_decompress_blocks, src, src_offsets, src_sizes, dst, dst_offsets,
                           dst_caps)

    def zstd_last_decode_stats(self):
        """(payloads, blocks of other encoders' frames listed for the block-parallel decoder, payloads given back to the serial
        decoder) of the last zstd_decompress_blocks call."""
        out = np.zeros(4, np.uint32)
        self._check(self.lib.dll.lthip_zstd_last_decode_stats(self.h, out.ctypes.data), "lthip_zstd_last_decode_stats")
        return tuple(int(v) for v in out)

    def zstd_debug_units(self, first: int, count: int):
        """Match-finder output of the last zstd_compress_blocks call: (meta[count,4] u32, lits[count,4096] u8,
        recs[count,1024] u64) for the 4 KiB units [first, first+count)."""
        meta = np.zeros((count, 4), np.uint32)
        lits = np.zeros((count, 4096), np.uint8)
        recs = np.zeros((count, 1024), np.uint64)
        self._check(self.lib.dll.lthip_zstd_debug_units(self.h, first, count, meta.ctypes.data, lits.ctypes.data,
                                                        recs.ctypes.data), "lthip_zstd_debug_units")
        return meta, lits, recs

    def build_version_index(self, asset_sizes, path_start_offsets, permissions, path_data: bytes, asset_chunk_counts,
                            chunk_hashes, chunk_lens, chunk_total: int, target_chunk_size: int, asset_tags=None,
                            hash_identifier: int = 0x626C6B33) -> bytes:
        """Serialized VersionIndex (== Longtail_WriteVersionIndexToBuffer) from device chunk lists; see longtail_hip.h."""
        n = len(asset_sizes)
        a_sz, a_off = _u64arr(asset_sizes), _u32arr(path_start_offsets)
        a_perm = np.ascontiguousarray(np.asarray(permissions, dtype=np.uint16))
        a_cnt = _u32arr(asset_chunk_counts)
        a_tag = _u32arr(asset_tags) if asset_tags is not None else None
        cap = self.lib.dll.lthip_version_index_size(n, chunk_total, chunk_total, len(path_data))
        out = np.zeros(cap + 16, np.uint8)
        size = C.c_size_t(0)
        err = self.lib.dll.lthip_build_version_index(
            self.h, n, a_sz.ctypes.data, a_off.ctypes.data, a_perm.ctypes.data, path_data, len(path_data), a_cnt.ctypes.data,
            chunk_total, _ptr(chunk_hashes), _ptr(chunk_lens), a_tag.ctypes.data if a_tag is not None else None, hash_identifier,
            target_chunk_size, out.ctypes.data, cap, C.byref(size))
        self._check(err, "lthip_build_version_index")
        return out[: size.value].tobytes()

    def write_stored_block_headers(self, block_first_chunk, chunk_hashes, chunk_lens, tag: int, raw_sizes, comp_sizes, arena,
                                   image_offsets, hash_identifier: int = 0x626C6B33):
        """BlockIndex + [raw][compressed] size words around payloads already compressed into `arena` (see longtail_hip.h)."""
        bf, rs, io = _u64arr(block_first_chunk), _u32arr(raw_sizes), _u64arr(image_offsets)
        self._check(self.lib.dll.lthip_write_stored_block_headers(self.h, len(io), bf.ctypes.data, _ptr(chunk_hashes), _ptr(chunk_lens),
                                                                  hash_identifier, tag, rs.ctypes.data, _ptr(comp_sizes), _ptr(arena),
                                                                  io.ctypes.data), "lthip_write_stored_block_headers")

    def create_missing_content(self, existing_hashes, chunk_hashes, chunk_lens, chunk_tags, max_block_size: int,
                               max_chunks_per_block: int, hash_identifier: int = 0x626C6B33) -> bytes:
        """Serialized StoreIndex of the version chunks a store with `existing_hashes` lacks (see longtail_hip.h)."""
        ne = int(existing_hashes.numel()) if existing_hashes is not None else 0
        n = int(chunk_hashes.numel())
        tags = _u32arr(chunk_tags) if chunk_tags is not None else None
        cap = 16 + 20 * n + 12 * n + 64
        out = np.zeros(cap, np.uint8)
        size = C.c_size_t(0)
        err = self.lib.dll.lthip_create_missing_content(
            self.h, ne, _ptr(existing_hashes) if ne else None, n, _ptr(chunk_hashes), _ptr(chunk_lens),
            tags.ctypes.data if tags is not None else None, hash_identifier, max_block_size, max_chunks_per_block, out.ctypes.data,
            cap, C.byref(size))
        self._check(err, "lthip_create_missing_content")
        return out[: size.value].tobytes()

    def get_existing_store_index(self, store_index: bytes, chunk_hashes, min_block_usage_percent: int) -> bytes:
        """Serialized StoreIndex of the store's blocks that cover `chunk_hashes` (device int64/uint64 tensor); see longtail_hip.h."""
        raw = np.frombuffer(store_index, np.uint8)
        n = int(chunk_hashes.numel()) if chunk_hashes is not None else 0
        out = np.zeros(len(raw) + 64, np.uint8)
        size = C.c_size_t(0)
        err = self.lib.dll.lthip_get_existing_store_index(self.h, raw.ctypes.data, len(raw), n, _ptr(chunk_hashes) if n else None,
                                                          min_block_usage_percent, out.ctypes.data, len(out), C.byref(size))
        self._check(err, "lthip_get_existing_store_index")
        return out[: size.value].tobytes()

    # -- block assembly --
    def link_copy(self, dst, src, nbytes: int):
        """dst[:nbytes] = src[:nbytes] by the compute units (pinned host <-> device, either direction; include/longtail_hip.h)."""
        self._check(self.lib.dll.lthip_link_copy(self.h, _ptr(dst), _ptr(src), nbytes), "lthip_link_copy")

    def gather_ranges(self, src, src_offsets, lens, dst, dst_offsets):
        n = int(src_offsets.numel())
        self._check(self.lib.dll.lthip_gather_ranges(self.h, _ptr(src), n, _ptr(src_offsets), _ptr(lens), _ptr(dst),
                                                     _ptr(dst_offsets)), "lthip_gather_ranges")

    def exchange_reorder(self, gathered, out, ranges):
        """out[dst ..) = gathered[src .. + cnt) for the (src, dst, cnt) element ranges of JobPartition.ranges (host arrays)."""
        r_src, r_dst, r_cnt = ranges
        assert gathered.dtype == out.dtype and gathered.is_contiguous() and out.is_contiguous()
        self._check(self.lib.dll.lthip_exchange_reorder(self.h, _ptr(gathered), _ptr(out), gathered.element_size(), len(r_src),
                                                        r_src.ctypes.data, r_dst.ctypes.data, r_cnt.ctypes.data), "lthip_exchange_reorder")

    def job_ordinals(self, local_first: np.ndarray, global_first: np.ndarray, local_chunks: int):
        """int32 tensor [local_chunks]: position in job order of every chunk of this rank (own job m: local_first[m] -> global_first[m])."""
        lf, gf = _u32arr(local_first), _u32arr(global_first)
        assert len(lf) == len(gf)
        out = self.torch.empty(max(1, local_chunks), dtype=self.torch.int32, device=self._dev())
        self._check(self.lib.dll.lthip_job_ordinals(self.h, len(lf), lf.ctypes.data, gf.ctypes.data, local_chunks, _ptr(out)),
                    "lthip_job_ordinals")
        return out[:local_chunks]

    # -- dedup --
    def dedup_first_seen_range(self, hashes, first: int, count: int):
        """All hashes inserted, first-occurrence (global) indices returned for [first, first+count) only; + distinct count."""
        torch = self.torch
        n = int(hashes.numel())
        out = torch.empty(max(1, count), dtype=torch.int32, device=self._dev())
        uniq = torch.zeros(1, dtype=torch.int64, device=self._dev())
        self._check(self.lib.dll.lthip_dedup_first_seen_range(self.h, n, _ptr(hashes), first, count, _ptr(out), _ptr(uniq)),
                    "lthip_dedup_first_seen_range")
        return out[:count], uniq

    def dedup_first_seen(self, hashes):
        torch = self.torch
        n = int(hashes.numel())
        first = torch.empty(max(1, n), dtype=torch.int32, device=self._dev())
        uniq = torch.zeros(1, dtype=torch.int64, device=self._dev())
        self._check(self.lib.dll.lthip_dedup_first_seen(self.h, n, _ptr(hashes), _ptr(first), _ptr(uniq)),
                    "lthip_dedup_first_seen")
        return first[:n], uniq

    def dedup_min_ordinal(self, hashes, ordinals, sync: bool = True):
        """The owner's side of the sharded first-seen table: first[j] = smallest ordinal among the items with the hash of item j
        (int32 tensor), number of distinct hashes (int, after a synchronisation; sync=False: a one-element int64 device tensor)."""
        torch = self.torch
        n = int(hashes.numel())
        first = torch.empty(max(1, n), dtype=torch.int32, device=self._dev())
        uniq = torch.zeros(1, dtype=torch.int64, device=self._dev())
        self._check(self.lib.dll.lthip_dedup_min_ordinal(self.h, n, _ptr(hashes.contiguous()), _ptr(ordinals.contiguous()), _ptr(first), _ptr(uniq)),
                    "lthip_dedup_min_ordinal")
        if not sync:
            return first[:n], uniq
        self.sync()
        return first[:n], int(uniq.item())


class Comm:
    """Communicator behind the C ABI (comm.hip), one process per GPU: RCCL, or -- when the id was made under
    LTHIP_COMM_TRANSPORT=shm -- the shared-memory stand-in for boxes without N GPUs.  `unique_id()` on rank 0, carried to the other
    ranks by the embedder (bench.py: a file, or the torch.distributed store), then `Comm(ctx, nranks, rank, id)` everywhere.
    `ctx=None` (shared-memory transport only): the tensors are CPU tensors (the tests without a GPU)."""

    ID_BYTES = 128
    TRANSPORTS = {1: "rccl", 2: "host-shm"}

    @staticmethod
    def unique_id(lib: Optional[HipLib] = None) -> bytes:
        lib = lib or load()
        buf = (C.c_ubyte * Comm.ID_BYTES)()
        err = lib.dll.lthip_comm_unique_id(C.addressof(buf))
        if err:
            raise LongtailHipError(err, "lthip_comm_unique_id")
        return bytes(buf)

    def __init__(self, ctx: "Optional[Context]", nranks: int, rank: int, unique_id: bytes, lib: Optional[HipLib] = None):
        assert len(unique_id) == Comm.ID_BYTES
        self.ctx, self.nranks, self.rank = ctx, nranks, rank
        self.lib = ctx.lib if ctx is not None else (lib or load())
        buf = (C.c_ubyte * Comm.ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        self._check(self.lib.dll.lthip_comm_create(self._ctx_h(), nranks, rank, C.addressof(buf), C.byref(h)), "lthip_comm_create")
        self.h = h

    def _ctx_h(self):
        return self.ctx.h if self.ctx is not None else None

    def _check(self, err, what):
        if self.ctx is not None:
            self.ctx._check(err, what)
        elif err:
            raise LongtailHipError(err, what)

    def sync(self):
        if self.ctx is not None:
            self.ctx.sync()

    def info(self) -> dict:
        """Size of the communicator as the TRANSPORT reports it (ncclCommCount), this rank, the transport's name."""
        n, r, t = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.dll.lthip_comm_info(self.h, C.byref(n), C.byref(r), C.byref(t)), "lthip_comm_info")
        return {"nranks": int(n.value), "rank": int(r.value), "transport": Comm.TRANSPORTS.get(int(t.value), str(t.value))}

    def allgather(self, send, recv=None):
        """`send`: contiguous tensor; returns `recv` = the ranks' tensors back to back (nranks * send.numel() elements)."""
        import torch

        send = send.contiguous()
        if recv is None:
            recv = torch.empty(send.numel() * self.nranks, dtype=send.dtype, device=send.device)
        self._check(self.lib.dll.lthip_comm_allgather(self._ctx_h(), self.h, _ptr(send), _ptr(recv), send.numel(), send.element_size()),
                    "lthip_comm_allgather")
        return recv

    def alltoallv(self, send, send_counts, recv_counts, recv=None):
        """`send`: this rank's elements grouped by destination rank (send_counts[p] of them for rank p, back to back); returns the
        elements received, grouped by source rank (recv_counts[p] from rank p).  Counts are host integers."""
        import torch

        send = send.contiguous()
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.uint64)
        assert len(sc) == self.nranks and len(rc) == self.nranks and int(sc.sum()) == send.numel()
        sd = np.zeros(self.nranks, np.uint64)
        rd = np.zeros(self.nranks, np.uint64)
        np.cumsum(sc[:-1], out=sd[1:])
        np.cumsum(rc[:-1], out=rd[1:])
        if recv is None:
            recv = torch.empty(int(rc.sum()), dtype=send.dtype, device=send.device)
        self._check(self.lib.dll.lthip_comm_alltoallv(self._ctx_h(), self.h, _ptr(send) if send.numel() else None, sc.ctypes.data, sd.ctypes.data,
                                                      _ptr(recv) if recv.numel() else None, rc.ctypes.data, rd.ctypes.data, send.element_size()),
                    "lthip_comm_alltoallv")
        return recv

    def close(self):
        if self.h:
            self.lib.dll.lthip_comm_destroy(self.h)
            self.h = None


class IngestConfig(C.Structure):
    _fields_ = [("target_chunk_size", C.c_uint32), ("hash_identifier", C.c_uint32), ("max_block_size", C.c_uint32),
                ("max_chunks_per_block", C.c_uint32), ("compression_type", C.c_uint32), ("codec", C.c_uint32),
                ("batch_bytes", C.c_uint64)]


class IngestTree(C.Structure):
    _fields_ = [("asset_count", C.c_uint32), ("asset_sizes", C.c_void_p), ("path_start_offsets", C.c_void_p),
                ("permissions", C.c_void_p), ("path_data", C.c_char_p), ("path_data_size", C.c_uint32), ("asset_tags", C.c_void_p),
                ("job_count", C.c_uint64), ("job_asset", C.c_void_p), ("job_first", C.c_void_p), ("my_job_count", C.c_uint64),
                ("my_jobs", C.c_void_p)]


class IngestResult(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("struct_size", "chunks_all", "unique_all", "chunks_local", "unique_local", "blocks", "raw_bytes",
                                          "compressed_bytes", "gathered_blocks", "version_index_size", "store_index_size", "gathered_bytes")]


CODECS = {"none": 0, "lz4": 1, "zstd": 2}
LZ4_TYPE = 0x6C7A3432    # 'lz42', lib/lz4/longtail_lz4.c:10
ZSTD_DEFAULT = 0x7A746432  # 'ztd2', lib/zstd/longtail_zstd.c:12-22 (ZSTD default quality)


class Ingest:
    """lthip_ingest: CreateVersionIndex tail + CreateMissingContent + WriteContent over device-resident chunk lists."""

    def __init__(self, ctx: "Context", target_chunk_size: int, max_block_size: int, max_chunks_per_block: int, codec: str,
                 compression_type: Optional[int] = None, batch_bytes: int = 0, hash_identifier: int = 0x626C6B33):
        self.ctx = ctx
        if compression_type is None:
            compression_type = LZ4_TYPE if codec == "lz4" else ZSTD_DEFAULT
        self.cfg = IngestConfig(target_chunk_size, hash_identifier, max_block_size, max_chunks_per_block, compression_type,
                                CODECS[codec], batch_bytes)
        h = C.c_void_p()
        ctx._check(ctx.lib.dll.lthip_ingest_create(ctx.h, C.byref(self.cfg), C.byref(h)), "lthip_ingest_create")
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.dll.lthip_ingest_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def tree(asset_sizes, path_start_offsets, permissions, path_data: bytes, job_asset, job_first, my_jobs=None, asset_tags=None):
        """-> (IngestTree, keep-alive list)."""
        a_sz, a_off = _u64arr(asset_sizes), _u32arr(path_start_offsets)
        a_perm = np.ascontiguousarray(np.asarray(permissions, dtype=np.uint16))
        j_as, j_first = _u32arr(job_asset), _u64arr(job_first)
        mine = _u64arr(my_jobs) if my_jobs is not None else None
        tags = _u32arr(asset_tags) if asset_tags is not None else None
        t = IngestTree(len(a_sz), a_sz.ctypes.data, a_off.ctypes.data, a_perm.ctypes.data, path_data, len(path_data),
                       tags.ctypes.data if tags is not None else None, len(j_as), j_as.ctypes.data, j_first.ctypes.data,
                       len(mine) if mine is not None else 0, mine.ctypes.data if mine is not None else None)
        keep = [a_sz, a_off, a_perm, path_data, j_as, j_first, mine, tags]
        t._keep = keep  # the arrays are read during index() only (the session copies what its helper thread needs later)
        return t, keep

    def index(self, tree: IngestTree, all_hashes, all_lens, all_chunks: int, local_offsets, local_part_first, local_chunks: int,
              version_index_out=None):
        """version_index_out: a (pinned) uint8 torch tensor or numpy array, or None."""
        cap = 0 if version_index_out is None else (version_index_out.numel() if hasattr(version_index_out, "numel") else len(version_index_out))
        err = self.ctx.lib.dll.lthip_ingest_index(self.h, C.byref(tree), _ptr(all_hashes), _ptr(all_lens), all_chunks, _ptr(local_offsets),
                                                  _ptr(local_part_first), local_chunks, _ptr(version_index_out) or None, cap)
        self._index_keep = (tree, all_hashes, all_lens, local_offsets, local_part_first, version_index_out)  # until finish()
        self.ctx._check(err, "lthip_ingest_index")

    def set_first_seen(self, first_index, unique_chunks: int):
        """first-seen index of every chunk from the sharded table (longtail_amd.dist.sharded_first_seen): the next index() uses it."""
        self._first_keep = first_index  # must stay alive until index() has returned
        self.ctx._check(self.ctx.lib.dll.lthip_ingest_set_first_seen(self.h, _ptr(first_index), C.c_uint64(int(unique_chunks))),
                        "lthip_ingest_set_first_seen")

    def write(self, data, arena):
        self.ctx._check(self.ctx.lib.dll.lthip_ingest_write(self.h, _ptr(data), _ptr(arena), int(arena.numel())), "lthip_ingest_write")

    def finish(self, store_index_out=None) -> IngestResult:
        cap = 0 if store_index_out is None else (store_index_out.numel() if hasattr(store_index_out, "numel") else len(store_index_out))
        res = IngestResult()
        res.struct_size = C.sizeof(IngestResult)
        err = self.ctx.lib.dll.lthip_ingest_finish(self.h, _ptr(store_index_out) or None, cap, C.byref(res))
        self.ctx._check(err, "lthip_ingest_finish")
        self._index_keep = None
        return res

    def images(self):
        """(first block, offsets into the arena [u64], image sizes [u32]) of the last codec batch: the stored-block images a host-fed
        embedder downloads (lthip_ingest_images; valid after finish())."""
        first, count, po, ps = C.c_uint64(), C.c_uint64(), C.c_void_p(), C.c_void_p()
        self.ctx._check(self.ctx.lib.dll.lthip_ingest_images(self.h, C.byref(first), C.byref(count), C.byref(po), C.byref(ps)), "lthip_ingest_images")
        n = int(count.value)
        if n == 0:
            return int(first.value), np.zeros(0, np.uint64), np.zeros(0, np.uint32)
        return (int(first.value), np.ctypeslib.as_array((C.c_uint64 * n).from_address(po.value)).copy(),
                np.ctypeslib.as_array((C.c_uint32 * n).from_address(ps.value)).copy())

    def compressed_sizes(self, nblocks: int) -> np.ndarray:
        p = self.ctx.lib.dll.lthip_ingest_compressed_sizes(self.h)
        if not p or nblocks == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array((C.c_uint32 * nblocks).from_address(p)).copy()


class Plan:
    def __init__(self, ctx: Context, part_offsets, part_sizes, min_chunk: int, avg_chunk: int, max_chunk: int):
        self.ctx = ctx
        o, s = _u64arr(part_offsets), _u64arr(part_sizes)
        assert len(o) == len(s)
        self.nparts = len(o)
        h = C.c_void_p()
        err = ctx.lib.dll.lthip_plan_create(ctx.h, self.nparts, o.ctypes.data, s.ctypes.data, min_chunk, avg_chunk,
                                            max_chunk, C.byref(h))
        ctx._check(err, "lthip_plan_create")
        self.h = h
        self.capacity = int(ctx.lib.dll.lthip_plan_chunk_capacity(h))
        self.total_bytes = int(s.sum()) if len(s) else 0

    @property
    def slices(self) -> int:
        """2 when chunk_hash runs this plan as two slices on two streams (per-kernel timings of scan and leaf hashing then overlap)."""
        return int(self.ctx.lib.dll.lthip_plan_slices(self.h))

    def reaim(self, part_offsets, part_sizes):
        """The plan aimed at another set of parts (lthip_plan_reaim): no more parts / 16 KiB tiles than it was created with; the part
        tables are recomputed and rewritten on the context's stream, nothing is allocated or waited for."""
        o, s = _u64arr(part_offsets), _u64arr(part_sizes)
        assert len(o) == len(s)
        self.ctx._check(self.ctx.lib.dll.lthip_plan_reaim(self.ctx.h, self.h, len(o), o.ctypes.data, s.ctypes.data), "lthip_plan_reaim")
        self.nparts = len(o)
        self.capacity = int(self.ctx.lib.dll.lthip_plan_chunk_capacity(self.h))
        self.total_bytes = int(s.sum()) if len(s) else 0

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.dll.lthip_plan_destroy(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_blocks(lens: np.ndarray, max_block_size: int, max_chunks_per_block: int, lib: Optional[HipLib] = None) -> np.ndarray:
    """block start indices (+ end) for chunk lengths `lens` (uint32), Longtail_CreateStoreIndex's greedy rule."""
    lib = lib or load()
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    cap = len(lens) + 2
    starts = np.zeros(cap, dtype=np.uint64)
    nb = C.c_uint64(0)
    err = lib.dll.lthip_pack_blocks(len(lens), lens.ctypes.data, max_block_size, max_chunks_per_block, starts.ctypes.data, cap,
                                    C.byref(nb))
    if err:
        raise LongtailHipError(err, "lthip_pack_blocks")
    return starts[: nb.value + 1].astype(np.int64)


class BatchPacker:
    """Resumable greedy packing (lthip_pack_blocks_batch): next() -> (starts[n+1], sizes[n]) of the next batch of blocks, or
    None when every chunk is packed.  The work arrays are allocated once."""

    def __init__(self, lens: np.ndarray, max_block_size: int, max_chunks_per_block: int, max_batch_bytes: int, arena_bytes: int,
                 bound_div: int, bound_add: int, lib: Optional[HipLib] = None):
        self.lib = lib or load()
        self.lens = np.ascontiguousarray(lens, dtype=np.uint32)
        self.args = (max_block_size, max_chunks_per_block, max_batch_bytes, arena_bytes, bound_div, bound_add)
        self.pos = 0
        # a batch of B bytes holds at most B / (limit/2) + 2 blocks that were closed by size, or one per max_chunks chunks
        self.cap = int(min(len(self.lens), max_batch_bytes // max(1, max_block_size // 2) + len(self.lens) // max_chunks_per_block + 8) + 2)
        self.starts = np.empty(self.cap, dtype=np.uint64)
        self.sizes = np.empty(self.cap, dtype=np.uint64)

    def next(self):
        if self.pos >= len(self.lens):
            return None
        nb, nxt = C.c_uint64(0), C.c_uint64(0)
        mb, mc, bb, ab, bd, ba = self.args
        err = self.lib.dll.lthip_pack_blocks_batch(len(self.lens), self.lens.ctypes.data, self.pos, mb, mc, bb, ab, bd, ba,
                                                   self.starts.ctypes.data, self.sizes.ctypes.data, self.cap, C.byref(nb), C.byref(nxt))
        if err:
            raise LongtailHipError(err, "lthip_pack_blocks_batch")
        self.pos = nxt.value
        n = nb.value
        return self.starts[: n + 1].astype(np.int64), self.sizes[:n].astype(np.int64)


def chunker_params(target_chunk_size: int, chunker_min: int = 48):
    """min/avg/max as DynamicChunking derives them (src/longtail.c:1985-1987, 2111-2113)."""
    mn = max(chunker_min, target_chunk_size // 8)
    av = max(chunker_min, target_chunk_size // 2)
    mx = max(chunker_min, target_chunk_size * 2)
    return mn, av, mx
