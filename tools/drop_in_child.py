#!/usr/bin/env python
"""bench.py's drop_in legs, in a process of their own: what an embedder gets that (a) makes Longtail_Hip_SetBlockingWaits(1) its first call
-- the device's wait policy must be set before the process touches the GPU, which bench.py's own process long has -- and (b) swaps the
three constructors.  argv[1] = JSON {cfg, sample_bytes, workers, target_chunk_size, block_size, max_chunks_per_block, reps};
prints ONE JSON line: the medians of the unmodified reference core with the HIP chunker + hash + codec at W and 2 W, with the HIP
chunker + hash and the reference's codec at W, and the bytes the HIP codec stored."""
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401  (its HIP runtime first; importing it does not touch the device)

from longtail_amd.lib import load

req = json.loads(sys.argv[1])
lib = load()
lib.dll.Longtail_Hip_SetBlockingWaits.restype = C.c_int
policy_rc = int(lib.dll.Longtail_Hip_SetBlockingWaits(1)) if req.get("blocking_waits", True) else None

import bench  # noqa: E402

cfg = req["cfg"]
args = bench.make_parser().parse_args(["--gib", str(max(1.0, req["sample_bytes"] / (1 << 30))), "--kind", cfg["kind"], "--codec", cfg["codec"], "--no-secondary",
                                       "--no-live-traffic", "--target-chunk-size", str(req["target_chunk_size"]), "--block-size", str(req["block_size"]),
                                       "--max-chunks-per-block", str(req["max_chunks_per_block"])])
b = bench.Bench(args)
cr = bench.CpuReference(b, args)
cr.REPS = int(req.get("reps", 3))
files, nbytes = cr.sample_files(cfg, int(req["sample_bytes"]))
r = cr.r
tag = r.lz4_type if cfg["codec"] == "lz4" else r.zstd_default
common = (args.target_chunk_size, args.block_size, args.max_chunks_per_block, tag)
tree = r.tree_create(files, tag)
out = {"nbytes": nbytes, "files": len(files), "blocking_waits_rc": policy_rc}
try:
    chunker, hasher, codec_api = cr.plugins(cfg["codec"])
    w = int(req["workers"])
    r.version_index(files[: min(256, len(files))] if len(files) > 1 else [(files[0][0], files[0][1][: 256 << 20])], args.target_chunk_size, w, 0, chunker, hasher)
    ws = [w] + [x for x in req.get("more_workers", [])]
    hip = r.ingest_sweep_tree(tree, *common, ws, cr.REPS, chunker, hasher, codec_api)
    if hip["err"]:
        out["error"] = f"errno {hip['err']}"
    else:
        out["hip"] = cr._median(hip, ws, nbytes)
        out["hip_raw_bytes"], out["hip_stored_bytes"], out["hip_blocks"] = hip["raw_bytes"], hip["stored_bytes"], hip["blocks"]
        mixed = r.ingest_sweep_tree(tree, *common, [w], cr.REPS, chunker, hasher, None)
        if not mixed["err"]:
            out["hip_chunker_hash_cpu_codec"] = cr._median(mixed, [w], nbytes)[str(w)]
finally:
    r.tree_destroy(tree)
    cr.close()
print(json.dumps(out), flush=True)
