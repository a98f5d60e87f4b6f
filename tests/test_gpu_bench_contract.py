"""-m gpu: bench.py end to end on a small tree: the JSON contract line, and that the result does not depend on how the
blocks were batched or assembled (blocks that span assets go through the device gather, the others are compressed where
they lie; src/longtail.c:4640-4721)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_bench(*args, extra_env=None):
    env = dict(os.environ, PYTHONPATH=str(ROOT), **(extra_env or {}))
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", *args],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_bench_line_and_batching_independence(codec):
    base = ["--gib", "0.5", "--tree", "mixed-sizes", "--kind", "mixed", "--codec", codec]
    a = run_bench(*base)
    b = run_bench(*base, "--batch-gib", "0.05")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in a, key
    assert a["n_gpus"] == 1 and a["higher_is_better"] is True and a["scaling"] == "weak" and a["value"] > 0
    assert set(a["phase_ms"]) == {"chunk_hash", "exchange", "index", "write_finish"}  # the three reference calls of SURVEY.md §8d
    assert a["result"]["version_index_bytes"] > 0 and a["result"]["store_index_bytes_rank0"] > 0
    assert a["config"]["device_block_assembly"] is True  # files are not multiples of 16 bytes: some blocks span assets
    # the codec entry points cut large calls into internal batches (LTHIP_BATCH_BYTES): same payloads whatever the cut
    c = run_bench(*base, extra_env={"LTHIP_BATCH_BYTES": str(48 << 20)})
    for key in ("chunks", "unique_chunks", "blocks", "compressed_bytes", "raw_bytes_written", "version_index_bytes", "store_index_bytes_rank0"):
        assert a["result"][key] == b["result"][key] == c["result"][key], key
    assert a["result"]["ratio"] > 1.3


def test_bench_equal_files_need_no_gather():
    j = run_bench("--gib", "0.25", "--kind", "random")
    assert j["config"]["device_block_assembly"] is False and 0.99 < j["result"]["ratio"] <= 1.0


@pytest.mark.parametrize("world,scaling,partition", [(2, "weak", "range"), (4, "strong", "lpt")])
def test_bench_multi_rank_flow_on_one_gpu(world, scaling, partition):
    """The N > 1 path end to end (job partition, all-gather of per-job chunk lists, reorder, ownership, per-rank WriteContent) with
    all ranks on the one GPU and the exchange over gloo: a functional check, not a measurement.  The tree-wide results must equal
    the single-rank run of the same tree."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    gib = 0.5
    total = gib * world if scaling == "weak" else gib
    env = dict(os.environ, PYTHONPATH=str(ROOT), LONGTAIL_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--gib", str(gib), "--scaling", scaling, "--partition", partition, "--tree", "mixed-sizes", "--kind", "mixed"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    multi = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    single = run_bench("--gib", str(total), "--tree", "mixed-sizes", "--kind", "mixed")
    assert multi["n_gpus"] == world and multi["scaling"] == scaling
    assert multi["config"]["tree_bytes"] == single["config"]["tree_bytes"]
    for key in ("chunks", "unique_chunks", "raw_bytes_written", "version_index_bytes"):
        assert multi["result"][key] == single["result"][key], key
    # blocks are packed per rank (each rank = CreateMissingContent against the others' chunks): at most world - 1 more blocks... per tag run
    assert single["result"]["blocks"] <= multi["result"]["blocks"] <= single["result"]["blocks"] + 4 * world


@pytest.mark.parametrize("launch", ["torch", "plain", "run8.sh", "plain-lpt3"])
def test_bench_starts_its_own_ranks(launch):
    """`python bench.py --gpus 2` as the driver types it (no launcher, WORLD_SIZE unset) starts two ranks itself, and the line says
    n_gpus 2 -- with torch.distributed (gloo standing in for RCCL on this 1-GPU box) and with the plain launch, where every collective
    -- the three all-gathers AND the sharded first-seen table's all-to-all -- goes through the C ABI (lthip_comm_*, the shared-memory
    transport standing in for RCCL).  Same tree-wide results as the single-rank run."""
    gib = 0.25
    world = 3 if launch == "plain-lpt3" else 2  # (LPT: rank-major order is NOT job order -- the device reorder has many runs)
    args = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--gib", str(gib), "--tree", "mixed-sizes", "--kind", "mixed",
            "--partition", "lpt" if launch == "plain-lpt3" else "range", "--dedup", "sharded"]
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if launch == "torch":
        env["LONGTAIL_DIST_BACKEND"] = "gloo"
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *args]
    elif launch in ("plain", "plain-lpt3"):
        env["LTHIP_COMM_TRANSPORT"] = "shm"
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--launch", "plain", *args]
    else:
        env["LTHIP_COMM_TRANSPORT"] = "shm"
        cmd = ["bash", str(ROOT / "tools" / "run8.sh"), "2", *args]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    multi = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert multi["n_gpus"] == world and multi["config"]["comm"]["nranks"] == world and multi["config"]["dedup_table"] == "sharded"
    assert multi["config"]["comm"]["transport"] == ("torch.distributed/gloo" if launch == "torch" else "host-shm")
    single = run_bench("--gib", str(world * gib), "--tree", "mixed-sizes", "--kind", "mixed")
    assert multi["config"]["tree_bytes"] == single["config"]["tree_bytes"]
    for key in ("chunks", "unique_chunks", "raw_bytes_written", "version_index_bytes"):
        assert multi["result"][key] == single["result"][key], key
