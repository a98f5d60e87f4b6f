"""longtail_amd -- MI355X (gfx950) implementation of longtail's chunk -> hash -> compress hot path.

The product is ``liblongtail_hip.so`` (hand-written HIP kernels + a plain-C plugin layer exposing longtail's own
``Longtail_ChunkerAPI`` / ``Longtail_HashAPI`` / ``Longtail_CompressionAPI`` structs, see ``include/longtail_hip.h``).
This Python package is only the harness around it: it builds the library in-tree (``build()``), binds its C ABI with
ctypes (``longtail_amd.lib``) and drives it from tests / ``bench.py`` using torch for device memory, streams and
``torch.distributed`` (RCCL).
"""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

REPO_ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = Path(__file__).resolve().parent / "liblongtail_hip.so"


def build(verbose: bool = False) -> Path:
    """Compile every HIP/C source for gfx950 into longtail_amd/liblongtail_hip.so (in-tree)."""
    cmd = ["make", "-C", str(REPO_ROOT), "-j", str(max(2, os.cpu_count() or 2)), "lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 or verbose:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building liblongtail_hip.so failed")
    return LIB_PATH


from .lib import HipLib, Context, LongtailHipError, load  # noqa: E402,F401
