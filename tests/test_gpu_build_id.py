"""The library the GPU tests load must be HEAD: lthip_build_id() (baked in by the Makefile) equals the hash recomputed
from the source tree that travelled with it (tools/build_id.py)."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def test_gpu_library_is_built_from_this_tree(gpu, hiplib):
    sys.path.insert(0, str(ROOT / "tools"))
    from build_id import build_id

    assert hiplib.build_id() == build_id(ROOT), "stale liblongtail_hip.so on the GPU box"
