"""pytest configuration: `-m gpu` = parity tests that need an MI355X, everything else runs on CPU."""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests._libs import oracle as _o

    return _o()


@pytest.fixture(scope="session")
def ref():
    from tests._libs import have_ref, ref as _r

    if not have_ref():
        pytest.skip("oracle/_ref/liblongtail_ref.so not built (needs /root/reference at build time)")
    return _r()


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    from tests._libs import GOLDEN

    j = json.load(open(GOLDEN / "reference_tests.json"))
    v = np.load(GOLDEN / "ref_vectors.npz")
    inp = np.fromfile(GOLDEN / "chunker.input", dtype=np.uint8)
    return {"tests": j, "vec": v, "chunker_input": inp, "cases": json.loads(str(v["chunk_cases"]))}


@pytest.fixture(scope="session")
def hiplib():
    from longtail_amd.lib import load

    p = ROOT / "longtail_amd" / "liblongtail_hip.so"
    if not p.exists():
        import __graft_entry__ as g

        g.build()
    return load()


@pytest.fixture(scope="session")
def gpu(hiplib):
    """A Context on cuda:0.  GPU tests must exercise the native library: no fallback, fail loudly."""
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert hiplib.device_count() > 0, "liblongtail_hip.so sees no GPU"
    from longtail_amd.lib import Context

    ctx = Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_abl(hiplib):
    """A Context of the ABLATION build (build/ablations/liblongtail_hip.so, `make ablations`): the differential tests that switch to an
    earlier formulation of a kernel or a debug path (LTHIP_* variables the product library does not read) run on it."""
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from longtail_amd.lib import ABLATIONS_LIB_PATH, Context, load_ablations

    if not ABLATIONS_LIB_PATH.exists() and not os.environ.get("LTHIP_LIB_PATH"):
        import subprocess

        subprocess.run(["make", "-C", str(ROOT), "-j", str(max(2, os.cpu_count() or 2)), "ablations"], check=True, capture_output=True)
    ctx = Context(0, lib=load_ablations())
    yield ctx
    ctx.close()


@pytest.fixture(autouse=True)
def _library_rereads_its_switches(request):
    """The library caches its environment switches per process; a test that sets one (monkeypatch) must not leak the cached value
    into the next test: read them again after every GPU test."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import longtail_amd.lib as L

        L.load().dll.lthip_debug_reload_env()
        if L._abl is not None:  # (loaded by a test of this session)
            L._abl.dll.lthip_debug_reload_env()
