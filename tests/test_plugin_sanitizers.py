"""The plain-C plugin layer (longtail_amd/csrc/plugin/*.c) under AddressSanitizer + UBSan on the CPU (SURVEY.md §5 asks for it):
tests/san/ builds it against a mock of the device API (the oracle's restatement instead of kernels) and drives it like the
reference core does -- window classes and moves, a tiny window pool shared by 12 threads, failing feeders, streaming hash contexts
with an injected allocation failure, Compress / Decompress from unaligned buffers.  Any sanitizer report or leak fails the run."""
import shutil
import subprocess
from pathlib import Path

import pytest

SAN = Path(__file__).resolve().parent / "san"


def test_plugin_layer_is_clean_under_asan_and_ubsan():
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    probe = subprocess.run(["gcc", "-fsanitize=address,undefined", "-x", "c", "-", "-o", "/dev/null"], input="int main(){return 0;}",
                           capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("gcc lacks the sanitizer runtimes")
    out = subprocess.run(["make", "-C", str(SAN), "run"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "plugin_san: all checks passed" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr


def test_plugin_layer_is_clean_under_tsan():
    """The same driver under ThreadSanitizer: 12 threads on one chunker API with a tiny window pool, the two batchers' hand-overs, the
    codec dispatcher, and chunkers DISPOSED FROM ANOTHER THREAD while the thread that drew their last chunk goes on calling HashBuffer
    (the registry's lock-free look-up counts itself in before it reads a slot; a writer waits for the count to drain)."""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    probe = subprocess.run(["gcc", "-fsanitize=thread", "-x", "c", "-", "-o", "/tmp/lthip_tsan_probe"], input="int main(){return 0;}",
                           capture_output=True, text=True)
    if probe.returncode != 0 or subprocess.run(["/tmp/lthip_tsan_probe"]).returncode != 0:
        pytest.skip("gcc lacks a working ThreadSanitizer runtime here")
    out = subprocess.run(["make", "-C", str(SAN), "run-tsan"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "plugin_san: all checks passed" in out.stdout
    assert "ThreadSanitizer" not in out.stderr
