#!/usr/bin/env python3
"""Identity of the product sources: sha256 over (relative path, contents) of every file under longtail_amd/csrc/ and
include/, in sorted path order.  The Makefile bakes the first 16 hex digits into liblongtail_hip.so
(lthip_build_id()); tests/test_abi.py and tests/test_gpu_build_id.py recompute it from the tree, so a library that
was not built from the sources it travels with fails loudly.

    python tools/build_id.py            print the id
    python tools/build_id.py --stamp F  rewrite F only when the id changed (keeps make from relinking needlessly)
"""
from __future__ import annotations

import hashlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SUFFIXES = {".hip", ".h", ".c", ".inc"}


def source_files(root: Path = ROOT):
    files = [p for d in ("longtail_amd/csrc", "include") for p in (root / d).rglob("*") if p.is_file() and p.suffix in SUFFIXES]
    return sorted(files, key=lambda p: p.relative_to(root).as_posix())


def build_id(root: Path = ROOT) -> str:
    h = hashlib.sha256()
    for p in source_files(root):
        h.update(p.relative_to(root).as_posix().encode() + b"\0")
        data = p.read_bytes()
        h.update(len(data).to_bytes(8, "little"))
        h.update(data)
    return h.hexdigest()[:16]


if __name__ == "__main__":
    bid = build_id()
    if len(sys.argv) == 3 and sys.argv[1] == "--stamp":
        stamp = Path(sys.argv[2])
        text = f'#define LTHIP_BUILD_ID "{bid}"\n'
        if not stamp.exists() or stamp.read_text() != text:
            stamp.parent.mkdir(parents=True, exist_ok=True)
            stamp.write_text(text)
    print(bid)
