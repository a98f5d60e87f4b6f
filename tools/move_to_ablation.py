#!/usr/bin/env python3
"""One-off refactoring helper (round 5): move a line range of a source file into an include file that only the ablation build
(-DLTHIP_ABLATIONS) compiles.   usage: move_to_ablation.py <file> <first> <last> <inc name> [<first> <last> <inc name> ...]
Ranges are 1-based, inclusive, and refer to the file BEFORE any move (they are applied from the bottom up)."""
import sys
from pathlib import Path

src = Path(sys.argv[1])
lines = src.read_text().split("\n")
moves = []
a = sys.argv[2:]
while a:
    moves.append((int(a[0]), int(a[1]), a[2]))
    a = a[3:]
abl = src.parent / "ablations"
abl.mkdir(exist_ok=True)
for first, last, name in sorted(moves, reverse=True):
    body = lines[first - 1:last]
    inc = abl / name
    head = [f"// {name} -- part of {src.name}, compiled by the ablation build only (-DLTHIP_ABLATIONS, `make ablations`):",
            "// an earlier formulation kept as a second implementation for differential tests and A/B runs; not in the product library.", ""]
    inc.write_text("\n".join(head + body) + "\n")
    lines[first - 1:last] = ["#ifdef LTHIP_ABLATIONS", f'#include "ablations/{name}"', "#endif"]
    print(f"{src.name}:{first}-{last} -> ablations/{name} ({len(body)} lines)")
src.write_text("\n".join(lines))
