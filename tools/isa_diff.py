#!/usr/bin/env python3
"""Compare the device code of two builds kernel by kernel (refactorings that must not change a shipped kernel).
usage: isa_diff.py <before.s> <after.s> [old_name_substring=new_name_substring ...]
Both files come from `hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S`.  Kernels are matched by demangled-ish name (the
mangled symbol with the given substitutions applied to the BEFORE side); instruction streams are compared with local labels
renumbered in order of appearance and comments stripped."""
import re
import sys


def functions(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L") and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                out[name] = body
                name = None
                continue
            t = line.split(";")[0].rstrip()
            if t.strip() and not t.strip().startswith((".p2align", ".loc", ".file", ".cfi")):
                body.append(t.strip())
    return out


def normalise(body):
    labels = {}

    def lab(m):
        return labels.setdefault(m.group(0), f".L{len(labels)}")

    return [re.sub(r"\.L[A-Za-z_]*\d+(_\d+)?", lab, t) for t in body]


def main():
    before, after = functions(sys.argv[1]), functions(sys.argv[2])
    subs = [a.split("=", 1) for a in sys.argv[3:]]
    renamed = {}
    for k, v in before.items():
        k2 = k
        for a, b in subs:
            k2 = k2.replace(a, b)
        renamed[k2] = (k, v)
    same = diff = 0
    for k in sorted(after):
        if k not in renamed:
            print(f"NEW      {k} ({len(after[k])} lines)")
            continue
        a, b = normalise(renamed[k][1]), normalise(after[k])
        if a == b:
            same += 1
        else:
            diff += 1
            first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
            print(f"DIFFERS  {k}: {len(a)} -> {len(b)} lines, first difference at {first}: {a[first] if first < len(a) else '-'} | {b[first] if first < len(b) else '-'}")
    for k in sorted(renamed):
        if k not in after:
            print(f"GONE     {renamed[k][0]} ({len(renamed[k][1])} lines)")
    print(f"{same} kernels identical, {diff} differ")


if __name__ == "__main__":
    main()
