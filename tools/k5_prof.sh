#!/bin/bash
# Per-phase wave cycles of the LZ4 lane parser (debug build build/prof/liblongtail_hip_prof.so with -DLTHIP_K5_PROF).
# usage: tools/k5_prof.sh <kind> [gib] [LTHIP_LZ4_DBG]
cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
LTHIP_LZ4_DBG=${3:-0} python - "$@" <<'PY'
import sys, os
sys.argv = [sys.argv[0]] + sys.argv[1:]
kind = sys.argv[1]; gib = sys.argv[2] if len(sys.argv) > 2 else "2"
sys.argv = ["k5_probe.py", gib, os.environ.get("LTHIP_LZ4_DBG", "0"), kind]
sys.path.insert(0, "tools")
import longtail_amd.lib as L
exec(open("tools/k5_probe.py").read())
L.load().dll.lthip_k5_prof_dump(1)
PY
cp build/cur.so longtail_amd/liblongtail_hip.so
