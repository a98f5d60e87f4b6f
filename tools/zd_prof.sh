#!/bin/bash
# Per-phase wave time of the zstd DEcoder (debug build build/prof/liblongtail_hip_prof.so, `make prof` first).
# usage: tools/zd_prof.sh [gib] [kind]    marks: 11 block start, 12 literals section, 13 sequence tables, 14 sequence decode, 15 copies
cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
python - "$@" <<'PY'
import sys, runpy
import longtail_amd.lib as L
sys.argv = ["decode_rate.py"] + sys.argv[1:]
L.load().dll.lthip_zb_prof_dump  # the symbol must exist
try:
    runpy.run_path("tools/decode_rate.py", run_name="__main__")
finally:
    L.load().dll.lthip_zb_prof_dump()
PY
cp build/cur.so longtail_amd/liblongtail_hip.so
