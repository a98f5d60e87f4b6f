#!/bin/bash
# Cycle breakdown of the LZ4 decoder (debug build build/prof/liblongtail_hip_prof.so with -DLTHIP_DEC_PROF; `make prof` first).
# usage: tools/dec_prof.sh [gib] [kind]
cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
python - "$@" <<'PY'
import sys, runpy
import longtail_amd.lib as L
sys.argv = ["decode_rate.py"] + sys.argv[1:]
try:
    runpy.run_path("tools/decode_rate.py", run_name="__main__")
finally:
    L.load().dll.lthip_dec_prof_dump()
PY
cp build/cur.so longtail_amd/liblongtail_hip.so
