#!/bin/bash
# Per-phase wave time of the zstd decoder kernels on payloads made by the reference encoder (debug build build/prof/liblongtail_hip_prof.so
# with -DLTHIP_ZB_PROF; `make prof` first).  usage: tools/zd_prof_ref.sh <blocks> [kind]
cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
python - "$@" <<'PY'
import sys, runpy
import longtail_amd.lib as L
sys.argv = ["decode_rate_ref.py"] + sys.argv[1:]
try:
    runpy.run_path("tools/decode_rate_ref.py", run_name="__main__")
finally:
    L.load().dll.lthip_zb_prof_dump()
PY
cp build/cur.so longtail_amd/liblongtail_hip.so
