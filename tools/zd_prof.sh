#!/bin/bash
# Per-phase wave time of the zstd decoder kernels (debug build build/prof/liblongtail_hip_prof.so, `make prof`):
# tools/zd_prof.sh <gib> <kind>.  Marks 1-10 belong to the encoder, 11-18 to the decoder core / k_zstd_prepare.
cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
python - "$@" <<'PY'
import sys, runpy
sys.argv = ["tools/decode_rate.py"] + sys.argv[1:]
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import longtail_amd.lib as L
g = runpy.run_path("tools/decode_rate.py")
L.load().dll.lthip_zb_prof_dump()
PY
cp build/cur.so longtail_amd/liblongtail_hip.so
