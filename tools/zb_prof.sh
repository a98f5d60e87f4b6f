#!/bin/bash
# Per-phase wave time of the zstd entropy kernel (debug build build/prof/liblongtail_hip_prof.so with -DLTHIP_ZB_PROF).
# usage: tools/zb_prof.sh <kind> [gib]
cp longtail_amd/liblongtail_hip.so build/cur.so
cp build/prof/liblongtail_hip_prof.so longtail_amd/liblongtail_hip.so
python - "$@" <<'PY'
import sys, subprocess, ctypes
import torch
sys.argv=[sys.argv[0]]+sys.argv[1:]
kind=sys.argv[1]; gib=sys.argv[2] if len(sys.argv)>2 else "2"
import bench, longtail_amd.lib as L
sys.argv=["bench.py","--gib",gib,"--steps","1","--warmup","0","--kind",kind,"--codec","zstd","--no-cpu-baseline"]
bench.main()
L.load().dll.lthip_zb_prof_dump()
PY
cp build/cur.so longtail_amd/liblongtail_hip.so
