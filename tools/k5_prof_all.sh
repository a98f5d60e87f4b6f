#!/bin/bash
# Per-phase wave cycles of the LZ4 lane parser on every kind (debug build build/prof/liblongtail_hip_prof.so: make prof, -DLTHIP_K5_PROF),
# blocks 12 345 bytes into the data (blocks at chunk offsets, like bench.py).  usage: tools/k5_prof_all.sh > profiles/<tag>_k5_phases_shifted.txt
export LTHIP_LIB_PATH=$(cd "$(dirname "$0")/.." && pwd)/build/prof/liblongtail_hip_prof.so
for kind in mixed tokens records lines; do
python - $kind <<'PY'
import sys, os
kind = sys.argv[1]
sys.argv = ["k5_probe.py", "2", "0", kind, "lz4", "12345"]
sys.path.insert(0, "tools")
import longtail_amd.lib as L
exec(open("tools/k5_probe.py").read())
L.load().dll.lthip_k5_prof_dump(1)
PY
done
