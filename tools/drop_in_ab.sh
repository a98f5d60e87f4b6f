#!/bin/bash
# The drop-in measurement alone (bench.py's cpu_baseline leg with the HIP plugin objects in the unmodified core), three workloads:
# random / compressible LZ4 / compressible zstd; for every library of $LIBS (default: the in-tree one) on the same box.
# usage: [LIBS="build/exp/a/liblongtail_hip.so build/exp/b/liblongtail_hip.so"] [ROUNDS=2] tools/drop_in_ab.sh <tag>
tag=${1:-dropin}
mkdir -p gpurun_out
for round in $(seq ${ROUNDS:-1}); do
for lib in ${LIBS:-longtail_amd/liblongtail_hip.so}; do
for leg in "random:--cpu-gib 8" "mixed:--kind mixed --cpu-gib 4" "zstd_mixed:--kind mixed --codec zstd --cpu-gib 4"; do
  name=${leg%%:*}; args=${leg#*:}
  out=gpurun_out/${tag}_dropin_$(basename $(dirname $lib))_$name
  LTHIP_LIB_PATH=$(pwd)/$lib python bench.py --gib 8 --steps 1 --warmup 0 --no-secondary --no-live-traffic $args > $out.json 2> $out.err
  python - "$(basename $(dirname $lib)) $name" $out.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
d = (j.get("secondary") or {}).get("drop_in") or {}
w = d.get("write_content_GBps") or {}
print(sys.argv[1], "| write hip", w.get("hip_plugins"), "cpu", w.get("cpu_plugins"), "| upsync", (d.get("upsync_GBps") or {}).get("hip_plugins"), d.get("error"))
PY
done; done; done
