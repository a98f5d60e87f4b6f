#!/bin/bash
# the private table's pre-seed (the aligned dwords of the unit before) on / off: LTHIP_LZ4_DBG bit 0, ablation build
export LTHIP_LIB_PATH=$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d[\"value\"], 'GB/s', d[\"ms_per_step\"], 'ms ratio', d[\"result\"][\"ratio\"], 'match finder', d[\"kernels\"][\"lz4_segments\"][\"ms_per_step\"])"; }
for codec in lz4 zstd; do for kind in mixed tokens records; do for w in 0 1; do
  echo -n "$codec $kind LTHIP_LZ4_DBG=$w: "
  LTHIP_LZ4_DBG=$w python bench.py --kind $kind --codec $codec --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | line
done; done; done
