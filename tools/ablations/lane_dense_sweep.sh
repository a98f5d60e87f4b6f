#!/bin/bash
# round 5: what the lane parser's miss stepping buys and costs on the compressible tree -- LTHIP_LZ4_DBG bits 29-30: one-byte steps
# after a hit before the parser steps by aligned dwords (4 default, 2, 1, 0); ratio against step time.  Ablation build.
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
for codec in lz4 zstd; do for d in 0 536870912 1073741824 1610612736; do
  echo -n "codec $codec LTHIP_LZ4_DBG=$d: "
  LTHIP_LZ4_DBG=$d python bench.py --kind mixed --codec $codec --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d[\"value\"], 'GB/s', d[\"ms_per_step\"], 'ms ratio', d[\"result\"][\"ratio\"], 'match finder', d[\"kernels\"][\"lz4_segments\"][\"ms_per_step\"])"
done; done
