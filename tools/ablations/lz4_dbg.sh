#!/bin/bash
# Ablations of the LZ4 segment encoder through LTHIP_LZ4_DBG (bit0 no pre-seed, bit1 no in-batch candidates, bit2 cooperative-only)
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
mkdir -p gpurun_out
for dbg in "$@"; do
  for kind in ${KINDS:-random mixed}; do
    LTHIP_LZ4_DBG=$dbg python bench.py --gib 8 --steps 3 --warmup 1 --kind $kind --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('dbg=$dbg $kind', 'value', j['value'], 'seg_ms', k['lz4_segments']['ms_per_step'], 'GBps', k['lz4_segments']['GBps'], 'stitch_ms', k['lz4_stitch']['ms_per_step'], 'ratio', j['result']['ratio'])
"
  done
done 2>&1 | tee gpurun_out/lz4_dbg.log
