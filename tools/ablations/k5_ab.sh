#!/bin/bash
# K5 ablations: LTHIP_LZ4_DBG bits (1 no pre-seed, 2 no in-batch, 4 cooperative only, 8 no twin probe, 16 no stride jump)
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
mkdir -p gpurun_out
for dbg in "$@"; do for kind in ${KINDS:-random mixed}; do
  LTHIP_LZ4_DBG=$dbg python bench.py --gib 8 --steps 3 --warmup 1 --kind $kind --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('dbg=$dbg $kind', 'value', j['value'], 'seg_ms', k['lz4_segments']['ms_per_step'], 'GBps', k['lz4_segments']['GBps'], 'ratio', j['result']['ratio'])
"
done; done 2>&1 | tee gpurun_out/k5_ab.log
