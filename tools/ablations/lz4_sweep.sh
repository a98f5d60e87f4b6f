#!/bin/bash
mkdir -p gpurun_out
for kind in random mixed; do for sl in 11 12 13; do
  python bench.py --gib 8 --steps 2 --warmup 1 --kind $kind --segment-log2 $sl --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$kind seg=$sl', 'value', j['value'], 'seg_ms', k['lz4_segments']['ms_per_step'], 'GBps', k['lz4_segments']['GBps'], 'stitch_ms', k['lz4_stitch']['ms_per_step'], 'ratio', j['result']['ratio'])
"
done; done 2>&1 | tee gpurun_out/lz4_sweep.log
