#!/bin/bash
# A/B of two builds of the library on the LZ4 path: tools/ablations/lz4_ab.sh build/a.so build/b.so
mkdir -p gpurun_out
cp longtail_amd/liblongtail_hip.so build/cur.so
for so in "$@"; do
  cp $so longtail_amd/liblongtail_hip.so
  for kind in random mixed; do for rep in 1 2; do
    python bench.py --gib 8 --steps 3 --warmup 1 --kind $kind --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$so $kind', 'value', j['value'], 'seg_ms', k['lz4_segments']['ms_per_step'], 'GBps', k['lz4_segments']['GBps'], 'stitch_ms', k['lz4_stitch']['ms_per_step'], 'ratio', j['result']['ratio'])
"
  done; done
done 2>&1 | tee gpurun_out/lz4_ab.log
cp build/cur.so longtail_amd/liblongtail_hip.so
