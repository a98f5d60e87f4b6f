#!/bin/bash
# Phase-2 codec comparison on 8 GiB per data kind: tools/ablations/codec_sweep.sh [kinds...]
mkdir -p gpurun_out
for kind in ${@:-random mixed records tokens lines}; do for codec in lz4 zstd; do
  python bench.py --gib 8 --steps 2 --warmup 1 --kind $kind --codec $codec --no-cpu-baseline 2>gpurun_out/codec_sweep.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels']
        print('$kind $codec', 'value', j['value'], 'ratio', j['result']['ratio'], 'phase2_ms', j['phase_ms']['pack_compress'], {n:v['ms_per_step'] for n,v in k.items() if n.startswith(('lz4','zstd','other'))})
"
done; done 2>&1 | tee gpurun_out/codec_sweep.log
tail -3 gpurun_out/codec_sweep.err
