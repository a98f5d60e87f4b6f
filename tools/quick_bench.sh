#!/bin/bash
# usage: tools/quick_bench.sh "<bench args>" ["<bench args>" ...]  -> one summary line per configuration
for a in "$@"; do
  python bench.py $a --no-cpu-baseline 2>gpurun_out/quick_err.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$a |', j['value'], 'GB/s', j['ms_per_step'], 'ms', json.dumps({k:v['ms_per_step'] for k,v in j['kernels'].items()}), j['phase_ms'], 'ratio', j['result']['ratio'])
"
  tail -2 gpurun_out/quick_err.log | grep -i -E "error|Traceback" 
done
