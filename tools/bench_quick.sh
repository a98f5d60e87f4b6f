#!/bin/bash
# quick bench lines: tools/bench_quick.sh <args...>
python bench.py "$@" --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], 'GB/s', j['ms_per_step'], 'ms ratio', j.get('result',{}).get('ratio'), {k:v['ms_per_step'] for k,v in j['kernels'].items()})
"
