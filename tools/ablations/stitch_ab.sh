#!/bin/bash
# same-box A/B of the stitch copy's grid (LTHIP_LZ4_STITCH_WGS = workgroups per CU; default: twice what is resident)
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
run() { python bench.py "$@" --no-cpu-baseline --no-live-traffic --no-secondary --steps 3 --warmup 1 2>&1 | grep -E "^\{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['value'], j['ms_per_step'], j['result']['ratio'], {k:v['ms_per_step'] for k,v in j['kernels'].items() if k.startswith('lz4')})
"; }
for w in 8 "" 7 ""; do echo "== mixed, stitch WGs/CU ${w:-default}"; LTHIP_LZ4_STITCH_WGS=$w run --kind mixed; done
for w in 8 "" 8 ""; do echo "== random, stitch WGs/CU ${w:-default}"; LTHIP_LZ4_STITCH_WGS=$w run; done
