#!/bin/bash
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
run() { python bench.py "$@" --no-cpu-baseline --no-live-traffic --no-secondary --steps 3 --warmup 1 2>&1 | grep -E "^\{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['value'], j['ms_per_step'], j['result']['ratio'], {k:v['ms_per_step'] for k,v in j['kernels'].items() if k.startswith('lz4') or k.startswith('zstd')})
"; }
for d in 0 16384 0 16384; do echo "== mixed lz4 LTHIP_LZ4_DBG=$d"; LTHIP_LZ4_DBG=$d run --kind mixed; done
for k in records tokens; do for d in 0 16384; do echo "== $k lz4 LTHIP_LZ4_DBG=$d"; LTHIP_LZ4_DBG=$d run --kind $k --gib 16; done; done
