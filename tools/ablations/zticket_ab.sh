#!/bin/bash
# same-box A/B of the zstd encoder's work distribution: tickets (default) against the fixed stride (LTHIP_ZSTD_TICKETS=0)
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
run() { python bench.py "$@" --no-cpu-baseline --no-live-traffic --no-secondary --steps 3 --warmup 1 2>&1 | grep -E "^\{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['value'], j['ms_per_step'], j['result']['ratio'], j['result']['compressed_bytes'], {k:v['ms_per_step'] for k,v in j['kernels'].items() if k.startswith('lz4') or k.startswith('zstd')})
"; }
for d in 0 1 0 1; do echo "== mixed zstd, LTHIP_ZSTD_TICKETS=$d"; LTHIP_ZSTD_TICKETS=$d run --kind mixed --codec zstd; done
