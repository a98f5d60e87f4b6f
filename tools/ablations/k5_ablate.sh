#!/bin/bash
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
for pv in 0 1; do for dbg in 0 1024 2048 4096 ; do echo "== PV $pv dbg $dbg"; LTHIP_LZ4_PV=$pv python tools/k5_probe.py 2 $dbg tokens,records 2>&1 | grep -v "amdgpu\|parser="; done; done
