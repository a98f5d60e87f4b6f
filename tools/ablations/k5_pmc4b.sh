#!/bin/bash
# usage: tools/ablations/k5_pmc4b.sh <kind> <pv> <dbg>: LDS counters + timing of the lane parser for one setting
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
kind=$1; export LTHIP_LZ4_PV=$2; dbg=$3
python tools/k5_probe.py 2 $dbg $kind 2>&1 | grep -v "amdgpu\|parser="
tools/pmc_cmd.sh pmc4b_${kind}_$2_$3 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "lz4_segments<16" python tools/k5_probe.py 2 $dbg $kind
