#!/bin/bash
# round 4: PMC of the lane parser on one kind: the round-3 formulation (LTHIP_LZ4_PV=0: k_lz4_segments<16,...,0>) and the shipped one
# (k_lz4_lanes2: register records + register window + padded LDS rows + half-groups); usage: tools/ablations/k5_pmc4.sh <kind>
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
kind=${1:-tokens}
for pv in 0 1; do
export LTHIP_LZ4_PV=$pv
pat=$([ $pv = 0 ] && echo "lz4_segments<16" || echo "lz4_lanes2")
python tools/k5_probe.py 2 0 $kind 2>&1 | grep -v "amdgpu\|parser="
tools/pmc_cmd.sh pmc4_${kind}_pv${pv}_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "$pat" python tools/k5_probe.py 2 0 $kind
tools/pmc_cmd.sh pmc4_${kind}_pv${pv}_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" "$pat" python tools/k5_probe.py 2 0 $kind
tools/pmc_cmd.sh pmc4_${kind}_pv${pv}_c "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "$pat" python tools/k5_probe.py 2 0 $kind
done
