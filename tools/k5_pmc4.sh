#!/bin/bash
# round 4: PMC of the lane parser on one kind, both formulations (LTHIP_LZ4_PV=0 / 1); usage: tools/k5_pmc4.sh <kind>
kind=${1:-tokens}
for pv in 0 1; do
export LTHIP_LZ4_PV=$pv
tools/pmc_cmd.sh pmc4_${kind}_pv${pv}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "lz4_segments<16" python tools/k5_probe.py 2 0 $kind
tools/pmc_cmd.sh pmc4_${kind}_pv${pv}_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR" "lz4_segments<16" python tools/k5_probe.py 2 0 $kind
tools/pmc_cmd.sh pmc4_${kind}_pv${pv}_c "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM" "lz4_segments<16" python tools/k5_probe.py 2 0 $kind
done
