#!/bin/bash
# PMC of the zstd entropy kernel (k_zstd_encode) on one data kind: usage tools/zb_pmc.sh <kind> [gib]
kind=${1:-mixed}; gib=${2:-8}
cmd="python bench.py --gib $gib --steps 1 --warmup 0 --kind $kind --codec zstd --no-cpu-baseline --no-secondary --no-live-traffic"
tools/pmc_cmd.sh zbpmc_${kind}_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "k_zstd_encode" $cmd
tools/pmc_cmd.sh zbpmc_${kind}_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" "k_zstd_encode" $cmd
tools/pmc_cmd.sh zbpmc_${kind}_c "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_VMEM" "k_zstd_encode" $cmd
