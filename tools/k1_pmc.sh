#!/bin/bash
# PMC of K1 per flavour: instruction counts and busy cycles (8 GiB, one step)
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
for f in dma16 roll; do
  echo "== LTHIP_K1=$f"
  export LTHIP_K1=$f
  tools/pmc_all.sh pmc_k1_${f}_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary | grep buzhash
  tools/pmc_all.sh pmc_k1_${f}_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary | grep buzhash
  tools/pmc_all.sh pmc_k1_${f}_c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary | grep buzhash
done
