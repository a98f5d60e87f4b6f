#!/bin/bash
# PMC of the LZ4 match finder on compressible data (2 GiB "mixed" as 8 MiB blocks, two calls: warm-up + timed): the same counters
# as the round-2 runs gpurun_out/pmc_k5a / pmc_k5b, so that profiles/r03_pmc_busy_k5.txt can set them side by side.
tools/pmc_cmd.sh pmc_k5_r3a "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVES" lz4_segments python tools/k5_probe.py 2 0 mixed
tools/pmc_cmd.sh pmc_k5_r3b "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" lz4_segments python tools/k5_probe.py 2 0 mixed
grep ratio gpurun_out/pmc_k5_r3a.log
