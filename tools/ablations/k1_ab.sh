#!/bin/bash
# A/B of the K1 flavours on the GPU box: parity tests of the chunker, then bench lines per flavour (16 GiB, kernels only)
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
mkdir -p gpurun_out
for f in dma16; do
  echo "== LTHIP_K1=$f"
  LTHIP_K1=$f timeout 900 python -m pytest tests/test_gpu_chunk_hash.py -m gpu -q -x --timeout=600 -p no:cacheprovider --tb=short 2>&1 | tail -3
  LTHIP_K1=$f tools/quick_bench.sh "--gib 16 --steps 3 --warmup 1 --no-secondary"
done
