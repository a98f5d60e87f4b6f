#!/bin/bash
# which K1 flavour is flaky? the batching-independence tests, several times per flavour
# (the LTHIP_* switches used here exist in the ablation build only: `make ablations`)
export LTHIP_LIB_PATH=${LTHIP_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so}
for f in "$@"; do
  for i in 1 2 3 4; do
    echo "== LTHIP_K1=$f run $i: $(LTHIP_K1=$f timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_configs4.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | grep -E "FAILED|passed|failed" | tr '\n' ' ')"
  done
done
