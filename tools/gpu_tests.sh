#!/bin/bash
# Runs the -m gpu test files in separate processes (a crash in one must not hide the others) on the GPU box.
# usage: tools/gpu_tests.sh [tag]     tag "abl": the whole suite on the ablation build (LTHIP_LIB_PATH=build/ablations/liblongtail_hip.so)
tag=${1:-}
mkdir -p gpurun_out
if [ "$tag" = "abl" ]; then export LTHIP_LIB_PATH=$(pwd)/build/ablations/liblongtail_hip.so; fi
for f in test_gpu_chunk_hash test_gpu_codecs test_gpu_plugins test_gpu_version_index test_gpu_full_size test_gpu_bench_contract test_gpu_ingest test_gpu_comm test_gpu_boundary test_gpu_configs4 test_gpu_build_id; do
  timeout 900 python -X faulthandler -m pytest tests/$f.py -m gpu -q --timeout=600 -p no:cacheprovider --tb=short > gpurun_out/$f$tag.log 2>&1
  echo "exit $?" >> gpurun_out/$f$tag.log
  echo "== $f $tag: $(grep -E 'passed|failed|exit' gpurun_out/$f$tag.log | tail -2 | tr '\n' ' ')"
done
