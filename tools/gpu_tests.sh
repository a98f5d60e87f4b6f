#!/bin/bash
# Runs the -m gpu test files in separate processes (a crash in one must not hide the others) on the GPU box.
mkdir -p gpurun_out
for f in test_gpu_chunk_hash test_gpu_codecs test_gpu_plugins test_gpu_version_index test_gpu_full_size test_gpu_bench_contract; do
  timeout 900 python -X faulthandler -m pytest tests/$f.py -m gpu -q --timeout=600 -p no:cacheprovider --tb=short > gpurun_out/$f.log 2>&1
  echo "exit $?" >> gpurun_out/$f.log
  echo "== $f: $(grep -E 'passed|failed|exit' gpurun_out/$f.log | tail -2 | tr '\n' ' ')"
done
