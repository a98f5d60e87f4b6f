export LTHIP_LIB_PATH=$(pwd)/build/ablations/liblongtail_hip.so
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d[\"value\"], 'GB/s', d[\"ms_per_step\"], 'ms ratio', d[\"result\"][\"ratio\"], 'match finder', d[\"kernels\"][\"lz4_segments\"][\"ms_per_step\"])"; }
for kind in mixed tokens; do for q in 1 2 3 4 6; do
  echo -n "$kind quiet=$q: "
  LTHIP_LZ4_DBG=$((q<<17)) timeout 300 python bench.py --kind $kind --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | line
done; done
