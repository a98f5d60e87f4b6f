#!/bin/bash
# round 5: the lane parser's adaptive miss stepping against the fixed rule, and the number of quiet rounds before the one-byte steps stop
# (LTHIP_LZ4_DBG bit 16: fixed rule; bits 17-19: quiet rounds, default LZ4_QUIET).  Ablation build.  -> profiles/r05_adaptive_stepping.txt
export LTHIP_LIB_PATH=$(cd "$(dirname "$0")/.." && pwd)/build/ablations/liblongtail_hip.so
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d[\"value\"], 'GB/s', d[\"ms_per_step\"], 'ms ratio', d[\"result\"][\"ratio\"], 'match finder', d[\"kernels\"][\"lz4_segments\"][\"ms_per_step\"])"; }
python tools/text_ratio_probe.py 2>&1 | grep -v amdgpu.ids
for codec in lz4 zstd; do for d in 65536 0 $((4<<17)) $((3<<17)) $((2<<17)) $((1<<17)) $((3<<29)); do
  echo -n "mixed $codec LTHIP_LZ4_DBG=$d: "
  LTHIP_LZ4_DBG=$d python bench.py --kind mixed --codec $codec --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | line
done; done
for d in 65536 0 $((2<<17)); do echo -n "dedup lz4 LTHIP_LZ4_DBG=$d: "; LTHIP_LZ4_DBG=$d python bench.py --kind mixed --dups --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | line; done
