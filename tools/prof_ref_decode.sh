#!/bin/bash
# per-kernel times of tools/decode_rate_ref.py (payloads made by the reference encoders) under rocprofv3: tools/prof_ref_decode.sh <blocks> <kind> <tag>
here=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $here
mkdir -p gpurun_out
rm -rf gpurun_out/prof_$3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$3 -o p -- python tools/decode_rate_ref.py $1 $2 > gpurun_out/prof_$3.log 2>&1
grep -v "amdgpu.ids\|simple_timer" gpurun_out/prof_$3.log | tail -2
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_$3/p_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>4s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
rm -f gpurun_out/prof_$3/p_kernel_trace.csv
