#!/bin/bash
# usage: tools/prof_stats.sh <tag> <bench args...>   -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
tag=$1; shift
here=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $here
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o p -- python bench.py "$@" > gpurun_out/${tag}_prof.log 2>&1
cp gpurun_out/${tag}_prof/p_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
rm -f gpurun_out/${tag}_prof/p_kernel_trace.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))[:14]:
    print(r["Name"].replace("(anonymous namespace)::","").split("(")[0][:40], r["Calls"], round(float(r["AverageNs"])/1e6,3), "ms avg", r["Percentage"])
PY
