#!/bin/bash
# usage: tools/pmc_all.sh <outdir> "<counters>" <bench args...>  -- like pmc.sh but prints every hot kernel
out=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/$out -o p -- python bench.py "$@" > gpurun_out/$out.log 2>&1
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/$out/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:30]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if any(x in k for x in ("lz4_seg","buzhash","blake3_leaves","blake3_parents","stitch_copy","zstd_encode")): print(k, {a:int(b) for a,b in v.items()})
PY
rm -f gpurun_out/$out/p_kernel_trace.csv
