#!/bin/bash
# usage: tools/pmc_cmd.sh <outdir> "<counters>" <kernel-name-substring> <command...>  -- PMC sums for kernels matching the substring
out=$1; shift; ctrs=$1; shift; pat=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/$out -o p -- "$@" > gpurun_out/$out.log 2>&1
python - "$pat" <<PY
import csv,collections,sys
rows=list(csv.DictReader(open("gpurun_out/$out/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
    if sys.argv[1] in k:
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items(): print(k, {a:int(b) for a,b in sorted(v.items())})
PY
rm -f gpurun_out/$out/p_kernel_trace.csv
