#!/bin/bash
# HBM traffic per kernel launch (guide: FETCH_SIZE and WRITE_SIZE in separate --pmc passes; KiB units; on gfx950
# FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled when comparing with byte counts).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic/$c -o p -- python bench.py "$@" > gpurun_out/traffic/$c.log 2>&1
done
python - <<'PY'
import csv,collections,json
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=list(csv.DictReader(open(f"gpurun_out/traffic/{c}/p_counter_collection.csv")))
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in rows:
        if r["Counter_Name"]!=c: continue
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        agg[k]+=float(r["Counter_Value"]); n[k]+=1
    for k in agg: out.setdefault(k,{})[c+"_KiB_per_launch"]=agg[k]/n[k]; out[k]["launches"]=n[k]
json.dump(out,open("gpurun_out/traffic/traffic.json","w"),indent=1)
for k,v in out.items():
    if any(x in k for x in ("lz4","buzhash","blake3","select")): print(k,v)
PY
rm -f gpurun_out/traffic/*/p_kernel_trace.csv
