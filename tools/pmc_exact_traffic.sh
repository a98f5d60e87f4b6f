#!/bin/bash
# Exact memory-side traffic per kernel from the L2's request-size counters (round 3): bytes read = 128*RDREQ_128B + 64*RDREQ_64B +
# 32*RDREQ_32B, bytes written = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B); RDREQ_DRAM = read requests that went to DRAM.
# usage: tools/pmc_exact_traffic.sh <out.json> <input bytes> <bench args...>   (two --pmc passes, each with --kernel-trace only)
out=$1; shift; inp=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/xtraffic
timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/xtraffic/rd -o p -- python bench.py "$@" > gpurun_out/xtraffic/rd.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum --output-format csv -d gpurun_out/xtraffic/wr -o p -- python bench.py "$@" > gpurun_out/xtraffic/wr.log 2>&1
python - "$out" "$inp" <<'PY'
import csv, collections, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for d in ("rd", "wr"):
    for r in csv.DictReader(open(f"gpurun_out/xtraffic/{d}/p_counter_collection.csv")):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
inp = int(sys.argv[2]); res = {"_about": "tools/pmc_exact_traffic.sh: sums over all launches of a kernel in one bench step; bytes from the L2's request-size counters", "input_bytes": inp, "kernels": {}}
for k, v in agg.items():
    rd = 128 * v["TCC_EA0_RDREQ_128B_sum"] + 64 * v["TCC_EA0_RDREQ_64B_sum"] + 32 * v["TCC_EA0_RDREQ_32B_sum"]
    wr = 64 * v["TCC_EA0_WRREQ_64B_sum"] + 32 * (v["TCC_EA0_WRREQ_sum"] - v["TCC_EA0_WRREQ_64B_sum"])
    if rd + wr < 0.005 * inp: continue
    res["kernels"][k] = {"launches": max(n[k].values()), "read_bytes": rd, "write_bytes": wr, "read_per_input_byte": round(rd / inp, 4), "write_per_input_byte": round(wr / inp, 4),
                         "requests": {a: v[a] for a in sorted(v)}}
    print(k, "read", round(rd / inp, 3), "write", round(wr / inp, 3), "B/B;  128B/64B/32B requests:", int(v["TCC_EA0_RDREQ_128B_sum"]), int(v["TCC_EA0_RDREQ_64B_sum"]), int(v["TCC_EA0_RDREQ_32B_sum"]), "RDREQ", int(v["TCC_EA0_RDREQ_sum"]), "DRAM", int(v["TCC_EA0_RDREQ_DRAM_sum"]))
json.dump(res, open(sys.argv[1], "w"), indent=1)
PY
rm -f gpurun_out/xtraffic/*/p_kernel_trace.csv
