#!/usr/bin/env python
"""gpurun_out/traffic/traffic.json (tools/pmc_traffic.sh) -> profiles-style summary with the MI355X guide's corrections.
usage: tools/pmc_traffic_json.py <input_bytes> <out.json>"""
import json
import sys

inp = int(sys.argv[1])
raw = json.load(open("gpurun_out/traffic/traffic.json"))
out = {"_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) around "
                 "`python bench.py --gib 8 --steps 1 --warmup 0 --no-cpu-baseline`. Counter unit KiB. Per "
                 "/opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half the bytes of wide "
                 "coalesced streaming reads: hbm_read = 2*FETCH_SIZE*1024 (corrected); WRITE_SIZE is taken 1:1. The byte "
                 "fields are per-launch averages; `corrected_per_input_byte` = traffic of ALL launches of the kernel in the step / "
                 "input_bytes (every kernel covers the input once per step in total, however many launches it takes).",
       "input_bytes": inp, "kernels": {}}
for k, v in raw.items():
    n = v.get("launches", 1)
    f = v.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024
    w = v.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    out["kernels"][k] = {"launches": n, "fetch_raw_bytes": f, "write_raw_bytes": w, "hbm_bytes_corrected": 2 * f + w,
                         "corrected_per_input_byte": round((2 * f + w) * n / inp, 4), "raw_per_input_byte": round((f + w) * n / inp, 4)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out["kernels"].items():
    if v["corrected_per_input_byte"] > 0.01:
        print(k, v["corrected_per_input_byte"], "raw", v["raw_per_input_byte"])
