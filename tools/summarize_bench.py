#!/usr/bin/env python3
"""One line per bench JSON of a directory: python tools/summarize_bench.py gpurun_out/<dir>  (also prints the DESIGN.md §9 table rows)."""
import glob
import json
import os
import sys

d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(os.path.basename(f), "no line:", e)
        continue
    r = j["roofline"]
    vm = r.get("valu_measured") or {}
    va = r.get("valu") or {}
    row = dict(wl=j["config"]["workload_key"], ms=round(j["ms_per_step"], 3), value="%.3g" % j["value"], frac=round(r["frac"], 3),
               traffic_over_alg=(round(r["traffic"] / r["algorithmic_bytes_per_launch"], 2) if r.get("traffic") else None),
               valu_frac=va.get("frac") and round(va["frac"], 2), busy_est=vm.get("busy_est") and round(vm["busy_est"], 2),
               complete=round(j["config"]["complete_fraction"], 3))
    for leg in ("turbo", "full_evaluation", "complete_only", "dataset_declared"):
        if j.get(leg):
            row[leg] = round(j[leg]["ms_per_step"], 3)
    if j.get("complete_only"):
        row["complete_only_frac"] = round(j["complete_only"]["roofline"]["frac"], 3)
    if j.get("value_executed"):
        row["value_executed"] = "%.3g" % j["value_executed"]
    if j.get("cpu_baseline"):
        row["cpu_all_cores"] = "%.3g" % j["cpu_baseline"]["value"]
        row["cpu_1_thread"] = "%.3g" % j["cpu_baseline"]["single_thread"]["value"]
        row["cores"] = j["cpu_baseline"]["cores"]
    print(json.dumps(row))
