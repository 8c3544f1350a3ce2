#!/usr/bin/env bash
# The bench lines of a round (run on the GPU box through gpurun):  bash tools/bench_lines.sh <tag>  -> gpurun_out/<tag>/bench_<workload>.json
# `headline` = the driver's command (python bench.py: every BASELINE config under `configs`); the others one workload each at steady clocks
# (50 warm-up steps: a 1 ms workload is not at steady clocks after 5 — DESIGN.md §9).
set -u
TAG=${1:-lines}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err || echo "bench headline failed rc=$?"
for wl in complete C2 C3 C4 C5 C5N C5Ng loss lossgrad C5pb turbo; do
  args="--workload $wl"; [ $wl = turbo ] && args="--workload headline --turbo"
  timeout 900 python bench.py $args --steps 20 --warmup 50 --no-cpu-baseline --no-configs > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed rc=$?"
done
python - <<PY
import json
for wl in "headline complete C2 C3 C4 C5 C5N C5Ng loss lossgrad C5pb turbo".split():
    try:
        d = json.loads(open("$O/bench_%s.json" % wl).readline())
        r = d["roofline"]
        print(wl, round(d["ms_per_step"], 3), "ms", "%.3g" % d["value"], "frac", round(r["frac"], 3), "traffic", r.get("traffic"), "kernel", r.get("kernel"))
        for k, v in (d.get("configs") or {}).items():
            print("   ", k, round(v["ms_per_step"], 3), "ms frac", round(v["roofline"]["frac"], 3), "traffic", v["roofline"].get("traffic"))
    except Exception as e:
        print(wl, "no line", e)
PY
