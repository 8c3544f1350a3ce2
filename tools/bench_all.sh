O=gpurun_out/$1; mkdir -p $O
for wl in C2 C3 C4 C5 C5N C5Ng loss lossgrad C5pb; do
  timeout 900 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$wl.json"))
    print("$wl", round(d["ms_per_step"], 3), "ms", "%.3g" % d["value"], "frac", round(d["roofline"]["frac"], 3), "turbo", (d.get("turbo") or {}).get("ms_per_step"), "full", (d.get("full_evaluation") or {}).get("ms_per_step"), "complete_only", (d.get("complete_only") or {}).get("ms_per_step"))
except Exception as e:
    print("$wl: no line", e)
PY
done
