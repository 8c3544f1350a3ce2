#!/usr/bin/env bash
# One GPU-box visit: the -m gpu tests, then the bench lines of every workload -> gpurun_out/<tag>/
#   gpurun --timeout 3000 -- 'bash tools/gpu_check.sh <tag> [pytest args...]'
set -u
TAG=${1:-check}; shift || true
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q -rs "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -25 $O/pytest.log
for wl in headline C2 C3 C4 C5 C5N C5Ng loss lossgrad C5pb; do
  extra=""; [ $wl = headline ] || extra="--no-cpu-baseline"
  timeout 900 python bench.py --workload $wl --steps 10 --warmup 2 $extra > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$wl.json"))
    print("$wl", round(d["ms_per_step"], 3), "ms", "%.3g" % d["value"], "frac", round(d["roofline"]["frac"], 3), "valu", (d["roofline"].get("valu") or {}).get("frac"), "turbo", (d.get("turbo") or {}).get("ms_per_step"))
except Exception as e:
    print("$wl: no line", e)
PY
done
