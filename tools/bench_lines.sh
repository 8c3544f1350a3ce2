#!/usr/bin/env bash
# Every workload's bench line of the current build -> gpurun_out/<tag>/bench_<workload>.json (no tests: tools/gpu_check.sh runs those too)
TAG=${1:-lines}; O=gpurun_out/$TAG; mkdir -p $O
for wl in headline C2 C3 C4 C5 C5N C5Ng loss lossgrad C5pb complete; do
  extra=""; [ $wl = headline ] || extra="--no-cpu-baseline"
  timeout 900 python bench.py --workload $wl --steps 10 --warmup 2 $extra > $O/bench_$wl.json 2> $O/bench_$wl.err || echo "bench $wl failed rc=$?"
  python tools/ms.py $wl < $O/bench_$wl.json
done
timeout 900 python bench.py --workload headline --turbo --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_turbo.json 2> $O/bench_turbo.err; python tools/ms.py turbo < $O/bench_turbo.json
