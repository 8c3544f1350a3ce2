#!/usr/bin/env bash
# Same-box A/B of an ENVIRONMENT switch (and optionally a library): tools/ab_env.sh "<VAR=a>" "<VAR=b>" <workload> [<workload> ...]
# each (setting, workload) twice, interleaved; ms per step of the main line.   gpurun -- 'bash tools/ab_env.sh DE_TAIL_SPLIT=1 DE_TAIL_SPLIT=4 C2 headline'
A=$1; B=$2; shift 2
for rep in 1 2; do
  for wl in "$@"; do
    for s in "$A" "$B"; do
      env $s python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-full-eval-leg --no-turbo-leg --no-complete-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$s $wl', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms_avg'],4), 'complete', round(d['config']['complete_fraction'],3))"
    done
  done
done
