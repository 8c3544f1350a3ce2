#!/usr/bin/env bash
# A/B of library builds on ONE box (gpurun): tools/ab_lib.sh <lib.so> [<lib.so> ...]; each runs the headline and C2 twice, interleaved
for rep in 1 2; do
  for lib in default "$@"; do
    for wl in headline C2; do
      if [ $lib = default ]; then unset DE_HIP_LIB; else export DE_HIP_LIB=$PWD/$lib; fi
      python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-full-eval-leg | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib $wl', round(d['ms_per_step'],3), 'turbo', round(d['turbo']['ms_per_step'],3), 'complete', round(d['config']['complete_fraction'],3))"
    done
  done
done
