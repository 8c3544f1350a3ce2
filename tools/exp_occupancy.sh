#!/usr/bin/env bash
# Occupancy sensitivity of the eval kernel under the early exit: pad the LDS allocation (DE_EXTRA_LDS_ROWS, 1040 B per row) and time the
# headline.  gpurun -- 'bash tools/exp_occupancy.sh > gpurun_out/occupancy.txt'
for rows in 0 2 4 8 12 20 0; do
  DE_EXTRA_LDS_ROWS=$rows python bench.py --workload headline --steps 10 --warmup 2 --no-cpu-baseline --no-full-eval-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('extra_rows $rows', 'ms', round(d['ms_per_step'],3), 'turbo', round(d['turbo']['ms_per_step'],3), 'lds', d['config'].get('lds_bytes'))"
done
