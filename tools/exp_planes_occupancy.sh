#!/usr/bin/env bash
# Occupancy sweep of the two-plane (default) and the one-plane (variants/libde_hip_tg1.so) eval kernels: DE_EXTRA_LDS_ROWS pads the
# workgroup's LDS, so fewer workgroups fit a CU.  Prints headline and complete_only ms per (library, extra rows).
#   gpurun -- 'bash tools/exp_planes_occupancy.sh > gpurun_out/planes_occupancy.txt'
for lib in default variants/libde_hip_tg1.so; do
  for extra in 0 1 2 3 5 8; do
    if [ $lib = default ]; then unset DE_HIP_LIB; else export DE_HIP_LIB=$PWD/$lib; fi
    DE_EXTRA_LDS_ROWS=$extra python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib extra_rows $extra headline', round(d['ms_per_step'],3), 'complete_only', round(d['complete_only']['ms_per_step'],3))"
  done
done
