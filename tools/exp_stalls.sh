#!/usr/bin/env bash
# Where does the time beyond the VALU floor go?  A/B runs of the headline workload (gpurun): output store removed, fewer /
# more workgroups per CU (padded / shrunk LDS: negative rows give wrong values, timing only), exact and turbo.
for env in "" "DE_DEBUG_NO_STORE=1" "DE_EXTRA_LDS_ROWS=-2" "DE_EXTRA_LDS_ROWS=-1" "DE_EXTRA_LDS_ROWS=1" "DE_EXTRA_LDS_ROWS=3" "DE_EXTRA_LDS_ROWS=7"; do
  for t in "" "--turbo"; do
    r=$(env $env python bench.py --workload headline --steps 10 --warmup 2 --no-cpu-baseline --no-turbo-leg $t | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
    echo "headline ${t:-exact} ${env:-default}: $r ms"
  done
done
