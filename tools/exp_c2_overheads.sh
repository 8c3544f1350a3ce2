for env in "" "DE_COMPACT=0" "DE_NO_PRIO_TILES=1" "DE_PRIO_PROBE_TPC=16" "DE_PRIO_PROBE_TPC=4" "DE_EVAL_TPC=32"; do
  r=$(env $env python bench.py --workload C2 --steps 30 --warmup 3 --no-cpu-baseline --no-turbo-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'declared', round(d['dataset_declared']['ms_per_step'],4), 'full', round(d['full_evaluation']['ms_per_step'],4), 'complete_only', round(d['complete_only']['ms_per_step'],4))")
  echo "C2 ${env:-default}: $r"
done
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2prof -o k -- python /root/repo/bench.py --workload C2 --steps 20 --warmup 2 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg > /dev/null 2>&1; f=$(find /tmp/c2prof -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-160
