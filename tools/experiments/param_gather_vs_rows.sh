# parameters as staged LDS rows (default) against the gather handlers (DE_NO_PARAM_ROWS=1): C5 (16 classes), C5N / C5Ng (per-sample parameters)
for wl in C5 C5N C5Ng; do
  for v in 0 1; do
    DE_NO_PARAM_ROWS=$v timeout 120 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('DE_NO_PARAM_ROWS=$v', '$wl', round(d['ms_per_step'],3))"
  done
done
