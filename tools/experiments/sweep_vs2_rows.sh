mkdir -p gpurun_out/r6e
for r in 15 18 20 22 24 28; do
  for wl in C5 C5Ng; do
    DE_GRAD_VS2_ROWS=$r timeout 120 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('VS2_ROWS=$r', '$wl', round(d['ms_per_step'],3))"
  done
done
