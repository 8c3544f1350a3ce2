# shared leaf rows in the forward-dual gradient kernels: DE_GRAD_SHARE = 0 (one copy of the leaf rows per wave) | 1 (one per workgroup) | unset (the rule: >= 8 leaf rows)
for wl in C5 C5Ng C3 lossgrad; do
  for v in 0 1 ""; do
    DE_GRAD_SHARE=$v; [ -z "$v" ] && unset DE_GRAD_SHARE || export DE_GRAD_SHARE
    timeout 300 python bench.py --workload $wl --steps 20 --warmup 30 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('DE_GRAD_SHARE=${v:-unset}', '$wl', round(d['ms_per_step'],3))"
  done
done
