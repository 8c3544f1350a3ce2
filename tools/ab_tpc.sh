for tpc in 64 128 256 32; do
  DE_EVAL_TPC=$tpc python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-eval-leg | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('TPC $tpc headline',round(d['ms_per_step'],3),'turbo',round(d['turbo']['ms_per_step'],3))"
  DE_EVAL_TPC=$tpc python bench.py --workload C2 --steps 10 --warmup 2 --no-cpu-baseline --no-full-eval-leg | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('TPC $tpc C2',round(d['ms_per_step'],3),'turbo',round(d['turbo']['ms_per_step'],3))"
done
python -m pytest tests/test_gpu_early_exit.py tests/test_gpu_ulp_f64.py tests/test_gpu_eval.py -m gpu -q -x 2>&1 | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/ulp_f64.json'))
print({k:v for k,v in d.items() if 'tan' in k or k=='^'})
PY
