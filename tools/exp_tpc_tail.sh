common="--steps 20 --warmup 3 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg"
for rep in 1 2; do
for tpc in 63 48 32 24 16; do
  DE_EVAL_TPC=$tpc python bench.py --workload C2 $common 2>/dev/null | python tools/ms.py "tpc $tpc C2"
done
for tpc in 63 32; do
  DE_EVAL_TPC=$tpc python bench.py --workload headline $common 2>/dev/null | python tools/ms.py "tpc $tpc headline"
done
done
