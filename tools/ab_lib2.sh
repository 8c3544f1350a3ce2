#!/usr/bin/env bash
# Same-box A/B of library builds (gpurun): tools/ab_lib2.sh <lib.so> [<lib.so> ...] — the default build and every variant run the headline,
# the complete-trees workload, C2, the fused loss and the turbo headline, three rounds, interleaved; ms per step.
common="--steps 10 --warmup 2 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg"
for rep in 1 2 3; do
  for lib in default "$@"; do
    if [ $lib = default ]; then unset DE_HIP_LIB; else export DE_HIP_LIB=$PWD/$lib; fi
    for wl in headline complete C2 loss; do
      python bench.py --workload $wl $common 2>/dev/null | python tools/ms.py "$lib $wl"
    done
    python bench.py --workload headline --turbo $common 2>/dev/null | python tools/ms.py "$lib turbo"
  done
done
