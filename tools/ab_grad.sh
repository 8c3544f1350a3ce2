#!/usr/bin/env bash
# Same-box A/B of library builds on the gradient workloads (gpurun): tools/ab_grad.sh <lib.so> [...]; C3, C5, C5Ng, lossgrad, C5pb; ms per step
common="--steps 10 --warmup 2 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg"
for rep in 1 2; do
  for lib in default "$@"; do
    if [ $lib = default ]; then unset DE_HIP_LIB; else export DE_HIP_LIB=$PWD/$lib; fi
    for wl in C3 C5 C5Ng lossgrad C5pb; do
      python bench.py --workload $wl $common 2>/dev/null | python tools/ms.py "$lib $wl"
    done
  done
done
