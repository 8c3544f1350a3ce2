#!/usr/bin/env bash
# Block order of the threaded eval kernel: chunk-fastest (DE_MAP_GROUP=0) against the chunk slower than a group of G sample tiles per XCD
# (map_block_grouped).  Prints ms per step of the headline, the complete-trees workload, C2 and the fused loss, two rounds, interleaved.
#   gpurun -- 'bash tools/exp_map_group.sh > gpurun_out/map_group.txt'
GROUPS_=${*:-"0 32 128 512 2048 1000000"}
common="--steps 10 --warmup 2 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg"
for rep in 1 2; do
  for g in $GROUPS_; do
    for wl in headline complete C2 loss; do
      DE_MAP_GROUP=$g python bench.py --workload $wl $common 2>/dev/null | python tools/ms.py "group $g $wl"
    done
  done
done
