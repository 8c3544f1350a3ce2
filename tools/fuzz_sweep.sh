#!/usr/bin/env bash
# Command-line fuzz sweep with NEW seeds (round 4: 41, 42; after the chain / sentinel / side-stream changes: 51 — `bash tools/fuzz_sweep.sh 51`), default thresholds and with the exit path (priority tiles + compaction) forced on
# every launch; one summary line per run -> gpurun_out/fuzz_sweep_r4.txt        gpurun --timeout 3000 -- 'bash tools/fuzz_sweep.sh'
SEEDS=${*:-"41 42"}
O=gpurun_out/fuzz_sweep_$(echo $SEEDS | tr ' ' '_').txt; : > $O
run() { # label, env..., -- script seed
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local t0=$(date +%s)
  local out; out=$(env "${envs[@]}" timeout 900 python "$@" 2>&1 | tail -4 | tr '\n' ' ')
  echo "$label $* ($(( $(date +%s) - t0 )) s): $out" | tee -a $O
}
for seed in $SEEDS; do
  run "default" DE_X=1 -- tests/fuzz/fuzz_gpu.py $seed
  run "exit-path-forced" DE_PRIO_MIN_TILES=1 DE_PRIO_MIN_TREES=1 -- tests/fuzz/fuzz_gpu.py $seed
  run "exit-path-forced" DE_PRIO_MIN_TILES=1 DE_PRIO_MIN_TREES=1 -- tests/fuzz/fuzz_hot.py $seed
  run "exit-path-forced" DE_PRIO_MIN_TILES=1 DE_PRIO_MIN_TREES=1 -- tests/fuzz/fuzz_param.py $seed
  run "exit-path-forced" DE_PRIO_MIN_TILES=1 DE_PRIO_MIN_TREES=1 -- tests/fuzz/fuzz_grad.py $seed
  run "default (fused reverse records, shared rows)" DE_X=1 -- tests/fuzz/fuzz_lossgrad.py $seed
  run "reverse unfused" DE_REV_NO_FUSE=1 -- tests/fuzz/fuzz_lossgrad.py $seed
done
