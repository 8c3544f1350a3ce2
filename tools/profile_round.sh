#!/usr/bin/env bash
# Per-round profile collection (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag> [workload ...]          (default: every workload of tools/gpu_check.sh + the turbo headline)
# -> gpurun_out/profiles_<tag>/<workload>/{stats,pmc_fetch,pmc_write}: rocprofv3 --kernel-trace --stats of `python bench.py
#    --workload W` and FETCH_SIZE / WRITE_SIZE PMC passes — each counter in its own run, never combined with a trace domain
#    (the pool refuses that).  tools/make_profiles.py condenses the result into profiles/.
set -u
TAG=$1; shift
WLS=${*:-"headline turbo complete C2 C3 C4 C5 C5N C5Ng loss lossgrad C5pb"}
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O; cd /tmp
for wl in $WLS; do
  args="--workload $wl"; [ $wl = turbo ] && args="--workload headline --turbo"
  common="--no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg --no-configs"
  mkdir -p $O/$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl/stats -o k -- python $R/bench.py $args --steps 5 --warmup 1 $common > $O/$wl/bench_under_rocprof.json 2>$O/$wl/stats.log
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/$wl/pmc_$c -o k -- python $R/bench.py $args --steps 2 --warmup 1 $common > /dev/null 2>$O/$wl/pmc_$c.log
  done
  # executed instruction mix of every kernel of the step (the VALU ceiling of bench.py's `roofline.valu_measured`)
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/$wl/pmc_SQ -o k -- python $R/bench.py $args --steps 2 --warmup 1 $common > /dev/null 2>$O/$wl/pmc_SQ.log
  echo "$wl: $(find $O/$wl -name '*.csv' | wc -l) csv files"
done
