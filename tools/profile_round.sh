#!/usr/bin/env bash
# Per-round profile collection (run on the GPU box through gpurun):
#   tools/profile_round.sh <round-tag>
# -> gpurun_out/profiles_<tag>/ : rocprofv3 --kernel-trace --stats summary of `python bench.py`
#    (headline workload) and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the pool requires).
set -u
TAG=$1
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o eval -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-turbo-leg > $O/bench_under_rocprof.json 2>$O/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o eval -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o eval -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_C3 -o grad -- python $R/bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline --no-turbo-leg > $O/bench_C3_under_rocprof.json 2>>$O/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_C3 -o grad -- python $R/bench.py --workload C3 --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_C3 -o grad -- python $R/bench.py --workload C3 --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg > /dev/null 2>&1
# the turbo variant of the headline (EvalContext(turbo=true)): kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_turbo -o eval -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --turbo > $O/bench_turbo_under_rocprof.json 2>>$O/stats.log
find $O -name "*.csv" | head -20
