#!/usr/bin/env bash
# Scalar-data-cache behaviour and wave-time split of the eval kernel (gpurun): tools/pmc_smem.sh <tag> [bench args...]
# (environment switches such as DE_MAP_GROUP pass through).  One rocprofv3 --pmc pass per counter set, never combined with a trace domain.
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O; cd /tmp
for set in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES" \
           "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  d=$O/$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $d -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg "$@" > /dev/null 2>$d.err
done
python $R/tools/pmc_parse.py $O $O/summary.txt
