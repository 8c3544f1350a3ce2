#!/usr/bin/env bash
# Collect SQ PMC passes for bench.py (run on the GPU box through gpurun).
# usage: tools/pmc.sh <tag> [bench args...]      -> gpurun_out/pmc_<tag>/{A,B,C}/...
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc_$TAG; cd /tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"
C="SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_CYCLES GRBM_GUI_ACTIVE"
for P in A B C; do
  eval CN=\$$P
  rocprofv3 --pmc $CN --output-format csv -d $R/gpurun_out/pmc_$TAG/$P -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg "$@" > /dev/null 2>&1
done
python $R/tools/pmc_parse.py $R/gpurun_out/pmc_$TAG $R/gpurun_out/pmc_$TAG/summary.txt
