#!/usr/bin/env bash
# Instruction-fetch / scalar-data cache behaviour of the eval kernel (gpurun): tools/pmc_icache.sh <tag> [bench args...]
# One rocprofv3 --pmc pass per counter set (never combined with a trace domain); a set with an unknown counter just
# leaves its .err behind and the summary lists the others.
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "\bSQC\?_[A-Z_0-9]*\b" | sort -u | tr '\n' ' ' > $O/avail.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_WAIT_IFETCH" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES" \
           "SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_INPUT_VALID_READYB" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  d=$O/$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $d -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg "$@" > /dev/null 2>$d.err
done
python $R/tools/pmc_parse.py $O $O/summary.txt
head -c 6000 $O/avail.txt
