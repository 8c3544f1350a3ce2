#!/usr/bin/env bash
# Instruction / scalar-data cache behaviour of the eval kernel (gpurun): tools/pmc_icache.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z_]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_WAIT_INST_[A-Z_]*\|SQ_BUSY_CU_CYCLES\|SQ_WAVES\b" | sort -u | tr '\n' ' ' > $O/avail.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  d=$O/$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $d -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg "$@" > /dev/null 2>$d.err
done
python $R/tools/pmc_parse.py $O $O/summary.txt
cat $O/avail.txt | head -c 3000
