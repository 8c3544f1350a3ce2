#!/usr/bin/env bash
# Round 5: where do the waves of the eval kernel wait?  Two SQ counter passes over `bench.py --workload <W>` (default: complete), one
# rocprofv3 run per pass, no trace domain beside the counters.   gpurun -- 'bash tools/pmc_stalls.sh complete > gpurun_out/r5_pmc_stalls.txt'
set -u
WL=${1:-complete}
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/pmc_stalls_$WL; mkdir -p $O; cd /tmp
A="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
B="SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_CYCLES"
for P in A B C; do
  eval CN=\$$P
  rocprofv3 --pmc $CN --output-format csv -d $O/$P -o p -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg > /dev/null 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "de_eval_threaded_kernel" in r["Kernel_Name"]]
    gmax = max((int(r["Grid_Size"]) for r in rows), default=0)
    for r in rows:
        if int(r["Grid_Size"]) * 2 >= gmax:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:34s} {tot[k] / max(n[k], 1):.4g} per launch ({n[k]} launches)")
PY
