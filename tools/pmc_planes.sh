#!/usr/bin/env bash
# Round 5: the SQ counters of the one-plane and the clean two-plane build on the one-feature proxy population (tools/exp_planes_equal_waves.py).
#   gpurun -- 'bash tools/pmc_planes.sh > gpurun_out/r5_pmc_planes.txt'
set -u
R=$PWD; export TMPDIR=/tmp; cd /tmp
A="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_IFETCH"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_TRANS_F32"
for lib in default variants/libde_hip_tg2nl.so; do
  O=$R/gpurun_out/pmc_planes_$(basename $lib .so); mkdir -p $O
  for P in A C; do
    eval CN=\$$P
    (cd $R && DE_HIP_LIB_SEL=$lib rocprofv3 --pmc $CN --output-format csv -d $O/$P -o p -- python tools/exp_planes_equal_waves.py > $O/$P.json 2>/dev/null)
  done
  echo "== $lib: $(grep -h '^{' $O/A.json | tail -1)"
  python - "$O" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "de_eval_threaded_kernel" in r["Kernel_Name"]]
    rows = rows[-12 * 8:]  # the timed population's launches are the last ones (12 launches x 8 counters)
    gmax = max((int(r["Grid_Size"]) for r in rows), default=0)
    for r in rows:
        if int(r["Grid_Size"]) * 2 >= gmax:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} {tot[k] / max(n[k], 1):.4g} per launch ({n[k]})")
PY
done
