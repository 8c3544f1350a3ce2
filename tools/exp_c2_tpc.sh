#!/usr/bin/env bash
# C2 (1000 trees x 10^6 samples) over the trees-per-chunk cap of the eval launch (DE_EVAL_TPC, default 63): shorter workgroups = finer tail
# of a launch that fills the chip only ~4 times, against one more staging of the X tile per chunk.   gpurun -- 'bash tools/exp_c2_tpc.sh'
O=gpurun_out/c2_tpc; mkdir -p $O
for tpc in ${TPCS:-63 48 40 32 24 16}; do
  DE_EVAL_TPC=$tpc timeout 600 python bench.py --workload ${WL:-C2} --steps 100 --warmup 5 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg > $O/tpc_$tpc.json 2> $O/tpc_$tpc.err
  python - $tpc $O/tpc_$tpc.json <<'P'
import json, sys
d = json.load(open(sys.argv[2]))
print("tpc", sys.argv[1], "ms %.4f" % d["ms_per_step"], "complete_only %.4f" % d.get("complete_only", {}).get("ms_per_step", float("nan")),
      "declared %.4f" % d.get("dataset_declared", {}).get("ms_per_step", float("nan")))
P
done
