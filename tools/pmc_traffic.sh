#!/usr/bin/env bash
# HBM write/fetch per step of every kernel of a bench workload (run through gpurun): tools/pmc_traffic.sh <workload> [env...]
# Traffic far above the algorithmic bytes = wasted re-reads or hidden writes (how the scratch traffic of DESIGN §4.2 was found).
W=$1; R=$PWD; export TMPDIR=/tmp; cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pt_x
    timeout 280 rocprofv3 --pmc $c --output-format csv -d /tmp/pt_x -o w -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    f=$(find /tmp/pt_x -name "*counter_collection.csv" 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" "$W" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-60:]
    tot[(r["Counter_Name"], k)] += float(r["Counter_Value"]); n[(r["Counter_Name"], k)] += 1
for (c, k), v in sorted(tot.items(), key=lambda kv: -kv[1])[:6]:
    print(f"{sys.argv[2]:9s} {c:11s} {v * 1024 / 3 / 1e9:9.3f} GB/step  {n[(c, k)]:4d} launches  {k}")
PY
done
