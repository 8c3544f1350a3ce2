# same-box A/B of the fused flag initialisation (round 6): the library before the change (csrc/libde_hip_prechain.so, built from the parent commit)
# against the shipped one, C2 and the 125-tree shard shape, alternating
P=$PWD/dynamicexpressions.jl_amd/csrc
for rep in 1 2 3; do
  for lib in libde_hip_prechain.so libde_hip.so; do
    DE_HIP_LIB=$P/$lib timeout 120 python bench.py --workload C2 --steps 40 --warmup 5 --no-cpu-baseline --no-turbo-leg --no-full-eval-leg --no-complete-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', 'C2', round(d['ms_per_step'],4))"
  done
done
