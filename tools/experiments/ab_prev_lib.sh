# same-box A/B of the gradient workloads: the library of the commit before the shared-leaf-row kernels (csrc/libde_hip_prev.so, built from a
# worktree of 30bde59 with DE_OUT_LIB) against the current one; three alternating pairs
for rep in 1 2 3; do
  for wl in lossgrad C3 C5; do
    for lib in prev cur; do
      L=$PWD/dynamicexpressions.jl_amd/csrc/libde_hip.so; [ $lib = prev ] && L=$PWD/dynamicexpressions.jl_amd/csrc/libde_hip_prev.so
      DE_HIP_LIB=$L timeout 300 python bench.py --workload $wl --steps 20 --warmup 30 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', '$wl', round(d['ms_per_step'],3))"
    done
  done
done
