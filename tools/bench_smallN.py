"""Many trees, few samples (the usual symbolic-regression shape): n_trees x N."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0)
lib = api.library()
for nt, N in ((10000, 1000), (10000, 10000), (1000, 1000), (1000, 100), (100000, 1000)):
    trees = de.synth.random_population(nt, seed=0xDE02)
    nodes = sum(de.count_nodes(t) for t in trees)
    t0 = time.perf_counter()
    pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
    t_create = time.perf_counter() - t0
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn((N, 5), generator=g, device="cuda").t()
    out = torch.empty((nt, N), device="cuda")
    ok = torch.empty(nt, device="cuda", dtype=torch.uint8)
    ms = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        ms.append((ctx.last_kernel_ms(), wall))
    k, w = np.median([m[0] for m in ms[2:]]), np.median([m[1] for m in ms[2:]])
    print(f"n_trees {nt:6d} N {N:6d}  kernel {k:7.3f} ms  wall {w:7.3f} ms  {nodes * N / (k * 1e-3):.3e} node-evals/s (kernel)  create {t_create:.2f} s  plan {pop.plan(N)}")
    pop.close()
