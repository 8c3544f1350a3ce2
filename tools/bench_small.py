"""Throughput of the eval path when few trees share X (HBM-bound regime): n_trees x N samples."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5 * 10**7
ctx = api.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((N, 5), generator=g, device="cuda").t()
lib = api.library()
for nt in (1, 2, 4, 8, 16, 64):
    trees = de.synth.random_population(nt, seed=0xDE02)
    pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
    out = torch.empty((nt, N), device="cuda")
    ok = torch.empty(nt, device="cuda", dtype=torch.uint8)
    ms = []
    for _ in range(6):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        ms.append(ctx.last_kernel_ms())
    t = np.median(ms[2:])
    hbm = (5 * 4 * N * max(1, -(-nt // pop.plan(N)["trees_per_chunk"])) + 4 * N * nt) / (t * 1e-3) / 1e9
    print(f"n_trees {nt:3d}  kernel {t:8.3f} ms  {nt * N / (t * 1e-3):.3e} tree-samples/s  plan {pop.plan(N)}  approx HBM {hbm:7.1f} GB/s")
    pop.close()
