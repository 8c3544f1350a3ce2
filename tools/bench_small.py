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
    out = torch.empty((nt, N), device="cuda")
    ok = torch.empty(nt, device="cuda", dtype=torch.uint8)
    line = f"n_trees {nt:3d}"
    for full in (False, True):  # default (early exit: incomplete trees are skipped) / DE_OPT_FULL_EVAL (every tree on every sample)
        pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx, eval_context=api.EvalContext(full_eval=full))
        ms = []
        for _ in range(6):
            ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
            ms.append(ctx.last_kernel_ms())
        t = np.median(ms[2:])
        if full:
            hbm = (5 * 4 * N * max(1, -(-nt // pop.plan(N)["trees_per_chunk"])) + 4 * N * nt) / (t * 1e-3) / 1e9
            line += f"  full eval {t:7.3f} ms = {hbm:6.0f} GB/s of X + out  plan {pop.plan(N)}"
        else:
            line += f"  kernel {t:7.3f} ms ({int(ok.sum().item())}/{nt} complete)"
        pop.close()
    print(line)
