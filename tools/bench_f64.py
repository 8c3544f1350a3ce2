"""Float64 rates (not a headline): 1000 trees x 10^6 rows, eval and gradient."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0); lib = api.library()
nt, N = 1000, 10**6
trees = de.synth.random_population(nt, seed=0xDE02, dtype=np.float64)
nodes = sum(de.count_nodes(t) for t in trees)
pop = api.Population(trees, ops, np.float64, n_features=5, ctx=ctx)
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float64).t()
out = torch.empty((nt, N), device="cuda", dtype=torch.float64); ok = torch.empty(nt, device="cuda", dtype=torch.uint8)
grad = torch.empty(nt * 5 * N, device="cuda", dtype=torch.float64)
for name, fn in (("eval", lambda: lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr())),
                 ("grad variable", lambda: lib.de_eval_grad(ctx._h, pop._h, X.data_ptr(), N, 5, None, 0, out.data_ptr(), N, grad.data_ptr(), None, ok.data_ptr()))):
    ms = []
    for _ in range(5):
        ctx.check(fn()); ms.append(ctx.last_kernel_ms())
    t = float(np.median(ms[1:]))
    print(f"f64 {name}: {t:.2f} ms  {nodes * N / (t * 1e-3):.3e} node-evals/s  kernel {ctx.last_kernel_name()}")
