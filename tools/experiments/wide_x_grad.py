#!/usr/bin/env python3
"""Constant-mode Jacobians (de_eval_grad, variable = false) over wide feature matrices: 1000 trees x 10^5 samples at F = 5 ... 120."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402

ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0)
lib = api.library()
N = 10**5
for F in (5, 20, 40, 60, 120):
    trees = de.synth.random_population(1000, seed=0xDE02 + F, nfeatures=F)
    g = torch.Generator(device="cuda")
    g.manual_seed(F)
    X = (torch.randn((N, F), generator=g, device="cuda") * 1.2).t()
    pop = api.Population(trees, ops, np.float32, n_features=F, ctx=ctx)
    ng = np.array([pop.n_grad(t, 1) for t in range(len(trees))], dtype=np.int64)
    goffs = np.zeros(len(trees), dtype=np.int64)
    np.cumsum(ng[:-1] * N, out=goffs[1:])
    grad = torch.empty(max(int((ng * N).sum()), 1), device="cuda")
    ok = torch.empty(1000, device="cuda", dtype=torch.uint8)
    res = {}
    for what in ("eval", "grad"):
        out = torch.empty((1000, N), device="cuda")

        def step():
            if what == "eval":
                ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
            else:
                ctx.check(lib.de_eval_grad(ctx._h, pop._h, X.data_ptr(), N, F, None, 1, None, N, grad.data_ptr(), goffs.ctypes.data, ok.data_ptr()))
        for _ in range(5):
            step()
        ctx.synchronize()
        ctx.timing_ring(10)
        for _ in range(10):
            step()
        ctx.synchronize()
        res[what] = float(np.mean([t for t in ctx.timing_read() if t is not None]))
        ctx.timing_ring(0)
    print(f"F {F}: eval {res['eval']:.3f} ms, constant-mode Jacobian {res['grad']:.3f} ms ({int(ng.sum())} rows), kernel {ctx.last_kernel_name()}", flush=True)
    pop.close()
