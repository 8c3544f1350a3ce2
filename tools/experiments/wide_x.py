#!/usr/bin/env python3
"""Very wide feature matrices: 1000 trees x 10^6 samples at F = 30 ... 120 (which kernel runs, how long)."""
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
N = 10**6
out = torch.empty((1000, N), device="cuda")
ok = torch.empty(1000, device="cuda", dtype=torch.uint8)
for F in (30, 36, 40, 60, 100, 120):
    trees = de.synth.random_population(1000, seed=0xDE02 + F, nfeatures=F)
    g = torch.Generator(device="cuda")
    g.manual_seed(F)
    X = (torch.randn((N, F), generator=g, device="cuda") * 1.2).t()
    pop = api.Population(trees, ops, np.float32, n_features=F, ctx=ctx)
    for _ in range(5):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
    ctx.synchronize()
    ctx.timing_ring(10)
    for _ in range(10):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
    ctx.synchronize()
    ms = np.mean([t for t in ctx.timing_read() if t is not None])
    ctx.timing_ring(0)
    print(f"F {F}: {ms:.3f} ms, kernel {ctx.last_kernel_name()}, waves {pop.meta(0)['waves']}, complete {int(ok.sum())}, checksum {float(out[ok.bool()].double().sum()):.10e}", flush=True)
    pop.close()
    del X
