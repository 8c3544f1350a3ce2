#!/usr/bin/env python3
"""GraphNode populations over a wide feature matrix (persistent shared rows: many slot rows): which W the rule picks, and the time of each W."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402
from test_lowering import random_graph  # noqa: E402

F, N = 20, 10**6
rng = de.synth.Xoshiro256ss(4711)
ops = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("cos", "exp", "safe_log", "square"))
trees = [random_graph(rng, ops, 6 + i % 24, F, 1 + i % 4, np.float32) for i in range(1000)]
ctx = api.Context(0)
lib = api.library()
g = torch.Generator(device="cuda")
g.manual_seed(5)
X = (torch.randn((N, F), generator=g, device="cuda") * 1.2).t()
out = torch.empty((len(trees), N), device="cuda")
ok = torch.empty(len(trees), device="cuda", dtype=torch.uint8)
for w in ("1", "2", "4", None):
    if w:
        os.environ["DE_EVAL_WAVES"] = w
    else:
        os.environ.pop("DE_EVAL_WAVES", None)
    pop = api.Population(trees, ops, np.float32, n_features=F, ctx=ctx)
    for _ in range(20):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
    ctx.synchronize()
    ctx.timing_ring(20)
    for _ in range(20):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
    ctx.synchronize()
    ms = np.mean([t for t in ctx.timing_read() if t is not None])
    ctx.timing_ring(0)
    print(f"DE_EVAL_WAVES={w}: slots {max(pop.meta(t)['n_slots'] for t in range(len(trees)))}, waves {pop.meta(0)['waves']}, {ms:.3f} ms, complete {int(ok.sum())}", flush=True)
    pop.close()
