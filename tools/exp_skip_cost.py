#!/usr/bin/env python3
"""Where the time of the early-exiting headline launch goes (gpurun): complete trees only / + the incomplete ones flagged by
the HOST from the start (a non-finite constant: pure skip-walk cost) / the real population (late-flagged trees are partly
evaluated).  -> gpurun_out/skip_cost.json"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(1000, seed=0xDE02)
N = 10**7
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
lib = api.library()


def run(sub, tag, res):
    pop = api.Population(sub, ops, np.float32, n_features=5)
    out = torch.empty((len(sub), N), device=dev, dtype=torch.float32)
    ok = torch.empty(len(sub), device=dev, dtype=torch.uint8)
    ctx = pop.ctx
    ms = []
    for i in range(7):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        torch.cuda.synchronize()
        if i >= 2:
            ms.append(ctx.last_kernel_ms())
    f = ok.cpu().numpy().astype(bool)
    res[tag] = dict(trees=len(sub), ms=float(np.median(ms)), complete=int(f.sum()))
    print(tag, res[tag], flush=True)
    pop.close()
    del out
    return f


res = {}
flags = run(trees, "real population", res)
comp = [t for t, f in zip(trees, flags) if f]
run(comp, "complete trees only", res)
B = {n: i + 1 for i, n in enumerate(ops.binops)}
hosted = [t if f else de.Node(B["+"], t, de.Node(val=float("inf"))) for t, f in zip(trees, flags)]
run(hosted, "incomplete trees flagged by the host (skipped everywhere)", res)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "skip_cost.json"), "w"), indent=1)
