#!/usr/bin/env python3
"""Upper bound of what chunk-level CSE of `unary(feature)` values could buy, measured without building it
(gpurun): the bench population against the same population with every cos(x_f) / exp(x_f) subtree replaced by a plain read of
x_f — the cost structure of a CSE read handler (one cheap dispatch per use) with the prologue that fills the rows left out
(it is priced separately: 5 cos + 5 exp per workgroup and wave).  Both run WITHOUT the early exit (every tree on every sample:
the replacement changes values, hence flags), on all trees and on the trees the real run completes.  -> gpurun_out/cse_bound.json"""
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


def strip(n):
    """unary(feature) -> feature"""
    if n.degree == 0:
        return n.copy()
    if n.degree == 1 and n.children[0].degree == 0 and not n.children[0].constant:
        return n.children[0].copy()
    return de.Node(n.op, *[strip(c) for c in n.children])


def count_uf(n):
    if n.degree == 0:
        return 0
    return (1 if n.degree == 1 and n.children[0].degree == 0 and not n.children[0].constant else 0) + sum(count_uf(c) for c in n.children)


def run(sub, full, turbo=False):
    pop = api.Population(sub, ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=full, turbo=turbo))
    out = torch.empty((len(sub), N), device=dev, dtype=torch.float32)
    ok = torch.empty(len(sub), device=dev, dtype=torch.uint8)
    ctx = pop.ctx
    ms = []
    for i in range(6):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        torch.cuda.synchronize()
        if i >= 2:
            ms.append(ctx.last_kernel_ms())
    f = ok.cpu().numpy().astype(bool)
    pop.close()
    del out
    return float(np.median(ms)), f


res = {"unary_of_feature_per_tree": sum(count_uf(t) for t in trees) / len(trees)}
t_real, flags = run(trees, False)
res["real_early_exit_ms"] = t_real
comp = [t for t, f in zip(trees, flags) if f]
res["unary_of_feature_per_complete_tree"] = sum(count_uf(t) for t in comp) / len(comp)
for tag, sub in (("all", trees), ("complete", comp)):
    a, _ = run(sub, True)
    b, _ = run([strip(t) for t in sub], True)
    res[tag] = dict(trees=len(sub), full_eval_ms=a, with_unary_of_feature_as_reads_ms=b, saving=1 - b / a)
    print(tag, res[tag], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "cse_bound.json"), "w"), indent=1)
print(res)
