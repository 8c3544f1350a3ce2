#!/usr/bin/env python3
"""Bound of "structurally zero Jacobian columns"  for config C3, measured with the SHIPPED kernels before anything is
built: a tree that names u of the F features has F - u Jacobian columns that are zero by construction.  A compacting kernel would carry
only the u columns (variant `noghost`: the zero columns written as +0 — the reference's dense update gives some of them the sign -0) or
u + 1 (variant `ghost`: ONE never-seeded column that is broadcast to every unused row — bit-identical to the dense update, NaN of 0 * Inf
included).  A never-seeded column is exactly an unused feature, so both variants can be timed today: split the C3 population by u, re-number
each tree's features to 1..u, and run each group as a population of its own over u (+ 1) features.  The groups run one after the other on
one stream (the built thing would spread them over the side streams like every bucket launch: this is the pessimistic side of the bound).

    gpurun -- 'python tools/exp_grad_compact_bound.py' -> gpurun_out/grad_compact_bound.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

N = int(float(os.environ.get("N", "1e6")))
STEPS = int(os.environ.get("STEPS", "10"))
ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(1000, seed=0xDE02)
Xh = de.synth.random_X(5, N, seed=1, dtype=np.float32)
Xd = torch.from_numpy(np.ascontiguousarray(np.asarray(Xh).T)).cuda()  # [N, 5] row-major = feature-fastest


def X_of(g):
    x = Xd[:, :g].contiguous()  # [N, g] row-major = feature-fastest, ldX = g
    torch.cuda.synchronize()  # (the library runs on the context's own stream)
    return x


def used(t):
    return sorted({n.feature for n in t if n.degree == 0 and not n.constant})


def renumber(t, m):
    c = t.copy()
    for n in c:
        if n.degree == 0 and not n.constant:
            n.feature = m[n.feature]
    return c


import time

ctx = api.Context(0)
lib = api.library()
out = torch.empty((1000, N), device="cuda", dtype=torch.float32)
grad = torch.empty(1000 * 5 * N, device="cuda", dtype=torch.float32)
okb = torch.empty(1000, device="cuda", dtype=torch.uint8)


def call(pop, Xg, width):
    """the bench's C3 step: the C ABI with caller-owned device buffers (no Python per tree)"""
    ctx.check(lib.de_eval_grad(ctx._h, pop._h, Xg.data_ptr(), N, width, None, 0, out.data_ptr(), N, grad.data_ptr(), None, okb.data_ptr()))


def timed(fn, steps=STEPS):
    """wall time per step of a free-running loop (what bench.py reports)"""
    for _ in range(2):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


res = {"N": N, "steps": STEPS}
dense = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
X5 = X_of(5)
call(dense, X5, 5)
ctx.synchronize()
ok_dense = okb.cpu().numpy().astype(bool).copy()
res["dense_ms"] = timed(lambda: call(dense, X5, 5))
res["complete_fraction"] = float(ok_dense.mean())
groups = {}
for i, t in enumerate(trees):
    groups.setdefault(len(used(t)), []).append(i)
res["trees_by_used_features"] = {str(u): len(v) for u, v in sorted(groups.items())}
for variant in ("ghost", "noghost"):
    pops = []
    for u, idx in sorted(groups.items()):
        width = min(5, u + 1) if variant == "ghost" else max(u, 1)
        if variant == "ghost" and u == 4:
            width = 5
        sub = []
        for i in idx:
            us = used(trees[i])
            sub.append(renumber(trees[i], {f: k + 1 for k, f in enumerate(us)}))
        pops.append((u, width, api.Population(sub, ops, np.float32, n_features=width, ctx=ctx), X_of(width), idx))
    per = {}
    flags_equal = True
    n_complete = 0
    for u, width, pop, Xg, idx in pops:
        call(pop, Xg, width)
        ctx.synchronize()
        ok = okb[:len(idx)].cpu().numpy().astype(bool)
        # (features are iid: a re-numbered tree sees other data, its flag may differ; counted, not asserted)
        flags_equal = flags_equal and bool((ok == ok_dense[idx]).all())
        n_complete += int(ok.sum())
        per[f"u={u} width={width} trees={len(idx)}"] = timed(lambda: call(pop, Xg, width))

    def all_groups():
        for u, width, pop, Xg, idx in pops:
            call(pop, Xg, width)

    res[variant] = {"per_group_ms": per, "sum_of_groups_ms": sum(per.values()), "one_step_all_groups_ms": timed(all_groups),
                    "flags_equal_to_dense": flags_equal, "complete_trees": n_complete}
    res[variant]["vs_dense"] = res[variant]["one_step_all_groups_ms"] / res["dense_ms"]
    for p in pops:
        p[2].close()
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "grad_compact_bound.json"), "w"), indent=1)
