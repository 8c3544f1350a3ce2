#!/usr/bin/env python3
"""NOTE: needs the experimental launcher of round 3 (capped grid: DE_GRID_LIFE / DE_GRID_STATIC), which was measured and removed
(profiles/r3_launch_order.json); kept as the record of how those numbers were taken.
Grid / work-distribution scan of the threaded eval kernel in ONE process (gpurun): the launcher reads DE_GRID_* at every launch.
-> gpurun_out/walk.json"""
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
lib = api.library()
KEYS = ("DE_GRID_LIFE", "DE_GRID_STATIC")


def timed(pop, X, N, out, ok, env, reps=5):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ctx = pop.ctx
    ms = []
    for i in range(reps + 1):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        torch.cuda.synchronize()
        if i >= 1:
            ms.append(ctx.last_kernel_ms())
    return float(np.median(ms))


def scan(tag, sub, N, res):
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
    pop = api.Population(sub, ops, np.float32, n_features=5)
    out = torch.empty((len(sub), N), device=dev, dtype=torch.float32)
    ok = torch.empty(len(sub), device=dev, dtype=torch.uint8)
    r = {}
    timed(pop, X, N, out, ok, {"DE_GRID_LIFE": 1})
    r["one workgroup per pair"] = [timed(pop, X, N, out, ok, {"DE_GRID_LIFE": 1}) for _ in range(3)]
    r["counters, life"] = {L: timed(pop, X, N, out, ok, {"DE_GRID_LIFE": L}) for L in (2, 4, 8, 16)}
    r["static walk, workgroups"] = {G: timed(pop, X, N, out, ok, {"DE_GRID_STATIC": G}) for G in range(16008, 200000, 2048)}
    r["one workgroup per pair (again)"] = timed(pop, X, N, out, ok, {"DE_GRID_LIFE": 1})
    res[tag] = r
    sw = r["static walk, workgroups"]
    best = min(sw, key=sw.get)
    print(tag, "1:1", r["one workgroup per pair"], "counters", r["counters, life"], "static best", best, sw[best], flush=True)
    print("  static:", " ".join(f"{G//1000}k:{v:.2f}" for G, v in sw.items()), flush=True)
    pop.close()
    del out, X
    torch.cuda.empty_cache()
    return ok.cpu().numpy().astype(bool)


res = {}
flags = scan("real population, N = 1e7", trees, 10**7, res)
comp = [t for t, f in zip(trees, flags) if f]
scan("complete trees only, N = 1e7", comp, 10**7, res)
scan("real population, N = 5e6", trees, 5 * 10**6, res)
scan("first 500 trees, N = 1e7", trees[:500], 10**7, res)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "walk.json"), "w"), indent=1)
