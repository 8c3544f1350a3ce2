#!/usr/bin/env python3
"""C2 (1000 trees x 1e6): where does the launch proper lose against its complete trees alone?  Times (a) the headline population, (b) its
complete trees only (443), (c) complete + the trees the probe launch misses (live after the probe), each with the dataset declared."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10**6
dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
lib = api.library(); ctx = api.Context(0)
ctx.declare_dataset(X)
trees = de.synth.random_population(1000, seed=0xDE02)

def run(tr, steps=30, warmup=3):
    pop = api.Population(tr, ops, np.float32, n_features=5, ctx=ctx)
    out = torch.empty((len(tr), N), device=dev, dtype=torch.float32); ok = torch.empty(len(tr), device=dev, dtype=torch.uint8)
    def step(): ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
    for _ in range(warmup): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / steps
    live = pop.last_live_trees(); okh = ok.cpu().numpy().astype(bool); pop.close()
    return ms, live, okh

ms_all, live, okh = run(trees)
comp = [t for t, k in zip(trees, okh) if k]
ms_comp, live_c, _ = run(comp)
print(json.dumps(dict(N=N, all=dict(ms=ms_all, live_after_probe=live, complete=int(okh.sum())), complete_only=dict(ms=ms_comp, n=len(comp), live_after_probe=live_c))))
