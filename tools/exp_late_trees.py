#!/usr/bin/env python3
"""The incomplete trees the probe launch of the priority tiles does NOT flag (headline population, 10^7 samples): on how many of 256
sample blocks does each of them fail?  A tree that fails on a tenth of the blocks would fall to a handful of random probe tiles; one that
fails on a single block cannot be found before that block runs.     gpurun -- 'python tools/exp_late_trees.py' -> gpurun_out/late_trees.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

ops = de.synth.BENCH_OPERATORS
N = 10**7
trees = de.synth.random_population(1000, seed=0xDE0C)
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
lib = api.library()
# 1. flags of the whole job, and the flags after the probe launch alone: a launch over the priority tiles only = DE_PRIO forced, then read
#    which trees were still live (the library reports the count; the set = trees that are incomplete but not flagged by a probe-only run)
pop = api.Population(trees, ops, np.float32, n_features=5)
out, ok = pop.eval(X)
torch.cuda.synchronize()
ok = ok.bool().cpu().numpy()
live = pop.last_live_trees()
del out
incomplete = [t for t in range(1000) if not ok[t]]
# 2. every incomplete tree on 256 blocks of samples, each block a call of its own (flags per block)
B = 256
blk = (N + B - 1) // B
sub = api.Population([trees[t] for t in incomplete], ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=True))
fails = np.zeros((len(incomplete), B), dtype=bool)
Xc = X.t().contiguous()  # [N, 5] feature-fastest
for b in range(B):
    a0, a1 = b * blk, min(N, (b + 1) * blk)
    if a0 >= a1:
        continue
    xb = Xc[a0:a1].t()
    o, k = sub.eval(xb)
    fails[:, b] = ~k.bool().cpu().numpy()
    del o
share = fails.mean(axis=1)
order = np.argsort(share)
hist = {"1 block": int((fails.sum(axis=1) == 1).sum()), "2-3 blocks": int(((fails.sum(axis=1) >= 2) & (fails.sum(axis=1) <= 3)).sum()),
        "4-25 blocks (< 10 %)": int(((fails.sum(axis=1) >= 4) & (fails.sum(axis=1) < 26)).sum()), ">= 10 % of the blocks": int((share >= 0.1).sum()),
        "every block": int((fails.sum(axis=1) == B).sum())}
res = dict(n_incomplete=len(incomplete), live_after_probe=int(live), complete=int(ok.sum()), late=int(live) - int(ok.sum()), blocks=B,
           incomplete_trees_by_failing_blocks=hist, fewest_failing_blocks=[int(fails[i].sum()) for i in order[:60]])
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "late_trees.json"), "w"), indent=1)
