#!/usr/bin/env python3
"""Does the eval kernel gain from MORE resident waves than the 5.25 per SIMD its LDS allows today?  A clean test: the complete trees of the
bench generator that need <= 1 spill slot, timed (a) as a population of their own — 6 LDS rows per workgroup: 6 waves per SIMD (the register
limit) — and (b) with ONE two-slot tree added — 7 rows: 5.25 waves, the same work + 1/n.  Also reports the share of trees per slot class
(what a launch split by slot need could use).   gpurun -- 'python tools/exp_slot_classes.py'"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

N = 10**7
dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
lib = api.library()
ctx = api.Context(0)
cand = de.synth.random_population(2400, seed=0xDE0C)


def slots_of(tree):
    w = np.zeros(4, dtype=np.uint32)
    p = api.Population([tree], ops, np.float32, n_features=5, ctx=ctx)
    lib.de_program_dump(p._h, 0, w.ctypes.data, 4, 1)
    p.close()
    return int(w[0])


def run(trees, steps=10, warmup=2):
    pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
    out = torch.empty((len(trees), N), device=dev, dtype=torch.float32)
    ok = torch.empty(len(trees), device=dev, dtype=torch.uint8)
    def step():
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    okh = ok.cpu().numpy().astype(bool)
    pop.close()
    del out
    return ms, okh


_, okc = run(cand[:1200], steps=1, warmup=0)
complete = [t for t, k in zip(cand[:1200], okc) if k]
sl = [slots_of(t) for t in complete]
hist = {k: sl.count(k) for k in sorted(set(sl))}
one = [t for t, k in zip(complete, sl) if k <= 1][:400]
two = [t for t, k in zip(complete, sl) if k >= 2]
res = dict(complete_trees=len(complete), slot_histogram=hist, n_one_slot=len(one))
a = [run(one)[0] for _ in range(3)]
b = [run(one + two[:1])[0] for _ in range(3)]
res["ms_6_rows_6_waves"] = a
res["ms_7_rows_5p25_waves_one_more_tree"] = b
res["ratio"] = min(a) / (min(b) * len(one) / (len(one) + 1))
print(json.dumps(res, indent=1))
