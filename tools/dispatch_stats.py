#!/usr/bin/env python3
"""Dispatch statistics of the bench population's fused program as the device runs it (after constant folding: needs the
GPU box): handler histogram and handler bigrams -> gpurun_out/dispatch_stats.json."""
import collections, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
n = 1000
trees = de.synth.random_population(n, seed=0xDE02)
turbo = "--turbo" in sys.argv
pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(turbo=turbo))
lib = api.library()
uni, big = collections.Counter(), collections.Counter()
seqs = []
for t in range(n):
    k = lib.de_program_dump(pop._h, t, None, 0, 3)
    w = np.zeros(int(k), dtype=np.uint32)
    lib.de_program_dump(pop._h, t, w.ctypes.data, w.size, 3)
    ids = [int(v) for v in w.reshape(-1, 4)[:, 0]]
    seqs.append(ids)
    uni.update(ids)
    big.update(zip(ids, ids[1:]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(n_trees=n, unigrams={str(k): v for k, v in uni.items()}, bigrams=[[a, b, c] for (a, b), c in big.most_common()], seqs=seqs),
          open(os.path.join(ROOT, "gpurun_out", "dispatch_stats.json"), "w"))
print("dispatches per tree", sum(uni.values()) / n)
