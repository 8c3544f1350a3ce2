#!/usr/bin/env python3
"""The pass over X that finds the priority tiles: scalar loop (DE_PRIO_VEC=0) against the vectorised one (packed, aligned Float32 X): ms per
eval step of the headline population at 10^7 / 10^6 / 2.6*10^5 samples, the trees still live behind the probe launch, and the flags.
    gpurun -- 'python tools/exp_prio_vec.py'"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
lib = api.library()
for N in (10**7, 10**6 + 3, 2**18 + 1):
    trees = de.synth.random_population(1000, seed=0xDE0C)
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
    res = {}
    for vec in ("0", "1"):
        os.environ["DE_PRIO_VEC"] = vec
        pop = api.Population(trees, ops, np.float32, n_features=5)
        out = torch.empty((1000, N), device="cuda", dtype=torch.float32)
        ok = torch.empty(1000, device="cuda", dtype=torch.uint8)
        pop.ctx.use_torch_stream()
        def step():
            pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        for _ in range(3): step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): step()
        b.record(); torch.cuda.synchronize()
        import ctypes as C
        nlive = C.c_int64(0)
        pop.ctx.check(lib.de_program_last_live_trees(pop._h, C.byref(nlive)))
        live = nlive.value
        res[vec] = (a.elapsed_time(b) / 20, int(live), ok.clone())
        pop.close(); del out
    print(N, "scalar ms %.4f live %d | vec ms %.4f live %d | flags equal %s" % (res["0"][0], res["0"][1], res["1"][0], res["1"][1], bool(torch.equal(res["0"][2], res["1"][2]))), flush=True)
