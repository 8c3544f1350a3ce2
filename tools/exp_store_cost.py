#!/usr/bin/env python3
"""What do the output stores cost?  (gpurun)  The headline population and its complete trees alone, with and without the stores
(DE_DEBUG_NO_STORE=1: the end of a tree keeps the value alive and writes nothing) -> gpurun_out/store_cost.json"""
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
res = {}


def run(sub, tag, envs):
    pop = api.Population(sub, ops, np.float32, n_features=5)
    out = torch.empty((len(sub), N), device=dev, dtype=torch.float32)
    ok = torch.empty(len(sub), device=dev, dtype=torch.uint8)
    ctx = pop.ctx
    for name, env in envs:
        for k in ("DE_DEBUG_NO_STORE", "DE_SKIP_PROTOCOL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = []
        for i in range(7):
            ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
            torch.cuda.synchronize()
            if i >= 2:
                ms.append(ctx.last_kernel_ms())
        res[f"{tag}: {name}"] = float(np.median(ms))
        print(tag, name, res[f"{tag}: {name}"], flush=True)
    f = ok.cpu().numpy().astype(bool)
    pop.close()
    del out
    return f


E = [("stores", {}), ("no stores", {"DE_DEBUG_NO_STORE": "1"}), ("no stores, agent-scope flags", {"DE_DEBUG_NO_STORE": "1", "DE_SKIP_PROTOCOL": "1"}),
     ("stores, agent-scope flags", {"DE_SKIP_PROTOCOL": "1"})]
flags = run(trees, "real population", E)
comp = [t for t, f in zip(trees, flags) if f]
run(comp, "complete trees only", E)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "store_cost.json"), "w"), indent=1)
