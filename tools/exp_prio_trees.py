import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
lib = api.library()
N = 10**7
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
for nt in (32, 64, 125, 250, 500):
    trees = de.synth.random_population(1000, seed=0xDE02)[:nt]
    pop = api.Population(trees, ops, np.float32, n_features=5)
    out = torch.empty((nt, N), device=dev, dtype=torch.float32)
    ok = torch.empty(nt, device=dev, dtype=torch.uint8)
    r = {}
    for tag, env in (("prio", {}), ("no prio", {"DE_NO_PRIO_TILES": "1"}), ("prio", {}), ("no prio", {"DE_NO_PRIO_TILES": "1"})):
        os.environ.pop("DE_NO_PRIO_TILES", None); os.environ.update(env)
        ms = []
        for i in range(6):
            pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
            torch.cuda.synchronize()
            if i >= 1: ms.append(pop.ctx.last_kernel_ms())
        r.setdefault(tag, []).append(round(float(np.median(ms)), 3))
    print(nt, "trees:", r, flush=True)
    pop.close(); del out
