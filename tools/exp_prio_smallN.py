#!/usr/bin/env python3
"""Priority tiles + probe launch at small sample counts (gpurun): 1000 bench trees, kernel ms with / without (DE_PRIO_MIN_TILES=1 forces them on)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
lib = api.library()
trees = de.synth.random_population(1000, seed=0xDE02)
pop = api.Population(trees, ops, np.float32, n_features=5)
for N in (2**14, 2**15, 2**16, 2**17, 2**18, 2**19, 2**20):
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
    out = torch.empty((1000, N), device=dev, dtype=torch.float32)
    ok = torch.empty(1000, device=dev, dtype=torch.uint8)
    r = {}
    for tag, env in (("on", {"DE_PRIO_MIN_TILES": "1"}), ("off", {"DE_NO_PRIO_TILES": "1"}), ("on", {"DE_PRIO_MIN_TILES": "1"}), ("off", {"DE_NO_PRIO_TILES": "1"})):
        for k in ("DE_PRIO_MIN_TILES", "DE_NO_PRIO_TILES"): os.environ.pop(k, None)
        os.environ.update(env)
        ms = []
        for i in range(12):
            pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
            torch.cuda.synchronize()
            if i >= 2: ms.append(pop.ctx.last_kernel_ms())
        r.setdefault(tag, []).append(round(float(np.median(ms)), 4))
    print(N, "samples (", (N + 255) // 256, "tiles ):", r, flush=True)
    del out, X
