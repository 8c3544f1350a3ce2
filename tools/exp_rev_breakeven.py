#!/usr/bin/env python3
"""Where reverse accumulation overtakes forward duals for the fused loss gradient (de_eval_loss_grad, constant mode): populations of 512
bench-style trees with EXACTLY r constants each, r = 3 ... 8, 10^6 samples; DE_LOSS_GRAD_REVERSE=0 / 1 forces the kernel.
    gpurun -- 'python tools/exp_rev_breakeven.py'   -> gpurun_out/rev_breakeven.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    import torch
    import dynamicexpressions_jl_amd as de
    from dynamicexpressions_jl_amd import api
    from dynamicexpressions_jl_amd.node import count_constant_nodes
    r = int(sys.argv[2])
    ops = de.synth.BENCH_OPERATORS
    cand = de.synth.random_population(60000, seed=0xBE01)
    trees = [t for t in cand if count_constant_nodes(t) == r][:512]
    assert len(trees) == 512, (r, len(trees))
    N = 10**6
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
    y = torch.randn(N, generator=g, device="cuda", dtype=torch.float32)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    for _ in range(3):
        out = pop.eval_loss_grad(X, y, variable=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        out = pop.eval_loss_grad(X, y, variable=False)
    b.record()
    torch.cuda.synchronize()
    print(json.dumps(dict(rows=r, ms=a.elapsed_time(b) / 10, complete=float(out[2].float().mean()))))
    sys.exit(0)

res = {}
for r in range(3, 9):  # (20-node trees: 60000 candidates hold 752 trees with 8 constants, 107 with 9)
    row = {}
    for rev in ("0", "1"):
        env = dict(os.environ, DE_LOSS_GRAD_REVERSE=rev)
        o = subprocess.run([sys.executable, __file__, "--one", str(r)], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("failed", r, rev, o.stderr[-400:])
            continue
        d = json.loads(line[-1])
        row["reverse" if rev == "1" else "forward"] = d["ms"]
        row["complete"] = d["complete"]
    res[r] = row
    print(r, row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "rev_breakeven.json"), "w"), indent=1)
