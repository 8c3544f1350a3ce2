import sys, os
sys.path.insert(0, os.environ.get("R", "."))
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(1000, seed=0xDE02)
N = 10**6
ctx = api.Context(0)
pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((N, 5), generator=g, device="cuda").t()
out = torch.empty((1000, N), device="cuda")
grad = torch.empty(1000 * 5 * N, device="cuda")
ok = torch.empty(1000, device="cuda", dtype=torch.uint8)
lib = api.library()
which = sys.argv[1]
o = out.data_ptr() if which == "out" else None
for _ in range(2):
    ctx.check(lib.de_eval_grad(ctx._h, pop._h, X.data_ptr(), N, 5, None, 0, o, N, grad.data_ptr(), None, ok.data_ptr()))
torch.cuda.synchronize()
