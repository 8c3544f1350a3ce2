import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(1000, seed=0xDE02)
N = 10**6
g = torch.Generator(device="cuda").manual_seed(1)
Xd = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
pop = api.Population(trees, ops, np.float32, n_features=5)
out, ok = pop.eval(Xd)
head, ok_h = pop.eval(Xd[:, :2048])
torch.cuda.synchronize()
a = out[:, :2048]; b = head
neq = ~((a == b) | (a.isnan() & b.isnan()))
print("neq elements", int(neq.sum()), "rows", int(neq.any(1).sum()), "complete rows", int(ok.sum()))
rows = neq.any(1).nonzero().flatten()[:10].tolist()
for r in rows:
    cols = neq[r].nonzero().flatten()
    print(r, bool(ok[r]), len(cols), cols[:8].tolist(), a[r, cols[:4]].tolist(), b[r, cols[:4]].tolist(), de.string_tree(trees[r], ops)[:100])
print("nan in complete rows:", int(out[ok].isnan().any(1).sum()), int((~out[ok].isfinite()).any(1).sum()))
