"""Experiment: a parametric population evaluated as a plain one over [X; parameters[:, classes]] (the reference's own
formulation, src/ParametricExpression.jl:381-389) against the in-kernel parameter path."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

def to_plain(t, F):
    if t.degree == 0:
        if getattr(t, "is_parameter", False):
            return de.Node(feature=F + t.parameter)
        return de.Node(val=t.val) if t.constant else de.Node(feature=t.feature)
    return de.Node(t.op, *[to_plain(c, F) for c in t.children])

ops = de.synth.BENCH_OPERATORS
F, P, C, N = 5, 8, 16, 10**6
trees = de.synth.random_population(1000, seed=0xDE05, node_type=de.ParametricNode, nparams=P)
plain = [to_plain(t, F) for t in trees]
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((N, F), generator=g, device="cuda").t()
params = torch.randn((C, P), generator=g, device="cuda").t()
classes = torch.randint(1, C + 1, (N,), generator=g, device="cuda", dtype=torch.int32)
pp = api.Population(trees, ops, np.float32, n_features=F, n_params=P)
pl = api.Population(plain, ops, np.float32, n_features=F + P)
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_gather = timeit(lambda: torch.cat([X.t(), params[:, (classes - 1).long()].t()], dim=1))
Xe = torch.cat([X.t(), params[:, (classes - 1).long()].t()], dim=1).contiguous().t()  # [F+P, N] feature-fastest
o1, k1 = pp.eval(X, params=params, classes=classes)
o2, k2 = pl.eval(Xe)
torch.cuda.synchronize()
print("flags equal:", bool(torch.equal(k1, k2)), " values equal where complete:", bool(torch.equal(o1[k1], o2[k1])))
print("in-kernel parameters: %.2f ms   plain over [X; params]: %.2f ms (+ gather %.2f ms)" % (
    timeit(lambda: pp.eval(X, params=params, classes=classes)), timeit(lambda: pl.eval(Xe)), t_gather))
for variable in (False, "both"):
    print("grad", variable, "%.2f ms vs plain %.2f ms" % (timeit(lambda: pp.eval_grad(X, variable, params=params, classes=classes), 2),
                                                         timeit(lambda: pl.eval_grad(Xe, variable), 2)))
