"""Experiment: cost of operators outside the hot set (generic handler) against the hot ones, eval and gradient."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
N = 10**6
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((N, 5), generator=g, device="cuda").t()
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, una in (("cos exp", ("cos", "exp")), ("square neg", ("square", "neg")), ("abs cube", ("abs", "cube")),
                  ("safe_log safe_sqrt", ("safe_log", "safe_sqrt")), ("tanh sin", ("tanh", "sin"))):
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=una)
    rng = de.synth.Xoshiro256ss(7)
    trees = [de.synth.gen_random_tree_fixed_size(20, ops, 5, rng, np.float32) for _ in range(1000)]
    pop = api.Population(trees, ops, np.float32, n_features=5)
    pop.eval(X); te = pop.ctx.last_kernel_ms()          # device time of the kernels of the call (hipEvents)
    pop.eval_grad(X, True); pop.eval_grad(X, True); tg = pop.ctx.last_kernel_ms()
    print(f"unary = {name:20s} eval {te:6.2f} ms   grad(variable) {tg:6.2f} ms")
    pop.close()
