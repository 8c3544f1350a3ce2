#!/usr/bin/env python3
"""Trace one Jacobian finding of tests/fuzz/fuzz_gpu.py 33 (rep 4, Float64, wide operator set, tree 49) against mpmath (gpurun):
which of the device / the oracle is off, by how much, and what the tolerance model says."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
import fuzzlib as FZ
from helpers import grad_tolerance
import mpmath
mpmath.mp.prec = 300

seed0, rep = 33, 4
rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
target = None
for ops, F in ((FZ.OPS_HOT, 5), (FZ.OPS_WIDE, 3), (FZ.OPS_HOT, 2)):
    for dtype in (np.float32, np.float64):
        trees = FZ.random_trees(rng, ops, F, dtype, 400, 33, rep)
        if ops is FZ.OPS_WIDE and dtype == np.float64:
            target = (trees[49], ops, F)
tree, ops, F = target
print(de.string_tree(tree, ops))
g = np.random.Generator(np.random.PCG64(seed0 + rep))
N = int(g.integers(1, 1500))
X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(np.float64))
pop = api.Population([tree], ops, np.float64, n_features=F)
out, grads, ok = pop.eval_grad(X, True)
tape, consts = de.flatten(tree, ops, np.float64)
y, go, oke = oracle.eval_grad_tree_array(tape, consts, X, oracle.GRAD_VARIABLE, elementwise=True)
G = np.asarray(grads[0], dtype=np.float64)
tol = grad_tolerance(tree, ops, X, np.float64, "variable")
err = np.abs(G - go)
ratio = np.where(np.isfinite(tol) & (tol > 0), err / tol, 0)
k = np.unravel_index(np.argmax(ratio), ratio.shape)
print("N", N, "worst err/tol", ratio[k], "entry", k, "gpu", repr(G[k]), "oracle", repr(go[k]), "tol", tol[k], "value gpu/oracle", repr(out[0][k[1]]), repr(y[k[1]]))
xv = [mpmath.mpf(float(v)) for v in X[:, k[1]]]
print("x", [float(v) for v in xv])


def f(a, b, c):
    pa = lambda x, yv: mpmath.e ** (yv * mpmath.log(abs(x)))
    t = mpmath.tanh(pa((a * c) ** 2, max(c, c)) + mpmath.mpf(2.3079707664371787) / mpmath.sin(c))
    return mpmath.sin(pa(mpmath.cos(mpmath.mpf(0.8577199987506753)), t))


args = list(xv)
def partial(i):
    return mpmath.diff(lambda v: f(*[v if q == i else args[q] for q in range(3)]), args[i])
d = partial(int(k[0]))
print("truth", mpmath.nstr(d, 20), " gpu err", float(abs(mpmath.mpf(float(G[k])) - d)), " oracle err", float(abs(mpmath.mpf(float(go[k])) - d)),
      " rel", float(abs(mpmath.mpf(float(G[k])) - d) / abs(d)) if d != 0 else None)
