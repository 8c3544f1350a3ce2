import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
from helpers import parity_tolerance
seed0, rep, F, dtype = 3, 0, 2, np.float64
ops = de.synth.BENCH_OPERATORS
rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
for o, f in ((ops, 5), (None, 3), (ops, 2)):
    for dt in (np.float32, np.float64):
        if o is None:
            opsw = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^", "mod", "rem", "greater"),
                           unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin", "safe_sqrt", "relu", "sign", "round", "atan"))
            trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 7 + rep) % 33, opsw, f, rng, dt) for i in range(400)]
        else:
            trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 7 + rep) % 33, o, f, rng, dt) for i in range(400)]
g = np.random.Generator(np.random.PCG64(seed0 + rep))
N = int(g.integers(1, 1500))
X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(dtype))
trees = trees[:150]
pop = api.Population(trees, ops, dtype, n_features=F)
for variable, om in ((True, oracle.GRAD_VARIABLE),):
    out, grads, ok = pop.eval_grad(X, variable)
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, dtype)
        y, gg, ok_el = oracle.eval_grad_tree_array(tape, consts, X, om, elementwise=True)
        if not ok_el: continue
        rel = np.abs(out[t] - y) / np.maximum(np.abs(y), 1e-300)
        frac = np.mean(rel <= 1e-11)
        if frac <= 0.99:
            tol = parity_tolerance(tree, ops, X, dtype, 7)
            print("tree", t, de.string_tree(tree, ops), "frac ok", frac, "max rel", rel.max(), "within parity tol:", np.mean(np.abs(out[t]-y) <= tol))
