import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=("neg", "square", "abs"))
rng = de.synth.Xoshiro256ss(17)
dtype = np.float32
trees = [de.synth.gen_random_tree_fixed_size(3 + i % 26, ops, 4, rng, dtype) for i in range(80)]
X = de.synth.random_X(4, 777, seed=12, dtype=dtype)
pop = api.Population(trees, ops, dtype, n_features=4)
out, grads, ok = pop.eval_grad(X, True)
n = 0
for t, tree in enumerate(trees):
    tape, consts = de.flatten(tree, ops, dtype)
    y, g, ok_el = oracle.eval_grad_tree_array(tape, consts, X, oracle.GRAD_VARIABLE, elementwise=True)
    if bool(ok[t]) != ok_el or (ok_el and not (np.array_equal(out[t], y) and np.array_equal(np.asarray(grads[t]), g))):
        n += 1
        if n <= 4:
            bad = np.nonzero(~np.isfinite(np.asarray(grads[t])).all(axis=0) | ~np.isfinite(out[t]))[0][:3]
            print("tree", t, de.string_tree(tree, ops), "ok gpu", bool(ok[t]), "oracle", ok_el, "nonfinite at", bad,
                  "vals equal", np.array_equal(out[t], y), "grad equal", np.array_equal(np.asarray(grads[t]), g))
            print(pop.dump(t)[:, 0] & 0xFF)
print("mismatches", n)
