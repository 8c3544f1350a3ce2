"""Debug aid: outputs of the threaded gradient kernel vs the oracle for single small trees (run from the repo root)."""
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
N = 6
X = de.synth.random_X(4, N, seed=12, dtype=dtype)
print(X)
for sel in ([1], [2], [3], [1, 2, 3], list(range(12))):
    sub = [trees[i] for i in sel]
    pop = api.Population(sub, ops, dtype, n_features=4)
    out, grads, ok = pop.eval_grad(X, True)
    for t, tree in enumerate(sub):
        tape, consts = de.flatten(tree, ops, dtype)
        y, g, ok_el = oracle.eval_grad_tree_array(tape, consts, X, oracle.GRAD_VARIABLE, elementwise=True)
        same = np.array_equal(out[t], y, equal_nan=True) and np.array_equal(np.asarray(grads[t]), g, equal_nan=True)
        print(sel, t, de.string_tree(tree, ops)[:60], "OK" if same else "BAD")
        if not same and len(sel) <= 3:
            print(" gpu out", out[t]); print(" ora out", y); print(" gpu g", np.asarray(grads[t])); print(" ora g", g)
            print(pop.dump(3)[:40] if hasattr(pop, "dump") else "")
