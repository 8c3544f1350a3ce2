import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^", "mod", "rem", "greater"),
                      unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                       "safe_sqrt", "relu", "sign", "round", "atan"))
rng = de.synth.Xoshiro256ss(99)
trees = [de.synth.gen_random_tree_fixed_size(5 + i % 20, ops, 3, rng, np.float32) for i in range(150)]
g = np.random.Generator(np.random.PCG64(3))
X = np.asfortranarray(g.standard_normal((3, 777)).astype(np.float32))
X[1, 5] = np.inf
X[0, 700] = np.nan
for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True)):
    pop = api.Population(trees, ops, np.float32, n_features=3, eval_context=ec)
    out, ok = pop.eval(X)
    opts = ec.option_bits(ops)
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, np.float32)
        y, ok_el = oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
        if ok_el and not np.array_equal(np.isfinite(out[t]), np.isfinite(y)):
            bad = np.nonzero(np.isfinite(out[t]) != np.isfinite(y))[0]
            print("opts", opts, "tree", t, de.string_tree(tree, ops), "idx", bad[:5], "gpu", out[t][bad[:5]], "oracle", y[bad[:5]], "X", X[:, bad[:3]].T)
