"""Differential fuzz of the gradient kernels on the GPU box: flags must equal the oracle's; values are compared
loosely (median relative error) — the strict comparisons live in tests/test_gpu_grad.py."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops_wide = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^"),
                           unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                            "safe_sqrt", "atan", "relu"))
MODES = {"variable": (True, oracle.GRAD_VARIABLE), "constant": (False, oracle.GRAD_CONSTANT), "both": ("both", oracle.GRAD_BOTH)}
bad = 0
for rep in range(4):
    rng = de.synth.Xoshiro256ss(seed0 * 100 + rep)
    for ops, F in ((de.synth.BENCH_OPERATORS, 5), (ops_wide, 3)):
        for dtype in (np.float32, np.float64):
            trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 5 + rep) % 31, ops, F, rng, dtype) for i in range(250)]
            g = np.random.Generator(np.random.PCG64(seed0 + rep))
            N = int(g.integers(1, 900))
            X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.3, 1, 4])).astype(dtype))
            pop = api.Population(trees, ops, dtype, n_features=F)
            for name, (variable, om) in MODES.items():
                out, grads, ok = pop.eval_grad(X, variable)
                for t, tree in enumerate(trees):
                    tape, consts = de.flatten(tree, ops, dtype)
                    y, gg, ok_el = oracle.eval_grad_tree_array(tape, consts, X, om, elementwise=True)
                    if bool(ok[t]) != ok_el:
                        fin_g = np.isfinite(np.asarray(grads[t])).all() and np.isfinite(out[t]).all()
                        fin_o = np.isfinite(gg).all() and np.isfinite(y).all()
                        print("FLAG", name, dtype.__name__, "tree", t, de.string_tree(tree, ops)[:150], "gpu", bool(ok[t]), "oracle", ok_el, "gpu all finite", fin_g, "oracle all finite", fin_o)
                        bad += 1
                        continue
                    if not ok_el or gg.size == 0:
                        continue
                    rel = np.abs(np.asarray(grads[t], dtype=np.float64) - gg) / (np.abs(gg) + 1e-30)
                    if np.median(rel) > (1e-3 if dtype == np.float32 else 1e-9):
                        print("VALUE", name, dtype.__name__, "tree", t, de.string_tree(tree, ops)[:150], "median rel err", np.median(rel))
                        bad += 1
            pop.close()
            print("done", rep, F, dtype.__name__, N, flush=True)
print("grad fuzz finished, findings:", bad)
