"""Reproduce tests/fuzz/fuzz_gpu.py and, at the first value mismatch, print the sample and every subtree's value."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
from helpers import parity_tolerance
seed0 = int(sys.argv[1])
HOT = len(sys.argv) > 2 and sys.argv[2] == "hot"  # reproduce tests/fuzz/fuzz_hot.py instead of tests/fuzz/fuzz_gpu.py
ops_wide = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^"),
                           unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                            "safe_sqrt", "atan", "relu"))
ops_hot = de.synth.BENCH_OPERATORS

def subtrees(t):
    out = [t]
    for c in t.children[:t.degree]:
        out += subtrees(c)
    return out

found = 0
for rep in range(5 if HOT else 6):
    rng = de.synth.Xoshiro256ss(seed0 * 31 + rep if HOT else seed0 * 1000 + rep)
    for ops, F in (((ops_hot, 1 + (seed0 + rep) % 7),) if HOT else ((ops_hot, 5), (ops_wide, 3), (ops_hot, 2))):
        for dtype in (np.float32, np.float64):
            if HOT:
                trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 3 + rep) % 40, ops, F, rng, dtype) for i in range(300)]
                g = np.random.Generator(np.random.PCG64(seed0 * 7 + rep))
                N = int(g.choice([1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 3000, 5121]))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.5, 1, 3])).astype(dtype))
            else:
                trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 7 + rep) % 33, ops, F, rng, dtype) for i in range(400)]
                g = np.random.Generator(np.random.PCG64(seed0 + rep))
                N = int(g.integers(1, 1500))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(dtype))
                if rep % 2:
                    X[0, N // 2] = np.inf
            for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True)):
                opts = ec.option_bits(ops)
                pop = api.Population(trees, ops, dtype, n_features=F, eval_context=ec)
                out, ok = pop.eval(X)
                for t, tree in enumerate(trees):
                    tape, consts = de.flatten(tree, ops, dtype)
                    y, ok_el = oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
                    if bool(ok[t]) != ok_el:
                        _, _ = 0, 0
                        yg_full = out[t]
                        # first sample where the finiteness differs
                        d = np.nonzero(np.isfinite(yg_full) != np.isfinite(y))[0]
                        print("FLAG MISMATCH rep", rep, "F", F, dtype.__name__, "opts", opts, "tree", t, de.string_tree(tree, ops), "gpu ok", bool(ok[t]), "oracle", ok_el, "finite-diff samples", d[:5])
                        j = int(d[0]) if len(d) else 0
                        Xj = np.asfortranarray(X[:, j:j + 1])
                        print(" sample", j, "X", X[:, j])
                        for st in subtrees(tree)[:40]:
                            tp, cs = de.flatten(st, ops, dtype)
                            yo, _ = oracle.eval_tree_array(tp, cs, Xj, 0, elementwise=True)
                            yg, _ = api.eval_tree_array(st, Xj, ops, eval_context=api.EvalContext(early_exit=False))
                            flag = "" if (yo[0] == yg[0] or (np.isnan(yo[0]) and np.isnan(yg[0]))) else "   <-- differs"
                            print("   ", de.string_tree(st, ops)[:70], "gpu", repr(yg[0]), "oracle", repr(yo[0]), flag)
                        found += 1
                        if found >= 1: sys.exit(0)
                        continue
                    if not ok_el or not ok[t]:
                        continue
                    m = np.isfinite(y)
                    tol = parity_tolerance(tree, ops, X, dtype, opts)
                    err = np.abs(out[t].astype(np.float64) - y)
                    bad = np.nonzero((m & np.isfinite(out[t]) & (err > tol)) | ((np.isfinite(out[t]) != m) & np.isfinite(tol)))[0]
                    err = np.where(np.isfinite(err), err, np.inf)
                    if len(bad):
                        j = bad[np.argmax(err[bad])]
                        print("MISMATCH rep", rep, "F", F, dtype.__name__, "opts", opts, "tree", t, de.string_tree(tree, ops))
                        print(" sample", j, "X", X[:, j], "gpu", out[t][j], "oracle", y[j], "tol", tol[j])
                        Xj = np.asfortranarray(X[:, j:j + 1])
                        for st in subtrees(tree)[:40]:
                            tp, cs = de.flatten(st, ops, dtype)
                            yo, _ = oracle.eval_tree_array(tp, cs, Xj, 0, elementwise=True)
                            yg, _ = api.eval_tree_array(st, Xj, ops, eval_context=api.EvalContext(early_exit=False))
                            flag = "" if (yo[0] == yg[0] or (np.isnan(yo[0]) and np.isnan(yg[0]))) else "   <-- differs"
                            print("   ", de.string_tree(st, ops)[:70], "gpu", repr(yg[0]), "oracle", repr(yo[0]), flag)
                        found += 1
                        if found >= 1: sys.exit(0)
                pop.close()
print("no mismatch")
