"""Differential fuzz of the fused loss gradient: reverse accumulation (de_rev_threaded.hip) against forward duals
(de_grad_threaded.hip) on the GPU box — same flags, same losses, gradient rows equal up to the conditioning of the
row (sum of |terms| of the forward Jacobian; rows whose paths cancel are only checked loosely)."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops_wide = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^"),
                           unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                            "safe_sqrt", "atan", "relu"),
                           ternary_operators=("fma", "clamp"))
bad = flagdiff = checked = 0
for rep in range(4):
    rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
    for ops, F, P in ((de.synth.BENCH_OPERATORS, 5, 0), (ops_wide, 3, 0), (de.synth.BENCH_OPERATORS, 2, 3)):
        for dtype in (np.float32, np.float64):
            nt = de.ParametricNode if P else de.Node
            args = (nt, P) if P else ()
            trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 5 + rep) % 31, ops, F, rng, dtype, *args) for i in range(200)]
            g = np.random.Generator(np.random.PCG64(seed0 * 10 + rep))
            N = int(g.integers(1, 1500))
            X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.3, 1, 3])).astype(dtype))
            y = g.standard_normal(N).astype(dtype)
            w = (g.random(N) > 0.15).astype(dtype)
            kw = {}
            if P:
                kw = dict(params=np.asfortranarray(g.standard_normal((P, 4)).astype(dtype)), classes=g.integers(1, 5, N))
            pop = api.Population(trees, ops, dtype, n_features=F, n_params=P)
            for variable in (False, True, "both"):
                for kind in ("L2", "pullback"):
                    os.environ["DE_LOSS_GRAD_REVERSE"] = "0"
                    lf, df, okf = pop.eval_loss_grad(X, y, weights=w, loss=kind, variable=variable, **kw)
                    os.environ["DE_LOSS_GRAD_REVERSE"] = "1"
                    lr, dr, okr = pop.eval_loss_grad(X, y, weights=w, loss=kind, variable=variable, **kw)
                    out, grads, okg = pop.eval_grad(X, variable, **kw)
                    eps = np.finfo(dtype).eps
                    for t in range(len(trees)):
                        if okf[t] != okr[t]:
                            flagdiff += 1
                            print("FLAG", dtype.__name__, variable, kind, okf[t], okr[t], de.string_tree(trees[t], ops)[:160])
                            continue
                        if not okf[t]:
                            continue
                        checked += 1
                        g64 = np.asarray(grads[t], dtype=np.float64)
                        lp = y.astype(np.float64) if kind == "pullback" else 2 * (out[t].astype(np.float64) - y)
                        mag = (np.abs(w * lp)[None, :] * np.abs(g64)).sum(axis=1)
                        err = np.abs(dr[t].astype(np.float64) - df[t].astype(np.float64))
                        # loose: a row whose paths cancel inside a sample has a larger conditioning than mag
                        # rows that are pure rounding noise (d/dx of (p*x)/x) carry the conditioning of their paths, of the order of the other rows'
                        M = mag.max(initial=0)
                        lim = 4096 * eps * np.where(mag > 1e-6 * M, mag, M) + 1e-3 * np.abs(df[t]) * (mag > 1e-6 * M) + 1e-30
                        fin = np.isfinite(df[t]) & np.isfinite(dr[t]) & (mag < 0.01 * np.finfo(dtype).max)
                        if np.any((err > lim) & fin) or (kind == "L2" and np.isfinite(lf[t]) and lf[t] != lr[t] and abs(lf[t] - lr[t]) > 64 * eps * abs(lf[t])):  # a pullback 'loss' is a cancelling sum
                            bad += 1
                            print("VALUE", dtype.__name__, variable, kind, de.string_tree(trees[t], ops)[:160], df[t][:6], dr[t][:6], lf[t], lr[t])
            pop.close()
    print("done", rep, checked, flush=True)
print("loss-grad fuzz finished: value findings", bad, "flag differences", flagdiff, "checked", checked)
