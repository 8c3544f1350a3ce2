#!/usr/bin/env python3
"""Trace the open findings of round 3's forced-priority-tile fuzz sweep to a CAUSE: for every value / Jacobian
entry beyond the suite's tolerance model, the TRUE value (mpmath, 200 bits, forward-mode duals through the tree with the exact Float32 /
Float64 inputs and constants) against what the device and the oracle returned.

    gpurun -- 'DE_PRIO_MIN_TILES=1 python tools/trace_findings.py > gpurun_out/findings_traced.jsonl'

Flavours re-run: tests/fuzz/fuzz_param.py 31 / 32 (eval values of ParametricExpression trees, the `cos(... - exp(exp(p2)) ...)` finding at
err / tol 1.29) and the Float32 Jacobians of tests/fuzz/fuzz_gpu.py 31 / 32 (rows at 1.2-1.6 x: safe_log / pow_abs2 chains).  One JSON line
per finding: who is further from the truth, in units of the modelled tolerance."""
import json
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import mpmath  # noqa: E402

import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402
from oracle import oracle  # noqa: E402
import fuzzlib as FZ  # noqa: E402
from helpers import grad_tolerance, parity_tolerance  # noqa: E402

mpmath.mp.prec = 200
mpf = mpmath.mpf


class D:
    """value + gradient vector (mpmath)"""
    def __init__(self, v, g):
        self.v, self.g = v, g


def _un(name, a):
    v = a.v
    if name == "cos": return mpmath.cos(v), -mpmath.sin(v)
    if name == "sin": return mpmath.sin(v), mpmath.cos(v)
    if name == "exp": return mpmath.exp(v), mpmath.exp(v)
    if name == "neg": return -v, mpf(-1)
    if name == "square": return v * v, 2 * v
    if name == "cube": return v * v * v, 3 * v * v
    if name == "abs": return abs(v), mpmath.sign(v)
    if name == "tanh": return mpmath.tanh(v), 1 - mpmath.tanh(v) ** 2
    if name == "atan": return mpmath.atan(v), 1 / (1 + v * v)
    if name == "relu": return (v if v > 0 else mpf(0)), (mpf(1) if v > 0 else mpf(0))
    if name == "safe_log": return (mpmath.log(v), 1 / v) if v > 0 else (mpmath.nan, mpmath.nan)
    if name == "safe_sqrt": return (mpmath.sqrt(v), 1 / (2 * mpmath.sqrt(v))) if v >= 0 else (mpmath.nan, mpmath.nan)
    raise KeyError(name)


def _bin(name, a, b):
    x, y = a.v, b.v
    if name == "+": return x + y, mpf(1), mpf(1)
    if name == "-": return x - y, mpf(1), mpf(-1)
    if name == "*": return x * y, y, x
    if name == "/": return x / y, 1 / y, -x / (y * y)
    if name == "max": return (x, mpf(1), mpf(0)) if x >= y else (y, mpf(0), mpf(1))
    if name == "min": return (x, mpf(1), mpf(0)) if x <= y else (y, mpf(0), mpf(1))
    if name == "pow_abs2":
        r = mpmath.exp(y * mpmath.log(abs(x)))
        return r, r * y / x, r * mpmath.log(abs(x))
    if name == "^":
        r = mpmath.power(x, y)
        return r, y * mpmath.power(x, y - 1), (r * mpmath.log(x) if x > 0 else mpf(0))
    raise KeyError(name)


def truth(tree, ops, leaf_value, n_grad, seed_of):
    """leaf_value(node) -> mpf; seed_of(node) -> gradient index or None."""
    def go(n):
        if n.degree == 0:
            g = [mpf(0)] * n_grad
            k = seed_of(n)
            if k is not None:
                g[k] = mpf(1)
            return D(leaf_value(n), g)
        if n.degree == 1:
            a = go(n.children[0])
            v, da = _un(ops.unaops[n.op - 1], a)
            return D(v, [da * q for q in a.g])
        a, b = go(n.children[0]), go(n.children[1])
        v, da, db = _bin(ops.binops[n.op - 1], a, b)
        return D(v, [da * p + db * q for p, q in zip(a.g, b.g)])
    return go(tree)


def const_order(tree):
    out = []
    def walk(n):
        if n.degree == 0:
            if n.constant:
                out.append(n)
        else:
            for c in n.children:
                walk(c)
    walk(tree)
    return out


def trace_grad(tree, ops, X, dtype, mode, j, k):
    """true d tree / d (gradient row k) at sample j"""
    F = X.shape[0]
    consts = const_order(tree)
    ids = {id(c): i for i, c in enumerate(consts)}
    n_grad = {"variable": F, "constant": len(consts), "both": F + len(consts)}[mode]
    def seed(n):
        if n.constant:
            return None if mode == "variable" else (ids[id(n)] if mode == "constant" else F + ids[id(n)])
        return None if mode == "constant" else n.feature - 1
    leaf = lambda n: mpf(float(dtype(n.val))) if n.constant else mpf(float(X[n.feature - 1, j]))  # noqa: E731
    return truth(tree, ops, leaf, n_grad, seed).g[k]


STATS = dict(trees_compared=0, entries_compared=0, worst_err_over_tol=0.0)


def jacobian_findings(seed0, out):
    for rep in range(6):
        rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
        for ops, F in ((FZ.OPS_HOT, 5), (FZ.OPS_WIDE, 3), (FZ.OPS_HOT, 2)):
            for dtype in (np.float32, np.float64):
                trees = FZ.random_trees(rng, ops, F, dtype, 400, 33, rep)
                g = np.random.Generator(np.random.PCG64(seed0 + rep))
                N = int(g.integers(1, 1500))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(dtype))
                if rep % 2:
                    X[0, N // 2] = np.inf
                if dtype != np.float32:
                    continue
                trees = trees[:150]
                pop = api.Population(trees, ops, dtype, n_features=F)
                for mode in ("variable", "constant", "both"):
                    variable, omode = FZ.GRAD_MODES[mode]
                    _, grads, ok = pop.eval_grad(X, variable)
                    for t, tree in enumerate(trees):
                        tape, consts = de.flatten(tree, ops, dtype)
                        _, go_, ok_el = oracle.eval_grad_tree_array(tape, consts, X, omode, elementwise=True)
                        if not ok_el or not ok[t] or go_.size == 0:
                            continue
                        tol = grad_tolerance(tree, ops, X, dtype, mode)
                        if tol is None:
                            continue
                        G = np.asarray(grads[t], dtype=np.float64)
                        err = np.abs(G - go_.astype(np.float64))
                        ratio = np.where(np.isfinite(tol) & (tol > 0), err / tol, 0)
                        STATS["trees_compared"] += 1
                        STATS["entries_compared"] += int(np.isfinite(tol).sum())
                        STATS["worst_err_over_tol"] = max(STATS["worst_err_over_tol"], float(ratio.max()))
                        if ratio.max() <= 1.0:
                            continue
                        k, j = np.unravel_index(np.argmax(ratio), ratio.shape)
                        try:
                            d = trace_grad(tree, ops, X, dtype, mode, int(j), int(k))
                            eg, eo = float(abs(mpf(float(G[k, j])) - d)), float(abs(mpf(float(go_[k, j])) - d))
                        except Exception as e:  # an operator the tracer does not know
                            d, eg, eo = None, None, str(e)
                        rec = dict(kind="jacobian", fuzz=f"fuzz_gpu {seed0} rep {rep}", mode=mode, tree=de.string_tree(tree, ops)[:200], entry=[int(k), int(j)],
                                   err_over_tol=float(ratio[k, j]), tol=float(tol[k, j]), gpu=float(G[k, j]), oracle=float(go_[k, j]),
                                   truth=None if d is None else float(d), gpu_err_over_tol=None if eg is None else eg / float(tol[k, j]),
                                   oracle_err_over_tol=eo / float(tol[k, j]) if isinstance(eo, float) else eo)
                        out.append(rec)
                        print(json.dumps(rec), flush=True)
                pop.close()


def gpu_findings(seed0, out):
    """tests/fuzz/fuzz_gpu.py <seed>: eval values (default context) and Jacobians, both element types."""
    for rep in range(6):
        rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
        for ops, F in ((FZ.OPS_HOT, 5), (FZ.OPS_WIDE, 3), (FZ.OPS_HOT, 2)):
            for dtype in (np.float32, np.float64):
                trees = FZ.random_trees(rng, ops, F, dtype, 400, 33, rep)
                g = np.random.Generator(np.random.PCG64(seed0 + rep))
                N = int(g.integers(1, 1500))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(dtype))
                if rep % 2:
                    X[0, N // 2] = np.inf
                pop = api.Population(trees, ops, dtype, n_features=F)
                o, ok = pop.eval(X)
                for t, tree in enumerate(trees):
                    tape, consts = de.flatten(tree, ops, dtype)
                    y, ok_el = oracle.eval_tree_array(tape, consts, X, elementwise=True)
                    if not ok_el or not ok[t]:
                        continue
                    tol = parity_tolerance(tree, ops, X, dtype)
                    m = np.isfinite(y) & np.isfinite(o[t]) & np.isfinite(tol)
                    err = np.abs(o[t].astype(np.float64) - y.astype(np.float64))
                    with np.errstate(invalid="ignore", divide="ignore"):
                        ratio = np.where(m & (tol > 0), err / np.where(tol > 0, tol, 1), 0)
                    STATS["trees_compared"] += 1
                    STATS["entries_compared"] += int(m.sum())
                    if ratio.max() <= 1.0:
                        continue
                    j = int(np.argmax(ratio))
                    leaf = lambda n: mpf(float(dtype(n.val))) if n.constant else mpf(float(X[n.feature - 1, j]))  # noqa: E731
                    try:
                        d = truth(tree, ops, leaf, 0, lambda n: None).v
                        eg, eo = float(abs(mpf(float(o[t][j])) - d)), float(abs(mpf(float(y[j])) - d))
                    except Exception as e:  # noqa: BLE001
                        d, eg, eo = None, None, str(e)
                    rec = dict(kind="value", fuzz=f"fuzz_gpu {seed0} rep {rep}", dtype=np.dtype(dtype).name, tree=de.string_tree(tree, ops)[:240], sample=j,
                               x=[float(v) for v in X[:, j]], err_over_tol=float(ratio[j]), tol=float(tol[j]), gpu=float(o[t][j]), oracle=float(y[j]),
                               truth=None if d is None else float(d), gpu_err_over_tol=None if eg is None else eg / float(tol[j]),
                               oracle_err_over_tol=eo / float(tol[j]) if isinstance(eo, float) else eo)
                    out.append(rec)
                    print(json.dumps(rec), flush=True)
                trees_g = trees[:150]
                popg = api.Population(trees_g, ops, dtype, n_features=F)
                for mode in ("variable", "constant", "both"):
                    variable, omode = FZ.GRAD_MODES[mode]
                    _, grads, okg = popg.eval_grad(X, variable)
                    for t, tree in enumerate(trees_g):
                        tape, consts = de.flatten(tree, ops, dtype)
                        _, go_, ok_el = oracle.eval_grad_tree_array(tape, consts, X, omode, elementwise=True)
                        if not ok_el or not okg[t] or go_.size == 0:
                            continue
                        tol = grad_tolerance(tree, ops, X, dtype, mode)
                        if tol is None:
                            continue
                        G = np.asarray(grads[t], dtype=np.float64)
                        err = np.abs(G - go_.astype(np.float64))
                        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
                            ratio = np.where(np.isfinite(tol) & (tol > 0), err / np.where(tol > 0, tol, 1), 0)
                        if not np.isfinite(ratio).all():
                            ratio = np.nan_to_num(ratio, nan=0.0, posinf=1e300)
                        if ratio.max() <= 1.0:
                            continue
                        k, j = np.unravel_index(np.argmax(ratio), ratio.shape)
                        try:
                            d = trace_grad(tree, ops, X, dtype, mode, int(j), int(k))
                            eg, eo = float(abs(mpf(float(G[k, j])) - d)), float(abs(mpf(float(go_[k, j])) - d))
                        except Exception as e:  # noqa: BLE001
                            d, eg, eo = None, None, str(e)
                        rec = dict(kind="jacobian", fuzz=f"fuzz_gpu {seed0} rep {rep}", dtype=np.dtype(dtype).name, mode=mode, tree=de.string_tree(tree, ops)[:240],
                                   entry=[int(k), int(j)], x=[float(v) for v in X[:, j]], err_over_tol=float(ratio[k, j]), tol=float(tol[k, j]), gpu=float(G[k, j]),
                                   oracle=float(go_[k, j]), truth=None if d is None else float(d),
                                   gpu_err_over_tol=None if eg is None else (eg / float(tol[k, j]) if tol[k, j] > 0 else None),
                                   oracle_err_over_tol=(eo / float(tol[k, j]) if isinstance(eo, float) and tol[k, j] > 0 else eo))
                        out.append(rec)
                        print(json.dumps(rec), flush=True)
                popg.close()
                pop.close()


def param_findings(seed0, out):
    for rep in range(4):
        rng = de.synth.Xoshiro256ss(seed0 * 77 + rep)
        ops_wide = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "pow_abs2"),
                                   unary_operators=("cos", "exp", "safe_log", "square", "abs", "tanh", "safe_sqrt"))
        for ops in (de.synth.BENCH_OPERATORS, ops_wide):
            for dtype in (np.float32, np.float64):
                P, F = 1 + rep % 3 * 3, 2 + rep
                trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 3 + rep) % 27, ops, F, rng, dtype, de.ParametricNode, P) for i in range(200)]
                g = np.random.Generator(np.random.PCG64(seed0 * 10 + rep))
                N, C = int(g.integers(1, 1300)), int(g.integers(1, 9))
                X = np.asfortranarray(g.standard_normal((F, N)).astype(dtype))
                params = np.asfortranarray((g.standard_normal((P, C)) * 2).astype(dtype))
                classes = g.integers(1, C + 1, N).astype(np.int64)
                for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False)):
                    opts = ec.option_bits(ops)
                    pop = api.Population(trees, ops, dtype, n_features=F, n_params=P, eval_context=ec)
                    o, ok = pop.eval(X, params, classes)
                    for t, tree in enumerate(trees):
                        tape, consts = de.flatten(tree, ops, dtype)
                        y, ok_el = oracle.eval_tree_array_parametric(tape, consts, X, params, classes.astype(np.int32), 1, opts, elementwise=True)
                        if not ok_el or not ok[t]:
                            continue
                        tol = parity_tolerance(tree, ops, X, dtype, opts, params, classes - 1)
                        m = np.isfinite(y) & np.isfinite(o[t]) & np.isfinite(tol)
                        err = np.abs(o[t].astype(np.float64) - y.astype(np.float64))
                        with np.errstate(invalid="ignore", divide="ignore"):
                            ratio = np.where(m & (tol > 0), err / np.where(tol > 0, tol, 1), 0)
                        STATS["trees_compared"] += 1
                        STATS["entries_compared"] += int(m.sum())
                        STATS["worst_err_over_tol"] = max(STATS["worst_err_over_tol"], float(ratio.max()))
                        if ratio.max() <= 1.0:
                            continue
                        j = int(np.argmax(ratio))
                        def leaf(n):
                            if n.constant:
                                return mpf(float(dtype(n.val)))
                            if getattr(n, "is_parameter", False):
                                return mpf(float(params[n.parameter - 1, classes[j] - 1]))
                            return mpf(float(X[n.feature - 1, j]))
                        try:
                            d = truth(tree, ops, leaf, 0, lambda n: None).v
                            eg, eo = float(abs(mpf(float(o[t][j])) - d)), float(abs(mpf(float(y[j])) - d))
                        except Exception as e:
                            d, eg, eo = None, None, str(e)
                        rec = dict(kind="value", fuzz=f"fuzz_param {seed0} rep {rep}", dtype=np.dtype(dtype).name, opts=int(opts), tree=de.string_tree(tree, ops)[:200],
                                   sample=j, err_over_tol=float(ratio[j]), tol=float(tol[j]), gpu=float(o[t][j]), oracle=float(y[j]),
                                   truth=None if d is None else float(d), gpu_err_over_tol=None if eg is None else eg / float(tol[j]),
                                   oracle_err_over_tol=eo / float(tol[j]) if isinstance(eo, float) else eo)
                        out.append(rec)
                        print(json.dumps(rec), flush=True)
                    pop.close()


if __name__ == "__main__":
    found = []
    which = sys.argv[1:] or ["param31", "param32", "jac31", "jac32"]
    for w in which:
        (param_findings if w.startswith("param") else (gpu_findings if w.startswith("gpu") else jacobian_findings))(int(w[-2:]), found)
    print(json.dumps(dict(summary=True, flavours=which, findings=len(found), **STATS,
                          device_is_the_outlier=sum(1 for r in found if isinstance(r.get("gpu_err_over_tol"), float) and isinstance(r.get("oracle_err_over_tol"), float)
                                                    and r["gpu_err_over_tol"] > 1.0 and r["gpu_err_over_tol"] > r["oracle_err_over_tol"]))))
