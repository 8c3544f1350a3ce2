"""Differential fuzzing against the oracle with CLASSIFIED findings (test infrastructure).

A finding is a flag or value difference between the HIP path and `oracle/` on a random (tree, X).  Two implementations
of the transcendentals that are each within an ulp or two legitimately differ — without bound where a tree is
ill-conditioned (`cos(exp(exp(x)))`, `c / (x - cos(y))` next to a pole: DESIGN.md §5) — so a fuzzer that stops at its
first difference stops on every seed and gates nothing.  Here every difference is put through the SAME
conditioned tolerance models the test-suite uses (`helpers.parity_tolerance`, `helpers.grad_tolerance`):

  * a VALUE beyond its finite tolerance, or a finite/non-finite pattern that differs on well-conditioned samples -> REAL
  * a FLAG difference on a tree that has no ill-conditioned sample (value model) / entry (gradient model)        -> REAL
  * a flag difference on a tree with ill-conditioned samples: the documented freedom (the flag inherits the last bit of
    a transcendental where the tree turns it into a discontinuity) -> counted as `ill_flags`
  * samples / entries the model classes as ill-conditioned: not compared, counted as `ill_values`

Runs go on to the end of the seed; `tests/test_gpu_fuzz.py` fails on any REAL finding and caps the ill-conditioned
shares.  The command-line fuzzers in tests/fuzz/ use the same functions with other seeds."""
import numpy as np

import dynamicexpressions_jl_amd as de
from helpers import grad_tolerance, parity_tolerance
from oracle import oracle

GRAD_MODES = {"variable": (True, oracle.GRAD_VARIABLE), "constant": (False, oracle.GRAD_CONSTANT), "both": ("both", oracle.GRAD_BOTH)}

OPS_HOT = de.synth.BENCH_OPERATORS
# continuous operators only: at a discontinuity (greater, mod, rem, round, sign at equality) a one-ulp difference of an operand
# legitimately flips the result and the relative-noise tolerance model cannot see it
OPS_WIDE = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^"),
                           unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                            "safe_sqrt", "atan", "relu"))


class Findings:
    def __init__(self):
        self.trees = self.compared = self.ill_values = self.ill_flags = self.flag_checks = 0
        self.real = []

    def add(self, other):
        for k in ("trees", "compared", "ill_values", "ill_flags", "flag_checks"):
            setattr(self, k, getattr(self, k) + getattr(other, k))
        self.real += other.real
        return self

    def summary(self):
        return (f"{self.trees} trees, {self.flag_checks} flags checked ({self.ill_flags} differ on ill-conditioned trees), "
                f"{self.compared} values bounded, {self.ill_values} ({100.0 * self.ill_values / max(self.compared, 1):.2f} %) ill-conditioned, "
                f"{len(self.real)} REAL finding(s)")


def _oracle_eval(tape, consts, X, opts, params, classes):
    if params is None:
        return oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
    return oracle.eval_tree_array_parametric(tape, consts, X, params, np.asarray(classes, dtype=np.int32), 1, opts, elementwise=True)


def fuzz_eval(api, trees, ops, X, dtype, eval_context=None, params=None, classes=None, label=""):
    """Population eval against the oracle; returns Findings (never asserts)."""
    f = Findings()
    ec = eval_context or api.EvalContext()
    P = 0 if params is None else params.shape[0]
    pop = api.Population(trees, ops, dtype, n_features=X.shape[0], n_params=P, eval_context=ec)
    out, ok = pop.eval(X, params, classes) if P else pop.eval(X)
    opts = ec.option_bits(ops) & 15  # the oracle knows the reference's options (turbo / full_eval are the device's business)
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, dtype)
        y, ok_el = _oracle_eval(tape, consts, X, opts, params, classes)
        f.trees += 1
        f.flag_checks += 1
        where = f"{label} {np.dtype(dtype).name} opts={opts} tree {t}: {de.string_tree(tree, ops)[:160]}"
        tol = None
        if bool(ok[t]) != ok_el:
            tol = parity_tolerance(tree, ops, X, dtype, opts, params, None if classes is None else np.asarray(classes) - 1)
            if np.isinf(tol).any():
                f.ill_flags += 1
            else:
                f.real.append(f"FLAG gpu={bool(ok[t])} oracle={ok_el} {where}")
            continue
        if not ok_el:
            continue
        tol = parity_tolerance(tree, ops, X, dtype, opts, params, None if classes is None else np.asarray(classes) - 1)
        wc = np.isfinite(tol)
        fin_o, fin_g = np.isfinite(y), np.isfinite(out[t])
        if not np.array_equal(fin_g[wc], fin_o[wc]):
            f.real.append(f"FINITENESS on a well-conditioned sample {where}")
            continue
        m = fin_o & fin_g
        err = np.abs(out[t][m].astype(np.float64) - y[m].astype(np.float64))
        f.compared += int(m.sum())
        f.ill_values += int(np.isinf(tol[m]).sum())
        if np.any(err > tol[m]):
            k = int(np.argmax(np.where(np.isfinite(tol[m]), err / tol[m], 0)))
            f.real.append(f"VALUE err {err[k]:.3g} > tol {tol[m][k]:.3g} {where}")
    pop.close()
    return f


def fuzz_grad(api, trees, ops, X, dtype, mode, params=None, classes=None, label=""):
    """Population Jacobians against the oracle (forward duals); returns Findings."""
    f = Findings()
    variable, omode = GRAD_MODES[mode]
    P = 0 if params is None else params.shape[0]
    pop = api.Population(trees, ops, dtype, n_features=X.shape[0], n_params=P)
    out, grads, ok = pop.eval_grad(X, variable, params, classes) if P else pop.eval_grad(X, variable)
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, dtype)
        if P:
            t2, PX = oracle.parametric_to_plain(tape, X, params, classes)
            y, g, ok_el = oracle.eval_grad_tree_array(t2, consts, PX, omode, elementwise=True)
        else:
            y, g, ok_el = oracle.eval_grad_tree_array(tape, consts, X, omode, elementwise=True)
        f.trees += 1
        f.flag_checks += 1
        where = f"{label} {np.dtype(dtype).name} [{mode}] tree {t}: {de.string_tree(tree, ops)[:160]}"
        kw = dict(params=params, classes=classes) if P else {}
        if bool(ok[t]) != ok_el:
            tol = grad_tolerance(tree, ops, X, dtype, mode, **kw)
            tolx = parity_tolerance(tree, ops, X, dtype, 7, params, None if classes is None else np.asarray(classes) - 1)
            if tol is None or np.isinf(tol).any() or np.isinf(tolx).any():
                f.ill_flags += 1
            else:
                f.real.append(f"GRAD FLAG gpu={bool(ok[t])} oracle={ok_el} {where}")
            continue
        if not ok_el:
            continue
        if tuple(np.shape(grads[t])) != tuple(g.shape):
            f.real.append(f"GRAD SHAPE {np.shape(grads[t])} vs {g.shape} {where}")
            continue
        if g.size == 0:
            continue
        tol = grad_tolerance(tree, ops, X, dtype, mode, **kw)
        if tol is None:
            f.real.append(f"no partial table for an operator of {where}")
            continue
        err = np.abs(np.asarray(grads[t], dtype=np.float64) - g.astype(np.float64))
        f.compared += tol.size
        f.ill_values += int(np.isinf(tol).sum())
        if np.any(err > tol):
            f.real.append(f"GRAD VALUE worst err/tol {np.nanmax(np.where(np.isfinite(tol), err / tol, 0)):.3g} {where}")
            continue
        tolx = parity_tolerance(tree, ops, X, dtype, 7, params, None if classes is None else np.asarray(classes) - 1)
        m = np.isfinite(tolx)
        if np.any(np.abs(out[t].astype(np.float64) - y)[m] > tolx[m]):
            f.real.append(f"VALUE (primal of the gradient call) {where}")
    pop.close()
    return f


def random_trees(rng, ops, F, dtype, n, max_nodes, offset=0, node_type=de.Node, nparams=0):
    args = (node_type, nparams) if nparams else ()
    return [de.synth.gen_random_tree_fixed_size(1 + (i * 7 + offset) % max_nodes, ops, F, rng, dtype, *args) for i in range(n)]


def gen_mixed_arity_tree(rng, ops, F, dtype, max_layers):
    """The reference's Supposition strategy (test/supposition_utils.jl:14-48, Data.Recursive over leaf | unary | binary |
    ternary wrappers): a leaf, or an operator of a uniformly drawn degree whose children are drawn the same way one layer down."""
    if max_layers <= 0 or rng.rand() < 0.3:
        return de.synth.make_random_leaf(F, rng, dtype)
    degs = [d for d in (1, 2, 3) if len(ops.ops[d - 1]) > 0]
    d = degs[rng.randint(len(degs)) - 1]
    op = rng.randint(len(ops.ops[d - 1]))
    # thin out below the top: keeps the expected size finite for ternary operators (the reference bounds layers, not nodes)
    kids = [gen_mixed_arity_tree(rng, ops, F, dtype, (max_layers - 1) if k == 0 else min(max_layers - 1, 3)) for k in range(d)]
    return de.Node(op, *kids)
